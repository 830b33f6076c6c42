cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/bf16_parity_figures.txt
( timeout 900 python -m pytest tests/test_gpu_parity_bf16.py -m gpu -q --tb=short 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" | tail -15 ) > gpurun_out/c26_parity.log 2>&1
tail -8 gpurun_out/c26_parity.log; grep "per-tensor gates" gpurun_out/bf16_parity_figures.txt | cut -c1-200
