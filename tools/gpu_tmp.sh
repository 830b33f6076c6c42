cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/s10_replay.txt
for v in 16 32 16 32 16 32 16 32 16 32 16 32; do
  HDU_BNB_SLOTS=$v timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "graph_replay_equals_eager_steps and end2end" 2>&1 | grep -E "graph replay vs eager|passed|failed" | cut -c1-420 | sed "s/^/slots=$v /" >> gpurun_out/s10_replay.txt
done
cat gpurun_out/s10_replay.txt
