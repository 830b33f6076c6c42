cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_backward_mode and 3dpart" 2>&1 | tail -8 ) > gpurun_out/s7_tests.log 2>&1
( time python bench.py ) > gpurun_out/s7_bench.json 2> gpurun_out/s7_bench.err
cp gpurun_out/bench_details.json gpurun_out/s7_bench_details.json
cat gpurun_out/s7_tests.log; grep -h "f32 forward exact" gpurun_out/bf16_parity_figures.txt | tail -1; wc -c gpurun_out/s7_bench.json; tail -4 gpurun_out/s7_bench.err
