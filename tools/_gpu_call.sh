cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2 ) > gpurun_out/c30_tests.log 2>&1
( time python bench.py ) > gpurun_out/c30_bench.json 2> gpurun_out/c30_bench.err
cp gpurun_out/bench_details.json gpurun_out/c30_bench_details.json
cat gpurun_out/c30_tests.log; wc -c gpurun_out/c30_bench.json; tail -4 gpurun_out/c30_bench.err; head -c 300 gpurun_out/c30_bench.json
