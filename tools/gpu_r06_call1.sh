#!/bin/bash
# Round 6, first call: (1) the grid-barrier micro-benchmark, (2) the bf16-only kernels on the BASELINE layer
# shapes against im2col (item 1e), (3) the default bench line of the tree as it stands (launch-ordered step traces included).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_gridbarrier tools/ubench_gridbarrier.hip 2> gpurun_out/c1_ubench_build.err
timeout 300 /tmp/ubench_gridbarrier 200 > gpurun_out/c1_ubench_gridbarrier.txt 2>&1
echo "ubench rc $?"; cat gpurun_out/c1_ubench_gridbarrier.txt
timeout 1500 python -m pytest tests/test_kernels_baseline_shapes.py -m gpu -q -x -s 2>&1 | tail -40 > gpurun_out/c1_baseline_shapes.txt
cat gpurun_out/c1_baseline_shapes.txt | tail -30
HDU_BENCH_TRACE=1 timeout 1500 python bench.py --steps 20 > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
cp gpurun_out/bench_details.json gpurun_out/c1_bench_details.json 2>/dev/null
wc -c gpurun_out/c1_bench.json; head -c 1500 gpurun_out/c1_bench.json; tail -5 gpurun_out/c1_bench.err
