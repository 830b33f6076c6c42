cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -7 ) > gpurun_out/c25_smoke.log 2>&1
tools/gpu_profile.sh 2d_s10 0 --config 2d --steps 10 --warmup 3 --no-roofline
tools/gpu_profile.sh 2d 1 --config 2d --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh 3dpart 0 --config 3dpart --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh 3dpart_s10 0 --config 3dpart --steps 10 --warmup 3 --no-roofline
tools/gpu_profile.sh end2end 0 --config end2end --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh end2end_s10 0 --config end2end --steps 10 --warmup 3 --no-roofline
cp gpurun_out/prof_2d/pmc_FETCH_SIZE.txt profiles/r03_pmc_FETCH_SIZE_2d_bf16.txt
cp gpurun_out/prof_2d/pmc_WRITE_SIZE.txt profiles/r03_pmc_WRITE_SIZE_2d_bf16.txt
( time python bench.py ) > gpurun_out/c25_bench.json 2> gpurun_out/c25_bench.err
cp gpurun_out/bench_details.json gpurun_out/c25_bench_details.json
cat gpurun_out/c25_smoke.log; wc -c gpurun_out/c25_bench.json; tail -4 gpurun_out/c25_bench.err; head -c 400 gpurun_out/c25_bench.json
