cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/gpu_profile.sh 2d_s10 0 --config 2d --steps 10 --warmup 3 --no-roofline
tools/gpu_profile.sh 2d 1 --config 2d --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh 3dpart 1 --config 3dpart --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh 3dpart_s10 0 --config 3dpart --steps 10 --warmup 3 --no-roofline
tools/gpu_profile.sh end2end 1 --config end2end --steps 30 --warmup 3 --no-roofline
tools/gpu_profile.sh end2end_s10 0 --config end2end --steps 10 --warmup 3 --no-roofline
for c in 2d 3dpart end2end; do
  cp gpurun_out/prof_$c/pmc_FETCH_SIZE.txt profiles/r03_pmc_FETCH_SIZE_${c}_bf16.txt
  cp gpurun_out/prof_$c/pmc_WRITE_SIZE.txt profiles/r03_pmc_WRITE_SIZE_${c}_bf16.txt
done
( time python bench.py ) > gpurun_out/c18_bench.json 2> gpurun_out/c18_bench.err
cp gpurun_out/bench_details.json gpurun_out/c18_bench_details.json
wc -c gpurun_out/c18_bench.json; tail -4 gpurun_out/c18_bench.err; head -c 600 gpurun_out/c18_bench.json
