cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in 2d 3dpart end2end; do
  HDU_BENCH_TRACE=1 timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --extras none > gpurun_out/s5_$cfg.json 2> gpurun_out/s5_$cfg.err
  cp gpurun_out/step_trace_0.json gpurun_out/s5_trace_$cfg.json
  head -c 200 gpurun_out/s5_$cfg.json; echo
done
