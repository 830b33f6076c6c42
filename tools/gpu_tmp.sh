cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/gpu_profile.sh ev_2d_f32x3b 1 --config 2d --dtype f32x3b --steps 10 --warmup 2
ls gpurun_out/prof_ev_2d_f32x3b | head; head -8 gpurun_out/prof_ev_2d_f32x3b/rocprofv3_kernel_stats.csv | cut -c1-160
