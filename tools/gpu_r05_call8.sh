#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for L in 3dconv_up4 3dconv_up0 3dconv_up1; do HW_ONLY=$L timeout 300 python tools/bench_halo_wide.py shard 3,8,5; done
  for L in conv_up4 conv_up0; do HW_ONLY=$L timeout 300 python tools/bench_halo_wide.py 2d 3,8,5; done
  for L in fianl_conv 3dconv_up4; do HW_ONLY=$L timeout 300 python tools/bench_halo_wide.py v224 3,8,5; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c8_p.txt
