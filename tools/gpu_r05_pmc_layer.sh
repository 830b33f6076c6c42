#!/bin/bash
# round 5: SQ counters of ONE layer / form / configuration of the halo-wide kernel (and the im2col kernel beside it)
#   tools/gpu_r05_pmc_layer.sh <set: 2d|shard|v224> <layer:form> <cfgs e.g. 1,4> <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$PWD
set_=$1; only=$2; cfgs=$3; tag=$4
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1)); rm -rf /tmp/rp_$i
  HW_ONLY=$only rocprofv3 --kernel-trace --pmc $ctr -d /tmp/rp_$i -o res -- python $ROOT/tools/bench_halo_wide.py $set_ $cfgs > /tmp/rp_$i.out 2> /tmp/rp_$i.err
  db=$(find /tmp/rp_$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $ROOT/tools/rocpd_pmc.py "$db" 6 | grep -A9 -E "conv_halo_wide|conv_igemm"; else echo "pass $i: no db"; tail -3 /tmp/rp_$i.err; fi
done > $ROOT/gpurun_out/pmc_layer_$tag.txt 2>&1
cat $ROOT/gpurun_out/pmc_layer_$tag.txt
