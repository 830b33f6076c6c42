#!/bin/bash
# Round-6 closing call of the FINAL tree: the GPU tier (its figures file carries the digest of the kernel sources), THEN the evidence run
# (PMC / kernel statistics / census per workload, the default bench line reading this call's figures and FETCH / WRITE summaries, the
# whole 512-cubed volume).  tools/collect_r06.sh copies the results into profiles/ afterwards.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/gpu_r06_tier.sh > /dev/null 2>&1
cp gpurun_out/bf16_parity_figures.txt profiles/r06_bf16_parity_figures.txt
tail -4 gpurun_out/tier_gpu.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/ev_smoke.txt 2>&1; tail -5 gpurun_out/ev_smoke.txt
bash tools/gpu_r06_evidence.sh
