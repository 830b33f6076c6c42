cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/s9_bnb_slots.txt; : > $out
for v in 32 16 8 4 32 16; do
  for cfg in 2d 3dpart end2end; do
    ms=$(HDU_BNB_SLOTS=$v timeout 300 python bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --extras none 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "BNB_SLOTS=$v $cfg $ms" >> $out
  done
done
cat $out
