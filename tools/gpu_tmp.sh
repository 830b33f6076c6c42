cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for f in 0 32 64 128 4 36 68 132; do
  HDU_SPLIT3_FORM=$f timeout 600 python bench.py --dtype f32x3b --steps 6 --warmup 2 --no-cpu-baseline --extras none > gpurun_out/s4_f$f.json 2> gpurun_out/s4_f$f.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_details.json'))
k=list(d['conv_kernels'].values())[0]
r=[v for n,v in k.items() if 'split3' in n][0]
print("form $f: split3 %.1f us %.0f GB/s; step %.3f ms" % (r['us_per_launch'], r['alg_gbs'], d['main']['ms_per_step']))
PY
done 2>&1 | tee gpurun_out/s4_split3_forms.txt
