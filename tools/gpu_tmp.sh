cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "weight_prep_batched or split3" 2>&1 | tail -3 ) > gpurun_out/s8_tests.log 2>&1
cat gpurun_out/s8_tests.log
for cfg in 2d 3dpart end2end; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --extras none > gpurun_out/s8_$cfg.json 2> gpurun_out/s8_$cfg.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_details.json'))
k=list(d['conv_kernels'].values())[0]
r=[v for n,v in k.items() if 'weight_prep' in n][0]
print("$cfg: weight_prep %.1f us %.0f GB/s; step %.3f ms" % (r['us_per_launch'], r['alg_gbs'], d['main']['ms_per_step']))
PY
done
