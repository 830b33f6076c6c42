cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in 3dpart end2end; do for dt in f32 f32x3b; do
  timeout 600 python bench.py --config $cfg --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --extras none > gpurun_out/s6_${cfg}_$dt.json 2> gpurun_out/s6_${cfg}_$dt.err
  cp gpurun_out/bench_details.json gpurun_out/s6_details_${cfg}_$dt.json
  python -c "
import json; d=json.load(open('gpurun_out/s6_${cfg}_$dt.json')); print('$cfg $dt', d['value'], d['ms_per_step'], d.get('config',{}).get('flops_check'))"
done; done
