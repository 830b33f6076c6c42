# developer A/B sweep of the tuning knobs inside ONE gpurun call (boxes differ by several percent between calls)
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline | cut -c53-60; }
for i in 1 2 3; do
run A=0
run HDU_RED_WGS=512
run HDU_RED_WGS=384
run HDU_RED_WGS=256
done
