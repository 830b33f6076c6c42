# developer A/B sweep of the tuning knobs inside ONE gpurun call (boxes differ by several percent between calls)
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline | cut -c53-60; }
run A=0
run HDU_BATCH_WGRAD_TARGET=768
run HDU_BATCH_WGRAD_TARGET=1024
run HDU_BATCH_WGRAD_TARGET=1536
run HDU_WGRAD_MIN_STEPS=2
run HDU_WGRAD_MIN_STEPS=8
run HDU_WGRAD_MIN_STEPS=16
run HDU_HALO_TARGET=256
run HDU_HALO_TARGET=1024
run HDU_WGRAD_TARGET=1536
run A=0
