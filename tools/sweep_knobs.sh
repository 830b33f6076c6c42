run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline | cut -c53-60; }
run A=0
run HDU_RED_WGS=512
run HDU_RED_WGS=2048
run HDU_ROW_WGS=1024
run HDU_ROW_WGS=4096
run A=0
run HDU_WGRAD_TARGET=512
run HDU_WGRAD_TARGET=1024
run HDU_WGRAD_MIN_STEPS=2
run HDU_WGRAD_MIN_STEPS=8
run HDU_HALO_TARGET=384
run HDU_HALO_TARGET=768
run A=0
