cd $GRAFT_REPO_ROOT
AB_STEPS=30 bash tools/gpu_ab.sh r06_rows_first 4 "2d 3dpart" "head=LIB=tools/libhdu_prev.so" "new_off=HDU_DEBUG_FLAGS=2048" "new="
