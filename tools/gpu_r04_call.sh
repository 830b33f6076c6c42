#!/bin/bash
# One gpurun call of round 4 (developer script): stages selected by "$@" (kernels ab bench prof parity parity8 full pmc).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${TAG:-c}
{ free -g | head -2; nproc; } > gpurun_out/${tag}_host.txt 2>&1
for stage in "$@"; do
  t0=$(date +%s)
  case $stage in
    kernels)
      timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv_fprop or conv_wgrad or splitk or plan" > gpurun_out/${tag}_kernels.log 2>&1 ;;
    ab)
      IFS='|' read -ra ARMS <<< "${AB_ARMS:-base=}"      # AB_ARMS="base=HDU_FUSE_PW=0|fuse=HDU_FUSE_PW=1"
      tools/gpu_ab.sh ${tag}_${AB_TAG:-ab} ${AB_ROUNDS:-2} "${AB_CONFIGS:-2d 3dpart end2end}" "${ARMS[@]}" > /dev/null 2>&1 ;;
    bench)
      timeout 1200 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
      cp gpurun_out/bench_details.json gpurun_out/${tag}_bench_details.json 2>/dev/null ;;
    prof)
      tools/gpu_profile.sh ${tag}_2d 0 --config 2d --steps 30 --warmup 3 ;;
    prof3d)
      tools/gpu_profile.sh ${tag}_3dpart 0 --config 3dpart --steps 30 --warmup 3
      tools/gpu_profile.sh ${tag}_end2end 0 --config end2end --steps 30 --warmup 3 ;;
    pmc)
      tools/gpu_profile.sh ${tag}_2d_pmc 1 --config 2d --steps 10 --warmup 2 ;;
    parity)
      timeout 1200 python -m pytest -m gpu -x -q "tests/test_gpu_parity.py::test_graph_replay_equals_eager_steps" \
        "tests/test_gpu_parity.py::test_shard_shape_forward_loss_f32" -s > gpurun_out/${tag}_parity.log 2>&1 ;;
    parity8)
      timeout 1800 python -m pytest -m gpu -x -q "tests/test_gpu_parity.py::test_full_forward_parity_f32[2d-denseunet-8-512-None-True]" \
        "tests/test_gpu_parity_bf16.py::test_bf16_train_step_parity_full_size[2d-8x512-mid]" -s > gpurun_out/${tag}_parity8.log 2>&1 ;;
    kernels_all)
      timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/${tag}_kernels_all.log 2>&1 ;;
    fromz)       # frozen-BN epilogue fusion in the training step: kernel test, the end2end bf16 / f32 parity gates
      timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "fused_bn_backward" > gpurun_out/${tag}_fromz_kernels.log 2>&1
      timeout 900 python -m pytest -m gpu -x -q -s "tests/test_gpu_parity_bf16.py::test_bf16_train_step_parity_full_size[end2end]" \
        > gpurun_out/${tag}_fromz_parity.log 2>&1 ;;
    halo3d)      # 3 x 3 x 3 filter gradients on the halo-tile kernel: kernel tests, then the 3D workloads incl. the shard shape
      timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "wgrad" > gpurun_out/${tag}_halo3d_kernels.log 2>&1
      for arm in ${HALO_ARMS:-2 0}; do
        HDU_NO_HALO=$arm timeout 600 python bench.py --config shard3d --steps 8 --warmup 2 --no-cpu-baseline --extras none \
          > gpurun_out/${tag}_halo3d_shard_$arm.json 2> gpurun_out/${tag}_halo3d_shard_$arm.err
        cp gpurun_out/bench_details.json gpurun_out/${tag}_halo3d_shard_details_$arm.json 2>/dev/null
      done ;;
    closing_parity)   # the model-level GPU gates of the paths changed last (frozen-BN epilogue in training, halo filter gradients in 3D)
      timeout 1200 python -m pytest -m gpu -x -q -s tests/test_gpu_parity.py -k "test_full_forward_parity_f32 and end2end" > gpurun_out/${tag}_closing_f32.log 2>&1
      timeout 1200 python -m pytest -m gpu -x -q -s "tests/test_gpu_parity_bf16.py::test_bf16_train_step_parity_full_size[3dpart]" \
        "tests/test_gpu_parity_bf16.py::test_bf16_train_step_parity_full_size[end2end]" \
        "tests/test_gpu_parity_bf16.py::test_bf16_train_step_parity_full_size[3d]" > gpurun_out/${tag}_closing_bf16.log 2>&1 ;;
    split_parity)
      timeout 900 python -m pytest -m gpu -x -q -s tests/test_gpu_parity.py \
        -k "test_f32_absolute_logit_error_from_trained_weights and (2d-denseunet or (3d and not 3dpart))" > gpurun_out/${tag}_split_parity.log 2>&1 ;;
    split)      # float32 storage with the split-bf16 contraction: logits beside the exact mode, then the timed steps
      HDU_PARITY_LOG=gpurun_out/${tag}_split_parity.txt timeout 900 python -m pytest -m gpu -x -q -s tests/test_gpu_parity.py \
        -k "test_f32_absolute_logit_error_from_trained_weights and (2d-denseunet or (3d and not 3dpart))" > gpurun_out/${tag}_split_parity.log 2>&1
      timeout 900 python bench.py --steps ${SPLIT_STEPS:-20} --warmup 3 --extras 2d:f32,2d:f32x3 --no-cpu-baseline \
        > gpurun_out/${tag}_split_bench.json 2> gpurun_out/${tag}_split_bench.err
      cp gpurun_out/bench_details.json gpurun_out/${tag}_split_bench_details.json 2>/dev/null ;;
    full)
      timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${tag}_gpu_tests.log 2>&1
      python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" >> gpurun_out/${tag}_gpu_tests.log 2>&1 ;;
  esac
  echo "$stage: $(( $(date +%s) - t0 )) s rc=$?" >> gpurun_out/${tag}_stages.txt
done
tail -n 3 gpurun_out/${tag}_*.log 2>/dev/null | tail -40
cat gpurun_out/${tag}_stages.txt
