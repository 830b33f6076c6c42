#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "stem or test_conv_wgrad" 2>&1 | tail -2 )
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_train_loss_decreases_bf16 or (test_graph_replay and end2end)" 2>&1 | tail -2 )
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
