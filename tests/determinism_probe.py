"""developer / evidence script (not a pytest module): are two trainings of the float32 product from the same seed BIT-EQUAL?
VERDICT r5 item 1f: every run of the bf16 parity tests trains its own weights, and with float atomics in the statistics / filter-gradient
/ BN-backward reductions the same commit draws different weights run to run.  `ORDERED` below is a recipe of EXISTING switches that
routes every one of those reductions through its two-level (per-workgroup partials, fixed-order finalize) or single-writer form:
    HDU_EPILOGUE_STATS=0        batch moments by the two-pass reduction (hdu_bn_stats) instead of conv-epilogue float atomics
    HDU_BN_BWD_FUSED=0          BN backward sums by hdu_bn_bwd_reduce_coef (partials + finalize) instead of the slot-table atomics
    HDU_FUSE_BN_BWD=0, HDU_BNB_SUMS_EPILOGUE=0, HDU_FUSE_BN_BWD_PW=0   no BN-backward sums in conv epilogues
    HDU_TUNE_WGRAD_TARGET_WGS=1 one pixel split per filter-gradient tile: every dw element has exactly one writer
    python tests/determinism_probe.py [2d|3d|hybrid] [steps]
prints, for the default launch list and for the ordered recipe, whether two trainings agree bit for bit and the step time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_utils as U  # noqa: E402

from parity_utils import ordered_reductions  # noqa: E402,F401


def train(kind, steps):
    ka = U.pkg("keras_api")
    if kind == "2d":
        m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 512), dtype="f32", seed=4321)
        x, y = U.synthetic_batch("2d", 2, 512, None, seed=77)
    elif kind == "3d":
        m = U.pkg("densenet3d_sharded").dense_net3d(U.make_args(1, 224, 12), dtype="f32", seed=4321)
        x, y = U.synthetic_batch("3d", 1, 224, 12, seed=77)
    else:
        m = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 224, 12), dtype="f32", seed=4321)
        x, y = U.synthetic_batch("hybrid", 1, 224, 12, seed=77)
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    m.train_on_batch(x, y)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps - 1):
        m.train_step_resident()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / max(1, steps - 1)
    return m.ctx.P.clone(), m.loss_value(), dt


if __name__ == "__main__":
    U.pkg("lib").load()
    torch.cuda.set_device(0)
    kind = sys.argv[1] if len(sys.argv) > 1 else "2d"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for name in ("default", "ordered"):
        runs = []
        for rep in range(2):
            if name == "ordered":
                with ordered_reductions():
                    runs.append(train(kind, steps))
            else:
                runs.append(train(kind, steps))
        a, b = runs[0][0], runs[1][0]
        nd = int((a != b).sum())
        rel = float((a - b).abs().max() / a.abs().max())
        print("%s/%s, %d steps: loss %.6f / %.6f; %d of %d parameters differ between two trainings (max |diff| / max |w| = %.2e); %.1f ms per step"
              % (kind, name, steps, runs[0][1], runs[1][1], nd, a.numel(), rel, runs[0][2] * 1e3))
        sys.stdout.flush()
