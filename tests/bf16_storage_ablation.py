"""Storage-point ablation of the bf16 Dice deficit (a script, not a pytest module; imports oracle/: TEST INFRASTRUCTURE).

DESIGN.md section 4(f) says the bf16 Dice deficit "is accumulated through the 161 layers, not made in the last tensors" without a
figure.  The bf16-storage oracle (oracle/torch_ref.py ParamStore.store_bf16: the product's storage points applied to the oracle's
float32 arithmetic; the GPU parity tests show the bf16 product equal to it) has a per-storage-point switch (store_policy).  For the
two cases whose deficit lies within 2 x of north_star's 1e-3 bound -- `2d/trained` and `end2end/trained` -- this script trains the
weights with the product's float32 mode on the GPU (the recipe of tests/test_gpu_parity_bf16.py), then runs `predict` of the oracle
on the host with
    all float32 (the reference) | all bf16 storage | float32 storage for ONE group, bf16 elsewhere | bf16 storage for ONE group only
and prints the Dice deficit per class of every variant against the float32 run, plus the max abs logit error.

    python tests/bf16_storage_ablation.py [2d] [end2end] > gpurun_out/bf16_storage_ablation.txt
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_utils as U  # noqa: E402
import test_gpu_parity_bf16 as B  # noqa: E402


def group_of(tag):
    """storage-point groups: tag = conv layer name, or the name of a q() call site"""
    t = tag
    if t in ("fea2d", "3dconv1", "3dconv1_relu") or t == "dense167classifer":
        # the 2D -> 3D hand-off: the 2D logits (x 250 into the 3D stem) and features as stored, the 4-channel stem input as read
        return "handoff"
    if t.startswith("3d") or t in ("fianl_conv", "2d3dclassifer"):
        if "conv_up" in t or t in ("fianl_conv", "2d3dclassifer"):
            return "decoder3d"
        return "dense3d"
    if t.startswith("conv_up") or t == "line0":
        return "decoder2d"
    if t.startswith("conv1"):
        return "stem2d"
    for s in (2, 3, 4, 5):
        if t.startswith("conv%d_" % s):
            return "block%d" % s
    return "other"


def run(kind, variant, b, size, cols):
    W = B.trained_weights(kind, variant, b, size, cols, B.FULL2D, B.FULL3D, *B.RECIPES["trained"])
    x, y = U.synthetic_batch(kind, b, size, cols, seed=1234)
    xt = torch.tensor(x)
    P, fwd = B.oracle_with(W, kind, variant, b, size, cols)
    ref = U.R.predict(P, fwd, xt).numpy()
    tags = set()

    def variant_run(policy):
        Pb, _ = B.oracle_with(W, kind, variant, b, size, cols)
        Pb.store_bf16 = True

        def pol(tag):
            tags.add(tag)
            return policy(group_of(tag))
        Pb.store_policy = pol
        out = U.R.predict(Pb, fwd, xt).numpy()
        dice = U.dice_vs_oracle(out, ref)
        return [1.0 - d for d in dice], float(np.abs(out - ref).max())

    name = "%s/%s/trained @%dx%d%s" % (kind, variant, b, size, "x%d" % cols if cols else "")
    print("[%s] max|logit| %.3f; Dice deficit per class vs the float32 oracle, max abs logit error" % (name, float(np.abs(ref).max())))
    d, e = variant_run(lambda g: True)
    print("  %-44s %s  logits %.3e" % ("bf16 storage everywhere", ["%.2e" % v for v in d], e))
    base = max(d)
    groups = ["stem2d", "block2", "block3", "block4", "block5", "decoder2d"]
    if kind != "2d":
        groups += ["handoff", "dense3d", "decoder3d"]
    rows = []
    for g in groups:
        d1, e1 = variant_run(lambda gg, g=g: gg != g)
        d2, e2 = variant_run(lambda gg, g=g: gg == g)
        rows.append((g, d1, e1, d2, e2))
        print("  float32 storage for %-12s only, bf16 elsewhere %s  logits %.3e   |  bf16 for %-12s only %s  logits %.3e" % (
            g, ["%.2e" % v for v in d1], e1, g, ["%.2e" % v for v in d2], e2))
        sys.stdout.flush()
    best = min(rows, key=lambda r: max(r[1]))
    worst_alone = max(rows, key=lambda r: max(r[3]))
    print("  => largest single-group relief: float32 storage for %s takes the worst-class deficit %.2e -> %.2e; "
          "largest single-group damage: bf16 for %s alone gives %.2e" % (best[0], base, max(best[1]), worst_alone[0], max(worst_alone[3])))
    print("  storage tags seen: %d (groups: %s)" % (len(tags), sorted(set(group_of(t) for t in tags))))


if __name__ == "__main__":
    U.pkg().lib.load()
    torch.cuda.set_device(0)
    which = sys.argv[1:] or ["2d", "end2end"]
    if "2d" in which:
        run("2d", "denseunet", 2, 512, None)
    if "end2end" in which:
        run("hybrid", "end2end", 1, 224, 12)
