"""developer script: under the ORDERED recipe of tests/determinism_probe.py, which tensors of ONE float32 training step still differ
between two executions from identical state?  Prints the first differing conv outputs (forward) and the parameter-gradient tensors
that differ, by layer and kind."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_utils as U  # noqa: E402
from determinism_probe import ordered_reductions  # noqa: E402

U.pkg("lib").load()
torch.cuda.set_device(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "2d"
ka = U.pkg("keras_api")
with ordered_reductions():
    if kind == "2d":
        m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 512), dtype="f32", seed=4321)
        x, y = U.synthetic_batch("2d", 2, 512, None, seed=77)
    else:
        m = U.pkg("densenet3d_sharded").dense_net3d(U.make_args(1, 224, 12), dtype="f32", seed=4321)
        x, y = U.synthetic_batch("3d", 1, 224, 12, seed=77)
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    m.train_on_batch(x, y)
    m.train_step_resident()
    torch.cuda.synchronize()
    ctx = m.ctx
    print("flags: epilogue_stats", ctx.epilogue_stats, "stats_sinks", len(ctx.stats_sinks), "bn_bwd_fused", ctx.bn_bwd_fused, "fuse_bn_bwd", ctx.fuse_bn_bwd,
          "wgrad_plan", ctx.wgrad_plan is not None, "sums_epilogue", ctx.bnb_sums_epilogue)
    state = (ctx.P.clone(), ctx.V.clone(), ctx.seed_dev.clone(), [(r.mean.clone(), r.var.clone()) for r in ctx.stat_roots])

    def restore():
        ctx.P.copy_(state[0]); ctx.V.copy_(state[1]); ctx.seed_dev.copy_(state[2])
        for r, (mu, va) in zip(ctx.stat_roots, state[3]):
            r.mean.copy_(mu); r.var.copy_(va)

    def snapshot():
        outs = {}
        for cv in ctx.convs:
            a = cv.out.act
            outs[cv.name] = a.buf[a.off:a.off + (a.M - 1) * a.ld + a.C].clone()
        return outs, ctx.G.clone(), ctx.P.clone()

    runs = []
    for rep in range(2):
        restore()
        m.train_step_resident()
        torch.cuda.synchronize()
        runs.append(snapshot())
    (o0, g0, p0), (o1, g1, p1) = runs
    ndiff = 0
    for cv in ctx.convs:
        if not torch.equal(o0[cv.name], o1[cv.name]):
            ndiff += 1
            if ndiff <= 6:
                d = (o0[cv.name].float() - o1[cv.name].float()).abs().max()
                print("forward output differs: %s (K %s, M %d) max |diff| %.3e" % (cv.name, cv.K, cv.out.act.M, float(d)))
    print("conv outputs that differ: %d of %d" % (ndiff, len(ctx.convs)))
    bad = {}
    for p in ctx.params:
        if not p.trainable:
            continue
        a, b = g0[p.offset:p.offset + p.numel], g1[p.offset:p.offset + p.numel]
        if not torch.equal(a, b):
            bad.setdefault(p.kind, []).append(p.layer)
    for k, v in bad.items():
        print("gradient differs: kind %s, %d tensors, e.g. %s" % (k, len(v), v[-4:]))
    print("updated parameters equal:", bool(torch.equal(p0, p1)))
