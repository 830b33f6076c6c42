"""developer diagnostic: is the training-phase forward reproducible from identical state?  (round 4: graph-vs-eager test)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as U
pkg = importlib.import_module("h-denseunet_amd"); pkg.lib.load()
ka = U.pkg("keras_api")
kind = sys.argv[1] if len(sys.argv) > 1 else "hybrid"
if kind == "2d":
    m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(8, 512), dtype="bf16"); lossfn = U.pkg("loss").weighted_crossentropy_2ddense
    x, y = U.synthetic_batch("2d", 8, 512, None)
else:
    m = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 224, 12), dtype="bf16"); lossfn = U.pkg("loss").weighted_crossentropy
    x, y = U.synthetic_batch("hybrid", 1, 224, 12)
m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[lossfn])
m.train_on_batch(x, y); m.train_step_resident(); torch.cuda.synchronize()
ctx = m.ctx
state = (ctx.P.clone(), ctx.V.clone(), ctx.seed_dev.clone(), [(r.mean.clone(), r.var.clone()) for r in ctx.stat_roots])
def restore():
    ctx.P.copy_(state[0]); ctx.V.copy_(state[1]); ctx.seed_dev.copy_(state[2])
    for r, (mu, va) in zip(ctx.stat_roots, state[3]): r.mean.copy_(mu); r.var.copy_(va)
def sums():
    out = []
    for cv in ctx.convs:
        a = cv.out.act
        t = a.buf[a.off:a.off + (a.M - 1) * a.ld + a.C].float()
        out.append((cv.name, float(t.double().abs().sum())))
    return out
runs = []
for rep in range(4):
    restore()
    if rep >= 2:
        m.train_step_resident()      # a full step in between (backward leaves its traces), then restore again
        restore()
    m._step_head(); torch.cuda.synchronize()
    runs.append((m.loss_value(), sums()))
print("losses", [r[0] for r in runs])
base = runs[0][1]
for rep in range(1, 4):
    first = None
    for (n, a), (_, b) in zip(base, runs[rep][1]):
        rel = abs(a - b) / (abs(a) + 1e-30)
        if rel > 1e-6 and first is None:
            first = (n, a, b, rel)
    print("rep", rep, "first conv output that differs by > 1e-6 relative:", first)
