"""worker of tests/test_dp_gloo.py: world_size-2 data parallel step on CPU (gloo) over the emulator build.
Checks, on every rank: identical weights after the step; all-reduced gradient == mean of the per-rank gradients
(each rank's loss is scaled by 1/world: loss.py:44 takes the mean over the merged batch, multi_gpu.py:65-68)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    pkg = importlib.import_module("h-denseunet_amd")
    import emu_bind
    emu_bind.use_emulator()
    import parity_utils as U
    par = U.pkg("parallel")
    ka = U.pkg("keras_api")
    rank, world = par.init_process_group_from_env("gloo")
    assert world == 2
    nb = (2, 2, 2, 2)
    args = U.make_args(1, 32)

    def make(seed):
        m = U.pkg("densenet").DenseUNet(reduction=0.5, args=args, dtype="f32", nb_layers=nb, seed=seed)
        m.ctx.dropout_enabled = False
        m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy_2ddense])
        return m

    x, y = U.synthetic_batch("2d", 1, 32, None, seed=100 + rank)
    # reference: local single-process gradient from the SAME initial weights (rank 0's, after broadcast)
    local = make(seed=7)                      # same seed on both ranks -> same weights
    local.train_on_batch(x, y)
    g_local = local.ctx.G[:local.ctx.n_trainable].clone()
    gl = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(gl, g_local)
    g_mean = (gl[0] + gl[1]) / world

    dp = make(seed=7 + rank)                  # different seeds: make_parallel must broadcast rank 0's weights
    ka.make_parallel(dp, 2, mini_batch=1)
    assert dp.world_size == 2
    # bucketed exchange (HDU_DP_BUCKETS set by the test): the buckets tile the flat gradient buffer and the backward
    bks = dp._buckets
    assert bks is not None and len(bks) >= 3, bks
    assert bks[0][0] == 0 and bks[-1][1] == len(dp.ctx.bwd) and bks[0][3] == dp.ctx.n_trainable and bks[-1][2] == 0
    for a, b in zip(bks[:-1], bks[1:]):
        assert a[1] == b[0] and a[2] == b[3] and a[0] < a[1] and a[2] < a[3], bks
    w0 = [torch.zeros_like(dp.ctx.P) for _ in range(world)]
    dist.all_gather(w0, dp.ctx.P)
    assert torch.equal(w0[0], w0[1]), "weights not broadcast"
    loss = dp.train_on_batch(x, y)
    g_dp = dp.ctx.G[:dp.ctx.n_trainable]
    err = float((g_dp - g_mean).abs().max() / (g_mean.abs().max() + 1e-12))
    assert err < 1e-5, "all-reduced gradient != mean of local gradients: %g" % err
    w1 = [torch.zeros_like(dp.ctx.P[:dp.ctx.n_trainable]) for _ in range(world)]
    dist.all_gather(w1, dp.ctx.P[:dp.ctx.n_trainable].clone())
    assert torch.equal(w1[0], w1[1]), "weights diverged after the step"
    # loss reported = mean over the global batch
    ll = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(ll, torch.tensor([local.loss_value()]))
    assert abs(loss - float((ll[0] + ll[1]) / 2)) < 1e-5 * abs(loss), (loss, ll)
    # depth sharding bookkeeping
    assert [par.shard_depth(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]
    dist.barrier()
    if rank == 0:
        print("DP_OK grad_err=%.2e loss=%.6f" % (err, loss))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
