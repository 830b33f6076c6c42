"""GPU tier: full-size nets through the gfx950 library vs the float32 oracle (reference arithmetic is float32,
K.backend/common.py:4).  Tolerances are BASELINE.json's: per-voxel logits <= 1e-4 (f32 parity mode), Dice of the
arg-max labels within 1e-3 of the oracle's (both modes)."""
import numpy as np
import pytest
import torch

import parity_utils as U

pytestmark = pytest.mark.gpu

FULL2D, FULL3D = (6, 12, 36, 24), (3, 4, 12, 8)


def _pair(kind, variant, b, size, cols, dtype):
    m, P, fwd = U.build_pair(kind, variant, b, size, cols, dtype, FULL2D, FULL3D, odtype=torch.float32)
    m.ctx.dropout_enabled = False
    return m, P, fwd


@pytest.mark.parametrize("kind,variant,b,size,cols,primed", [
    ("2d", "denseunet", 1, 512, None, False),          # BASELINE configs[0]: single 512x512 slice
    ("2d", "denseunet", 1, 512, None, True),           # same, batch statistics taken in the conv epilogues
    ("2d", "denseunet", 8, 512, None, True),           # BASELINE configs[1]: the batch bench.py times --
                                                       # M = 8 x 512^2 picks other tiles / split-K counts than 1-2 slices do
    ("2d", "densenet", 2, 224, None, False),
    ("hybrid", "3dpart", 1, 224, 12, False),           # configs[2]
    ("hybrid", "end2end", 1, 224, 12, False),          # configs[3]
    ("hybrid", "end2end", 1, 224, 12, True),
    ("3d", "3dpart", 1, 224, 12, False),               # the per-shard network of configs[4] (unsharded)
])
def test_full_forward_parity_f32(hip_lib, kind, variant, b, size, cols, primed):
    m, P, fwd = _pair(kind, variant, b, size, cols, "f32")
    x, y = U.synthetic_batch(kind, b, size, cols)
    xt = torch.tensor(x)
    # the oracle's predict (on a copy of the parameters) runs on a second host thread beside its training step: independent
    # runs of thousands of small CPU ops that do not fill the box (GPU tier time)
    import copy
    import threading
    Pp, box = copy.deepcopy(P), {}

    def run_predict():
        try:
            box["ref"] = U.R.predict(Pp, fwd, xt).numpy()
        except BaseException as e_:      # noqa: BLE001 -- re-raised below
            box["err"] = e_
    th = threading.Thread(target=run_predict)
    th.start()
    ref_loss, ref_grads, ref_logits = U.R.train_step(P, fwd, U.loss_fn_for(kind), xt, torch.tensor(y), {})
    th.join()
    if "err" in box:
        raise box["err"]
    ref = box["ref"]
    got = m.predict(x)
    scale = max(1.0, float(np.abs(ref).max()))
    e = float(np.abs(got - ref).max())
    assert e <= 1e-4 * scale, "predict logits: max abs err %.3e (scale %.3g)" % (e, scale)
    assert min(U.dice_vs_oracle(got, ref)) >= 1 - 1e-3
    # training-phase forward (batch statistics) + loss
    ka = U.pkg("keras_api")
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    w_before = m.get_weights_dict()
    if primed:
        # steady-state path: a training-phase forward leaves every layer's batch mean behind (the shift of the one-pass
        # epilogue moments); weights and moving statistics are put back, so the step below starts from the same state
        # as the oracle's but runs the primed launch list (what every step after the first runs)
        assert len(m.ctx.stats_sinks) > 5
        P0 = m.ctx.P.clone()
        m.forward_train_mode(x)
        m.ctx.P.copy_(P0)
        assert all(s.primed for s in m.ctx.stats_sinks)
    loss = m.train_on_batch(x, y)
    got_l = m._download_logits().cpu().numpy()
    rl = ref_logits.numpy()
    scale = max(1.0, float(np.abs(rl).max()))
    e = float(np.abs(got_l - rl).max())
    assert e <= 2e-4 * scale, "train-mode logits: max abs err %.3e (scale %.3g)" % (e, scale)
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss)
    assert min(U.dice_vs_oracle(got_l, rl)) >= 1 - 1e-3
    # gradients: relative L2 error per tensor (max-norm is dominated by single ReLU flips at f32 roundoff).
    # Calibration: the float32 oracle differs from ITSELF by this much when only its convolution algorithm changes
    # (torch mkldnn on/off, 224x224x12): end2end worst 1.0 % / median 0.07 %, 3dpart worst 3.3 % / median 1.4 % (the
    # frozen-2D-net graph is chaotic: 1.4e-4 on the logits already).  Measured here (MI355X): 2d-512 0.44 % / 0.03 %,
    # densenet 1.8 % / 0.5 %, 3dpart 2.9 % / 1.7 %, end2end 1.1 % / 0.09 %, 3d 2.3 % / 0.26 %.  The per-config bounds
    # are ~2x those; a kernel that dropped 4 of 128 tile rows (3 % of the pixels) measured 5.8-10 % / 0.9 % and a
    # norm ratio of 0.96, i.e. it fails all three checks.
    tol_worst, tol_median = {("2d", "denseunet"): (2e-2, 3e-3), ("2d", "densenet"): (4e-2, 1.5e-2),
                             ("hybrid", "3dpart"): (6e-2, 4e-2), ("hybrid", "end2end"): (3e-2, 5e-3),
                             ("3d", "3dpart"): (5e-2, 1e-2)}[(kind, variant)]
    gg = m.get_grads_dict()
    worst = (0.0, None)
    rels, ratios = [], []
    rms_max = max(float(np.sqrt((g.numpy().astype(np.float64) ** 2).mean())) for g in ref_grads.values())
    for (name, i), g in ref_grads.items():
        a, r = gg[name][i].astype(np.float64), g.numpy().astype(np.float64)
        # floor: gradients that are mathematically zero (e.g. the bias of a conv feeding a batch-stat BN) are
        # roundoff noise in both implementations
        den = max(np.linalg.norm(r), 1e-3 * rms_max * np.sqrt(r.size))
        rel = np.linalg.norm(a - r) / den
        rels.append(rel)
        if np.linalg.norm(r) > 1e-2 * rms_max * np.sqrt(r.size):
            ratios.append(np.linalg.norm(a) / np.linalg.norm(r))
        if rel > worst[0]:
            worst = (rel, (name, i))
    assert worst[0] < tol_worst, "gradient L2 mismatch: %s" % (worst,)
    assert float(np.median(rels)) < tol_median, "median gradient error %.4f" % float(np.median(rels))
    # a systematic deficit (dropped pixels / taps) shows as a norm ratio != 1 on average; noise averages out
    assert abs(float(np.mean(ratios)) - 1.0) < 5e-3, "mean |got|/|ref| = %.4f" % float(np.mean(ratios))
    # SGD update direction: updated weights moved by (-lr*g*(1+momentum)) -> compare deltas on the classifier
    w_after = m.get_weights_dict()
    last = {"2d": "dense167classifer", "hybrid": "2d3dclassifer", "3d": "3dclassifer"}[kind]
    d_got = w_after[last][0] - w_before[last][0]
    d_ref = P.numpy()[last][0] - w_before[last][0]
    assert np.linalg.norm(d_got - d_ref) <= 2e-2 * np.linalg.norm(d_ref) + 1e-9


def test_shard_shape_forward_loss_f32(hip_lib):
    """the per-GPU shard shape of BASELINE configs[4] -- 512 x 512 planes of the stand-alone 3D DenseNet,
    16 depth planes here (bench.py's `shard3d` runs 64: the same large-grid kernel forms, M = 65 K ... 4.2 M pixels per layer)
    -- training-phase forward (batch statistics) + loss.py's loss in float32 against the float32 oracle."""
    kind, variant, b, size, cols = "3d", "3dpart", 1, 512, 16
    m, P, fwd = _pair(kind, variant, b, size, cols, "f32")
    x, y = U.synthetic_batch(kind, b, size, cols)
    xt = torch.tensor(x)
    P.learning_phase = 1
    with torch.no_grad():
        ref = fwd(P, xt)
        ref_loss = float(U.loss_fn_for(kind)(torch.tensor(y), ref))
    P.bn_batch_means = {}
    ref = ref.numpy()
    m.loss_layer.set_labels(m._labels_internal(y))
    got = m.forward_train_mode(x)
    m.loss_layer.run(False)
    loss = m.loss_layer.value()
    scale = max(1.0, float(np.abs(ref).max()))
    e = float(np.abs(got - ref).max())
    print("shard shape 512x512x16 f32: train-mode logits max abs err %.3e (scale %.3g), loss %.6f (oracle %.6f)" % (e, scale, loss, ref_loss))
    assert e <= 2e-4 * scale, "train-mode logits: max abs err %.3e (scale %.3g)" % (e, scale)
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss), (loss, ref_loss)
    assert min(U.dice_vs_oracle(got, ref)) >= 1 - 1e-3


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_depth_halo_form_world1_equals_unsharded_gpu(hip_lib, monkeypatch, dtype):
    """the depth-sharded launch list in ONE process (HDU_FORCE_DEPTH_HALO=1, world-1 ShardInfo: every 3D layer
    over an input that stores its halo planes, which stay zero without neighbours = the unsharded network) on the GPU, full depth,
    224 x 224 x 12 -- what bench.py times as the per-rank compute of the sharded step was compared on the emulator only
    (tests/test_model_parity.py).  A shard decides like the whole layer (tile form, split-K, kernel family), so the two launch
    lists sum every output element in the same order: float32 logits / loss / gradients equal to float-atomics noise; bf16 (the
    depth-valid / cropped forms of the halo-tile kernels) equal to a few bf16 ulps of the logits."""
    shard_mod = U.pkg("shard")
    ctor = U.pkg("densenet3d_sharded").dense_net3d
    x, y = U.synthetic_batch("3d", 1, 224, 12, seed=5)

    def step(halo):
        monkeypatch.setenv("HDU_FORCE_DEPTH_HALO", "1" if halo else "0")
        m = ctor(U.make_args(1, 224, 12), dtype=dtype, nb_layers3d=FULL3D, seed=9, shard=shard_mod.ShardInfo(0, 1) if halo else None)
        assert any(cv.halo for cv in m.ctx.convs) == halo
        m.ctx.dropout_enabled = False
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
        loss = m.train_on_batch(x, y)
        names = [U.pkg("ops").conv_kernel_name(cv.d_f, 0) for cv in m.ctx.convs]
        return loss, m._download_logits().cpu().numpy(), m.ctx.G[:m.ctx.n_trainable].clone(), names

    l_h, z_h, g_h, n_h = step(True)
    l_u, z_u, g_u, n_u = step(False)
    if dtype == "bf16":
        assert any(nm.startswith("conv_halo_wide_kernel") for nm in n_h) and any(nm.startswith("conv_halo_wide_kernel") for nm in n_u)
    scale = max(1.0, float(np.abs(z_u).max()))
    ez, eg = float(np.abs(z_h - z_u).max()), float((g_h - g_u).norm() / g_u.norm())
    print("depth-halo form vs unsharded, %s, 224x224x12: logits max abs diff %.3e (scale %.3g), loss %.6f vs %.6f, gradient rel-L2 %.3e"
          % (dtype, ez, scale, l_h, l_u, eg))
    if dtype == "f32":
        assert ez <= 1e-4 * scale and abs(l_h - l_u) <= 1e-5 * abs(l_u) and eg <= 1e-3, (ez, l_h, l_u, eg)
    else:
        assert ez <= 2e-2 * scale and abs(l_h - l_u) <= 2e-3 * abs(l_u) and eg <= 5e-2, (ez, l_h, l_u, eg)


@pytest.mark.parametrize("dtype,kind,variant,b,size,cols", [
    ("f32", "2d", "denseunet", 2, 256, None),
    ("bf16", "2d", "denseunet", 8, 512, None),          # the benchmarked configuration itself
    ("bf16", "hybrid", "end2end", 1, 224, 12),
], ids=["f32-2d", "bf16-2d-8x512", "bf16-end2end"])
def test_graph_replay_equals_eager_steps(hip_lib, dtype, kind, variant, b, size, cols):
    """what bench.py times is a REPLAYED hipGraph of the step; every parity test drives eager steps.
    Same model, same start state (weights, velocities, moving statistics, dropout counter): three eager steps, then the state
    is restored, the step is captured and replayed three times.  The two runs launch the same kernels on the same buffers, so
    they differ only by the order of the float atomics (epilogue statistics, pixel-split filter gradients).  That noise is
    MEASURED in the test (the eager steps are run twice from the same state) and the replayed graph is held to it; a captured
    launch with a stale pointer / argument (learning rate, seed, a buffer rebuilt after capture) is off by O(1) of the update."""
    ka = U.pkg("keras_api")
    if kind == "2d":
        m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(b, size), dtype=dtype)
        lossfn = U.pkg("loss").weighted_crossentropy_2ddense
    else:
        m = U.pkg("hybridnet").dense_rnn_net(U.make_args(b, size, cols), dtype=dtype)
        lossfn = U.pkg("loss").weighted_crossentropy
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[lossfn])
    x, y = U.synthetic_batch(kind, b, size, cols)
    m.train_on_batch(x, y)                   # primes the epilogue statistics, builds the step tables (dropout stays ON)
    m.train_step_resident()
    torch.cuda.synchronize()
    ctx = m.ctx
    state = (ctx.P.clone(), ctx.V.clone(), ctx.seed_dev.clone(), m.optimizer.iterations,
             [(r.mean.clone(), r.var.clone()) for r in ctx.stat_roots])

    def restore():
        ctx.P.copy_(state[0]); ctx.V.copy_(state[1]); ctx.seed_dev.copy_(state[2])
        m.optimizer.iterations = state[3]
        for r, (mu, va) in zip(ctx.stat_roots, state[4]):
            r.mean.copy_(mu); r.var.copy_(va)

    def three_steps():
        losses = []
        for _ in range(3):
            m.train_step_resident()
            losses.append(m.loss_value())
        torch.cuda.synchronize()
        return losses, ctx.P.clone()

    assert m._graph is None
    l_eager, p_eager = three_steps()
    restore()
    l_again, p_again = three_steps()          # the SAME eager steps once more: the run-to-run noise of the float atomics
    restore()
    m.capture_graph(warmup=0)
    assert m._graph is not None
    restore()                                 # (capture executes nothing, but be explicit)
    l_graph, p_graph = three_steps()
    upd = float((p_eager - state[0]).double().norm())
    noise = float((p_again - p_eager).double().norm())
    diff = float((p_graph - p_eager).double().norm())
    print("graph replay vs eager (%s %s/%s): losses %s vs %s (eager again %s); |P_graph - P_eager| = %.3e, |P_eager' - P_eager| = %.3e, "
          "|update| %.3e" % (dtype, kind, variant, ["%.6f" % v for v in l_graph], ["%.6f" % v for v in l_eager],
                             ["%.6f" % v for v in l_again], diff, noise, upd))
    assert upd > 0 and np.isfinite(diff)
    # a fresh random-init net amplifies roundoff by orders of magnitude per step (measured: two eager runs of the float32 net
    # differ by 4e-4 of the update after three steps), so the bound is the eager-vs-eager distance itself: the replayed graph
    # must not be further from an eager run than ~ another eager run is
    # (round 4, five runs of the end2end case on MI355X: the eager-vs-eager distance itself came out between 0.3 % and 3.4 % of the
    # update's norm -- the float atomics of the 3D branch's batch statistics perturb the first forward by 1e-6 relative, the
    # x250 coupling of the hybrid amplifies it, tools/diag_determinism.py -- so ONE repeat is a weak estimate of the noise: the
    # bound also has a floor of 5 % of the update.  A replay with a stale argument or pointer -- a frozen dropout mask, an ignored
    # learning rate, a table rebuilt after capture -- moves the parameters by tens of percent of the update.)
    assert diff <= max(4.0 * noise, 0.05 * upd), (diff, noise, upd)
    # (round 6, 13 runs of the end2end case on MI355X, profiles/r06_graph_replay_vs_eager_noise.txt: the third-step loss of two EAGER runs
    # differs by 0.01-0.2 %, the replay's from an eager run by 0.002-0.86 % -- one repeat is a weak estimate here too, and one run in
    # 13 failed a 0.5 % floor with nothing wrong; the floor grows with the step like the amplification does: 0.5 / 1 / 2 %)
    for k, (a, g, a2) in enumerate(zip(l_eager, l_graph, l_again)):
        assert abs(a - g) <= 4.0 * abs(a - a2) + 5e-3 * (1 << k) * abs(a), (l_eager, l_graph, l_again)
    # and the replayed step is not a no-op: the three losses differ from each other
    assert len({round(v, 7) for v in l_graph}) == 3


@pytest.mark.parametrize("kind,variant,b,size,cols", [
    ("2d", "denseunet", 2, 512, None),             # BASELINE configs[0] / [1] network at 512 x 512
    ("hybrid", "3dpart", 1, 224, 12),              # configs[2] (its 2D branch is densenet.py's DenseNet)
    ("hybrid", "end2end", 1, 224, 12),             # configs[3]
    ("3d", "3dpart", 1, 224, 12),                  # per-shard network of configs[4]
], ids=["2d-denseunet", "3dpart", "end2end", "3d"])
def test_f32_absolute_logit_error_from_trained_weights(hip_lib, kind, variant, b, size, cols):
    """north_star: per-voxel logits within 1e-4 ABSOLUTE of the reference's.  On random-init weights the logits of a
    161-layer net reach |3.5e3| and only a relative statement is possible (test_full_forward_parity_f32); here the nets are
    first trained with the reference's recipe (tests/test_gpu_parity_bf16.py: trained_weights), so max|logit| is 2-14, and
    the float32 product is held to  max|got - oracle| <= 1e-4 ABSOLUTE  on every voxel of predict (measured on MI355X,
    profiles/r03_bf16_parity_figures.txt: 3.6e-6 ... 3.8e-5, the same size as the float32 oracle's own distance from the
    SAME graph run in float64, printed beside it), and to max(1e-4, 2.5e-5 * max|logit|) on the training-phase forward
    (batch statistics: measured 2.6e-5 ... 1.25e-4 at max|logit| 5 ... 14).
    Round 4: the 2D and the 3D per-shard nets are run once more with the split-bf16 contraction (include/hdu.h
    HDU_TUNE_F32_SPLIT) and its distance from the same oracle is logged beside the exact mode's and held to 1e-3."""
    import os
    from test_gpu_parity_bf16 import trained_weights, oracle_with, product_with, _log
    W = trained_weights(kind, variant, b, size, cols)
    seed = 77 if kind != "2d" else 1234          # the nets that were fitted to ONE volume are evaluated on it (decided logits)
    x, y = U.synthetic_batch(kind, b, size, cols, seed=seed)
    P, fwd = oracle_with(W, kind, variant, b, size, cols)
    xt = torch.tensor(x)
    # the float64 run of the same graph (two forwards of a 161-layer net in double) on a second host thread beside the float32
    # oracle: independent ParamStores, thousands of small CPU ops that do not fill the box (GPU tier time)
    import threading
    box64 = {}

    def run64():
        try:
            P64 = U.R.ParamStore(seed=1, dtype=torch.float64, perturb=False)
            with torch.no_grad():
                fwd(P64, xt.double())
            P64.bn_batch_means = {}
            for name in P64.w:
                P64.w[name] = [torch.tensor(np.asarray(a, np.float64)) for a in W[name]]
            box64["ref64"] = U.R.predict(P64, fwd, xt.double()).numpy()
        except BaseException as e:      # noqa: BLE001 -- re-raised below
            box64["err"] = e
    th64 = threading.Thread(target=run64)
    th64.start()
    ref = U.R.predict(P, fwd, xt).numpy()
    th64.join()
    if "err" in box64:
        raise box64["err"]
    ref64 = box64["ref64"]
    m = product_with(W, kind, variant, b, size, cols, "f32")
    got = m.predict(x)
    split = kind in ("2d", "3d")          # the split-bf16 contraction beside the exact mode (below)
    if split:
        lib = U.pkg("lib")
        prev = lib.set_f32_contraction("bf16x3")
        try:
            got3 = m.predict(x)           # (before forward_train_mode below moves the stored statistics)
        finally:
            lib.set_f32_contraction(prev)
    mx = float(np.abs(ref64).max())
    e_got, e_ref, e_got64 = float(np.abs(got - ref).max()), float(np.abs(ref - ref64).max()), float(np.abs(got - ref64).max())
    dice = U.dice_vs_oracle(got, ref)
    # training-phase forward (batch statistics)
    ka = U.pkg("keras_api")
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
    P.learning_phase = 1
    with torch.no_grad():
        ref_t = fwd(P, xt).numpy()
    P.bn_batch_means = {}
    got_t = m.forward_train_mode(x)
    mx_t = float(np.abs(ref_t).max())
    e_t = float(np.abs(got_t - ref_t).max())
    _log("[f32 absolute %s/%s] predict: max|logit| %.3f, product vs float32 oracle %.3e, float32 oracle vs float64 oracle %.3e, "
         "product vs float64 oracle %.3e; Dice vs oracle %s; training-phase forward: max|logit| %.3f, product vs float32 oracle %.3e"
         % (kind, variant, mx, e_got, e_ref, e_got64, ["%.6f" % d for d in dice], mx_t, e_t))
    # the same product with the split-bf16 contraction (lib.set_f32_contraction("bf16x3"): float32 storage, three bf16 MFMAs per
    # product, <= 3 * 2^-18 relative per product instead of float32's 2^-24): measured beside the exact mode, on the 2D net of
    # the headline and the 3D per-shard net
    e3 = e3_t = None
    if split:
        prev = lib.set_f32_contraction("bf16x3")
        try:
            got3_t = m.forward_train_mode(x)
        finally:
            lib.set_f32_contraction(prev)
        e3, e3_t = float(np.abs(got3 - ref).max()), float(np.abs(got3_t - ref_t).max())
        dice3 = U.dice_vs_oracle(got3, ref)
        _log("[f32 storage, bf16x3 contraction %s/%s] predict: product vs float32 oracle %.3e (exact mode %.3e; bound of the exact mode 1e-4: %s), "
             "Dice vs oracle %s; training-phase forward: %.3e (exact mode %.3e)"
             % (kind, variant, e3, e_got, "MET" if e3 <= 1e-4 else "NOT MET", ["%.6f" % d for d in dice3], e3_t, e_t))
        assert not np.array_equal(got3, got), "the split contraction did not run"
    if os.environ.get("HDU_PARITY_MEASURE_ONLY") == "1":
        return
    if e3 is not None:
        # measured on MI355X (profiles/r04_f32_split_contraction.txt): predict 1.3e-4 (2D) / 2.6e-5 (3D) against the exact mode's
        # 1.2e-5 / 3.7e-6, training-phase forward 2.6e-4 / 2.8e-4 at max|logit| 5 / 16 -- about 10x the float32 product's own error,
        # 1000x inside the bf16 product's 0.1 ... 0.3; NOT a mode that meets the 1e-4 bound on every net
        # round 5: gated at what it measures (1.5 x the figures above), not at a loose 1e-3
        assert e3 <= 2e-4 and e3_t <= 4e-4, (e3, e3_t)
        assert min(dice3) >= 1 - 1e-3
    assert mx <= 40.0, "the recipe is meant to give O(10) logits"
    assert e_got <= 1e-4, "predict logits: max abs err %.3e at max|logit| %.3f" % (e_got, mx)
    assert e_t <= max(1e-4, 2.5e-5 * mx_t), "training-phase logits: max abs err %.3e at max|logit| %.3f" % (e_t, mx_t)
    assert min(dice) >= 1 - 1e-3


@pytest.mark.parametrize("kind,variant,b,size,cols", [
    ("2d", "denseunet", 2, 512, None),             # BASELINE configs[1] shape family
    ("hybrid", "end2end", 1, 224, 12),             # configs[3]
    ("hybrid", "3dpart", 1, 224, 12),              # configs[2]
], ids=["2d-denseunet", "end2end", "3dpart"])
def test_f32_exact_forward_split_backward_mode(hip_lib, kind, variant, b, size, cols):
    """Round 6 (VERDICT r5 item 5 / Missing 3: "a tolerance-meeting mode that is fast"): lib.set_f32_contraction("bf16x3_bwd") --
    float32 storage, the FORWARD convolutions in exact float32 (so predict and the training-phase logits are the parity mode's: held
    bit-equal here), the data and filter gradients of the backward pass with the split-bf16 contraction (<= 3 * 2^-18 per product).
    One training step from the test's perturbed weights against the float32 oracle with the exact mode's gates (per-tensor relative
    L2 worst / median, mean norm ratio, the head's SGD delta), and against the exact mode itself (logged)."""
    from test_gpu_parity_bf16 import _log
    lib = U.pkg("lib")
    ka = U.pkg("keras_api")
    x, y = U.synthetic_batch(kind, b, size, cols)
    xt = torch.tensor(x)
    runs = {}
    for mode in ("exact", "bf16x3_bwd"):
        prev = lib.set_f32_contraction(mode)
        try:
            m, P, fwd = _pair(kind, variant, b, size, cols, "f32")
            pred = m.predict(x)
            m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
            w_before = m.get_weights_dict()
            loss = m.train_on_batch(x, y)
            logits = m._download_logits().cpu().numpy()
            runs[mode] = dict(pred=pred, loss=loss, logits=logits, grads=m.get_grads_dict(), w_before=w_before, w_after=m.get_weights_dict(), P=P, fwd=fwd)
        finally:
            lib.set_f32_contraction(prev)
        del m
    ex, sp = runs["exact"], runs["bf16x3_bwd"]
    assert np.array_equal(ex["pred"], sp["pred"]), "predict must run the exact float32 kernels in the bf16x3_bwd mode"
    # (the training-phase forward takes its batch statistics with float atomics in the conv epilogues: equal to their order)
    e_fwd = float(np.abs(ex["logits"] - sp["logits"]).max())
    assert e_fwd <= 2e-5 * max(1.0, float(np.abs(ex["logits"]).max())), e_fwd
    P, fwd = sp["P"], sp["fwd"]
    ref_loss, ref_grads, ref_logits = U.R.train_step(P, fwd, U.loss_fn_for(kind), xt, torch.tensor(y), {})
    tol_worst, tol_median = {("2d", "denseunet"): (2e-2, 3e-3), ("hybrid", "end2end"): (3e-2, 5e-3), ("hybrid", "3dpart"): (6e-2, 4e-2)}[(kind, variant)]      # (the exact mode's own gates, test_full_forward_parity_f32)
    rms_max = max(float(np.sqrt((g.numpy().astype(np.float64) ** 2).mean())) for g in ref_grads.values())
    fig = {}
    for mode, r in runs.items():
        rels, ratios, dex = [], [], []
        for (name, i), g in ref_grads.items():
            a, rr = r["grads"][name][i].astype(np.float64), g.numpy().astype(np.float64)
            den = max(np.linalg.norm(rr), 1e-3 * rms_max * np.sqrt(rr.size))
            rels.append(np.linalg.norm(a - rr) / den)
            dex.append(np.linalg.norm(a - ex["grads"][name][i].astype(np.float64)) / den)
            if np.linalg.norm(rr) > 1e-2 * rms_max * np.sqrt(rr.size):
                ratios.append(np.linalg.norm(a) / np.linalg.norm(rr))
        fig[mode] = (max(rels), float(np.median(rels)), float(np.mean(ratios)), max(dex), float(np.median(dex)))
    _log("[f32 forward exact, bf16x3 backward %s/%s] predict bit-equal to the exact mode; training-phase logits vs the exact mode %.3e; "
         "gradients vs float32 oracle rel-L2 worst %.4f / median %.5f, mean norm ratio %.5f (exact mode: %.4f / %.5f, %.5f); "
         "vs the exact mode's gradients worst %.2e / median %.2e"
         % (kind, variant, e_fwd, fig["bf16x3_bwd"][0], fig["bf16x3_bwd"][1], fig["bf16x3_bwd"][2], fig["exact"][0], fig["exact"][1],
            fig["exact"][2], fig["bf16x3_bwd"][3], fig["bf16x3_bwd"][4]))
    assert abs(sp["loss"] - ref_loss) <= 1e-4 * abs(ref_loss), (sp["loss"], ref_loss)
    assert fig["bf16x3_bwd"][0] < tol_worst and fig["bf16x3_bwd"][1] < tol_median, fig
    # ... and no further from the oracle than the exact mode is (3dpart's float32 gradients are 1.7 % from the oracle's in BOTH modes:
    # the split contraction adds 1e-5 of that)
    assert fig["bf16x3_bwd"][0] <= fig["exact"][0] + 5e-4 and fig["bf16x3_bwd"][1] <= fig["exact"][1] + 1e-4, fig
    assert abs(fig["bf16x3_bwd"][2] - 1.0) < 5e-3, fig
    assert fig["bf16x3_bwd"][4] > 0.0, "the split contraction did not run in the backward pass"
    last = {"2d": "dense167classifer", "hybrid": "2d3dclassifer"}[kind]
    d_got = sp["w_after"][last][0] - sp["w_before"][last][0]
    d_ref = P.numpy()[last][0] - sp["w_before"][last][0]
    assert np.linalg.norm(d_got - d_ref) <= 2e-2 * np.linalg.norm(d_ref) + 1e-9


@pytest.mark.parametrize("kind,variant,b,size,cols", [
    ("2d", "denseunet", 2, 512, None),
    ("hybrid", "end2end", 1, 224, 12),
])
def test_full_dice_parity_bf16(hip_lib, kind, variant, b, size, cols):
    """bf16 storage / f32 accumulate (the throughput mode): Dice of arg-max labels vs the float32 oracle.
    Random-init logits are nearly tied between classes, so Dice is evaluated where the oracle's top-2 margin
    exceeds the bf16 logit resolution; the fraction of such voxels is asserted too."""
    m, P, fwd = _pair(kind, variant, b, size, cols, "bf16")
    x, y = U.synthetic_batch(kind, b, size, cols)
    ref = U.R.predict(P, fwd, torch.tensor(x)).numpy()
    got = m.predict(x)
    scale = max(1.0, float(np.abs(ref).max()))
    e = float(np.abs(got - ref).max())
    srt = np.sort(ref, -1)
    margin = srt[..., -1] - srt[..., -2]
    sure = margin > 0.05 * scale
    print("bf16 %s: max logit err %.3e (scale %.3g), confident voxels %.1f%%" % (variant, e, scale, 100 * sure.mean()))
    assert e <= 0.08 * scale
    gl, rl = np.argmax(got, -1)[sure], np.argmax(ref, -1)[sure]
    assert min(U.R.dice_per_class(gl, rl)) >= 1 - 1e-3
    assert (np.argmax(got, -1) == np.argmax(ref, -1)).mean() > 0.97


def test_train_loss_decreases_bf16(hip_lib):
    """a few SGD steps on one synthetic batch reduce the loss (optimizers_test.py:25-40 analogue) -- with dropout on
    and through a captured hipGraph, i.e. exactly the path bench.py times."""
    ka = U.pkg("keras_api")
    m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 128), dtype="bf16", nb_layers=(2, 2, 2, 2))
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy_2ddense])
    x, y = U.synthetic_batch("2d", 2, 128, None)
    l0 = m.train_on_batch(x, y)
    m.capture_graph(warmup=1)
    for _ in range(30):
        m.train_step_resident()
    l1 = m.loss_value()
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < 0.7 * l0, (l0, l1)


def test_rccl_data_parallel_path_world1(hip_lib):
    """The multi-process code path (process group over RCCL, parameter broadcast, the eager flat all-reduce between
    the two captured hipGraphs) with a world of ONE -- what a single-GPU box can exercise; world-2 numerics are covered
    by the gloo test on CPU (tests/test_dp_gloo.py)."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, HDU_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--size", "64", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and np.isfinite(rec["config"]["loss"])
