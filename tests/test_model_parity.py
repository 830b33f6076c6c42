"""Whole-model parity of the HIP path against the oracle restatement of the reference graphs: training-mode
forward logits, loss, every parameter gradient, the SGD-updated weights, the BN moving statistics and predict().
CPU tier: reduced-depth nets at 32x32 under the x86 emulator build of the kernels (same sources as the GPU
library); GPU tier (tests/test_gpu_parity.py) runs the full nets."""
import numpy as np
import pytest
import torch

import parity_utils as U

NB2D, NB3D = (2, 2, 2, 2), (1, 1, 2, 1)

CASES = [
    pytest.param("2d", "denseunet", 2, 32, None, id="denseunet-skips"),
    pytest.param("2d", "densenet", 1, 64, None, id="densenet-noskips"),
    pytest.param("hybrid", "3dpart", 1, 32, 8, id="denseunet_3d-3dpart"),
    pytest.param("hybrid", "end2end", 1, 32, 8, id="dense_rnn_net-end2end"),
    pytest.param("3d", "3dpart", 1, 32, 8, id="densenet3d-standalone"),
]


def run_step_compare(kind, variant, b, size, cols, dtype, tol_logit, tol_grad):
    # the reference computes in float32 (K.backend/common.py:4): the model-level oracle runs in float32 too, so that
    # ReLU-mask decisions on near-zero activations are not an artefact of comparing f32 with f64 (measured: the f32
    # and f64 oracles differ from each other by up to 3% on single gradient tensors of these tiny nets)
    odt = torch.float32
    m, P, fwd = U.build_pair(kind, variant, b, size, cols, dtype, NB2D, NB3D, odtype=odt)
    m.ctx.dropout_enabled = False   # TF's dropout RNG cannot be reproduced: parity runs at rate 0
    x, y = U.synthetic_batch(kind, b, size, cols)
    xt, yt = torch.tensor(x, dtype=odt), torch.tensor(y)
    # ---- predict (moving statistics)
    ref_pred = U.R.predict(P, fwd, xt).numpy()
    got_pred = m.predict(x)
    assert got_pred.shape == ref_pred.shape
    e = float(np.abs(got_pred - ref_pred).max())
    assert e <= tol_logit * max(1.0, float(np.abs(ref_pred).max())), "predict logits: max abs err %.3e" % e
    # ---- one training step
    m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
              loss=[U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense])
    w_before = m.get_weights_dict()
    vel = {}
    ref_loss, ref_grads, ref_logits = U.R.train_step(P, fwd, U.loss_fn_for(kind), xt, yt, vel)
    loss = m.train_on_batch(x, y)
    got_logits = m._download_logits().cpu().numpy()
    e = float(np.abs(got_logits - ref_logits.numpy()).max())
    assert e <= tol_logit * max(1.0, float(np.abs(ref_logits.numpy()).max())), "train-mode logits: max abs err %.3e" % e
    assert abs(loss - ref_loss) <= 10 * tol_logit * abs(ref_loss) + 1e-7, (loss, ref_loss)
    # ---- gradients of every trainable weight
    got_grads = m.get_grads_dict()

    def worst_vs(grads):
        gmax = max(float(g.abs().max()) for g in grads.values())
        worst = (0.0, None)
        for (name, i), g in grads.items():
            err = float(np.abs(got_grads[name][i] - g.numpy()).max())
            scale = max(float(g.abs().max()), 1e-3 * gmax)
            if err / scale > worst[0]:
                worst = (err / scale, (name, i, err, scale))
        return worst

    worst = worst_vs(ref_grads)
    if worst[0] > tol_grad and dtype == "f32":
        # A ReLU whose pre-activation lies within float32 roundoff of zero is decided by the summation ORDER of the conv
        # that feeds it, and at these sizes (4..64 pixels per channel) one such decision moves a per-channel gradient by
        # percents.  The float32 oracle (sequential accumulation, like the unsplit kernels) and the float64 oracle
        # disagree on such elements; the split-K kernels accumulate in shorter chains and side with float64 (measured:
        # bn_up1 beta 3.4 % off the float32 oracle, 9e-4 off the float64 one; unsplit: the other way round).  Either
        # oracle is a valid reference for an element that is zero to float32 precision.
        P64, fwd64 = U.build_pair(kind, variant, b, size, cols, dtype, NB2D, NB3D, odtype=torch.float64)[1:]
        g64 = U.R.train_step(P64, fwd64, U.loss_fn_for(kind), torch.tensor(x, dtype=torch.float64), yt, {})[1]
        worst = min(worst, worst_vs(g64), key=lambda w: w[0])
    assert worst[0] <= tol_grad, "gradient mismatch: %s" % (worst,)
    check_updates = worst_vs(ref_grads)[0] <= tol_grad
    # ---- updated weights + BN moving statistics
    w_after = m.get_weights_dict()
    ow = P.numpy() if check_updates else {}      # (the float32 oracle's updates carry its own ReLU decisions)
    for name, arrs in ow.items():
        for i, a in enumerate(arrs):
            d_ref = a - w_before[name][i]
            d_got = w_after[name][i] - w_before[name][i]
            s = max(float(np.abs(d_ref).max()), 1e-9)
            ulp = 2.5e-7 * max(float(np.abs(a).max()), 1.0)   # the product keeps float32 master weights
            assert float(np.abs(d_got - d_ref).max()) <= tol_grad * s + ulp, (name, i)
    return m


@pytest.mark.parametrize("kind,variant,b,size,cols", CASES)
def test_step_parity_f32_emu(emu_lib, kind, variant, b, size, cols):
    run_step_compare(kind, variant, b, size, cols, "f32", tol_logit=1e-4, tol_grad=2e-2)


@pytest.mark.parametrize("kind,variant,b,size,cols", CASES[:1] + CASES[3:])
def test_step_parity_bf16_emu(emu_lib, kind, variant, b, size, cols):
    """bf16 storage: only coarse agreement is expected; Dice on arg-max labels is the gate (BASELINE.json)"""
    m, P, fwd = U.build_pair(kind, variant, b, size, cols, "bf16", NB2D, NB3D)
    x, _ = U.synthetic_batch(kind, b, size, cols)
    ref = U.R.predict(P, fwd, torch.tensor(x, dtype=torch.float64)).numpy()
    got = m.predict(x)
    assert float(np.abs(got - ref).max()) <= 0.1 * max(1.0, float(np.abs(ref).max()))
    assert min(U.dice_vs_oracle(got, ref)) >= 0.9


def test_weights_roundtrip_and_names(emu_lib, tmp_path):
    """Keras-shaped get/set round trip + save/load by name (tests/test_model_saving.py:176-283 analogue);
    layer names are part of the by_name contract."""
    m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(1, 32), dtype="f32", nb_layers=NB2D)
    names = m.layer_names()
    for n in ("conv1", "conv1_bn", "conv1_scale", "conv2_1_x1_bn", "conv2_1_x1_scale", "conv2_1_x1", "conv2_1_x2",
              "conv2_blk", "conv5_blk_bn", "line0", "conv_up0", "bn_up4", "dense167classifer"):
        assert n in names, n
    w = m.get_weights_dict()
    assert w["conv1"][0].shape == (7, 7, 3, 96) and w["dense167classifer"][0].shape == (1, 1, 64, 3)
    assert [a.shape for a in w["conv1_bn"]] == [(96,)] * 4 and [a.shape for a in w["conv1_scale"]] == [(96,)] * 2
    rng = np.random.default_rng(0)
    w2 = {n: [rng.normal(size=a.shape).astype(np.float32) for a in arrs] for n, arrs in w.items()}
    m.set_weights_dict(w2)
    w3 = m.get_weights_dict()
    for n in w2:
        for a, b in zip(w2[n], w3[n]):
            np.testing.assert_array_equal(a, b)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    m2 = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(1, 32), dtype="f32", nb_layers=NB2D, seed=7)
    m2.load_weights(path, by_name=True)
    for n, arrs in m2.get_weights_dict().items():
        for a, b in zip(arrs, w2[n]):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        m.set_weights_dict({"conv1": [np.zeros((7, 7, 3, 95), np.float32)]})


def test_hybrid_names_and_b1(emu_lib):
    m = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 32, 8), dtype="f32", nb_layers2d=NB2D, nb_layers3d=NB3D)
    names = m.layer_names()
    for n in ("3dconv1", "3dconv1_bn", "3dconv1_scale", "3dconv2_1_x1", "3dconv2_blk", "3dconv_up4", "3dbn_up4",
              "fianl_conv", "final_bn", "2d3dclassifer", "conv1", "dense167classifer"):
        assert n in names, n
    assert "3dclassifer" not in names   # built but never part of the Keras graph (hybridnet.py:176-178)
    assert "line0" not in names
    w = m.get_weights_dict()
    assert w["3dconv1"][0].shape == (7, 7, 7, 4, 96) and w["fianl_conv"][0].shape == (3, 3, 3, 64, 64)
    with pytest.raises(ValueError):
        U.pkg("hybridnet").dense_rnn_net(U.make_args(2, 32, 8), dtype="f32", nb_layers2d=NB2D, nb_layers3d=NB3D)


def test_batched_filter_gradients_equal_per_layer_launches(emu_lib):
    """bf16 training step: the deferred, batched filter gradients (ops.WgradPlan, one launch per kernel family at the
    end of the backward pass) give the same flat gradient as one hdu_conv_wgrad launch per layer."""
    grads = []
    for batched in (True, False):
        m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 32), dtype="bf16", nb_layers=NB2D, seed=5)
        m.ctx.dropout_enabled = False
        if not batched:
            m.ctx.set_batch_wgrad(False)
        assert (m.ctx.wgrad_plan is not None) == batched
        if batched:
            assert len(m.ctx.wgrad_plan) >= 20 and sum(cv.in_plan for cv in m.ctx.convs) == len(m.ctx.wgrad_plan)
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy_2ddense])
        x, y = U.synthetic_batch("2d", 2, 32, None)
        m.train_on_batch(x, y)
        grads.append(m.ctx.G[:m.ctx.n_trainable].clone())
    a, b = grads
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("kind,variant,b,size,cols", [("2d", "denseunet", 2, 32, None)])      # (end2end: the GPU tier, test_gpu_parity.py)
def test_f32_split_filter_gradients_on_bf16_triples(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """Round 6: a float32 network in the "bf16x3_bwd" mode takes its filter gradients from bf16 hi / lo image triples (one split
    launch + the bf16 batched filter-gradient plan at the end of the backward pass, engine.Ctx._build_split_wgrad_plan) instead of
    conv_wgrad_kernel<float>'s in-kernel split.  One training step: every layer with whole bf16 channel chunks is in the plan; the
    flat gradient equals the in-kernel split's (HDU_F32_SPLIT_WGRAD=0) and the exact mode's to the split's 2^-16 per product; the
    logits are the exact mode's."""
    monkeypatch.setenv("HIPEMU_THREADS", "1")      # (same order of the float atomics in every run)
    lib = U.pkg("lib")
    lossf = U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense
    x, y = U.synthetic_batch(kind, b, size, cols)
    out = {}
    for tag, mode, on in (("exact", "exact", "1"), ("triples", "bf16x3_bwd", "1"), ("inkernel", "bf16x3_bwd", "0")):
        monkeypatch.setenv("HDU_F32_SPLIT_WGRAD", on)
        prev = lib.set_f32_contraction(mode)
        try:
            m = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D)[0]
            m.ctx.dropout_enabled = False
            m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[lossf])
            m.train_on_batch(x, y)
            out[tag] = (m.ctx.G[:m.ctx.n_trainable].clone(), m._download_logits().clone())
            if tag == "triples":
                sp, plan, own, _ = m.ctx._split_plan
                taken = [cv for cv in m.ctx.convs if getattr(cv, "in_split_plan", False)]
                skipped = [cv.name for cv in m.ctx.convs if cv.trainable and cv.out.root.needs_grad and not getattr(cv, "in_split_plan", False)]
                assert len(sp) == 2 * len(taken) == 2 * (len(plan) + len(own)) and len(taken) >= 20
                assert all(("conv1" in n or "classifer" in n) for n in skipped), skipped      # 3 / 4-channel input, 3-class heads
            else:
                assert m.ctx._split_plan is None
        finally:
            lib.set_f32_contraction(prev)
    g_ex, g_tr, g_ik = out["exact"][0], out["triples"][0], out["inkernel"][0]
    scale = float(g_ex.abs().max())
    assert scale > 0
    assert torch.equal(out["triples"][1], out["exact"][1])
    assert not torch.equal(g_tr, g_ex) and not torch.equal(g_tr, g_ik)
    assert float((g_tr - g_ik).abs().max()) <= 1e-4 * scale and float((g_tr - g_ex).abs().max()) <= 1e-4 * scale
    n2 = lambda t: float(t.double().norm())
    # (both split forms sit ~4e-5 from the exact mode -- the split DATA gradients they share -- and much closer to each other)
    assert n2(g_tr - g_ex) <= 1e-4 * n2(g_ex), (n2(g_tr - g_ex) / n2(g_ex), n2(g_ik - g_ex) / n2(g_ex))
    assert n2(g_tr - g_ik) <= 1e-5 * n2(g_ex), n2(g_tr - g_ik) / n2(g_ex)


@pytest.mark.parametrize("kind,variant,b,size,cols", [("3d", "3dpart", 1, 32, 8), ("2d", "denseunet", 2, 32, None)])
def test_halo_tile_filter_gradients_equal_im2col_form(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """Round 4: 3 x 3 x 3 layers and convs behind a fused up-sampling take the halo-tile filter gradient (three plane-shifted
    2D problems / the up-sampling resolved in the tile addressing).  bf16 training step of the 3D net and of the 2D net
    (conv_up4): same flat gradient as with those layers on the im2col form (HDU_TUNE_NO_HALO bits 1 + 2), and the halo families
    of the plan really hold such layers."""
    lib = emu_lib.lib.get()
    # one emulator thread: workgroups run in launch order, so the float atomics of both runs add in the same order and the two
    # bf16 steps differ ONLY in the filter-gradient kernel (with threads the statistics' atomics reorder from run to run, bf16
    # roundings flip, and two runs of the SAME configuration already differ by ~1 % under load)
    monkeypatch.setenv("HIPEMU_THREADS", "1")
    grads, fams = [], []
    try:
        for off in (0, 6):
            lib.hdu_set_tuning(8, off)
            m = U.build_pair(kind, variant, b, size, cols, "bf16", NB2D, NB3D)[0]
            m.ctx.dropout_enabled = False
            lossf = U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense
            m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[lossf])
            x, y = U.synthetic_batch(kind, b, size, cols)
            m.train_on_batch(x, y)
            grads.append(m.ctx.G[:m.ctx.n_trainable].clone())
            plan = m.ctx.wgrad_plan
            fams.append(sum(1 for v, ds in plan.descs.items() if v >= 8 for d in ds if d.KD == 3 or (d.ud | d.uh | d.uw)))
    finally:
        lib.hdu_set_tuning(8, 0)
    assert fams[0] >= 1 and fams[1] == 0, fams
    a, c = grads
    assert float(c.abs().max()) > 0
    assert float((a - c).abs().max()) <= 1e-4 * float(c.abs().max())


def test_depth_halo_form_world1_equals_unsharded_and_its_halo_filter_gradients(emu_lib, monkeypatch):
    """The depth-sharded launch list in ONE process (HDU_FORCE_DEPTH_HALO=1, world-1 ShardInfo: every 3D layer over an input that
    stores its halo planes, which stay zero without neighbours = the unsharded network).  float32: logits / loss / gradients of
    the halo form equal the unsharded net's.  bf16: the halo form's 3 x 3 x 3 filter gradients on the halo-tile kernel (depth
    "valid", and depth padding -1 behind the decoder's depth up-sampling) equal the im2col form's."""
    shard_mod = U.pkg("shard")
    ctor = U.pkg("densenet3d_sharded").dense_net3d
    lib = emu_lib.lib.get()
    x, y = U.synthetic_batch("hybrid", 1, 32, 8, seed=5)
    x = np.concatenate([x, np.random.default_rng(3).normal(0, 60, x.shape[:4] + (3,)).astype(np.float32)], -1)   # 4 input channels

    def step(dtype, halo, no_halo_wgrad=0):
        monkeypatch.setenv("HDU_FORCE_DEPTH_HALO", "1" if halo else "0")
        lib.hdu_set_tuning(8, no_halo_wgrad)
        try:
            m = ctor(U.make_args(1, 32, 8), dtype=dtype, nb_layers3d=NB3D, seed=9,
                     shard=shard_mod.ShardInfo(0, 1) if halo else None)
            assert any(cv.halo for cv in m.ctx.convs) == halo
            m.ctx.dropout_enabled = False
            m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
            loss = m.train_on_batch(x, y)
            fam = 0
            if m.ctx.wgrad_plan is not None:
                fam = sum(1 for v, ds in m.ctx.wgrad_plan.descs.items() if v >= 8 for d in ds if d.KD == 3 and d.pd <= 0)
            return loss, m._download_logits().cpu().numpy(), m.ctx.G[:m.ctx.n_trainable].clone(), fam
        finally:
            lib.hdu_set_tuning(8, 0)

    l_h, z_h, g_h, _ = step("f32", True)
    l_u, z_u, g_u, _ = step("f32", False)
    assert float(np.abs(z_h - z_u).max()) <= 1e-4 * max(1.0, float(np.abs(z_u).max()))
    assert abs(l_h - l_u) <= 1e-5 * abs(l_u)
    assert float((g_h - g_u).norm() / g_u.norm()) <= 1e-3
    monkeypatch.setenv("HIPEMU_THREADS", "1")        # (deterministic atomics order: see test_halo_tile_filter_gradients_equal_im2col_form)
    _, _, g_on, fam_on = step("bf16", True)
    _, _, g_off, fam_off = step("bf16", True, no_halo_wgrad=1)
    assert fam_on >= 1 and fam_off == 0, (fam_on, fam_off)
    assert float((g_on - g_off).abs().max()) <= 1e-4 * float(g_off.abs().max())


def test_epilogue_statistics_equal_reduction_pass(emu_lib, monkeypatch):
    """second training step (the first one primes the shift): batch statistics taken in the conv epilogues give the same
    logits, loss and gradient as the separate reduction pass (HDU_EPILOGUE_STATS=0).  64x64, batch 2: at 32x32 the
    deepest BN sees 2 samples and any rounding difference is amplified to percents (measured 2.6 %; here 1e-4)."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("HDU_EPILOGUE_STATS", on)
        m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 64), dtype="f32", nb_layers=NB2D, seed=5)
        assert (len(m.ctx.stats_sinks) > 20) == (on == "1")
        m.ctx.dropout_enabled = False
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy_2ddense])
        x, y = U.synthetic_batch("2d", 2, 64, None)
        m.train_on_batch(x, y)
        if on == "1":
            assert all(s.primed for s in m.ctx.stats_sinks)
        x2, y2 = U.synthetic_batch("2d", 2, 64, None, seed=77)
        loss = m.train_on_batch(x2, y2)
        res.append((loss, m._download_logits().cpu().numpy(), m.ctx.G[:m.ctx.n_trainable].clone()))
    (l1, z1, g1), (l0, z0, g0) = res
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert float(np.abs(z1 - z0).max()) <= 1e-4 * max(1.0, float(np.abs(z0).max()))
    rel = float((g1 - g0).norm() / g0.norm())
    assert rel <= 2e-3, rel


def test_splitk_step_vs_oracles(emu_lib):
    """whole training step with the K loops of the small-grid convs dealt to 2 / 3 / the library's number of workgroups
    (hdu_conv_desc.splitk_ws): every gradient tensor within 5e-3 (max-norm) of the float64 OR the float32 oracle.  One
    ReLU / max-pool decision of this tiny net sits within float32 roundoff of a tie, and which way a run takes it depends
    on the summation order, i.e. on the split count (measured: S=2 lands 2.5e-5 from float64; S=3 and the unsplit
    kernels take the float32 oracle's side and are 3 % from float64 on bn_up1 beta) -- see run_step_compare."""
    lib = emu_lib.lib.get()
    try:
        for S in (2, 3, 0):
            lib.hdu_set_tuning(13, S)
            worst = {}
            for odt in (torch.float64, torch.float32):
                m, P, fwd = U.build_pair("2d", "densenet", 1, 64, None, "f32", NB2D, NB3D, odtype=odt)
                m.ctx.dropout_enabled = False
                x, y = U.synthetic_batch("2d", 1, 64, None)
                m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                          loss=[U.pkg("loss").weighted_crossentropy_2ddense])
                rl, rg, rlog = U.R.train_step(P, fwd, U.loss_fn_for("2d"), torch.tensor(x, dtype=odt), torch.tensor(y), {})
                loss = m.train_on_batch(x, y)
                assert abs(loss - rl) <= 1e-5 * abs(rl)
                assert float(np.abs(m._download_logits().cpu().numpy() - rlog.numpy()).max()) <= 1e-4 * float(rlog.abs().max())
                gg = m.get_grads_dict()
                gmax = max(float(g.abs().max()) for g in rg.values())
                worst[odt] = max(float(np.abs(gg[n][i] - g.numpy()).max()) / max(float(g.abs().max()), 1e-3 * gmax)
                                 for (n, i), g in rg.items())
                if worst[odt] <= 5e-3:
                    break
            assert min(worst.values()) <= 5e-3, (S, worst)
    finally:
        lib.hdu_set_tuning(13, 0)


@pytest.mark.parametrize("kind,variant,b,size,cols", [("2d", "denseunet", 2, 64, None), ("hybrid", "end2end", 1, 32, 8)])
def test_fused_bn_backward_equals_separate_passes(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """BN(+Scale)+ReLU backward in the epilogue of the data-gradient launch (hdu_conv_desc.bnb_*: a*g stored / added
    directly, S1 / S2 in slot rows, the mean terms deferred to hdu_bn_bwd_correct over the producer's own channels) vs
    the separate reduction + apply passes (HDU_FUSE_BN_BWD=0): same loss, same gradient of every parameter -- batch
    statistics BNs (2D net), inference-mode BNs with trainable Scale and batch-statistics 3D decoder BNs (end2end)."""
    res = []
    for on in ("2", "0"):          # 2 = every BN (the default, 1, fuses the inference-mode BNs only: engine.Ctx)
        monkeypatch.setenv("HDU_FUSE_BN_BWD", on)
        m, P, fwd = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D, odtype=torch.float32)
        nf = sum(1 for cv in m.ctx.convs if cv.bnb_fused)
        assert (nf >= 8) == (on == "2"), nf
        m.ctx.dropout_enabled = False
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy])
        x, y = U.synthetic_batch(kind, b, size, cols)
        loss = m.train_on_batch(x, y)
        res.append((loss, m.get_grads_dict()))
    (l1, g1), (l0, g0) = res
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    gmax = max(float(np.abs(a).max()) for gs in g0.values() for a in gs)
    worst = (0.0, None)
    for n, gs in g0.items():
        for i, a in enumerate(gs):
            sc = max(float(np.abs(a).max()), 1e-3 * gmax)
            e = float(np.abs(g1[n][i] - a).max()) / sc
            if e > worst[0]:
                worst = (e, (n, i))
    assert worst[0] <= 2e-4, worst

@pytest.mark.parametrize("kind,variant,b,size,cols", [("2d", "denseunet", 1, 64, None), ("hybrid", "3dpart", 1, 32, 8)])
def test_bn_in_producer_epilogue_equals_materialize_pass(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """hdu_conv_desc.epi_*: the bottleneck 1x1 conv applies the following BN(+Scale)+ReLU in its epilogue and writes the
    3x3 conv's operand directly -- in every predict, and in the training step of the 3dpart hybrid's frozen 2D branch.
    Same logits / loss / gradients as the separate materialise pass (HDU_FUSE_BN_EPILOGUE=0)."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("HDU_FUSE_BN_EPILOGUE", on)
        m = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D, odtype=torch.float32)[0]
        m.ctx.dropout_enabled = False
        fused = [c for c in m.ctx.convs if c.epi_consumer is not None]
        assert (len(fused) >= 8) == (on == "1")
        x, y = U.synthetic_batch(kind, b, size, cols)
        pred = m.predict(x)
        if on == "1":
            m.ctx.learning_phase = 0
            assert all(c.epi_consumer.epi_active() for c in fused)
            m.ctx.learning_phase = 1
            # training: only where no gradient flows through the BN (the frozen 2D branch of 3dpart)
            act = [c for c in fused if c.epi_consumer.epi_active()]
            assert (len(act) > 0) == (variant == "3dpart") and len(act) < len(fused) + (variant == "3dpart")
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense])
        loss = m.train_on_batch(x, y)
        res.append((pred, loss, m.ctx.G[:m.ctx.n_trainable].clone()))
    (p1, l1, g1), (p0, l0, g0) = res
    assert float(np.abs(p1 - p0).max()) <= 1e-5 * max(1.0, float(np.abs(p0).max()))
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert float((g1 - g0).norm() / g0.norm()) <= 1e-4


def test_frozen_bn_epilogue_in_training_equals_materialize_pass(emu_lib, monkeypatch):
    """Round 4: in dense_rnn_net (end2end) every dense-block BN runs on stored statistics AND has a backward pass
    (hybridnet.py:11-97,182-354).  The bottleneck 1x1 conv applies the following BN(+Scale)+ReLU in its epilogue in the
    TRAINING step too: it stores z = relu(a*u + b), u is never written, and the fused BN backward of the 3x3 data gradient takes
    the mask and the normalised input from z (hdu_conv_desc.bnb_relu bit 1).  Same loss and gradients as the materialise pass
    (HDU_FUSE_BN_EPILOGUE_TRAIN=0), and one launch per dense layer fewer."""
    kind, variant, b, size, cols = "hybrid", "end2end", 1, 32, 8
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("HDU_FUSE_BN_EPILOGUE_TRAIN", on)
        m = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D, odtype=torch.float32)[0]
        m.ctx.dropout_enabled = False
        fused = [c for c in m.ctx.convs if c.epi_consumer is not None]
        act = [c for c in fused if c.epi_consumer.epi_active()]                  # (learning phase 1)
        grad_through = [c for c in act if c.epi_consumer.need_input_grad]
        assert (len(grad_through) >= 8) == (on == "1"), (len(fused), len(act), len(grad_through))
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
        x, y = U.synthetic_batch(kind, b, size, cols)
        U.pkg("lib").profile_begin()
        loss = m.train_on_batch(x, y)
        recs, _ = U.pkg("lib").profile_end()
        res.append((loss, m.ctx.G[:m.ctx.n_trainable].clone(), len(recs), m.get_grads_dict()))
    (l1, g1, n1, d1), (l0, g0, n0, d0) = res
    assert n1 <= n0 - 8, (n1, n0)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert float((g1 - g0).norm() / g0.norm()) <= 1e-4
    gmax = max(float(np.abs(a).max()) for gs in d0.values() for a in gs)
    for n, gs in d0.items():
        for i, a in enumerate(gs):
            sc = max(float(np.abs(a).max()), 1e-3 * gmax)
            assert float(np.abs(d1[n][i] - a).max()) <= 5e-4 * sc, (n, i)


@pytest.mark.parametrize("kind,variant,b,size,cols", [("2d", "denseunet", 2, 64, None)])     # (3D dense blocks: same StatsOp code)
def test_finalize_folds_next_bn_equals_two_launches(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """hdu_bn_stats_finalize_fold_next (the finalize launch of a dense layer's epilogue statistics also folds the next
    layer's first BN over the whole slab) == finalize + bn_fold as two launches (HDU_FOLD_NEXT=0): same loss, logits,
    gradients, weights and moving statistics after the second (primed) training step."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("HDU_FOLD_NEXT", on)
        m = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D, odtype=torch.float32)[0]
        m.ctx.dropout_enabled = False
        linked = [s for s in m.ctx.stats_sinks if s.fold_next is not None]
        assert (len(linked) >= 4) == (on == "1")
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense])
        x, y = U.synthetic_batch(kind, b, size, cols)
        m.train_on_batch(x, y)
        x2, y2 = U.synthetic_batch(kind, b, size, cols, seed=77)
        loss = m.train_on_batch(x2, y2)
        res.append((loss, m._download_logits().cpu().numpy(), m.ctx.G[:m.ctx.n_trainable].clone(), m.ctx.P.clone()))
    (l1, z1, g1, p1), (l0, z0, g0, p0) = res
    # (same arithmetic; the epilogue statistics themselves are float atomics, so two runs agree to roundoff only)
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert float(np.abs(z1 - z0).max()) <= 1e-4 * max(1.0, float(np.abs(z0).max()))
    assert float((g1 - g0).norm() / g0.norm()) <= 2e-3
    assert float((p1 - p0).abs().max()) <= 1e-5


@pytest.mark.parametrize("kind,variant,b,size,cols", [("2d", "denseunet", 2, 64, None), ("3d", "3dpart", 1, 32, 8)])
def test_no_finalize_launches_equal_the_three_launch_forms(emu_lib, monkeypatch, kind, variant, b, size, cols):
    """Round 3's launch-count reductions == the launches they replace: the BN fold inside the consumer's materialize pass
    (hdu_materialize_stats, HDU_ABSORB_STATS) vs a finalize launch, and the two-launch BN backward (hdu_bn_bwd_fused,
    HDU_BN_BWD_FUSED) vs reduce / finalize / apply -- same loss, logits, gradients, weights and moving statistics after the
    third training step (the first one primes the statistics shifts)."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("HDU_ABSORB_STATS", on)
        monkeypatch.setenv("HDU_BN_BWD_FUSED", on)
        m = U.build_pair(kind, variant, b, size, cols, "f32", NB2D, NB3D, odtype=torch.float32)[0]
        m.ctx.dropout_enabled = False
        assert (sum(1 for s in m.ctx.stats_sinks if s.absorber is not None) >= 4)
        assert (sum(1 for bn in m.ctx.bns if bn.bsum is not None) >= 4) == (on == "1")
        m.compile(optimizer=U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True),
                  loss=[U.pkg("loss").weighted_crossentropy if kind != "2d" else U.pkg("loss").weighted_crossentropy_2ddense])
        # (the reduced-depth 3D net normalises over 2-16 pixels per channel and amplifies the float atomics' summation order:
        # two IDENTICAL runs of it differ by 2e-5 in the third step's loss and 6e-3 in its gradient -- it is compared after
        # the second step, whose forward and whose predecessor's backward already ran every new launch)
        for seed in (1234, 77, 5)[:3 if kind == "2d" else 2]:
            x, y = U.synthetic_batch(kind, b, size, cols, seed=seed)
            loss = m.train_on_batch(x, y)
        res.append((loss, m._download_logits().cpu().numpy().copy(), m.ctx.G[:m.ctx.n_trainable].clone(), m.ctx.P.clone()))
    (l1, z1, g1, p1), (l0, z0, g0, p0) = res
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    assert float(np.abs(z1 - z0).max()) <= 1e-4 * max(1.0, float(np.abs(z0).max()))
    assert float((g1 - g0).norm() / g0.norm()) <= (2e-3 if kind == "2d" else 2e-2)
    # (moving variances of the CT stem are O(1e3): relative; the 3D net turns the 1e-5 gradient roundoff of step 1 into 6e-5)
    assert float(((p1 - p0).abs() / p0.abs().clamp_min(1.0)).max()) <= (2e-5 if kind == "2d" else 5e-4)
