#!/usr/bin/env python
"""bench.py -- CT slices/s of one fwd+bwd+SGD step of the H-DenseUNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2d|3dpart|end2end|shard3d] [--dtype bf16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

BASELINE.json's metric is "CT slices/sec fwd+bwd (2D 512^2 & 3D 224x224x12)".  The JSON line's top-level `value` is the
2D half on the configuration the metric is quoted on (configs[1]: 2D DenseUNet-161 training step, batch 8 x 512 x 512,
bf16 storage / f32 accumulate, dropout ON, one hipGraph per step); the 3D half -- `denseunet_3d` (configs[2]) and
`dense_rnn_net` end2end (configs[3]) at 224 x 224 x 12 --, the 512 x 512 x 64 per-GPU shard of configs[4] (single GPU
only) and the float32 parity-mode 2D step are timed by the SAME function in the same process and reported, compacted,
under `config.extra_workloads` (value / ms_per_step / roofline fraction each).  The line stays under 6 KB (the driver
keeps 8 KB of stdout); the per-kernel tables go to gpurun_out/bench_details.json (stderr under HDU_BENCH_VERBOSE).
Under N>1 every rank runs the same per-GPU batch (weak scaling) and gradients are summed with one flat RCCL
all-reduce.  Inputs and labels are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8(d) / BASELINE.md section 2: conv FLOPs (2*MAC) per slice, fwd + dgrad + wgrad (stem dgrad skipped).
# "shard3d" is the STAND-ALONE 3D DenseNet (models.py: no HFF head, no `fianl_conv`): BASELINE.md config 5, 459.3 GFLOP per 512 x 512
# slice (235 157 GFLOP per 512^3 volume) -- rounds 3-5 priced it with the 3D net + HFF head figure (121.3 per 224 x 224 slice, i.e.
# 633.7 per 512 x 512 slice: 1.38 x too much).  Every figure is cross-checked against the sum of the per-kernel model
# of the instrumented step (check_step_flops).
TRAIN_GFLOP_PER_SLICE = {"2d": 580.5, "3dpart": 156.6, "end2end": 227.3, "shard3d": 459.3}
REF_PLANE = {"2d": 512, "3dpart": 224, "end2end": 224, "shard3d": 512}      # plane edge the per-slice figure is quoted at
FLOPS_TOLERANCE = 0.03
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3,   # MI355X_MICROARCH.md: dense MFMA peaks
               # "f32x3" = float32 storage, bf16 hi/lo split operands, THREE v_mfma_f32_16x16x16_bf16 per product (lib.set_f32_contraction):
               # the K=16 form moves half the K of the K=32 form per issue, so a product-equivalent peak of 2500 / 2 / 3
               "f32x3": 2500.0 / 6.0,
               # "f32x3b" = lib.set_f32_contraction("bf16x3_bwd"): exact float32 forward (1/3 of the conv FLOPs at the f32 peak), split
               # backward (2/3 at the f32x3 figure): the harmonic mix
               "f32x3b": 1.0 / ((1.0 / 3.0) / 157.3 + (2.0 / 3.0) / (2500.0 / 6.0))}
PEAK_HBM_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)
PROFILE_ROUND = "r06"

# tests/bench_dryrun.py (tests/test_bench_flow_gloo.py only) sets these two: the same control flow on CPU -- x86 emulator build
# of the kernels, gloo instead of RCCL, reduced-depth nets -- so that the multi-rank sequence of collectives of this script is
# checked without a multi-GPU node.  Never a measurement; the JSON line says so.  No environment variable or flag of bench.py
# itself reaches this path.
DRYRUN = False
_dryrun_bind = None


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class _HostEvent:
    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def physical_cores():
    """physical core count of the host (north_star: 'core count stated'); falls back to the logical count"""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        seen = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except Exception:
        pass
    return os.cpu_count() or 1


def build(config, dtype, b, size, cols):
    args = argparse.Namespace(b=b, input_size=size, input_cols=cols)
    ka = importlib.import_module("h-denseunet_amd.keras_api")
    small2d, small3d = (2, 2, 2, 2), (1, 1, 2, 1)
    if config == "2d":
        kw = {"nb_layers": small2d} if DRYRUN else {}
        m = importlib.import_module("h-denseunet_amd.denseunet").DenseUNet(reduction=0.5, args=args, dtype=dtype, **kw)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy_2ddense
    elif config == "3dpart":
        kw = {"nb_layers2d": small2d, "nb_layers3d": small3d} if DRYRUN else {}
        m = importlib.import_module("h-denseunet_amd.denseunet3d").denseunet_3d(args, dtype=dtype, **kw)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    elif config == "shard3d":
        par = importlib.import_module("h-denseunet_amd.parallel")
        sh = par.depth_shard_info("gloo" if DRYRUN else "nccl")
        kw = {"nb_layers3d": small3d} if DRYRUN else {}
        m = importlib.import_module("h-denseunet_amd.densenet3d_sharded").dense_net3d(args, dtype=dtype, shard=sh, **kw)
        par.attach_depth_shard(m)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    else:
        kw = {"nb_layers2d": small2d, "nb_layers3d": small3d} if DRYRUN else {}
        m = importlib.import_module("h-denseunet_amd.hybridnet").dense_rnn_net(args, dtype=dtype, **kw)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[loss])
    return m


def _esz(dtype):
    return 2 if dtype == 0 else 4


def step_gflop(config, slices_per_rank, size):
    """conv GFLOP of one training step on one rank from the table above (conv FLOPs scale with the plane area)"""
    return TRAIN_GFLOP_PER_SLICE[config] * slices_per_rank * (size * size) / float(REF_PLANE[config] ** 2)


def check_step_flops(agg, gflop):
    """the whole-step FLOP figure against the SUM of the per-kernel algorithmic model: `agg` is
    instrumented_step()'s {kernel: [launches, ms, flops, bytes]}.  Returns (sum of the kernels in GFLOP, relative difference, ok)."""
    ksum = sum(v[2] for v in agg.values()) / 1e9
    rel = (ksum - gflop) / gflop
    return ksum, rel, abs(rel) <= FLOPS_TOLERANCE


def _conv_work(d, op, scale):
    """algorithmic work of one conv launch: FLOPs = 2 * M * Cout * taps * Cin (logical channels);
    bytes = every stored input element and every filter element read once, every output element written once (and read
    once more in accumulate mode); the filter gradient reads x and dy once and writes the float32 gradient once"""
    esz = _esz(d.dtype)
    m_in = d.N * d.Di * d.Hi * d.Wi
    m_out = d.N * d.Do * d.Ho * d.Wo
    taps = d.KD * d.KH * d.KW
    fl = 2.0 * m_out * d.Cout * taps * d.Cin * scale.get(d.w, 1.0)
    if op == 1:
        return fl, (m_in * d.Cin + m_out * d.Cout) * esz + d.Cout * taps * d.Cin * 4.0
    nb = (m_in * d.Cin + d.Cout * taps * d.Cin) * esz + m_out * d.Cout * esz * (2 if d.accumulate else 1)
    if d.bnb_u:          # fused BN backward (or its sums alone): the BN input is read as well
        nb += m_out * d.Cout * esz
    return fl, nb


def _row_work(name, a, ctx):
    """algorithmic HBM bytes of the launches of one row-kernel entry point (include/hdu.h argument order), one entry per
    kernel the call launches: every element the operation has to read / write, once.
    None = no model (small per-channel kernels)."""
    v = lambda x: getattr(x, "value", x)
    if name in ("hdu_materialize", "hdu_materialize_stats"):
        dt, N, D, H, W, C = v(a[0]), v(a[3]), v(a[4]), v(a[5]), v(a[6]), v(a[7])
        k = 9 if name == "hdu_materialize_stats" else 10
        ud, uh, uw, skip = v(a[k + 1]), v(a[k + 2]), v(a[k + 3]), a[k + 4]
        m_in, m_out = N * D * H * W, N * (D << ud) * (H << uh) * (W << uw)
        return [(m_in * C + m_out * C * (2 if v(skip) else 1)) * _esz(dt)]
    if name == "hdu_affine_act":
        return [2.0 * v(a[3]) * v(a[4]) * _esz(v(a[0]))]
    if name == "hdu_bn_bwd_fused":       # reduction (dz, x) + apply (dz, x, old gradient in accumulate mode -> dx)
        dt, M, C, acc = v(a[0]), v(a[5]), v(a[6]), v(a[24])
        e = _esz(dt)
        return [2.0 * M * C * e, (3.0 + (1 if acc else 0)) * M * C * e]
    if name == "hdu_bn_bwd_apply_sums":  # the apply half alone (the sums came from the data-gradient epilogue)
        dt, M, C, acc = v(a[0]), v(a[5]), v(a[6]), v(a[24])
        return [(3.0 + (1 if acc else 0)) * M * C * _esz(dt)]
    if name == "hdu_bn_bwd_apply":
        dt, M, C, acc = v(a[0]), v(a[5]), v(a[6]), v(a[16])
        return [(3.0 + (1 if acc else 0)) * M * C * _esz(dt)]
    if name in ("hdu_bn_bwd_reduce_coef", "hdu_bn_bwd_reduce"):
        return [2.0 * v(a[5]) * v(a[6]) * _esz(v(a[0])), None]
    if name in ("hdu_bn_stats", "hdu_bn_stats_fold", "hdu_colsum"):
        return [1.0 * v(a[3]) * v(a[4]) * _esz(v(a[0])), None]
    if name == "hdu_bn_bwd_correct":
        return [3.0 * v(a[3]) * v(a[4]) * _esz(v(a[0]))]
    if name in ("hdu_maxpool3s2_fwd", "hdu_avgpool2_fwd"):
        dt, N, D, H, W, C = (v(a[i]) for i in (0, 3, 4, 5, 6, 7))
        m = N * D * H * W
        return [m * C * _esz(dt) * 1.25]
    if name in ("hdu_maxpool3s2_bwd", "hdu_avgpool2_bwd"):
        off = 1 if name == "hdu_maxpool3s2_bwd" else 0
        dt, N, D, H, W, C = v(a[0]), v(a[3 + off]), v(a[4 + off]), v(a[5 + off]), v(a[6 + off]), v(a[7 + off])
        return [N * D * H * W * C * _esz(dt) * 1.25]
    if name == "hdu_upsample_bwd":
        dt, N, D, H, W, C, ud, uh, uw = (v(a[i]) for i in (0, 3, 4, 5, 6, 7, 8, 9, 10))
        m = N * D * H * W
        return [(m * (1 << (ud + uh + uw)) + m) * C * _esz(dt)]
    if name == "hdu_sgd_nesterov":
        return [5.0 * v(a[3]) * 4]
    if name == "hdu_weight_prep_batched":
        return [sum(cv.kernel.numel for cv in ctx.convs) * 4.0 + ctx.Wc.numel() * ctx.Wc.element_size()]
    if name == "hdu_wce_loss":
        return [v(a[4]) * (2 * 8 * _esz(v(a[0])) + 1.0)]
    if name == "hdu_split3_batched":     # float32 read once, three bf16 planes written (ops.Split3Plan of the float32 split modes)
        sp = ctx._split_plan[0] if getattr(ctx, "_split_plan", None) else None
        return [sum(src.M * src.C * (4.0 + 6.0) for src, _, _, _, _ in sp.items)] if sp is not None else None
    return None


def _is_3d_dense_block_conv(layer):
    """conv_block3d's two convolutions (denseunet3d.py:18-53, hybridnet.py:11-46): 3dconv<stage>_<i>_x1 (1x1x1) / _x2 (3x3x3)"""
    return layer.startswith("3dconv") and (layer.endswith("_x1") or layer.endswith("_x2"))


def step_roofline(agg, dtype):
    """time-weighted roofline fraction of the WHOLE step (a one-kernel roofline says little for a step whose top
    kernel is 11 %): every kernel with an algorithmic-work model is priced against the roof its arithmetic intensity puts it
    under (HBM 8 TB/s or the dense MFMA peak), weighted by its share of the step's kernel time; kernels without a model (small
    per-channel folds / finalizes) count as 0 and their share is reported as `unmodelled_time_share`."""
    ridge = PEAK_TFLOPS[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    tot = sum(v[1] for v in agg.values())
    acc = mf = hb = un = 0.0
    for n, tms, fl, nb in agg.values():
        sec = tms * 1e-3
        hbm = fl == 0.0 or (nb and fl / nb < ridge)
        if hbm and nb:
            acc += tms * min(1.0, nb / sec / 1e9 / PEAK_HBM_GBS); hb += tms
        elif not hbm:
            acc += tms * min(1.0, fl / sec / 1e12 / PEAK_TFLOPS[dtype]); mf += tms
        else:
            un += tms
    return {"time_weighted_frac": round(acc / tot, 4), "mfma_bound_time_share": round(mf / tot, 3),
            "hbm_bound_time_share": round(hb / tot, 3), "unmodelled_time_share": round(un / tot, 3)}


def source_digest():
    """identity of the kernel sources of the running tree (csrc + the C-ABI header): the figures / N = 1 files the line quotes carry
    the digest of the tree that produced them, and a quote from another tree is marked stale"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "h-denseunet_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "hdu.h")]):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _figures_file():
    """the committed figures of the GPU parity tests: this round's, else the previous round's (then every quote is stale by definition)"""
    for rnd in (PROFILE_ROUND, "r05"):
        path = os.path.join(ROOT, "profiles", "%s_bf16_parity_figures.txt" % rnd)
        if os.path.exists(path):
            return path, rnd
    return None, None


def parity_of_timed_mode(config, dtype, batch=None):
    """fidelity of the mode this workload is timed in, from the committed figures of the GPU parity tests (a
    bf16 slices/s is never quoted without it): Dice deficit per class and max abs logit error of the product against the FLOAT32
    oracle (tests/test_gpu_parity_bf16.py, tests/test_gpu_parity.py -> profiles/<round>_bf16_parity_figures.txt).
    Cases are keyed by the bracketed tag of their line, which since round 6 carries the batch / shape ("2d/denseunet/mid @2x512" and
    "... @8x512" are two cases: VERDICT r5 W3a); a tag seen twice keeps its WORST figure.  dtype "f32x3" reads the split-contraction
    line of its own mode and never falls back to the exact mode's."""
    import re
    path, rnd = _figures_file()
    tag = {"2d": "2d/denseunet", "3dpart": "hybrid/3dpart", "end2end": "hybrid/end2end", "shard3d": "3d/3dpart"}[config]
    if path is None:
        return None
    bounds = {"dice": 1e-3, "logits": 1e-4}
    recs, worst, digest = {}, {}, None
    f32rec = gradrec = exact_rec = None
    for ln in open(path):
        if ln.startswith("# source_digest"):
            digest = ln.split()[2]
        elif dtype == "bf16" and ln.startswith("[" + tag + "/") and "north_star tolerances" in ln:
            mm = re.search(r"Dice deficit per class \[([^\]]*)\].*max abs err ([0-9.e+-]+)", ln)
            if mm:
                case = ln[1:ln.index("]")]
                r = {"dtype": "bf16", "vs": "float32 oracle", "case": case,
                     "dice_deficit_per_class": [float(v.strip(" '")) for v in mm.group(1).split(",")],
                     "logit_max_abs_err": float(mm.group(2)), "north_star_bounds": bounds}
                if case not in recs or max(r["dice_deficit_per_class"]) > worst[case]:
                    recs[case] = r
                worst[case] = max(worst.get(case, 0.0), max(r["dice_deficit_per_class"]))
        elif dtype == "f32x3b" and ln.startswith("[f32 forward exact, bf16x3 backward " + tag.split("/", 1)[0]) and tag.split("/", 1)[1] in ln:
            mm = re.search(r"gradients vs float32 oracle rel-L2 worst ([0-9.e+-]+) / median ([0-9.e+-]+).*\(exact mode: ([0-9.e+-]+) / ([0-9.e+-]+)", ln)
            if mm:
                gradrec = {"grad_rel_l2_worst_median": [float(mm.group(1)), float(mm.group(2))],
                           "exact_mode_worst_median": [float(mm.group(3)), float(mm.group(4))]}
        elif dtype in ("f32", "f32x3b") and ln.startswith("[f32 absolute " + tag.split("/", 1)[0]) and tag.split("/", 1)[1] in ln:
            mm = re.search(r"product vs float32 oracle ([0-9.e+-]+).*Dice vs oracle \[([^\]]*)\]", ln)
            if mm:
                exact_rec = {"dtype": "f32", "vs": "float32 oracle", "case": ln[1:ln.index("]")],
                             "dice_deficit_per_class": [round(1.0 - float(v.strip(" '")), 7) for v in mm.group(2).split(",")],
                             "logit_max_abs_err": float(mm.group(1)), "north_star_bounds": bounds}
                if dtype == "f32":
                    f32rec = exact_rec
        elif dtype == "f32x3" and ln.startswith("[f32 storage, bf16x3 contraction " + tag.split("/", 1)[0]) and tag.split("/", 1)[1] in ln:
            mm = re.search(r"product vs float32 oracle ([0-9.e+-]+).*Dice vs oracle \[([^\]]*)\]", ln)
            if mm:
                f32rec = {"dtype": "f32x3", "vs": "float32 oracle", "case": ln[1:ln.index("]")],
                          "dice_deficit_per_class": [round(1.0 - float(v.strip(" '")), 7) for v in mm.group(2).split(",")],
                          "logit_max_abs_err": float(mm.group(1)), "north_star_bounds": bounds}
    rec = f32rec
    if dtype == "f32x3b" and gradrec is not None and exact_rec is not None:
        rec = dict(exact_rec, dtype="f32x3b", forward="exact float32 kernels (predict bit-equal to the f32 mode)", backward=gradrec)
    if dtype == "bf16" and recs:
        # primary figure: the case run at the benchmarked batch / shape -- for the 2D net the mid-training case at the benchmarked batch
        # (else the largest batch on file), for the shard shape its mid case; the hybrids' cases all run 224 x 224 x 12: the reference's
        # full recipe ("trained") is quoted
        def rank(case):
            mid = "/mid" in case
            if config in ("2d", "shard3d"):
                mb = re.search(r"@(\d+)x", case)
                nb = int(mb.group(1)) if mb else 0
                return (mid, nb == batch if batch else False, nb)
            return (not mid, 0, 0)
        rec = dict(recs[max(recs, key=rank)])
        if len(recs) > 1:
            rec["max_dice_deficit_by_case"] = {c: worst[c] for c in recs}
            rec["max_dice_deficit"] = max(worst.values())
    if rec is not None:
        rec["source"] = "profiles/%s_bf16_parity_figures.txt" % rnd
        if digest is None or digest != source_digest():
            rec["stale"] = "figures are of %s" % ("another tree (%s)" % digest if digest else "round %s: no digest on file" % rnd)
    return rec


def instrumented_step(m):
    """ONE eager step with the library's launch profiler armed (include/hdu.h: hdu_profile_*): EVERY kernel the step
    launches is recorded with its own begin-to-end time (start / stop events attached to the dispatch itself -- the figure
    rocprofv3 --kernel-trace reports, no marker-packet overhead to subtract) and its instantiated name; the C-ABI call log
    attaches the algorithmic FLOPs / HBM bytes to each record.
    Returns {kernel_name: [n_launches, total_ms, algorithmic_flops, algorithmic_bytes (or None)]}."""
    lib = importlib.import_module("h-denseunet_amd.lib")
    ctx = m.ctx
    scale, owner = {}, {}
    for cv in ctx.convs:
        ks = cv.kernel.keras_shape
        sc = (ks[-2] / cv.cin_p) * (ks[-1] / cv.cout_p)
        scale[cv.wf_ptr.value] = sc
        owner[cv.wf_ptr.value] = cv.name
        if cv.wd_ptr is not None:
            scale[cv.wd_ptr.value] = sc
            owner[cv.wd_ptr.value] = cv.name
    dense3d = lambda d: _is_3d_dense_block_conv(owner.get(d.w, ""))
    db = [0.0, 0.0, 0]           # north_star's quantity: FLOPs, ms, launches of the 3D dense-block convs (forward + data + filter gradient)
    g = m._graph
    m._graph = None
    # this extra step runs on rank 0 ONLY: it must not enter the gradient all-reduce (the other ranks are already
    # waiting in the final barrier)
    dp = (m._allreduce, m._allreduce_async, m._buckets)
    m._allreduce = m._allreduce_async = m._buckets = None
    _sync()
    lib.profile_begin()
    try:
        m.train_step_resident()
    finally:
        recs, calls = lib.profile_end()
        m._graph = g
        m._allreduce, m._allreduce_async, m._buckets = dp
    _sync()
    work = [None] * len(recs)           # (flops, bytes) per record
    plan = ctx.wgrad_plan
    triples = 1.0
    if plan is None and getattr(ctx, "_split_plan", None) is not None and ctx._split_now:
        # float32 split modes: the filter gradients run as bf16 launches over N' = 3 N images (hi / lo triples) -- the
        # ALGORITHMIC FLOPs are the layer's, a third of what the descriptors of the triples say
        plan, triples = ctx._split_plan[1], 3.0
    for name, args, n0, n1 in calls:
        if name in ("hdu_conv_fprop", "hdu_conv_wgrad", "hdu_conv_dgrad_strided"):
            d = args[0]._obj
            work[n0] = _conv_work(d, 1 if name == "hdu_conv_wgrad" else 0, scale)
            if triples > 1 and name == "hdu_conv_wgrad" and d.dtype == 0 and n1 > n0:
                work[n0] = (work[n0][0] / triples, work[n0][1])
            if dense3d(d):
                db[0] += work[n0][0]; db[1] += sum(r[1] for r in recs[n0:n1]); db[2] += n1 - n0
        elif name == "hdu_wgrad_plan_run" and plan is not None:
            ds = plan.descs[getattr(args[0], "value", args[0])]
            ws = [_conv_work(d, 1, scale) for d in ds]
            ws = [(w[0] / triples, w[1]) for w in ws]
            work[n0] = (sum(w[0] for w in ws), sum(w[1] for w in ws))
            fd = sum(w[0] for d, w in zip(ds, ws) if dense3d(d))
            if fd > 0:             # a batched launch covers many layers: its time is shared out by FLOPs
                db[0] += fd; db[1] += recs[n0][1] * fd / max(work[n0][0], 1.0); db[2] += 1
        else:
            rw = _row_work(name, args, ctx)
            if rw is not None:
                for i, nb in enumerate(rw[:n1 - n0]):
                    if nb is not None:
                        work[n0 + i] = (0.0, nb)
    if os.environ.get("HDU_BENCH_TRACE"):
        # developer view: the step as a launch-ordered list (kernel, us, entry point, shape), gpurun_out/step_trace_<tag>.json
        owner = [None] * len(recs)
        for name, args, n0, n1 in calls:
            v = lambda x: getattr(x, "value", x)
            shape = None
            if name in ("hdu_conv_fprop", "hdu_conv_wgrad", "hdu_conv_dgrad_strided"):
                d = args[0]._obj
                shape = dict(M=d.N * d.Do * d.Ho * d.Wo, Cout=d.Cout, Cin=d.Cin, taps=d.KD * d.KH * d.KW, acc=d.accumulate)
            elif name in ("hdu_materialize", "hdu_materialize_stats"):
                shape = dict(M=v(args[3]) * v(args[4]) * v(args[5]) * v(args[6]), C=v(args[7]))
            elif name in ("hdu_bn_bwd_fused", "hdu_bn_bwd_apply", "hdu_bn_bwd_reduce_coef"):
                shape = dict(M=v(args[5]), C=v(args[6]))
            elif name in ("hdu_bn_stats", "hdu_bn_stats_fold", "hdu_colsum", "hdu_affine_act"):
                shape = dict(M=v(args[3]), C=v(args[4]))
            for i in range(n0, n1):
                owner[i] = (name, shape)
        trace = [dict(k=kn, us=round(ms * 1e3, 2), api=(owner[i] or (None, None))[0], shape=(owner[i] or (None, None))[1])
                 for i, (kn, ms) in enumerate(recs)]
        try:
            with open(os.path.join(ROOT, "gpurun_out", "step_trace_%d.json" % len(DETAILS)), "w") as f:   # 0 = main workload, 1.. = extras
                json.dump(trace, f)
        except OSError:
            pass
    agg = {}
    for (kname, ms), w in zip(recs, work):
        a = agg.setdefault(kname, [0, 0.0, 0.0, 0.0, True])
        a[0] += 1
        a[1] += max(ms, 1e-5)
        if w is None:
            a[4] = False
        else:
            a[2] += w[0]
            a[3] += w[1]
    out = {}
    for k, a in agg.items():
        out[k] = [a[0], a[1], a[2], a[3] if a[4] else None]
    instrumented_step.dense_blocks_3d = tuple(db)
    if os.environ.get("HDU_BENCH_VERBOSE"):
        tot = sum(v[1] for v in out.values())
        for k, v in sorted(out.items(), key=lambda kv: -kv[1][1]):
            print("%-100s launches=%4d ms=%7.3f (%5.2f%%) us/launch=%7.1f" % (k[:100], v[0], v[1], 100 * v[1] / tot, v[1] / v[0] * 1e3),
                  file=sys.stderr)
    return out


def cpu_baseline(config, size, cols, samples=2):
    """the reference's Keras/TF CPU path cannot run here; stand-in = the float32 torch-CPU
    restatement of the same graph (oracle/torch_ref.py), training steps on ONE slice / ONE volume, on the host's
    PHYSICAL cores (torch threads pinned to that count)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import torch_ref as R
    import parity_utils as U
    kind = "2d" if config == "2d" else ("3d" if config == "standalone3d" else "hybrid")
    variant = {"2d": "denseunet", "3dpart": "3dpart", "end2end": "end2end", "standalone3d": "3dpart"}[config]
    b = 1
    phys = physical_cores()
    prev = torch.get_num_threads()
    # Round 5: the thread count is PROBED -- this graph is thousands of small ops, and on the 128-core GPU host a team of all physical
    # cores spends its time in fork / join (tools/oracle_pair_timing.py: 59.5 s with 128 threads, 23.6 s with 64, 15.0 s with 32).  Rounds
    # 1-4 reported the 128-thread figure, i.e. a CPU baseline ~4x slower than the host can do.  One step per candidate count, then the
    # remaining samples at the best one; `cores` = the threads of the reported figure.
    cands = [c for c in (8, 16, 32, 64) if c <= phys] or [phys]
    times = {}
    try:
        x, y = U.synthetic_batch(kind, b, size, cols)
        P = R.ParamStore(seed=4321, dtype=torch.float32, perturb=False)
        fwd = U.oracle_forward_fn(kind, variant, (6, 12, 36, 24), (3, 4, 12, 8))
        torch.set_num_threads(cands[len(cands) // 2])
        with torch.no_grad():
            fwd(P, torch.tensor(x))      # creates the parameters (not timed)
        P.bn_batch_means = {}
        vel = {}

        def one(c):
            torch.set_num_threads(c)
            t1 = time.time()
            R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), vel)
            times.setdefault(c, []).append(time.time() - t1)
        for c in cands:
            one(c)
        best_c = min(times, key=lambda c: min(times[c]))
        for _ in range(max(0, samples - 1)):
            one(best_c)
    finally:
        torch.set_num_threads(prev)
    slices = b if kind == "2d" else cols
    best = min(times[best_c])
    return {"value": round(slices / best, 4), "unit": "slices/s", "cores": best_c, "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "kind": "port",
            "sample": "fwd+bwd+SGD steps of the float32 torch-CPU restatement of the reference graph (not TensorFlow) on %s; "
                      "thread count probed, s/step by threads %s, best used" %
                      ("1x%dx%d" % (size, size) if kind == "2d" else "one %dx%dx%d volume" % (size, size, cols),
                       {c: [round(t, 1) for t in ts] for c, ts in times.items()})}


def pmc_traffic(kernel, config, dtype):
    """HBM bytes per launch of `kernel` from the committed PMC summaries (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes over this same bench command, profiles/): FETCH_SIZE [KB] x2 (gfx950 counts a 128-B request as 64 B,
    MI355X_MICROARCH.md "HBM") + WRITE_SIZE [KB].  None when no summary of this config / kernel is committed."""
    for rnd in (PROFILE_ROUND,):          # (summaries of THIS round's kernels only: an older round's figures belong to other code)
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(ROOT, "profiles", "%s_pmc_%s_%s_%s.txt" % (rnd, ctr, config, dtype))
            if not os.path.exists(path):
                break
            lines = open(path).read().split("\n")
            for i, ln in enumerate(lines):
                if ln.startswith("void " + kernel + "(") and i + 1 < len(lines) and ctr in lines[i + 1]:
                    vals[ctr] = float(lines[i + 1].split()[-1])
        if len(vals) == 2:
            return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), rnd
    return None, None


WORKLOAD_TEXT = {
    "2d": "2D DenseUNet-161 train step, batch %(b)d x %(size)dx%(size)d per GPU (BASELINE configs[1])",
    "3dpart": "denseunet_3d train step, %(size)dx%(size)dx%(cols)d (BASELINE configs[2])",
    "end2end": "dense_rnn_net end2end train step, %(size)dx%(size)dx%(cols)d (BASELINE configs[3])",
    "shard3d": "3D DenseNet train step on ONE %(size)dx%(size)dx%(gcols)d volume, depth-sharded over %(world)d rank(s), "
               "%(cols)d planes each (BASELINE configs[4]: 512^3 over 8 GPUs = 64 planes per GPU)",
}

DETAILS = {}          # per-workload conv-kernel tables: written to gpurun_out/ (and stderr under HDU_BENCH_VERBOSE), not the JSON line


def roofline_record(agg, name, config, dtype):
    """SURVEY.md section 8(d) / DESIGN.md section 3: the kernel with the largest total time in the step (over ALL kernels, row
    kernels included): its algorithmic FLOPs and algorithmic HBM bytes per launch over its own begin-to-end launch time
    (library launch profiler, live).  `bound` is the roof its arithmetic intensity puts it under (ridge = MFMA peak / HBM
    peak; a kernel without FLOPs is HBM-bound); `achieved` / `peak` / `frac` are in that roof's unit; both fractions are
    reported."""
    n, tms, fl, nbytes = agg[name]
    sec = tms * 1e-3
    step_ms = sum(v[1] for v in agg.values())
    peak_tf = PEAK_TFLOPS[dtype]
    tf = fl / sec / 1e12
    gbs = (nbytes / sec / 1e9) if nbytes else None
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    ai = (fl / nbytes) if nbytes else None
    hbm_bound = fl == 0.0 or (ai is not None and ai < ridge)
    traffic, rnd = pmc_traffic(name, config, dtype)
    if hbm_bound:
        ach, peak, unit = (round(gbs, 2) if gbs is not None else None), PEAK_HBM_GBS, "GB/s"
        frac = round(gbs / PEAK_HBM_GBS, 4) if gbs is not None else None
    else:
        ach, peak, unit, frac = round(tf, 2), peak_tf, "TFLOP/s", round(tf / peak_tf, 4)
    r = {"bound": "hbm" if hbm_bound else "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": frac,
         "traffic": traffic, "kernel": name, "launches_per_step": n, "avg_launch_us": round(tms / n * 1e3, 2),
         "share_of_step_kernel_time": round(tms / step_ms, 4),
         "mfma_tflops": round(tf, 2), "mfma_frac": round(tf / peak_tf, 4),
         "hbm_gbs_algorithmic": round(gbs, 1) if gbs is not None else None,
         "hbm_frac": round(gbs / PEAK_HBM_GBS, 4) if gbs is not None else None,
         "flop_per_byte": round(ai, 1) if ai is not None else None,
         "algorithmic_bytes_per_launch": int(nbytes / n) if nbytes else None,
         "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3)}
    if traffic is not None:
        r["traffic_source"] = "profiles/%s_pmc_{FETCH,WRITE}_SIZE_%s_%s.txt (FETCH x2 + WRITE, per launch)" % (rnd, config, dtype)
        r["hbm_gbs_counters"] = round(traffic / (sec / n) / 1e9, 1)
    return r


def top_kernels(agg, dtype, k=3):
    """the k kernels with the largest total time, compact: (name, launches, us per step, share, fraction of its roof)"""
    tot = sum(v[1] for v in agg.values())
    ridge = PEAK_TFLOPS[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    out = []
    for name, (n, tms, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:k]:
        sec = tms * 1e-3
        hbm = fl == 0.0 or (nb and fl / nb < ridge)
        frac = None
        if hbm and nb:
            frac = nb / sec / 1e9 / PEAK_HBM_GBS
        elif not hbm:
            frac = fl / sec / 1e12 / PEAK_TFLOPS[dtype]
        out.append({"kernel": name[:72], "n": n, "us": round(tms * 1e3, 1), "share": round(tms / tot, 3),
                    "bound": "hbm" if hbm else "mfma", "frac": round(frac, 3) if frac is not None else None})
    return out


def run_workload(config, dtype, b, size, cols, steps, warmup, rank, world, use_graph, roofline):
    """dtype "f32x3": the float32 network with the split-bf16 contraction mode switched on for the duration of the workload;
    "f32x3b": exact float32 forward (the parity mode's logits), split contraction in the backward pass only"""
    if dtype not in ("f32x3", "f32x3b"):
        return _run_workload(config, dtype, dtype, b, size, cols, steps, warmup, rank, world, use_graph, roofline)
    lib = importlib.import_module("h-denseunet_amd").lib
    prev = lib.set_f32_contraction("bf16x3" if dtype == "f32x3" else "bf16x3_bwd")
    try:
        return _run_workload(config, dtype, "f32", b, size, cols, steps, warmup, rank, world, use_graph, roofline)
    finally:
        lib.set_f32_contraction(prev)


def _run_workload(config, dtype, store, b, size, cols, steps, warmup, rank, world, use_graph, roofline):
    """build, make resident, (capture,) warm up, time `steps` steps between barriers, MAX over ranks.  Returns the
    record of this workload (rank 0 adds the roofline of its dominant conv kernel)."""
    par = importlib.import_module("h-denseunet_amd.parallel")
    synth = importlib.import_module("h-denseunet_amd.synth")
    gcols = cols
    if config == "shard3d":
        # ONE volume of `cols` depth planes split over the ranks (strong scaling); every rank builds the same phantom
        # and keeps its own planes.  world > 1: no hipGraph, the step contains the neighbour exchanges; world 1 (the
        # per-shard shape on one GPU) has no exchange and is captured like every other workload.
        assert cols % (4 * world) == 0, "--cols must be a multiple of 4*world"
        gcols, cols = cols, cols // world
        if world > 1:
            use_graph = False
    if torch.cuda.is_available():
        torch.cuda.reset_peak_memory_stats()
    m = build(config, store, b, size, cols)
    if (world > 1 or os.environ.get("HDU_FORCE_DP") == "1") and config != "shard3d":
        par.attach_data_parallel(m)
    kind = "2d" if config == "2d" else "hybrid"
    if config == "shard3d":
        xv, yv = synth.synthetic_batch("hybrid", 1, size, gcols, seed=1234)
        rng = np.random.default_rng(99)
        xv = np.concatenate([xv, rng.normal(0, 60, xv.shape[:4] + (3,)).astype(np.float32)], -1)
        x, y = xv[:, :, :, rank * cols:(rank + 1) * cols], yv[:, :, :, rank * cols:(rank + 1) * cols]
    else:
        x, y = synth.synthetic_batch(kind, b, size, cols, seed=1234 + rank)
    m._upload_x(x)
    m.loss_layer.set_labels(m._labels_internal(y))
    _sync()

    if use_graph:
        m.capture_graph(warmup=1)
    for _ in range(warmup):
        m.train_step_resident()

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        _sync()

    shard_mod = importlib.import_module("h-denseunet_amd.shard")
    shard_mod.reset_counts()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step_resident()
    barrier()
    dt = time.perf_counter() - t0
    coll = shard_mod.counts()
    if dist_on:
        t = torch.tensor([dt], device="cpu" if DRYRUN else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / steps * 1e3
    slices_per_step = (b if kind == "2d" else cols) * world   # shard3d: cols is per rank -> the whole volume
    loss = m.loss_value()
    gflop = step_gflop(config, slices_per_step / world, size)
    rec = {
        "workload": WORKLOAD_TEXT[config] % dict(b=b, size=size, cols=cols or 0, gcols=gcols or 0, world=world),
        "value": round(slices_per_step / (ms / 1e3), 2), "unit": "slices/s", "ms_per_step": round(ms, 3),
        "steps": steps, "warmup": warmup, "dtype": dtype, "global_batch_slices": slices_per_step,
        "hipgraph": bool(use_graph), "loss": round(loss, 5),
        "step_conv_tflops": round(gflop / ms, 2),
        "step_frac_of_mfma_peak": round(gflop / ms / PEAK_TFLOPS[dtype], 4),
    }
    rec["parallelism"] = ("depth-shard%d" if config == "shard3d" else "dp%d") % world
    if config == "shard3d" and world > 1:
        # what one depth-sharded step exchanges (h-denseunet_amd/shard.py COUNTS over the timed steps): all of it is EXPOSED today --
        # issued in program order, nothing computes under it
        rec["collectives_per_step"] = {"allreduce": coll["allreduce"] // steps, "allreduce_kb": coll["allreduce_bytes"] // steps // 1024,
                                       "neighbour_exchange": coll["neighbour_exchange"] // steps,
                                       "neighbour_mb": round(coll["neighbour_bytes"] / steps / 2.0 ** 20, 2), "hidden": 0}
    par_rec = parity_of_timed_mode(config, dtype, b)
    if par_rec is not None:
        rec["parity"] = par_rec
    if config == "shard3d" and world > 1 and gcols == 512 and size == 512:
        # strong scaling of BASELINE configs[4] against the SAME volume on ONE GPU (profiles/: measured this round, world 1)
        for rnd in (PROFILE_ROUND, "r05"):
            ref = os.path.join(ROOT, "profiles", "%s_full_512cubed_world1.json" % rnd)
            if os.path.exists(ref):
                n1 = json.load(open(ref))
                rec["strong_scaling"] = {"n1_ms_per_step": n1["ms_per_step"], "speedup_vs_n1": round(n1["ms_per_step"] / ms, 3),
                                         "efficiency": round(n1["ms_per_step"] / ms / world, 3), "n1_source": "profiles/%s_full_512cubed_world1.json" % rnd}
                if (n1.get("source_digest") or n1.get("config", {}).get("source_digest")) != source_digest():     # (an N = 1 time of another tree's kernels is marked, not hidden)
                    rec["strong_scaling"]["n1_stale"] = True
                break
    if torch.cuda.is_available():
        rec["peak_hbm_gib"] = round(torch.cuda.max_memory_allocated() / 2.0 ** 30, 2)
    # the instrumented step is rank-0-only and must not enter a collective: the depth-sharded step always does
    # (halo exchange, sync-BN), so it is skipped there when world > 1
    if rank == 0 and roofline and not (config == "shard3d" and world > 1):
        agg = instrumented_step(m)
        # the table's whole-step FLOPs must be the sum of the per-kernel model; if they ever part by more than 3 %
        # the line quotes the SUM OF THE KERNELS and says so -- a wrong denominator is never printed silently again
        ksum, rel, ok = check_step_flops(agg, gflop)
        rec["flops_check"] = {"table_gflop": round(gflop, 1), "sum_of_kernels_gflop": round(ksum, 1), "rel_diff": round(rel, 4), "ok": ok}
        if not ok:
            print("bench.py: step FLOPs of %s: table %.1f GFLOP vs sum of kernels %.1f GFLOP (%.1f %%) -- quoting the sum of kernels"
                  % (config, gflop, ksum, 100 * rel), file=sys.stderr)
            rec["step_conv_tflops"] = round(ksum / ms, 2)
            rec["step_frac_of_mfma_peak"] = round(ksum / ms / PEAK_TFLOPS[dtype], 4)
        name = max(agg.items(), key=lambda kv: kv[1][1])[0]       # largest total time over ALL kernels of the step
        rec["roofline"] = roofline_record(agg, name, config, dtype)
        rec["top_kernels"] = top_kernels(agg, dtype)
        rec["step_roofline"] = step_roofline(agg, dtype)
        fl3, ms3, n3 = instrumented_step.dense_blocks_3d
        if n3 and ms3 > 0:       # north_star: "MFMA roofline on the 3D dense-block fwd+bwd" -- the convs of conv_block3d, all three passes
            rec["dense_blocks_3d"] = {"conv_launches": n3, "ms": round(ms3, 3), "gflop": round(fl3 / 1e9, 1),
                                      "mfma_frac": round(fl3 / (ms3 * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4),
                                      "note": "convolutions of conv_block3d (1x1x1 + 3x3x3; forward, data gradient, filter gradient) only"}
        kms = sum(v[1] for v in agg.values())
        rec["step_kernel_ms_eager_profiled"] = round(kms, 3)
        rec["launches_per_step"] = sum(v[0] for v in agg.values())
        DETAILS["%s:%s" % (config, dtype)] = {
            k: {"launches": v[0], "ms": round(v[1], 4), "us_per_launch": round(v[1] / v[0] * 1e3, 2),
                "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                "alg_gbs": round(v[3] / (v[1] * 1e-3) / 1e9, 1) if v[3] else None}
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    del m
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return rec


def compact(rec):
    """what the ONE JSON line carries per extra workload (the driver keeps 8 KB of stdout: the whole metric must fit)"""
    out = {k: rec[k] for k in ("workload", "value", "ms_per_step", "steps", "dtype", "global_batch_slices",
                               "step_frac_of_mfma_peak", "peak_hbm_gib", "error", "parallelism", "strong_scaling",
                               "collectives_per_step") if k in rec}      # (unit: slices/s, hipGraph: as the main workload)
    if "step_roofline" in rec:
        out["step_roofline_frac"] = rec["step_roofline"]["time_weighted_frac"]
    if "dense_blocks_3d" in rec:
        out["dense_blocks_3d_mfma_frac"] = rec["dense_blocks_3d"]["mfma_frac"]
    if "parity" in rec:
        out["parity"] = {k: rec["parity"][k] for k in ("dtype", "case", "dice_deficit_per_class", "logit_max_abs_err", "max_dice_deficit",
                                                        "source", "stale", "backward") if k in rec["parity"]}
        out["parity"]["source"] = out["parity"]["source"].split("/")[-1].split("_")[0]      # the round tag: profiles/<round>_bf16_parity_figures.txt
        if "stale" in out["parity"]:
            out["parity"]["stale"] = True
    if "flops_check" in rec:
        out["flops_check"] = {k: rec["flops_check"][k] for k in ("rel_diff", "ok")}
    out["workload"] = rec["workload"][:80]
    if "roofline" in rec and rec.get("dtype") == "bf16":      # (the float32 modes' dominant kernels: gpurun_out/bench_details.json; the line has 8 KB)
        r = rec["roofline"]
        out["roofline"] = {k: r[k] for k in ("bound", "achieved", "unit", "frac", "traffic", "kernel", "launches_per_step",
                                             "avg_launch_us", "share_of_step_kernel_time")}
        out["roofline"]["kernel"] = out["roofline"]["kernel"][:64]
    if "cpu_baseline" in rec:
        c = rec["cpu_baseline"]
        out["cpu_baseline"] = {k: c[k] for k in ("value", "cores", "kind", "note") if k in c}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="2d", choices=["2d", "3dpart", "end2end", "shard3d"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f32x3", "f32x3b"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--extras", default=None,
                    help="comma list of extra workloads timed after the main one (config[:dtype]); default for the "
                         "default 2d/bf16 run: 3dpart,end2end,shard3d,2d:f32,2d:f32x3b (shard3d = the 512x512x64 per-GPU shard of "
                         "BASELINE configs[4], single GPU only); 'none' disables")
    a = ap.parse_args()

    pkg = importlib.import_module("h-denseunet_amd")
    if DRYRUN:
        _dryrun_bind()
        a.no_graph, a.no_cpu_baseline = True, True
    else:
        pkg.lib.load()   # gfx950 library or a loud failure: there is no CPU fallback
    par = importlib.import_module("h-denseunet_amd.parallel")
    rank, world = par.init_process_group_from_env("gloo" if DRYRUN else "nccl")
    if world != a.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
    if world == 1 and not DRYRUN:
        torch.cuda.set_device(0)
    b = a.batch or (8 if a.config == "2d" else 1)
    size = a.size or (512 if a.config in ("2d", "shard3d") else 224)
    cols = (a.cols or (64 * world if a.config == "shard3d" else 12)) if a.config != "2d" else None

    main_rec = run_workload(a.config, a.dtype, b, size, cols, a.steps, a.warmup, rank, world, not a.no_graph,
                            not a.no_roofline)
    extras = a.extras
    if extras is None:
        default_run = a.config == "2d" and a.dtype == "bf16" and a.batch is None and a.size is None
        extras = "none"
        if default_run and not DRYRUN:
            # the 512x512x64 shard shape runs where a whole 512^3 volume cannot (one GPU); under N > 1 the driver's
            # weak-scaling run keeps to the data-parallel workloads
            # (round 6: the tolerance-meeting mode -- exact float32 forward, split backward -- on every BASELINE workload)
            extras = ("3dpart,end2end,shard3d,2d:f32,2d:f32x3b,3dpart:f32x3b,end2end:f32x3b" if world == 1
                      else "3dpart,end2end,2d:f32,2d:f32x3b")
    extra_recs = []
    if extras != "none":
        for spec in extras.split(","):
            cfg, _, dt = spec.partition(":")
            dt = dt or "bf16"
            # the 3D half of the metric at the shape BASELINE names; fewer steps for the slow workloads
            if cfg == "2d":
                e_b, e_size, e_cols = b, size, None
            elif cfg == "shard3d":
                e_b, e_size, e_cols = 1, 64 if DRYRUN else 512, (8 if DRYRUN else 64) * world
            else:
                e_b, e_size, e_cols = 1, 32 if DRYRUN else 224, 8 if DRYRUN else 12
            cap = {"shard3d": 10}.get(cfg, 30 if dt == "bf16" else 10)       # every workload is timed over >= 10 steps
            e_steps = max(2, min(a.steps, cap))
            e_warm = max(1, min(a.warmup, 3 if (dt == "bf16" and cfg != "shard3d") else 1))
            extra_recs.append(run_workload(cfg, dt, e_b, e_size, e_cols, e_steps, e_warm, rank, world, not a.no_graph,
                                           not a.no_roofline))

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    emitted = [False]

    def emit():
        """rank 0 prints THE line (once)"""
        if emitted[0]:
            return
        emitted[0] = True
        out = {
            "metric": "CT slices/sec fwd+bwd (%s)" % ("2D 512^2" if a.config == "2d" else
                                                       ("3D %dx%dx%d depth-sharded" % (size, size, cols) if a.config == "shard3d"
                                                        else "3D 224x224x12")),
            "value": main_rec["value"], "unit": "slices/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_rec["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if a.config == "shard3d" else "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic CT phantom (seeded), random-init weights, dropout on" + (" -- CPU DRY RUN, not a measurement" if DRYRUN else ""),
            "config": {"workload": main_rec["workload"], "global_batch_slices": main_rec["global_batch_slices"],
                       "parallelism": ("depth-shard%d" if a.config == "shard3d" else "dp%d") % world,
                       "hipgraph": main_rec["hipgraph"], "loss": main_rec["loss"],
                       "step_conv_tflops": main_rec["step_conv_tflops"],
                       "step_frac_of_mfma_peak": main_rec["step_frac_of_mfma_peak"]},
        }
        if "peak_hbm_gib" in main_rec:
            out["config"]["peak_hbm_gib"] = main_rec["peak_hbm_gib"]
        out["config"]["source_digest"] = source_digest()      # which kernel sources produced this line (quotes from another tree are marked stale)
        if dist_on:     # one process per GPU over RCCL (torch.distributed backend "nccl" IS RCCL on ROCm)
            out["config"]["collectives"] = {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                                            "data_plane": os.environ.get("HDU_COMM", "torch.distributed")}
        if rank == 0:
            if "roofline" in main_rec:
                out["roofline"] = main_rec["roofline"]
                out["config"]["top_kernels"] = main_rec.get("top_kernels")
                out["config"]["launches_per_step"] = main_rec.get("launches_per_step")
                out["config"]["step_roofline"] = main_rec.get("step_roofline")
                if "dense_blocks_3d" in main_rec:
                    out["config"]["dense_blocks_3d"] = main_rec["dense_blocks_3d"]
            if "parity" in main_rec:
                out["parity"] = main_rec["parity"]
            if not a.no_cpu_baseline and world == 1 and a.config != "shard3d":
                out["cpu_baseline"] = cpu_baseline(a.config, size, cols)
                cpu_done = {}
                for r in extra_recs:              # EVERY extra carries a CPU baseline
                    if "error" in r:
                        continue
                    hyb = "3dpart" if r["workload"].startswith("denseunet_3d") else ("end2end" if r["workload"].startswith("dense_rnn_net") else None)
                    if hyb is not None:            # the 3D half beside its own CPU baseline (timed once per workload, whatever the mode)
                        if hyb not in cpu_done:
                            cpu_done[hyb] = cpu_baseline(hyb, 224, 12, samples=1)
                            r["cpu_baseline"] = cpu_done[hyb]
                        else:
                            r["cpu_baseline"] = dict(cpu_done[hyb], note="same workload as the bf16 extra")
                    elif r["workload"].startswith("3D DenseNet"):
                        # config 5's CPU side is not run (78 TFLOP forward per volume); the stand-alone 3D net is
                        # timed on ONE 224 x 224 x 12 volume and the per-slice rate scaled by the plane area (conv FLOPs per slice)
                        c = cpu_baseline("standalone3d", 224, 12, samples=1)
                        c["value"] = round(c["value"] * (224.0 * 224.0) / (512.0 * 512.0), 4)
                        c["note"] = "extrapolated: 224x224x12 sample x (224/512)^2"
                        c["sample"] += "; value = that per-slice rate x (224/512)^2 (BASELINE.md section 3: config 5 is extrapolated)"
                        r["cpu_baseline"] = c
                    elif r["workload"].startswith("2D DenseUNet") and a.config == "2d":
                        c = dict(out["cpu_baseline"])       # the same workload in another storage / contraction mode: the same CPU figure
                        c["note"] = "same workload as the main line"
                        r["cpu_baseline"] = c
            if extra_recs:
                out["config"]["extra_workloads"] = [compact(r) for r in extra_recs]
            line = json.dumps(out)
            # the driver keeps 8 KB of stdout: never let the line outgrow it.  Dropped in this order, only if needed: the top-kernel
            # table, the extras' step counts / parity case names, and last the extras' CPU baselines.
            for drop in ("top_kernels", "extras_small", "extras_cpu"):
                if len(line) <= 7600:
                    break
                if drop == "top_kernels":
                    out["config"].pop("top_kernels", None)
                for e in out["config"].get("extra_workloads", []):
                    if drop == "extras_small":
                        e.pop("steps", None)
                        e.get("parity", {}).pop("case", None)
                        e.get("cpu_baseline", {}).pop("note", None)
                    elif drop == "extras_cpu":
                        e.pop("cpu_baseline", None)
                line = json.dumps(out)
            # full per-kernel tables and uncompacted records: scratch file (copied to profiles/ for the judged runs) + stderr
            detail = {"main": main_rec, "extras": extra_recs, "conv_kernels": DETAILS}
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w") as f:
                    json.dump(detail, f, indent=1)
            except OSError:
                pass
            if os.environ.get("HDU_BENCH_VERBOSE"):
                print(json.dumps(detail), file=sys.stderr)
            print(line)

    # Depth sharding under N > 1 (BASELINE configs[4]: ONE 512 x 512 x 512 volume over the ranks, strong scaling) rides along
    # with the data-parallel line when the driver launches `bench.py --gpus N`.  It is the one workload whose step contains
    # neighbour exchanges, and this path has only ever run multi-rank over gloo (tests/test_depth_shard_gloo.py): a watchdog
    # bounds it, so that a stuck collective costs this extra record, never the line.
    s3 = os.environ.get("HDU_BENCH_SHARD3D", "1")             # "0" = never, "force" = also outside the default run (tests)
    s3_size, s3_cols = (32, 8 * world) if DRYRUN else (512, 512)
    if (world > 1 and a.config == "2d" and s3_cols % (4 * world) == 0 and
            (s3 == "force" or (s3 == "1" and a.extras is None and not DRYRUN and a.dtype == "bf16"))):
        import threading
        limit = float(os.environ.get("HDU_BENCH_SHARD3D_TIMEOUT", "240"))

        def expire():
            extra_recs.append({"workload": WORKLOAD_TEXT["shard3d"] % dict(b=1, size=s3_size, cols=s3_cols // world, gcols=s3_cols, world=world),
                               "error": "no result within %g s (watchdog): the depth-sharded step did not complete" % limit})
            try:
                emit()
                sys.stdout.flush()
            finally:
                os._exit(0)
        wd = threading.Timer(limit, expire)
        wd.daemon = True
        wd.start()
        try:
            extra_recs.append(run_workload("shard3d", a.dtype, 1, s3_size, s3_cols, max(2, min(a.steps, 10)), 1, rank, world, False, False))
        except Exception as e:      # noqa: BLE001 -- whatever it is, the data-parallel line must still be printed
            extra_recs.append({"workload": "shard3d 512x512x512 over %d ranks" % world, "error": ("%s: %s" % (type(e).__name__, e))[:300]})
        wd.cancel()
    emit()
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
