#!/usr/bin/env python
"""bench.py -- CT slices/s of one fwd+bwd+SGD step of the H-DenseUNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2d|3dpart|end2end] [--dtype bf16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N=1 workload (BASELINE.json configs[1], the configuration the metric is quoted on): 2D DenseUNet-161 training step,
batch 8 x 512 x 512, bf16 storage / f32 accumulate, synthetic CT phantom, random-init weights, dropout ON.  Under
N>1 every rank runs the same per-GPU batch (weak scaling) and gradients are summed with one flat RCCL all-reduce.
Inputs and labels are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# conv FLOPs (2*MAC) per slice, fwd + dgrad + wgrad (stem dgrad skipped)
TRAIN_GFLOP_PER_SLICE = {"2d": 580.5, "3dpart": 156.6, "end2end": 227.3, "shard3d": 121.3}
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks


# HDU_BENCH_DRYRUN=1 (tests/test_bench_flow_gloo.py only): the same control flow on CPU -- x86 emulator build of the
# kernels, gloo instead of RCCL, a reduced-depth net -- so that the multi-rank sequence of collectives of this script is
# checked without a multi-GPU node.  Never a measurement; the JSON line says so.
DRYRUN = os.environ.get("HDU_BENCH_DRYRUN") == "1"


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


def build(config, dtype, b, size, cols):
    args = argparse.Namespace(b=b, input_size=size, input_cols=cols)
    ka = importlib.import_module("h-denseunet_amd.keras_api")
    if config == "2d":
        kw = {"nb_layers": (2, 2, 2, 2)} if DRYRUN else {}
        m = importlib.import_module("h-denseunet_amd.denseunet").DenseUNet(reduction=0.5, args=args, dtype=dtype, **kw)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy_2ddense
    elif config == "3dpart":
        m = importlib.import_module("h-denseunet_amd.denseunet3d").denseunet_3d(args, dtype=dtype)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    elif config == "shard3d":
        par = importlib.import_module("h-denseunet_amd.parallel")
        sh = par.depth_shard_info("gloo" if DRYRUN else "nccl")
        m = importlib.import_module("h-denseunet_amd.densenet3d_sharded").dense_net3d(args, dtype=dtype, shard=sh)
        par.attach_depth_shard(m)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    else:
        m = importlib.import_module("h-denseunet_amd.hybridnet").dense_rnn_net(args, dtype=dtype)
        loss = importlib.import_module("h-denseunet_amd.loss").weighted_crossentropy
    m.compile(optimizer=ka.SGD(lr=1e-3, momentum=0.9, nesterov=True), loss=[loss])
    return m


def instrumented_step(m):
    """one eager step with a HIP-event pair around every conv launch (events on the launch stream = torch's current
    stream); returns {kernel_name: [n_launches, total_ms, total_algorithmic_flops]}"""
    ops = importlib.import_module("h-denseunet_amd.ops")
    eng = importlib.import_module("h-denseunet_amd.engine")
    ctx = m.ctx
    scale = {}
    for cv in ctx.convs:
        ks = cv.kernel.keras_shape
        s = (ks[-2] / cv.cin_p) * (ks[-1] / cv.cout_p)
        scale[cv.wf_ptr.value] = s
        if cv.wd_ptr is not None:
            scale[cv.wd_ptr.value] = s
    recs = []
    orig_f, orig_w = ops.conv_fprop, ops.conv_wgrad

    def flops(d, op):
        m_out = d.N * d.Do * d.Ho * d.Wo
        return 2.0 * m_out * d.Cout * d.KD * d.KH * d.KW * d.Cin * scale.get(d.w, 1.0)

    def wrap(fn, op):
        def f(d, *a):
            Ev = torch.cuda.Event if torch.cuda.is_available() else _HostEvent
            e0, e1 = Ev(enable_timing=True), Ev(enable_timing=True)
            e0.record()
            fn(d, *a)
            e1.record()
            recs.append((ops.conv_kernel_name(d, op), flops(d, op), e0, e1, d.N * d.Do * d.Ho * d.Wo))
        return f

    ops.conv_fprop, ops.conv_wgrad = wrap(orig_f, 0), wrap(orig_w, 1)
    batched = ctx.wgrad_plan is not None
    try:
        g = m._graph
        m._graph = None
        if batched:
            ctx.set_batch_wgrad(False)     # per-layer filter-gradient launches so that each one can be timed
        # this extra step runs on rank 0 ONLY: it must not enter the gradient all-reduce (the other ranks are already
        # waiting in the final barrier)
        dp = (m._allreduce, m._allreduce_async, m._buckets)
        m._allreduce = m._allreduce_async = m._buckets = None
        try:
            m.train_step_resident()
        finally:
            m._graph = g
            m._allreduce, m._allreduce_async, m._buckets = dp
            if batched:
                ctx.set_batch_wgrad(True)
        _sync()
    finally:
        ops.conv_fprop, ops.conv_wgrad = orig_f, orig_w
    agg = {}
    bym = {}
    for name, fl, e0, e1, mm in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
        b = bym.setdefault((mm, name), [0, 0.0, 0.0])
        b[0] += 1
        b[1] += e0.elapsed_time(e1)
        b[2] += fl
    if os.environ.get("HDU_BENCH_VERBOSE"):
        for mm in sorted(bym):
            b = bym[mm]
            print("M=%8d %-44s launches=%4d ms=%7.3f us/launch=%7.1f TF=%6.1f" %
                  (mm[0], mm[1], b[0], b[1], b[1] / b[0] * 1e3, b[2] / (b[1] * 1e-3) / 1e12), file=sys.stderr)
    return agg


def cpu_baseline(config, size, cols):
    """the reference's Keras/TF CPU path cannot run here; stand-in = the float32 torch-CPU
    restatement of the same graph (oracle/torch_ref.py), ONE training step on ONE slice / ONE volume."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import torch_ref as R
    import parity_utils as U
    kind = "2d" if config == "2d" else "hybrid"
    variant = {"2d": "denseunet", "3dpart": "3dpart", "end2end": "end2end"}[config]
    b = 1
    x, y = U.synthetic_batch(kind, b, size, cols)
    P = R.ParamStore(seed=4321, dtype=torch.float32, perturb=False)
    fwd = U.oracle_forward_fn(kind, variant, (6, 12, 36, 24), (3, 4, 12, 8))
    t0 = time.time()
    R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), {})
    t1 = time.time()   # includes parameter creation + one plain forward
    R.train_step(P, fwd, U.loss_fn_for(kind), torch.tensor(x), torch.tensor(y), {})
    t2 = time.time()
    slices = b if kind == "2d" else cols
    return {"value": round(slices / (t2 - t1), 4), "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 training step (fwd+bwd+SGD) of the float32 torch-CPU restatement of the reference graph "
                      "(not TensorFlow) on %s, %.1f s" % ("1x%dx%d" % (size, size) if kind == "2d" else
                                                        "one %dx%dx%d volume" % (size, size, cols), t2 - t1)}


def pmc_traffic(kernel, config, dtype):
    """HBM bytes per launch of `kernel` from the committed PMC summaries (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes over this same bench command, profiles/): FETCH_SIZE [KB] x2 (gfx950 counts a 128-B request as 64 B,
    MI355X_MICROARCH.md "HBM") + WRITE_SIZE [KB].  None when no summary of this config / kernel is committed."""
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_%s_%s_%s.txt" % (ctr, config, dtype))
        if not os.path.exists(path):
            return None
        lines = open(path).read().split("\n")
        for i, ln in enumerate(lines):
            if ln.startswith("void " + kernel + "(") and i + 1 < len(lines) and ctr in lines[i + 1]:
                vals[ctr] = float(lines[i + 1].split()[-1])
    if len(vals) != 2:
        return None
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2d", choices=["2d", "3dpart", "end2end", "shard3d"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--cols", type=int, default=12)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    pkg = importlib.import_module("h-denseunet_amd")
    if DRYRUN:
        pkg.lib.use_emulator_for_tests()
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
    size = a.size or (512 if a.config == "2d" else 224)
    cols = a.cols if a.config != "2d" else None

    if a.config == "shard3d":
        # ONE volume of --cols depth planes split over the ranks (strong scaling); every rank builds the same phantom
        # and keeps its own planes.  No hipGraph: the step contains the neighbour exchanges.
        assert cols % (4 * world) == 0, "--cols must be a multiple of 4*world"
        gcols, cols = cols, cols // world
        a.no_graph = True
    m = build(a.config, a.dtype, b, size, cols)
    if (world > 1 or os.environ.get("HDU_FORCE_DP") == "1") and a.config != "shard3d":
        par.attach_data_parallel(m)
    synth = importlib.import_module("h-denseunet_amd.synth")
    kind = "2d" if a.config == "2d" else "hybrid"
    if a.config == "shard3d":
        xv, yv = synth.synthetic_batch("hybrid", 1, size, gcols, seed=1234)
        rng = np.random.default_rng(99)
        xv = np.concatenate([xv, rng.normal(0, 60, xv.shape[:4] + (3,)).astype(np.float32)], -1)
        x, y = xv[:, :, :, rank * cols:(rank + 1) * cols], yv[:, :, :, rank * cols:(rank + 1) * cols]
    else:
        x, y = synth.synthetic_batch(kind, b, size, cols, seed=1234 + rank)
    m._upload_x(x)
    m.loss_layer.set_labels(m._labels_internal(y))
    _sync()

    if not a.no_graph:
        m.capture_graph(warmup=1)
    for _ in range(a.warmup):
        m.train_step_resident()

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        _sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.train_step_resident()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device="cpu" if DRYRUN else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / a.steps * 1e3
    slices_per_step = (b if kind == "2d" else cols) * world   # shard3d: cols is per rank -> the whole volume
    value = slices_per_step / (ms / 1e3)
    loss = m.loss_value()

    out = {
        "metric": "CT slices/sec fwd+bwd (%s)" % ("2D 512^2" if a.config == "2d" else
                                                   ("3D %dx%dx%d depth-sharded" % (size, size, cols * world) if a.config == "shard3d"
                                                    else "3D 224x224x12")),
        "value": round(value, 2), "unit": "slices/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "strong" if a.config == "shard3d" else "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic CT phantom (seeded), random-init weights, dropout on" + (" -- CPU DRY RUN, not a measurement" if DRYRUN else ""),
        "config": {"workload": {"2d": "2D DenseUNet-161 train step, batch %d x %dx%d per GPU (BASELINE configs[1])" % (b, size, size),
                                "3dpart": "denseunet_3d train step, %dx%dx%d (BASELINE configs[2])" % (size, size, cols or 0),
                                "end2end": "dense_rnn_net end2end train step, %dx%dx%d (BASELINE configs[3])" % (size, size, cols or 0),
                                "shard3d": "3D DenseNet train step on ONE %dx%dx%d volume, depth-sharded (BASELINE configs[4] shape family)" % (size, size, (cols or 0) * world)}[a.config],
                   "global_batch_slices": slices_per_step, "parallelism": "dp%d" % world, "hipgraph": not a.no_graph,
                   "loss": round(loss, 5),
                   "step_conv_tflops": round(TRAIN_GFLOP_PER_SLICE[a.config] * slices_per_step / world / ms, 2)},
    }
    if rank == 0:
        if not a.no_roofline:
            agg = instrumented_step(m)
            name, (n, tms, fl) = max(agg.items(), key=lambda kv: kv[1][1])
            ach = fl / (tms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[a.dtype]
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": pmc_traffic(name, a.config, a.dtype),
                               "traffic_unit": "HBM bytes per launch, PMC (profiles/r01_pmc_*), avg over the layer shapes",
                               "kernel": name, "launches_per_step": n,
                               "avg_launch_ms": round(tms / n, 4),
                               "all_conv_kernels": {k: {"launches": v[0], "ms": round(v[1], 3),
                                                        "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                                                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
        if not a.no_cpu_baseline and world == 1 and a.config != "shard3d":
            out["cpu_baseline"] = cpu_baseline(a.config, size, cols)
        print(json.dumps(out))
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
