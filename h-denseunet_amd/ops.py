"""Thin Python wrappers over the C-ABI (include/hdu.h).  torch tensors are storage only: every wrapper
hands raw device pointers + sizes to libhdu.so and launches on torch's current stream."""
import ctypes

import torch

from . import lib as _l
from .lib import HDU_BF16, HDU_F32, ConvDesc, check

_TORCH_DT = {HDU_BF16: torch.bfloat16, HDU_F32: torch.float32}
CHUNK = {HDU_BF16: 8, HDU_F32: 4}


def device():
    """Storage device of the bound library: cuda for the product, cpu only under the test emulator."""
    if _l.is_emulator():
        return torch.device("cpu")
    if not torch.cuda.is_available():
        raise _l.HduError("libhdu.so (gfx950) is bound but no GPU is visible; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    if _l.is_emulator():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cpad(c, dtype):
    ch = CHUNK[dtype]
    return (c + ch - 1) // ch * ch


def fptr(t):
    """pointer of a float32 tensor (or None)"""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


class Act:
    """Channels-last activation view [N][D][H][W][C] with pixel stride `ld` inside a flat buffer."""
    __slots__ = ("buf", "off", "N", "D", "H", "W", "C", "ld", "dtype")

    def __init__(self, buf, off, N, D, H, W, C, ld, dtype):
        self.buf, self.off, self.N, self.D, self.H, self.W, self.C, self.ld, self.dtype = buf, off, N, D, H, W, C, ld, dtype

    @staticmethod
    def alloc(N, D, H, W, C, dtype, ld=None, zero=False):
        ld = ld or C
        n = N * D * H * W * ld
        buf = (torch.zeros if zero else torch.empty)(n, dtype=_TORCH_DT[dtype], device=device())
        return Act(buf, 0, N, D, H, W, C, ld, dtype)

    @property
    def M(self):
        return self.N * self.D * self.H * self.W

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + self.off * self.buf.element_size())

    def slab(self, c0, C):
        assert c0 % CHUNK[self.dtype] == 0 and c0 + C <= self.ld
        return Act(self.buf, self.off + c0, self.N, self.D, self.H, self.W, C, self.ld, self.dtype)

    def like(self, C=None, zero=False):
        return Act.alloc(self.N, self.D, self.H, self.W, C or self.C, self.dtype, zero=zero)

    def rows(self, m0, m1):
        """row sub-range (whole leading-dimension slices only make sense for N/D splits)"""
        a = Act(self.buf, self.off + m0 * self.ld, 1, 1, 1, m1 - m0, self.C, self.ld, self.dtype)
        return a

    def to_torch(self):
        """float32 [N,D,H,W,C] copy (host-side helper for tests / boundary)"""
        full = self.buf[self.off:self.off + (self.M - 1) * self.ld + self.C] if self.M else self.buf[:0]
        idx = torch.arange(self.M, device=self.buf.device)[:, None] * self.ld + torch.arange(self.C, device=self.buf.device)[None, :]
        return full[idx].float().reshape(self.N, self.D, self.H, self.W, self.C)

    def from_torch(self, t):
        t = t.reshape(self.M, self.C).to(self.buf.dtype).to(self.buf.device)
        idx = torch.arange(self.M, device=self.buf.device)[:, None] * self.ld + torch.arange(self.C, device=self.buf.device)[None, :]
        self.buf[self.off:][idx.reshape(-1)] = t.reshape(-1)
        return self


# split-K scratch of hdu_conv_fprop (include/hdu.h): one float32 buffer + ticket counters per process, shared by every
# launch.  SINGLE-STREAM CONTRACT: every hdu_conv_fprop launch of the process must be ordered after the previous one
# (one stream, or streams joined by events as torch's graph capture does) -- two concurrent launches would race on the
# partial tiles and tickets.  The engine issues all convs of a model on torch's current stream; an overlap scheme that
# puts convs on a second stream has to give that stream its own scratch (conv_desc(..., splitk=(ws, counters))).
# Sized for the library's worst case (tile count x 16 splits x 64x128 float32 outputs).
_SPLITK = None
SPLITK_BYTES = 320 * 16 * 64 * 128 * 4
PRO_CMAX = 2304      # conv_igemm.hip HDU_PRO_CMAX: widest contraction whose BN prologue the async-DMA pointwise kernels take


def splitk_scratch():
    global _SPLITK
    if _SPLITK is None:
        _SPLITK = (torch.empty(SPLITK_BYTES // 4, dtype=torch.float32, device=device()),
                   torch.zeros(512, dtype=torch.int32, device=device()))
    return _SPLITK


def conv_desc(x, w_ptr, y, K, stride=(1, 1, 1), pad=(0, 0, 0), up=(0, 0, 0), skip=None, pro=None, relu=True,
              bias=None, accumulate=False, drop_keep=1.0, drop_seed=0, drop_seed_dev=None, epi=None, splitk=None, halo_out=0,
              shard_world=1):
    """x: Act (stored input), y: Act (output), K=(KD,KH,KW); epi = (a, b, relu): output affine of the BN that follows;
    splitk = (float32 scratch tensor of SPLITK_BYTES, int32[512] zeroed counters) for launches on another stream;
    halo_out = depth planes of y that a neighbouring depth shard computes too (data gradient w.r.t. an input with halos);
    shard_world = depth shards the layer is split over (the OWNING model's: engine.Ctx.conv_desc passes it -- ADVICE r3: a
    process-wide setting leaked into descriptors of other models)."""
    d = ConvDesc()
    if shard_world > 1:      # the unsharded layer's output pixels: the library decides tiles / split-K for those
        d.layer_rows = y.N * (y.D - halo_out) * shard_world * y.H * y.W
    ws, cnt = splitk if splitk is not None else splitk_scratch()
    d.splitk_ws, d.splitk_ws_bytes, d.splitk_counters = ws.data_ptr(), SPLITK_BYTES, cnt.data_ptr()
    d.dtype = x.dtype
    d.x, d.ldx = x.ptr, x.ld
    d.N, d.Di, d.Hi, d.Wi, d.Cin = x.N, x.D, x.H, x.W, x.C
    d.ud, d.uh, d.uw = up
    if skip is not None:
        d.skip, d.ldskip = skip.ptr, skip.ld
    if pro is not None:
        d.pro_a, d.pro_b, d.pro_relu = fptr(pro[0]), fptr(pro[1]), 1 if relu else 0
    d.w = w_ptr
    d.KD, d.KH, d.KW = K
    d.sd, d.sh, d.sw = stride
    d.pd, d.ph, d.pw = pad
    d.y, d.ldy = y.ptr, y.ld
    d.Do, d.Ho, d.Wo, d.Cout = y.D, y.H, y.W, y.C
    d.bias = fptr(bias)
    if epi is not None:
        d.epi_a, d.epi_b, d.epi_relu = fptr(epi[0]), fptr(epi[1]), 1 if epi[2] else 0
    d.accumulate = 1 if accumulate else 0
    d.drop_keep = drop_keep
    d.drop_seed = drop_seed
    if drop_seed_dev is not None:
        d.drop_seed_dev = ctypes.c_void_p(drop_seed_dev.data_ptr())
    return d


def conv_fprop(d):
    check(_l.get().hdu_conv_fprop(ctypes.byref(d), stream()), "hdu_conv_fprop")


def conv_splitk_ws_bytes(d):
    return _l.get().hdu_conv_splitk_ws_bytes(ctypes.byref(d))


def conv_wgrad(d, dw):
    check(_l.get().hdu_conv_wgrad(ctypes.byref(d), fptr(dw), stream()), "hdu_conv_wgrad")


class WgradPlan:
    """Deferred, batched filter gradients (hdu_wgrad_plan_*): add(desc, dw) for every layer at build time, run() once
    per backward pass.  One launch per kernel family; tables live in device memory (uploaded once).  (Launching the plan in
    groups on a second stream while the backward chain continues was measured slower in rounds 1 and 3:
    profiles/r03_experiment_wgrad_side_stream_overlap.txt.)

    Pixel splits are sized from the WHOLE launch (round 4): a workgroup ends with Cout x 128 float atomics on its partial
    tile, and inside a batched launch the OTHER layers fill the GPU, so every workgroup of a DMA family takes the same number
    of 64-pixel steps S = clamp(sum over the family layers of tiles x steps / LAUNCH_WGS, 8, 128) instead of the 8 a layer
    launched alone would need.  Swept on MI355X (profiles/r04_experiment_wgrad_split_sizing.txt): 2D best at 64-128 steps
    (18.1 -> 17.6 ms), end2end at 32-64, denseunet_3d (few small layers) at 8 -- all three fall out of LAUNCH_WGS = 512 (1024 and 2048 within 0.1 ms)."""

    LAUNCH_WGS = 512

    def __init__(self, target_wgs=0, min_steps=None):
        self.target = target_wgs
        self.min_steps = min_steps        # None: sized per family as described above; an int forces it (tests / A-B runs)
        self.layers = []                  # (descriptor, dw): descriptors (and the tensors they point to) must outlive the plan
        self.by_variant = {}
        self.descs = {}
        self.steps_of = {}
        self.tables = None

    def add(self, d, dw):
        self.layers.append((d, dw))
        self.tables = None

    def __len__(self):
        return len(self.layers)

    def finalize(self):
        import numpy as np
        import os
        lib = _l.get()
        forced = os.environ.get("HDU_WGRAD_PLAN_STEPS")
        shapes, work = [], {}
        for d, dw in self.layers:
            variant, tiles, steps = ctypes.c_int(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
            check(lib.hdu_wgrad_plan_shape(ctypes.byref(d), ctypes.byref(variant), ctypes.byref(tiles), ctypes.byref(steps)),
                  "hdu_wgrad_plan_shape")
            shapes.append(variant.value)
            work[variant.value] = work.get(variant.value, 0) + tiles.value * steps.value
        self.steps_of = {}
        for v, w in work.items():
            if v >= 8:
                # halo families: `steps` counts 4 x 32-pixel spatial tiles (two 64-pixel steps each); the same whole-launch rule,
                # 4 ... 64 tiles per workgroup (HDU_WGRAD_HALO_TILES forces a count; 2 = the per-layer sizing of rounds 2-3)
                ht = os.environ.get("HDU_WGRAD_HALO_TILES")
                wgs = int(os.environ.get("HDU_WGRAD_LAUNCH_WGS", self.LAUNCH_WGS))
                self.steps_of[v] = int(ht) if ht else (int(self.min_steps) if self.min_steps is not None
                                                      else max(4, min(64, (w + wgs - 1) // wgs)))
            elif forced:
                self.steps_of[v] = int(forced)
            elif self.min_steps is not None:
                self.steps_of[v] = int(self.min_steps)
            else:
                wgs = int(os.environ.get("HDU_WGRAD_LAUNCH_WGS", self.LAUNCH_WGS))
                self.steps_of[v] = max(8, min(128, (w + wgs - 1) // wgs))
        nb = lib.hdu_wgrad_plan_entry_bytes()
        self.by_variant, self.descs = {}, {}
        for (d, dw), v in zip(self.layers, shapes):
            ent = (ctypes.c_ubyte * nb)()
            variant, nblk = ctypes.c_int(0), ctypes.c_uint32(0)
            check(lib.hdu_wgrad_plan_fill(ctypes.byref(d), fptr(dw), self.target, self.steps_of[v], ctypes.cast(ent, ctypes.c_void_p),
                                          ctypes.byref(variant), ctypes.byref(nblk)), "hdu_wgrad_plan_fill")
            assert variant.value == v
            self.by_variant.setdefault(v, []).append((bytes(ent), nblk.value))
            self.descs.setdefault(v, []).append(d)       # (bench.py: algorithmic FLOPs of a batched launch)
        self.tables = []
        for variant, ents in sorted(self.by_variant.items()):
            raw = np.frombuffer(b"".join(e for e, _ in ents), dtype=np.uint8).copy()
            begins = np.zeros(len(ents), dtype=np.uint32)
            tot = 0
            for i, (_, nb_) in enumerate(ents):
                begins[i] = tot
                tot += nb_
            assert tot < 2 ** 31
            self.tables.append((variant, torch.from_numpy(raw).to(device()),
                                torch.from_numpy(begins.view(np.int32)).to(device()), len(ents), tot))

    def run(self, around=None):
        """around(variant, launch): optional hook that performs the launch itself (bench.py brackets it with events)"""
        if self.tables is None:
            self.finalize()
        lib = _l.get()
        for variant, tab, begins, n, tot in self.tables:
            def launch(variant=variant, tab=tab, begins=begins, n=n, tot=tot):
                check(lib.hdu_wgrad_plan_run(variant, ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(begins.data_ptr()), n,
                                             tot, stream()), "hdu_wgrad_plan_run")
            if around is None:
                launch()
            else:
                around(variant, launch)


class ZeroPlan:
    """hdu_zero_regions: ONE launch clears a fixed list of device buffers (and bumps the step counter).  Built once per
    list; the table lives in device memory."""

    def __init__(self, tensors, copies=()):
        """tensors: buffers to clear; copies: (dst, src) pairs of equal size filled by the same launch"""
        import numpy as np
        self.keep = [t for t in tensors if t is not None and t.numel() > 0]
        self.keep_copies = [(d, s_) for d, s_ in copies if d is not None and d.numel() > 0]
        ents, blk = [], 0
        for t, src in [(t, None) for t in self.keep] + self.keep_copies:
            nb = t.numel() * t.element_size()
            assert t.is_contiguous() and t.data_ptr() % 16 == 0 and nb % 4 == 0, "zero plan: 16-byte aligned, whole dwords"
            if src is not None:
                assert src.is_contiguous() and src.data_ptr() % 16 == 0 and src.numel() * src.element_size() == nb
            ents.append(_l.ZeroEntry(t.data_ptr(), nb, blk, 0, src.data_ptr() if src is not None else None))
            blk += (nb + _l.ZERO_BLOCK_BYTES - 1) // _l.ZERO_BLOCK_BYTES
        self.n, self.blocks = len(ents), blk
        self.table = None
        if ents:
            arr = (_l.ZeroEntry * len(ents))(*ents)
            self.table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device())

    def run(self, counter=None, inc=0):
        if self.table is None:
            return
        check(_l.get().hdu_zero_regions(ctypes.c_void_p(self.table.data_ptr()), self.n, self.blocks,
                                        ctypes.c_void_p(counter.data_ptr()) if counter is not None else None, inc, stream()),
              "hdu_zero_regions")


class Split3Plan:
    """hdu_split3_batched: ONE launch writes the bf16 (hi, lo, hi) / (hi, hi, lo) image triples of a fixed list of float32
    tensors (the operands and output gradients of every filter gradient of a float32 network in the split-bf16 modes).
    add() at build time, run() once per backward pass; the table lives in device memory."""

    def __init__(self):
        self.items = []
        self.table = None
        self.keep = []

    def add(self, src, pattern, dst, pro=None, relu=False):
        """src: float32 Act (any pixel stride); dst: bf16 Act with N = 3 * src.N, dense; pro = (a, b) float32 vectors or None"""
        assert src.dtype == HDU_F32 and dst.dtype == HDU_BF16 and dst.ld == dst.C == src.C and dst.M == 3 * src.M
        self.items.append((src, pattern, dst, pro, relu))
        self.table = None

    def __len__(self):
        return len(self.items)

    def finalize(self):
        import numpy as np
        lib = _l.get()
        ents, blk = [], 0
        for src, pattern, dst, pro, relu in self.items:
            e, nb = _l.Split3Entry(), ctypes.c_uint32(0)
            check(lib.hdu_split3_entry_fill(ctypes.byref(e), src.ptr, src.ld, src.M, src.C, fptr(pro[0]) if pro else None,
                                            fptr(pro[1]) if pro else None, 1 if relu else 0, pattern, dst.ptr, blk,
                                            ctypes.byref(nb)), "hdu_split3_entry_fill")
            ents.append(e)
            blk += nb.value
        self.n, self.blocks = len(ents), blk
        self.chl = ents[0].chl
        assert all(e.chl == self.chl for e in ents)
        arr = (_l.Split3Entry * len(ents))(*ents)
        self.table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device())

    def run(self):
        if not self.items:
            return
        if self.table is None:
            self.finalize()
        check(_l.get().hdu_split3_batched(ctypes.c_void_p(self.table.data_ptr()), self.n, self.blocks, self.chl, stream()),
              "hdu_split3_batched")


def zero_tensor(t):
    check(_l.get().hdu_zero(ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), stream()), "hdu_zero")


def conv_dgrad_strided(d):
    check(_l.get().hdu_conv_dgrad_strided(ctypes.byref(d), stream()), "hdu_conv_dgrad_strided")


class Stride2Dgrad:
    """Data gradient of a stride-2 (per axis: 1 or 2) convolution as 2^d stride-1 implicit GEMMs (include/hdu.h,
    hdu_stride2_dgrad_filters / hdu_parity_interleave): built once per layer, run() per backward pass."""

    def __init__(self, dtype, w_master, dy, dx_dims, Cin_p, K, stride, pad, shard_world=1):
        """w_master: float32 master filter [Cout_p][KD][KH][KW][Cin_p] (flat tensor view); dy: Act of the output gradient;
        dx_dims = (N, Di, Hi, Wi) of the input"""
        N, Di, Hi, Wi = dx_dims
        for a in range(3):
            if stride[a] not in (1, 2) or (Di, Hi, Wi)[a] % stride[a]:
                raise ValueError("stride-2 data gradient: strides 1 or 2 and input dims divisible by them")
        self.dtype, self.w_master, self.dy, self.dims, self.C = dtype, w_master, dy, dx_dims, Cin_p
        self.K, self.stride, self.pad = K, stride, pad
        self.Cout = dy.C
        q = (Di // stride[0], Hi // stride[1], Wi // stride[2])
        classes, off = [], 0
        for rd in range(stride[0]):
            for rh in range(stride[1]):
                for rw in range(stride[2]):
                    nt, pl = [], []
                    for a, r in enumerate((rd, rh, rw)):
                        kmax = K[a] - 1
                        if stride[a] == 2 and ((kmax - r - pad[a]) & 1):
                            kmax -= 1
                        nt.append(kmax // stride[a] + 1)
                        pl.append(-(r + pad[a] - kmax) // stride[a])
                    classes.append((tuple(nt), tuple(pl), off))
                    off += Cin_p * nt[0] * nt[1] * nt[2] * self.Cout
        tdt = _TORCH_DT[dtype]
        self.wsub = torch.zeros(off, dtype=tdt, device=device())
        cls_elems = N * q[0] * q[1] * q[2] * Cin_p
        self.cls = torch.zeros(len(classes) * cls_elems, dtype=tdt, device=device())
        esz = self.wsub.element_size()
        self.descs = []
        for i, (nt, pl, woff) in enumerate(classes):
            y = Act(self.cls, i * cls_elems, N, q[0], q[1], q[2], Cin_p, Cin_p, dtype)
            self.descs.append(conv_desc(dy, ctypes.c_void_p(self.wsub.data_ptr() + woff * esz), y, nt, (1, 1, 1), pl, shard_world=shard_world))

    def run(self, dx, accumulate=False):
        lib = _l.get()
        K, st, pd = self.K, self.stride, self.pad
        check(lib.hdu_stride2_dgrad_filters(self.dtype, fptr(self.w_master), self.Cout, K[0], K[1], K[2], self.C, st[0], st[1],
                                            st[2], pd[0], pd[1], pd[2], ctypes.c_void_p(self.wsub.data_ptr()), stream()),
              "hdu_stride2_dgrad_filters")
        for d in self.descs:
            conv_fprop(d)
        N, Di, Hi, Wi = self.dims
        check(lib.hdu_parity_interleave(self.dtype, ctypes.c_void_p(self.cls.data_ptr()), N, Di, Hi, Wi, self.C, st[0], st[1],
                                        st[2], dx.ptr, dx.ld, 1 if accumulate else 0, stream()), "hdu_parity_interleave")


def conv_kernel_name(d, op=0):
    buf = ctypes.create_string_buffer(128)
    check(_l.get().hdu_conv_kernel_name(ctypes.byref(d), op, buf, 128), "hdu_conv_kernel_name")
    return buf.value.decode()


def weight_prep(dtype, w_master, Cout, T, Cin, w_f, w_d):
    check(_l.get().hdu_weight_prep(dtype, fptr(w_master), Cout, T, Cin,
                                   ctypes.c_void_p(w_f.data_ptr()) if w_f is not None else None,
                                   ctypes.c_void_p(w_d.data_ptr()) if w_d is not None else None, stream()),
          "hdu_weight_prep")


def weight_prep_batched(dtype, table_dev, n, total_tiles, master, wc):
    check(_l.get().hdu_weight_prep_batched(dtype, ctypes.c_void_p(table_dev.data_ptr()), n, total_tiles, fptr(master),
                                           ctypes.c_void_p(wc.data_ptr()), stream()), "hdu_weight_prep_batched")


class Workspace:
    """Reduction scratch shared by all stats / loss calls of one model (stream-ordered reuse)."""

    def __init__(self, nbytes):
        self.nbytes = max(int(nbytes), 1 << 16)
        self.buf = torch.empty(self.nbytes // 4 + 64, dtype=torch.float32, device=device())

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr())


def reduce_ws_bytes(M, C):
    return _l.get().hdu_reduce_ws_bytes(M, C)


def bn_stats(x, mean, var, ws):
    check(_l.get().hdu_bn_stats(x.dtype, x.ptr, x.ld, x.M, x.C, fptr(mean), fptr(var), ws.ptr, ws.nbytes, stream()),
          "hdu_bn_stats")


def bn_fold(C, mean, var, gamma, beta, eps, sgamma, sbeta, a, b, rstd, mov_mean=None, mov_var=None, momentum=0.99):
    check(_l.get().hdu_bn_fold(C, fptr(mean), fptr(var), fptr(gamma), fptr(beta), eps, fptr(sgamma), fptr(sbeta),
                               fptr(a), fptr(b), fptr(rstd), fptr(mov_mean), fptr(mov_var), momentum, stream()),
          "hdu_bn_fold")


class FoldPlan:
    """inference-mode BN(+Scale) folds of many layers as ONE launch (hdu_bn_fold_batched); built once, tables on device"""

    def __init__(self, entries):
        """entries: [(C, eps, mean, var, gamma, beta, sgamma|None, sbeta|None, a, b, rstd)] of float32 tensors"""
        import numpy as np
        self.keep = entries
        n = len(entries)
        tab = (_l.FoldEntry * n)()
        begins = np.zeros(n, dtype=np.uint32)
        tot = 0
        for i, (C, eps, mean, var, g, be, sg, sb, a, b, r) in enumerate(entries):
            p = lambda t: t.data_ptr() if t is not None else None
            tab[i] = _l.FoldEntry(p(mean), p(var), p(g), p(be), p(sg), p(sb), p(a), p(b), p(r), C, eps)
            begins[i] = tot
            tot += (C + 255) // 256
        self.n, self.total = n, tot
        self.table = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy()).to(device())
        self.begins = torch.from_numpy(begins.view(np.int32)).to(device())

    def run(self):
        check(_l.get().hdu_bn_fold_batched(ctypes.c_void_p(self.table.data_ptr()), ctypes.c_void_p(self.begins.data_ptr()),
                                           self.n, self.total, stream()), "hdu_bn_fold_batched")


class BnBwdPlan:
    """parameter gradients of many inference-mode BN(+Scale) layers from their fused-epilogue slot sums as ONE launch
    (hdu_bn_bwd_finalize_batched); built once, tables on device"""

    def __init__(self, entries):
        """entries: [(partial, slots, C, gamma, beta, sgamma|None, dgamma|None, dbeta|None, dsgamma|None, dsbeta|None)]"""
        import numpy as np
        self.keep = entries
        n = len(entries)
        tab = (_l.BnBwdEntry * n)()
        begins = np.zeros(n, dtype=np.uint32)
        tot = 0
        p = lambda t: t.data_ptr() if t is not None else None
        for i, (part, slots, C, g, be, sg, dg, db, dsg, dsb) in enumerate(entries):
            tab[i] = _l.BnBwdEntry(p(part), slots, C, p(g), p(be), p(sg), p(dg), p(db), p(dsg), p(dsb))
            begins[i] = tot
            tot += (C + 7) // 8
        self.n, self.total = n, tot
        self.table = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy()).to(device())
        self.begins = torch.from_numpy(begins.view(np.int32)).to(device())

    def run(self):
        check(_l.get().hdu_bn_bwd_finalize_batched(ctypes.c_void_p(self.table.data_ptr()), ctypes.c_void_p(self.begins.data_ptr()),
                                                   self.n, self.total, stream()), "hdu_bn_bwd_finalize_batched")


def bn_stats_fold(x, mean, var, gamma, beta, eps, sgamma, sbeta, a, b, rstd, mov_mean, mov_var, momentum, ws):
    check(_l.get().hdu_bn_stats_fold(x.dtype, x.ptr, x.ld, x.M, x.C, fptr(mean), fptr(var), fptr(gamma), fptr(beta), eps,
                                     fptr(sgamma), fptr(sbeta), fptr(a), fptr(b), fptr(rstd), fptr(mov_mean),
                                     fptr(mov_var), momentum, ws.ptr, ws.nbytes, stream()), "hdu_bn_stats_fold")


def bn_bwd_reduce_coef(dz, x, a, b, relu, mean, rstd, batch_stats, gamma, beta, sgamma, s1, s2, k1, k2, k3, dgamma,
                       dbeta, dsgamma, dsbeta, ws):
    check(_l.get().hdu_bn_bwd_reduce_coef(x.dtype, dz.ptr, dz.ld, x.ptr, x.ld, x.M, x.C, fptr(a), fptr(b),
                                          1 if relu else 0, fptr(mean), fptr(rstd), 1 if batch_stats else 0,
                                          fptr(gamma), fptr(beta), fptr(sgamma), fptr(s1), fptr(s2), fptr(k1), fptr(k2),
                                          fptr(k3), fptr(dgamma), fptr(dbeta), fptr(dsgamma), fptr(dsbeta), ws.ptr,
                                          ws.nbytes, stream()), "hdu_bn_bwd_reduce_coef")


def bn_bwd_reduce(dz, x, a, b, relu, mean, rstd, s1, s2, ws):
    check(_l.get().hdu_bn_bwd_reduce(x.dtype, dz.ptr, dz.ld, x.ptr, x.ld, x.M, x.C, fptr(a), fptr(b), 1 if relu else 0,
                                     fptr(mean), fptr(rstd), fptr(s1), fptr(s2), ws.ptr, ws.nbytes, stream()),
          "hdu_bn_bwd_reduce")


def bn_bwd_coef(C, M, batch_stats, s1, s2, gamma, beta, sgamma, rstd, k1, k2, k3, dgamma=None, dbeta=None,
                dsgamma=None, dsbeta=None):
    check(_l.get().hdu_bn_bwd_coef(C, M, 1 if batch_stats else 0, fptr(s1), fptr(s2), fptr(gamma), fptr(beta),
                                   fptr(sgamma), fptr(rstd), fptr(k1), fptr(k2), fptr(k3), fptr(dgamma), fptr(dbeta),
                                   fptr(dsgamma), fptr(dsbeta), stream()), "hdu_bn_bwd_coef")


def bn_bwd_apply(dz, x, a, b, relu, mean, k1, k2, k3, dx, accumulate=False, drop_keep=1.0, drop_seed=0,
                 drop_seed_dev=None):
    check(_l.get().hdu_bn_bwd_apply(x.dtype, dz.ptr, dz.ld, x.ptr, x.ld, x.M, x.C, fptr(a), fptr(b), 1 if relu else 0,
                                    fptr(mean), fptr(k1), fptr(k2), fptr(k3), dx.ptr, dx.ld, 1 if accumulate else 0,
                                    drop_keep, drop_seed,
                                    ctypes.c_void_p(drop_seed_dev.data_ptr()) if drop_seed_dev is not None else None,
                                    stream()), "hdu_bn_bwd_apply")


def bn_bwd_fused(dz, x, a, b, relu, mean, rstd, batch_stats, gamma, beta, sgamma, sums, slots, dgamma, dbeta, dsgamma,
                 dsbeta, dx, accumulate=False, drop_keep=1.0, drop_seed=0, drop_seed_dev=None, sums_ready=False):
    """reduction (float atomics into the zeroed [slots][2][C] table `sums`) + coefficients / parameter gradients / dx: two
    launches (include/hdu.h: hdu_bn_bwd_fused); sums_ready: the data-gradient epilogue that produced dz filled `sums`
    (hdu_conv_desc.bnb_relu bit 2) -- the apply launch alone (hdu_bn_bwd_apply_sums)"""
    fn = _l.get().hdu_bn_bwd_apply_sums if sums_ready else _l.get().hdu_bn_bwd_fused
    check(fn(x.dtype, dz.ptr, dz.ld, x.ptr, x.ld, x.M, x.C, fptr(a), fptr(b), 1 if relu else 0,
             fptr(mean), fptr(rstd), 1 if batch_stats else 0, fptr(gamma), fptr(beta), fptr(sgamma),
             fptr(sums), slots, fptr(dgamma), fptr(dbeta), fptr(dsgamma), fptr(dsbeta), dx.ptr, dx.ld,
             1 if accumulate else 0, drop_keep, drop_seed,
             ctypes.c_void_p(drop_seed_dev.data_ptr()) if drop_seed_dev is not None else None,
             stream()), "hdu_bn_bwd_fused")


def bn_bwd_finalize(partial, slots, M, C, batch_stats, gamma, beta, sgamma, mean, rstd, dgamma, dbeta, dsgamma, dsbeta,
                    corr3, corr4):
    check(_l.get().hdu_bn_bwd_finalize(fptr(partial), slots, M, C, 1 if batch_stats else 0, fptr(gamma), fptr(beta),
                                       fptr(sgamma), fptr(mean), fptr(rstd), fptr(dgamma), fptr(dbeta), fptr(dsgamma),
                                       fptr(dsbeta), fptr(corr3), fptr(corr4), stream()), "hdu_bn_bwd_finalize")


def bn_bwd_finalize_correct(partial, slots, M, C, gamma, beta, sgamma, mean, rstd, dgamma, dbeta, dsgamma, dsbeta, corr3, corr4,
                            cs0, u, du):
    """hdu_bn_bwd_finalize (batch statistics) of a BN over C channels + hdu_bn_bwd_correct of the channels [cs0, cs0 + u.C) of the
    tensor it normalises (u / du: Act slabs starting at channel cs0), one launch"""
    check(_l.get().hdu_bn_bwd_finalize_correct(u.dtype, fptr(partial), slots, M, C, fptr(gamma), fptr(beta), fptr(sgamma), fptr(mean),
                                               fptr(rstd), fptr(dgamma), fptr(dbeta), fptr(dsgamma), fptr(dsbeta), fptr(corr3),
                                               fptr(corr4), cs0, u.C, u.ptr, u.ld, du.ptr, du.ld, stream()),
          "hdu_bn_bwd_finalize_correct")


def bn_bwd_correct(u, corr3, corr4, du):
    check(_l.get().hdu_bn_bwd_correct(u.dtype, u.ptr, u.ld, u.M, u.C, fptr(corr3), fptr(corr4), du.ptr, du.ld, stream()),
          "hdu_bn_bwd_correct")


def affine_act(x, a, b, relu, z):
    check(_l.get().hdu_affine_act(x.dtype, x.ptr, x.ld, x.M, x.C, fptr(a), fptr(b), 1 if relu else 0, z.ptr, z.ld,
                                  stream()), "hdu_affine_act")


def materialize(x, a, b, relu, up, skip, out):
    check(_l.get().hdu_materialize(x.dtype, x.ptr, x.ld, x.N, x.D, x.H, x.W, x.C, fptr(a), fptr(b), 1 if relu else 0,
                                   up[0], up[1], up[2], skip.ptr if skip is not None else None,
                                   skip.ld if skip is not None else 0, out.ptr, out.ld, stream()), "hdu_materialize")


def bn_stats_finalize(partial, slots, M, C, shift, mean, var, fold=None):
    """fold = (gamma, beta, eps, sgamma, sbeta, a, b, rstd, mov_mean, mov_var, momentum) or None"""
    g, be, eps, sg, sb, a, b, r, mm, mv, mom = fold if fold is not None else (None, None, 0.0, None, None, None, None,
                                                                                 None, None, None, 0.0)
    check(_l.get().hdu_bn_stats_finalize(fptr(partial), slots, M, C, fptr(shift), fptr(mean), fptr(var), fptr(g), fptr(be),
                                         eps, fptr(sg), fptr(sb), fptr(a), fptr(b), fptr(r), fptr(mm), fptr(mv), mom,
                                         stream()), "hdu_bn_stats_finalize")


def bn_stats_finalize_fold_next(partial, slots, M, Cseg, seg_c0, C_all, shift_all, mean_all, var_all, fold):
    """finalize the segment's epilogue statistics and fold the next BN over [0, C_all) in one launch (fold as above)"""
    g, be, eps, sg, sb, a, b, r, mm, mv, mom = fold
    check(_l.get().hdu_bn_stats_finalize_fold_next(fptr(partial), slots, M, Cseg, seg_c0, C_all, fptr(shift_all),
                                                   fptr(mean_all), fptr(var_all), fptr(g), fptr(be), eps, fptr(sg), fptr(sb),
                                                   fptr(a), fptr(b), fptr(r), fptr(mm), fptr(mv), mom, stream()),
          "hdu_bn_stats_finalize_fold_next")


def materialize_stats(x, stats, relu, up, skip, out):
    """hdu_materialize with the BN folded inside the launch from a conv epilogue's sums.
    stats = (partial, slots, M, Cseg, seg_c0, shift, mean, var, fold) with fold = (gamma, beta, eps, sgamma, sbeta, a, b, rstd,
    mov_mean, mov_var, momentum); shift / mean / var are indexed by the BN's channel (= channel of x)."""
    partial, slots, M, Cseg, seg_c0, shift, mean, var, fold = stats
    g, be, eps, sg, sb, a, b, r, mm, mv, mom = fold
    f = _l.BnStatsFold()
    f.partial, f.slots, f.Cseg, f.seg_c0, f.M = fptr(partial), slots, Cseg, seg_c0, M
    f.shift, f.mean, f.var = fptr(shift), fptr(mean), fptr(var)
    f.gamma, f.beta, f.sgamma, f.sbeta, f.eps, f.momentum = fptr(g), fptr(be), fptr(sg), fptr(sb), eps, mom
    f.a, f.b, f.rstd, f.mov_mean, f.mov_var = fptr(a), fptr(b), fptr(r), fptr(mm), fptr(mv)
    check(_l.get().hdu_materialize_stats(x.dtype, x.ptr, x.ld, x.N, x.D, x.H, x.W, x.C, ctypes.byref(f), 1 if relu else 0,
                                         up[0], up[1], up[2], skip.ptr if skip is not None else None,
                                         skip.ld if skip is not None else 0, out.ptr, out.ld, stream()), "hdu_materialize_stats")


def colsum(x, out, ws):
    check(_l.get().hdu_colsum(x.dtype, x.ptr, x.ld, x.M, x.C, fptr(out), ws.ptr, ws.nbytes, stream()), "hdu_colsum")


def maxpool_fwd(x, y, argmax=None, pad_d=1):
    check(_l.get().hdu_maxpool3s2_fwd(x.dtype, x.ptr, x.ld, x.N, x.D, x.H, x.W, x.C, y.ptr, y.ld,
                                      ctypes.c_void_p(argmax.data_ptr()) if argmax is not None else None, pad_d,
                                      stream()),
          "hdu_maxpool3s2_fwd")


def maxpool_bwd(argmax, dy, dx, accumulate=False, pad_d=1):
    check(_l.get().hdu_maxpool3s2_bwd(dx.dtype, ctypes.c_void_p(argmax.data_ptr()), dy.ptr, dy.ld, dx.N, dx.D, dx.H,
                                      dx.W, dx.C, dx.ptr, dx.ld, 1 if accumulate else 0, pad_d, stream()),
          "hdu_maxpool3s2_bwd")


def avgpool_fwd(x, y):
    check(_l.get().hdu_avgpool2_fwd(x.dtype, x.ptr, x.ld, x.N, x.D, x.H, x.W, x.C, y.ptr, y.ld, stream()),
          "hdu_avgpool2_fwd")


def avgpool_bwd(dy, dx, accumulate=False):
    check(_l.get().hdu_avgpool2_bwd(dx.dtype, dy.ptr, dy.ld, dx.N, dx.D, dx.H, dx.W, dx.C, dx.ptr, dx.ld,
                                    1 if accumulate else 0, stream()), "hdu_avgpool2_bwd")


def upsample_bwd(dxe, dz, up, accumulate=False):
    check(_l.get().hdu_upsample_bwd(dz.dtype, dxe.ptr, dxe.ld, dz.N, dz.D, dz.H, dz.W, dz.C, up[0], up[1], up[2],
                                    dz.ptr, dz.ld, 1 if accumulate else 0, stream()), "hdu_upsample_bwd")


def wce_loss(logits, labels_u8, row0, M, weights, grad_scale, dlogits, loss_sum, class_count, ws):
    esz = logits.buf.element_size()
    lp = ctypes.c_void_p(logits.buf.data_ptr() + (logits.off + row0 * logits.ld) * esz)
    dp = None
    ldd = 0
    cp = 0
    if dlogits is not None:
        dp = ctypes.c_void_p(dlogits.buf.data_ptr() + (dlogits.off + row0 * dlogits.ld) * esz)
        ldd, cp = dlogits.ld, dlogits.C
    check(_l.get().hdu_wce_loss(logits.dtype, lp, logits.ld, ctypes.c_void_p(labels_u8.data_ptr() + row0), M,
                                weights[0], weights[1], weights[2], grad_scale, dp, ldd, cp, fptr(loss_sum),
                                fptr(class_count), ws.ptr, ws.nbytes, stream()), "hdu_wce_loss")


def sgd_nesterov(p, v, g, lr, momentum, grad_scale=1.0):
    check(_l.get().hdu_sgd_nesterov(fptr(p), fptr(v), fptr(g), p.numel(), lr, momentum, grad_scale, stream()),
          "hdu_sgd_nesterov")


def slab25d(vol_f32, D, H, W, out):
    check(_l.get().hdu_slab25d(out.dtype, fptr(vol_f32), D, H, W, out.ptr, out.ld, stream()), "hdu_slab25d")


def make_input3d(vol_f32, logits2d, scale, out):
    check(_l.get().hdu_make_input3d(out.dtype, fptr(vol_f32), logits2d.ptr, logits2d.ld, scale, out.D, out.H, out.W,
                                    out.ptr, out.ld, stream()), "hdu_make_input3d")


def make_input3d_bwd(dinput3d, scale, dlogits2d, accumulate=False):
    check(_l.get().hdu_make_input3d_bwd(dinput3d.dtype, dinput3d.ptr, dinput3d.ld, scale, dinput3d.M, dlogits2d.ptr,
                                        dlogits2d.ld, dlogits2d.C, 1 if accumulate else 0, stream()),
          "hdu_make_input3d_bwd")


def cast_pad(src_f32, M, C, dst):
    check(_l.get().hdu_cast_pad(dst.dtype, fptr(src_f32), M, C, dst.ptr, dst.ld, dst.C, stream()), "hdu_cast_pad")


def cast_out(src, C, dst_f32):
    check(_l.get().hdu_cast_out(src.dtype, src.ptr, src.ld, src.M, C, fptr(dst_f32), stream()), "hdu_cast_out")


def softmax_accumulate(logits, row0, M, num, score):
    """score[m][j] += softmax(logits[row0 + m][0:3])[j], j < num (include/hdu.h: hdu_softmax_accumulate); score: float32 view [M*num]"""
    esz = logits.buf.element_size()
    lp = ctypes.c_void_p(logits.buf.data_ptr() + (logits.off + row0 * logits.ld) * esz)
    assert score.dtype == torch.float32 and score.is_contiguous() and score.numel() == M * num
    check(_l.get().hdu_softmax_accumulate(logits.dtype, lp, logits.ld, M, num, ctypes.c_void_p(score.data_ptr()), stream()),
          "hdu_softmax_accumulate")
