"""Host-side execution engine of the hot path (replaces the slice of Keras-2.0.8/keras/engine/training.py
that is one train / predict step: :948-967 _make_train_function, :1715-1766 train_on_batch, :1659-1713 predict).

There is no graph tracer and no autodiff: a model constructor emits a static list of forward launches and, in
reverse, hand-written backward launches; every buffer is allocated once at build time so a whole step is a
fixed launch sequence that can be captured in one hipGraph.

Layout: activations channels-last [N][D][H][W][C] in the compute dtype (bf16, or f32 in parity mode), depth-major
for 3D (the reference's N,H,W,D,C is converted at the boundary).  Parameters live in ONE flat float32 buffer
(trainables first) with matching gradient and velocity buffers, so the optimiser and the data-parallel gradient
all-reduce are single flat operations.  Conv filters are stored [Cout][KD][KH][KW][Cin] (converted from Keras'
(k..,Cin,Cout) in set_weights / get_weights).
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from . import shard as _sh
from .lib import HDU_BF16, HDU_F32


class Param:
    __slots__ = ("layer", "idx", "kind", "keras_shape", "shape", "trainable", "init", "offset", "numel", "ctx", "meta")

    def __init__(self, ctx, layer, idx, kind, keras_shape, shape, trainable, init, meta=None):
        self.ctx, self.layer, self.idx, self.kind = ctx, layer, idx, kind
        self.keras_shape, self.shape, self.trainable, self.init = tuple(keras_shape), tuple(shape), trainable, init
        self.numel = int(np.prod(shape))
        self.offset = None
        self.meta = meta or {}

    @property
    def data(self):
        return self.ctx.P[self.offset:self.offset + self.numel]

    @property
    def grad(self):
        return self.ctx.G[self.offset:self.offset + self.numel]


class Var:
    """An activation (or a channel slab of a dense-block buffer) plus its gradient / statistics bookkeeping."""

    def __init__(self, ctx, act, root=None, c0=0):
        self.ctx, self.act, self.c0 = ctx, act, c0
        self.root = root or self
        if root is None:
            self.grad_act = None
            self.written = False
            self.mean = None
            self.var = None
            self.shift = None
            self.needs_grad = False
            self.drop = None  # (keep, seed) if produced through dropout
            self.corr_off = None   # offset of this tensor's [2][ld] deferred-BN-backward accumulators in Ctx.corr_acc
        self.C = act.C

    def slab(self, c0, C):
        return Var(self.ctx, self.act.slab(c0, C), self.root, self.c0 + c0)

    # -- gradient storage (allocated lazily at build time)
    def require_grad(self):
        r = self.root
        if not r.needs_grad:
            r.needs_grad = True
            a = r.act
            r.grad_act = ops.Act.alloc(a.N, a.D, a.H, a.W, a.ld, a.dtype, zero=True)
        return self

    @property
    def grad(self):
        return self.root.grad_act.slab(self.c0, self.C)

    def dy(self):
        """the gradient of these channels as a READER takes it (the producing layer's backward): first the deferred,
        reduction-dependent part of the BN backward of every fused consumer of these channels (`-k3*u + k4`, summed over
        the consumers in corr3 / corr4 by hdu_bn_bwd_finalize) is added -- once, over just these channels."""
        r = self.root
        ctx = self.ctx
        if r.corr_off is not None and ctx.corr_acc is not None and ctx.fuse_bn_bwd_now:
            pend = ctx._pending_fin
            # (hdu_bn_bwd_finalize_correct addresses 16-byte chunks: channel offsets, widths and pixel strides must be chunk
            # multiples and the slabs 16-byte aligned -- anything else takes the two-launch path below instead of raising in the
            # middle of a backward pass)
            ch = ops.CHUNK[self.act.dtype]
            aligned = (self.c0 - (pend["c0"] if pend else 0)) % ch == 0 and self.C % ch == 0 and self.act.ld % ch == 0 \
                and self.grad.ld % ch == 0 and self.act.ptr.value % 16 == 0 and self.grad.ptr.value % 16 == 0
            if pend is not None and pend["root"] is r and pend["c0"] <= self.c0 and self.c0 + self.C <= pend["c0"] + pend["C"] \
                    and self.act.M == pend["M"] and aligned:
                # the finalize of the consumer BN that ran last and this correction: one launch (hdu_bn_bwd_finalize_correct)
                ctx._pending_fin = None
                ops.bn_bwd_finalize_correct(*pend["args"], self.c0 - pend["c0"], self.act, self.grad)
                return self.grad
            ctx.flush_pending_finalize()
            ld = r.act.ld
            base = r.corr_off + self.c0
            acc = ctx.corr_acc
            ops.bn_bwd_correct(self.act, acc[base:base + self.C], acc[base + ld:base + ld + self.C], self.grad)
        return self.grad

    def grad_mode(self):
        """returns accumulate flag for a writer into this var's gradient; first writer overwrites"""
        r = self.root
        acc = r.written
        if not acc:
            # a first write must cover the buffer from channel 0 (asserted by construction of the nets)
            assert self.c0 == 0, "first gradient write into a shared buffer must start at channel 0"
            if self.C != r.act.ld:
                # partial first write: clear the rest so later slab readers see zeros (part of the step-head launch once
                # the buffer is known, Ctx.step_zero)
                self.ctx.zero_partial_grad(r.grad_act.buf)
        r.written = True
        return acc

    # -- batch statistics of the channels of this var (shared by every consumer BN: SURVEY.md section 7)
    def stats(self):
        r = self.root
        if r.mean is None:
            dev = ops.device()
            r.mean = torch.zeros(r.act.ld, dtype=torch.float32, device=dev)
            r.var = torch.ones(r.act.ld, dtype=torch.float32, device=dev)
            # shift of the conv-epilogue statistics (sums of y - shift): last step's mean, copied by the step-head launch --
            # a separate array because the launch that finalizes a segment's moments reads the shift in every workgroup
            # while one of them publishes the new mean (hdu_materialize_stats)
            r.shift = torch.zeros(r.act.ld, dtype=torch.float32, device=dev)
            r.ctx.stat_roots.append(r)
        return r.mean[self.c0:self.c0 + self.C], r.var[self.c0:self.c0 + self.C]

    def stats_shift(self):
        self.stats()
        return self.root.shift[self.c0:self.c0 + self.C]


class _BwdList(list):
    """backward closures in forward order; remembers how many parameters existed when each one was appended, so that a
    suffix of the list (= a prefix of the backward pass) can be mapped to the range of the flat gradient buffer it
    completes (gradient buckets of the data-parallel exchange)"""

    def __init__(self, ctx):
        super().__init__()
        self._ctx = ctx
        self.mark = []

    def append(self, fn):
        super().append(fn)
        self.mark.append(len(self._ctx.params))


class Ctx:
    """Build + run context of one model."""

    def __init__(self, dtype, batch_shape):
        self.dtype = dtype
        self.ch = ops.CHUNK[dtype]
        self.params = []
        self.by_layer = OrderedDict()
        self.layer_kind = OrderedDict()
        self.fwd = []          # list of callables
        self.bwd = _BwdList(self)   # appended in forward order, executed reversed
        self.vars = []
        self.learning_phase = 1
        self.pass_id = 0
        self.P = self.G = self.V = None
        self.n_trainable = 0
        self.convs = []
        self.bns = []
        self._fold_plans = {}
        self.prefolded_pass = -1
        # inference-mode BN folds depend on parameters only: all of them (every BN in a predict pass, the frozen ones in
        # a training pass) run as ONE launch at the head of the forward instead of one launch per layer
        self.batch_fold = os.environ.get("HDU_BATCH_FOLD", "1") == "1"
        # BN(+Scale)+ReLU backward fused into the epilogue of the data-gradient launch that produces dz (include/hdu.h,
        # hdu_conv_desc.bnb_*): no dz round trip, no reduction pass, and the full-width apply pass of a dense-block layer
        # (C0 + l*growth channels) shrinks to a correction over the producer's own channels
        # Measured on MI355X (profiles/r02_experiment_fused_bn_backward.txt): for INFERENCE-mode BNs (the hybrids' 2D
        # branch and 3D dense blocks: no mean terms, so nothing is deferred) the step gains 3.8 % (end2end 28.6 -> 27.5 ms);
        # for batch-statistics BNs the 3.2 ms of reduction + apply passes it removes from the 2D step come back as +2.4 ms
        # of data-gradient epilogues (the u / gradient-slab loads are exposed: 2-3 workgroups per CU cannot hide them
        # the way the 2048-workgroup streaming kernels do) and +1.0 ms of correction launches: neutral (23.1 vs 23.2 ms).
        # Default 1 = inference-mode BNs only; 2 = every BN (tests cover both); 0 = off.
        self.defer_bnb_finalize = os.environ.get("HDU_DEFER_BNB_FINALIZE", "1") == "1"
        self._bnb_deferred, self._bnb_plan = [], None
        self._pending_fin = None
        self.merge_fin_correct = os.environ.get("HDU_MERGE_FIN_CORRECT", "1") == "1"
        self.absorb_stats = os.environ.get("HDU_ABSORB_STATS", "1") == "1"     # BN fold inside the consumer's materialize pass
        self.bn_bwd_fused = os.environ.get("HDU_BN_BWD_FUSED", "1") == "1"      # two-launch BN backward (hdu_bn_bwd_fused)
        self.fuse_bn_bwd_mode = int(os.environ.get("HDU_FUSE_BN_BWD", "1"))
        self.fuse_bn_bwd = self.fuse_bn_bwd_mode > 0
        self.fuse_bn_bwd_pw = os.environ.get("HDU_FUSE_BN_BWD_PW", "1") == "1"
        self.bnb_sums_epilogue = os.environ.get("HDU_BNB_SUMS_EPILOGUE", "1") == "1"      # round 6: ConvLayer._epilogue_sums
        self.fuse_bn_bwd_now = False
        self.bnb_sinks = []          # (layer, offset into bnb_acc)
        self.bnb_acc = None
        self.corr_acc = None
        self._corr_total = 0
        self._scratch = {}
        self.ws_bytes = 1 << 16
        self.ws = None
        self.dev = ops.device()
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)  # device step counter (dropout seed)
        self.drop_layers = 0
        self.dropout_enabled = True
        self.grad_enabled = True
        self.fuse_bn_epilogue = os.environ.get("HDU_FUSE_BN_EPILOGUE", "1") == "1"
        self.fuse_bn_epilogue_train = os.environ.get("HDU_FUSE_BN_EPILOGUE_TRAIN", "1") == "1"
        self.fold_next = os.environ.get("HDU_FOLD_NEXT", "1") == "1"
        self.shard = None          # shard.ShardInfo when one volume is split on the depth axis
        self.fuse_prologue = os.environ.get("HDU_FUSE_PROLOGUE", "0") == "1"
        # Round 4: a POINTWISE conv behind a BN(+Scale)+ReLU (every dense-block bottleneck, every transition) reads the raw
        # tensor and applies the affine to its operand fragments in registers (conv_igemm.hip: PRO kernels; the filter
        # gradient recomputes it the same way), so the normalised copy of the O(L^2)-wide concatenated slab is never written.
        # HDU_FUSE_PW=0 restores the materialised operand of rounds 1-3 (A/B runs).
        self.fuse_pw = os.environ.get("HDU_FUSE_PW", "1") == "1"
        # filter gradients deferred to the end of the backward pass and run as ONE launch per kernel family
        # (ops.WgradPlan); bf16 + materialised inputs only; off under depth sharding / bucketed data parallelism
        self.batch_wgrad = os.environ.get("HDU_BATCH_WGRAD", "1") == "1"
        self.wgrad_plan = None
        # Round 6: float32 networks in a split-bf16 mode (lib.set_f32_contraction "bf16x3" / "bf16x3_bwd") take their filter
        # gradients from bf16 hi / lo image triples written once per backward pass (ops.Split3Plan) and contracted by the bf16
        # batched filter-gradient kernels on 3 N images (_build_split_wgrad_plan).  HDU_F32_SPLIT_WGRAD=0: the in-kernel split of
        # conv_wgrad_kernel<float> (rounds 4-5) for an A/B.
        self.split_wgrad = os.environ.get("HDU_F32_SPLIT_WGRAD", "1") == "1"
        self._split_plan = None      # (Split3Plan, WgradPlan, [(descriptor, dw) of the layers with their own kernel])
        self._split_now = False
        # batch statistics of a conv output taken in the conv's epilogue (hdu_conv_desc.stats_*) instead of a separate
        # reduction pass; the StatsOp then only runs hdu_bn_stats_finalize over the 32 slot rows
        self.epilogue_stats = os.environ.get("HDU_EPILOGUE_STATS", "1") == "1"
        self.stats_sinks = []
        self.stat_roots = []        # tensors with batch moments: their statistics shifts are refreshed at every step head
        self.stats_acc = None
        self.finalized = False
        # per-step accumulators live in ONE arena cleared by one launch (ops.ZeroPlan / hdu_zero_regions)
        self.arena = None
        self.loss_layers = []
        self._grad_zero = []           # slab gradient buffers whose first writer covers only part of the channels
        self._grad_zero_ids = set()
        self._zp_arena = self._zp_bwd = self._zp_step = None
        self._zp_keep = []
        self._zeroed_fwd_pass = -1     # pass whose forward accumulators are already clear
        self._zeroed_bwd_pass = -1
        self._zeroed_grad_pass = -1

    # ---------------- parameters
    def add_param(self, layer, kind, idx, keras_shape, shape, trainable, init, meta=None):
        p = Param(self, layer, idx, kind, keras_shape, shape, trainable, init, meta)
        self.params.append(p)
        self.by_layer.setdefault(layer, []).append(p)
        self.layer_kind[layer] = kind
        return p

    def need_ws(self, M, C):
        self.ws_bytes = max(self.ws_bytes, ops.reduce_ws_bytes(M, C))

    def scratch(self, slot, N, D, H, W, C):
        """shared backward scratch (stream-ordered reuse): one buffer per slot sized to the largest request"""
        n = N * D * H * W * C
        ent = self._scratch.setdefault(slot, {"n": 0, "buf": None})
        ent["n"] = max(ent["n"], n)
        return lambda: ops.Act(ent["buf"], 0, N, D, H, W, C, C, self.dtype)

    def new_var(self, N, D, H, W, C, ld=None):
        v = Var(self, ops.Act.alloc(N, D, H, W, C, self.dtype, ld=ld, zero=True))
        self.vars.append(v)
        return v

    def fvec(self, C, fill=0.0):
        return torch.full((C,), fill, dtype=torch.float32, device=self.dev)

    def finalize(self, seed=4321):
        order = [p for p in self.params if p.trainable] + [p for p in self.params if not p.trainable]
        off = 0
        for p in order:
            p.offset = off
            off += (p.numel + 3) // 4 * 4   # keep every parameter 16-byte aligned
            if p.trainable:
                self.n_trainable = off
        self.P = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.G = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.V = torch.zeros(max(self.n_trainable, 4), dtype=torch.float32, device=self.dev)
        self.ws = ops.Workspace(self.ws_bytes)
        tdt = torch.bfloat16 if self.dtype == HDU_BF16 else torch.float32
        for ent in self._scratch.values():
            ent["buf"] = torch.zeros(ent["n"], dtype=tdt, device=self.dev)
        # compute-dtype filter copies
        tot = 0
        for cv in self.convs:
            cv.wf_off = tot
            tot += cv.kernel.numel
            if cv.need_dgrad_filter:
                cv.wd_off = tot
                tot += cv.kernel.numel
        self.Wc = torch.zeros(max(tot, 8), dtype=tdt, device=self.dev)
        self.init_weights(seed)
        n_stats = 0
        for st in self.stats_sinks:
            st.acc_off = n_stats
            n_stats += st.SLOTS * 2 * st.var.C
        n_bnb = 0
        for cv in self.convs:
            if cv.bnb_fused:
                cv.bnb_off = n_bnb
                n_bnb += cv.BNB_SLOTS * 2 * cv.bn.C
                r = cv.x.root
                if cv.bn.mode == "batch" and r.corr_off is None:
                    r.corr_off = self._corr_total
                    self._corr_total += 2 * r.act.ld
        # slot tables of the two-launch BN backward (hdu_bn_bwd_fused): every BN that will need its sums
        n_bsum = 0
        bsum_off = {}
        if self.bn_bwd_fused and self.grad_enabled:
            for bn in self.bns:
                if bn.mode == "batch" or bn.any_trainable():
                    bsum_off[bn] = n_bsum
                    n_bsum += bn.BSUM_SLOTS * 2 * bn.C
        # arena layout: [loss sums | epilogue statistics | fused-BN-backward slot rows | deferred corrections | BN-backward
        # slot tables], every part a multiple of 4 floats so that the parts stay 16-byte aligned
        up4 = lambda n: (n + 3) // 4 * 4
        o_loss, n_loss = 0, 4 * max(1, len(self.loss_layers))
        o_stats = o_loss + n_loss
        o_bnb = o_stats + up4(n_stats)
        o_corr = o_bnb + up4(n_bnb)
        o_bsum = o_corr + up4(self._corr_total)
        total = o_bsum + up4(n_bsum)
        self.arena = torch.zeros(total, dtype=torch.float32, device=self.dev)
        for i, ll in enumerate(self.loss_layers):
            ll.loss_sum = self.arena[4 * i:4 * i + 1]
            ll.class_count = self.arena[4 * i + 1:4 * i + 4]
        self.stats_acc = self.arena[o_stats:o_stats + n_stats] if n_stats else None
        self.bnb_acc = self.arena[o_bnb:o_bnb + n_bnb] if n_bnb else None
        self.corr_acc = self.arena[o_corr:o_corr + self._corr_total] if self._corr_total else None
        for bn, o in bsum_off.items():
            bn.bsum = self.arena[o_bsum + o:o_bsum + o + bn.BSUM_SLOTS * 2 * bn.C]
        self._shift_copies = [(r.shift, r.mean) for r in self.stat_roots] if self.stats_sinks else []
        self._zp_arena = ops.ZeroPlan([self.arena], self._shift_copies)
        self._zp_bwd = ops.ZeroPlan([self.arena[o_bnb:total]]) if total > o_bnb else None
        for cv in self.convs:
            cv.bind()
        self._build_prep_table()
        self._build_wgrad_plan()
        self.finalized = True

    def _build_wgrad_plan(self):
        self.wgrad_plan = None
        for cv in self.convs:
            cv.in_plan = False
        if not (self.batch_wgrad and self.dtype == HDU_BF16 and self.grad_enabled and
                (self.shard is None or self.shard.world == 1)):
            return
        plan = ops.WgradPlan(int(os.environ.get("HDU_BATCH_WGRAD_TARGET", "0")))
        for cv in self.convs:
            if not (cv.trainable and cv.out.root.needs_grad):
                continue
            if cv.xin is not None:          # materialised input
                d = self.conv_desc(cv.xin.act, cv.wf_ptr, cv.out.grad, cv.K, cv.stride, cv.pad, cv.conv_up)
            elif cv.bn is None and cv.skip is None:      # the conv reads its producer directly (stems, 1x1 heads)
                d = self.conv_desc(cv.x.act, cv.wf_ptr, cv.out.grad, cv.K, cv.stride, cv.pad, cv.up)
            elif cv.pw_fused:                            # pointwise over relu(a * x + b): recomputed on the x fragments
                d = self.conv_desc(cv.x.act, cv.wf_ptr, cv.out.grad, cv.K, cv.stride, cv.pad, cv.up, None,
                                  (cv.bn.a, cv.bn.b), cv.bn.relu)
            else:
                continue                    # fused prologue: per-layer launch
            if ops.conv_kernel_name(d, 1) == "conv_stem_wgrad_kernel":
                continue                    # the 7 x 7 (x 7) stem has its own filter-gradient kernel (round 5): per-layer launch
            plan.add(d, cv.kernel.grad)
            cv.in_plan = True
        if len(plan):
            plan.finalize()
            self.wgrad_plan = plan

    def split_wgrad_active(self):
        """do the filter gradients of this pass run on the bf16 image triples? (float32 storage, a split contraction mode, one
        device, deferred filter gradients allowed)"""
        return bool(self.dtype == HDU_F32 and self.split_wgrad and self.batch_wgrad and self.grad_enabled and self.finalized
                    and ops._l.f32_contraction() != "exact" and (self.shard is None or self.shard.world == 1))

    def _build_split_wgrad_plan(self):
        """x = hi + lo in bfloat16: dW ~ dyh.xh + dyh.xl + dyl.xh, and three products contracted over the pixels are ONE filter
        gradient over three times the images -- operand triple (hi, lo, hi), gradient triple (hi, hi, lo), both written by one
        table-driven launch at the END of the backward pass (every operand and output gradient of the pass is final and still
        stored then, exactly what the bf16 networks' deferred plan relies on), followed by the bf16 plan's one launch per kernel
        family.  conv_wgrad_kernel<float> split its operands per consuming wave (80 TF, 21 of the 56 ms of a 2D step); the bf16
        kernels run at 0.7-1.1 PF.  Layers taken: the ones _build_wgrad_plan takes, with channel counts that are whole bf16 chunks
        (the 3-channel stem input and the 3-class head keep the float32 kernel)."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the split filter-gradient plan is built by the first EAGER backward pass in a split-bf16 mode: run one "
                               "step (or call Ctx._build_split_wgrad_plan) before capturing")
        sp, plan, own = ops.Split3Plan(), ops.WgradPlan(int(os.environ.get("HDU_BATCH_WGRAD_TARGET", "0"))), []
        keep = []
        for cv in self.convs:
            cv.in_split_plan = False
            if not (cv.trainable and cv.out.root.needs_grad) or cv.halo or cv.cin_p % 8 or cv.cout_p % 8:
                continue
            if cv.xin is not None:          # materialised input
                src, pro, relu, up = cv.xin.act, None, False, cv.conv_up
            elif cv.bn is None and cv.skip is None:
                src, pro, relu, up = cv.x.act, None, False, cv.up
            elif cv.pw_fused:               # pointwise over relu(a * x + b): applied (in float32) by the split launch
                src, pro, relu, up = cv.x.act, (cv.bn.a, cv.bn.b), cv.bn.relu, cv.up
            else:
                continue
            g = cv.out.grad
            xs = ops.Act.alloc(3 * src.N, src.D, src.H, src.W, src.C, HDU_BF16)
            dys = ops.Act.alloc(3 * g.N, g.D, g.H, g.W, g.C, HDU_BF16)
            d16 = ops.conv_desc(xs, cv.wf_ptr, dys, cv.K, cv.stride, cv.pad, up)
            sp.add(src, ops._l.SPLIT3_OPERAND, xs, pro, relu)
            sp.add(g, ops._l.SPLIT3_GRADIENT, dys)
            if ops.conv_kernel_name(d16, 1) == "conv_stem_wgrad_kernel":
                own.append((d16, cv.kernel.grad))
            else:
                plan.add(d16, cv.kernel.grad)
            keep.append((xs, dys))
            cv.in_split_plan = True
        if len(sp):
            sp.finalize()
        if len(plan):
            plan.finalize()
        self._split_plan = (sp, plan, own, keep)

    def flush_pending_finalize(self):
        """the finalize of a fused batch-statistics BN backward is held back until the next reader of its tensor's gradient, so
        that both run as one launch (Var.dy); whatever is still pending runs on its own here"""
        pend, self._pending_fin = self._pending_fin, None
        if pend is not None:
            (part, slots, M, C, g, be, sg, mean, rstd, dg, db, dsg, dsb, c3, c4) = pend["args"]
            ops.bn_bwd_finalize(part, slots, M, C, True, g, be, sg, mean, rstd, dg, db, dsg, dsb, c3, c4)

    def unprime_stats(self):
        """new weights: the stored means are no longer a good shift for the one-pass epilogue moments"""
        for st in self.stats_sinks:
            st.primed = False

    def set_batch_wgrad(self, on):
        """(re)build or drop the deferred filter-gradient plan (dropped e.g. for bucketed data parallelism, where a
        gradient bucket must be complete when its backward segment ends)"""
        self.batch_wgrad = on
        if self.finalized:
            self._build_wgrad_plan()

    def init_weights(self, seed):
        """Keras initialisers (K.initializers.py): glorot_uniform for convs (:332), 'normal' = RandomNormal(0.05)
        (:72,:429) where the reference asks for it; BN gamma=1 beta=0 mean=0 var=1; Scale gamma=1 beta=0."""
        rng = np.random.default_rng(seed)
        for layer, ps in self.by_layer.items():
            arrs = []
            for p in ps:
                if p.init == "zeros":
                    a = np.zeros(p.keras_shape, np.float32)
                elif p.init == "ones":
                    a = np.ones(p.keras_shape, np.float32)
                elif p.init == "normal":
                    a = rng.normal(0.0, 0.05, p.keras_shape).astype(np.float32)
                else:
                    ks = p.keras_shape
                    fan_in = int(np.prod(ks[:-1]))
                    fan_out = int(np.prod(ks[:-2])) * ks[-1]
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    a = rng.uniform(-lim, lim, ks).astype(np.float32)
                arrs.append(a)
            self.set_layer_weights(layer, arrs)

    # ---------------- Keras-shaped weight exchange (topology.py:2847-2873 ordering per layer)
    def _to_internal(self, p, a):
        a = np.asarray(a, np.float32)
        if a.shape != p.keras_shape:
            raise ValueError("layer %s weight %d: expected shape %s, got %s" % (p.layer, p.idx, p.keras_shape, a.shape))
        if p.kind == "conv_kernel":
            nd = a.ndim - 2
            cin, cout = a.shape[-2], a.shape[-1]
            cout_p, kd, kh, kw, cin_p = p.shape
            if nd == 2:
                k = a.transpose(3, 0, 1, 2)[:, None]            # (Cout,1,kh,kw,Cin)
            else:
                k = a.transpose(4, 2, 0, 1, 3)                   # Keras (kH,kW,kD,Cin,Cout) -> (Cout,kD,kH,kW,Cin)
            out = np.zeros(p.shape, np.float32)
            out[:cout, :, :, :, :cin] = k
            return out
        if p.kind == "conv_bias":
            out = np.zeros(p.shape, np.float32)
            out[:a.shape[0]] = a
            return out
        return a

    def _to_keras(self, p, t):
        a = t.detach().float().cpu().numpy().reshape(p.shape)
        if p.kind == "conv_kernel":
            nd = len(p.keras_shape) - 2
            cin, cout = p.keras_shape[-2], p.keras_shape[-1]
            k = a[:cout, :, :, :, :cin]
            if nd == 2:
                return np.ascontiguousarray(k[:, 0].transpose(1, 2, 3, 0))
            return np.ascontiguousarray(k.transpose(2, 3, 1, 4, 0))
        if p.kind == "conv_bias":
            return a[:p.keras_shape[0]].copy()
        return a.copy()

    def set_layer_weights(self, layer, arrs):
        ps = self.by_layer[layer]
        if len(arrs) != len(ps):
            raise ValueError("layer %s expects %d weight arrays, got %d" % (layer, len(ps), len(arrs)))
        for p, a in zip(ps, arrs):
            t = torch.from_numpy(self._to_internal(p, a).reshape(-1)).to(self.dev)
            self.P[p.offset:p.offset + p.numel] = t

    def get_layer_weights(self, layer):
        return [self._to_keras(p, p.data) for p in self.by_layer[layer]]

    def get_layer_grads(self, layer):
        return [self._to_keras(p, p.grad) for p in self.by_layer[layer] if p.trainable]

    # ---------------- step pieces
    def step_zero(self):
        """head of a training step: ONE launch clears the accumulator arena, the flat gradient buffer (the filter-gradient
        kernels add into it) and the slab gradient buffers with a partial first writer, and advances the dropout seed"""
        if self._zp_step is None:
            self._zp_step = ops.ZeroPlan([self.arena, self.G[:self.n_trainable]] + self._grad_zero, self._shift_copies)
            self._zp_keep.append(self._zp_step)          # a captured graph may still hold an older table
        self._zp_step.run(self.seed_dev, 1)
        self._zeroed_fwd_pass = self._zeroed_bwd_pass = self.pass_id + 1     # the pass run_forward is about to start
        self._zeroed_grad_pass = self.pass_id + 1

    def zero_partial_grad(self, buf):
        key = buf.data_ptr()
        if key in self._grad_zero_ids and self._zeroed_grad_pass == self.pass_id:
            return                                        # cleared by this step's step_zero launch
        ops.zero_tensor(buf)
        if key not in self._grad_zero_ids:
            self._grad_zero_ids.add(key)
            self._grad_zero.append(buf)
            self._zp_step = None                          # rebuilt (with this buffer) at the next step head

    def prep_weights(self):
        """float32 master filters -> compute-dtype forward / data-gradient copies, all layers in one launch"""
        if self._prep_n:
            ops.weight_prep_batched(self.dtype, self._prep_table, self._prep_n, self._prep_tiles, self.P, self.Wc)

    def _build_prep_table(self):
        from .lib import PrepEntry
        ents = []
        tiles = 0
        for cv in self.convs:
            wf = -1 if self.dtype == HDU_F32 else cv.wf_off
            wd = cv.wd_off if cv.need_dgrad_filter else -1
            if wf < 0 and wd < 0:
                continue
            ents.append(PrepEntry(cv.kernel.offset, wf, wd, tiles, cv.cout_p, cv.T, cv.cin_p, 0))
            tiles += cv.T * ((cv.cout_p + 63) // 64) * ((cv.cin_p + 63) // 64)      # include/hdu.h HDU_PREP_TILE
        self._prep_n = len(ents)
        self._prep_tiles = tiles
        if ents:
            arr = (PrepEntry * len(ents))(*ents)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self._prep_table = torch.from_numpy(raw).to(self.dev)

    def _fold_plan(self, phase):
        if phase not in self._fold_plans:
            ents = []
            for bn in self.bns:
                if bn.mode == "batch" and phase == 1:
                    continue
                ents.append((bn.C, bn.eps, bn.mm.data, bn.mv.data, bn.gamma.data, bn.beta.data,
                             bn.sg.data if bn.sg else None, bn.sb.data if bn.sb else None, bn.a, bn.b, bn.rstd))
            self._fold_plans[phase] = ops.FoldPlan(ents) if ents else None
        return self._fold_plans[phase]

    def shard_world(self):
        return self.shard.world if (self.shard is not None and self.shard.world > 1) else 1

    def conv_desc(self, *a, **k):
        """ops.conv_desc for a layer of THIS model: a depth shard computes 1 / world of every layer and decides like the whole layer"""
        return ops.conv_desc(*a, shard_world=self.shard_world(), **k)

    def run_forward(self):
        self.pass_id += 1
        if self.batch_fold and self.finalized:
            plan = self._fold_plan(self.learning_phase)
            if plan is not None:
                plan.run()
                self.prefolded_pass = self.pass_id
        if self.learning_phase == 1 and self._zeroed_fwd_pass != self.pass_id and self.stats_acc is not None:
            self._zp_arena.run()            # (a training step cleared everything in its step-head launch: step_zero)
        for f in self.fwd:
            f()

    def bnb_fusable(self):
        """do the data-gradient launches of the pass carry their consumer BN's backward? (not under depth sharding: the halo
        planes of dz are reduced across ranks before the BN backward)"""
        return self.fuse_bn_bwd and (self.shard is None or self.shard.world == 1)

    def run_backward(self, seg=None):
        """the whole backward pass, or positions [seg[0], seg[1]) of it (in execution order) -- see grad_buckets"""
        lo, hi = seg if seg is not None else (0, len(self.bwd))
        if lo == 0:
            for v in self.vars:
                v.written = False
            self.fuse_bn_bwd_now = self.bnb_fusable()
            if self._zeroed_bwd_pass != self.pass_id and self._zp_bwd is not None:
                self._zp_bwd.run()         # (a training step's head launch has already cleared the whole arena)
                self._zeroed_bwd_pass = self.pass_id
        order = list(reversed(self.bwd))
        if lo == 0:
            self._bnb_deferred = []
            self._split_now = self.split_wgrad_active()
            if self._split_now and self._split_plan is None:
                self._build_split_wgrad_plan()
        # float32 networks in the "bf16x3_bwd" mode (lib.set_f32_contraction): the forward contracts in exact float32, the data and
        # filter gradients of this pass with the split-bf16 contraction (read by the library at launch time)
        split_bwd = self.dtype == HDU_F32 and ops._l.f32_split_in_backward_only()
        if split_bwd:
            ops._l.set_f32_split_now(True)
        try:
            for f in order[lo:hi]:
                f()
            self.flush_pending_finalize()
            if hi == len(self.bwd) and self._bnb_deferred:
                keys = tuple(k for k, _ in self._bnb_deferred)
                if self._bnb_plan is None or self._bnb_plan[0] != keys:       # (the set is fixed by the model: built once)
                    self._bnb_plan = (keys, ops.BnBwdPlan([e for _, e in self._bnb_deferred]))
                self._bnb_plan[1].run()
            if hi == len(self.bwd) and self.wgrad_plan is not None:
                self.wgrad_plan.run()
            if hi == len(self.bwd) and self._split_now:
                sp, plan, own, _ = self._split_plan
                sp.run()
                if len(plan):
                    plan.run()
                for d16, dw in own:
                    ops.conv_wgrad(d16, dw)
        finally:
            if split_bwd:
                ops._l.set_f32_split_now(False)

    def grad_buckets(self, fractions):
        """Cut the backward pass where the first-completed `fractions` of the trainable parameters have their final
        gradient (layers own their parameters, parameters are laid out in creation order, the backward runs the
        layers in reverse): returns [(pos_lo, pos_hi, off_lo, off_hi)] in execution order -- after positions
        [pos_lo, pos_hi) of the backward have run, G[off_lo:off_hi] is complete and can be exchanged while the rest
        of the backward computes."""
        n = len(self.bwd)
        tr = [p for p in self.params if p.trainable]
        index_of = {id(p): i for i, p in enumerate(self.params)}
        total = sum(p.numel for p in tr)
        cuts = []          # (position in execution order, flat offset): everything at offsets >= off is complete
        for f in sorted(fractions):
            want, done, pos, off = f * total, 0, n, self.n_trainable
            # walking the backward from the last layer: after bwd[k] ran, parameters created after mark[k-1] are done
            for k in range(n - 1, -1, -1):
                first = self.bwd.mark[k - 1] if k > 0 else 0
                own = [p for p in tr if first <= index_of[id(p)]]
                done = sum(p.numel for p in own)
                if done >= want:
                    pos, off = n - k, min([p.offset for p in own] + [self.n_trainable])
                    break
            if 0 < pos < n and (not cuts or pos > cuts[-1][0]):
                cuts.append((pos, off))
        out, prev_pos, prev_off = [], 0, self.n_trainable
        for pos, off in cuts:
            out.append((prev_pos, pos, off, prev_off))
            prev_pos, prev_off = pos, off
        out.append((prev_pos, n, 0, prev_off))
        return out


# ======================================================================================= layers
class BNLayer:
    """BatchNormalization (+ optional Scale) + optional ReLU folded to a per-channel affine that the consumer
    applies on load.  K.layers/normalization.py:126-190, lib/custom_layers.py:63-69."""
    BSUM_SLOTS = int(os.environ.get("HDU_BSUM_SLOTS", "16"))      # slot rows the backward reduction spreads its float atomics over (workgroup % slots; <= 32)

    def __init__(self, ctx, name, C, eps=1e-3, momentum=0.99, mode="batch", trainable=True, scale_name=None,
                 scale_trainable=True, relu=True, scale=None):
        """scale: the custom_layers.Scale that follows this BatchNormalization in the reference graph (its name and trainable
        flag are taken from it; scale_name / scale_trainable are the same two facts given directly)"""
        if scale is not None:
            scale_name, scale_trainable = scale.name, scale.trainable
        self.scale_layer = scale
        self.ctx, self.name, self.C, self.eps, self.momentum, self.mode, self.relu = ctx, name, C, eps, momentum, mode, relu
        self.trainable = trainable
        self.gamma = ctx.add_param(name, "bn", 0, (C,), (C,), trainable, "ones")
        self.beta = ctx.add_param(name, "bn", 1, (C,), (C,), trainable, "zeros")
        self.mm = ctx.add_param(name, "bn", 2, (C,), (C,), False, "zeros")
        self.mv = ctx.add_param(name, "bn", 3, (C,), (C,), False, "ones")
        self.sg = self.sb = None
        self.scale_trainable = scale_trainable
        if scale_name:
            self.sg = ctx.add_param(scale_name, "scale", 0, (C,), (C,), scale_trainable, "ones")
            self.sb = ctx.add_param(scale_name, "scale", 1, (C,), (C,), scale_trainable, "zeros")
            if scale is not None:
                scale.gamma, scale.beta = self.sg, self.sb
        v = ctx.fvec
        self.a, self.b, self.rstd = v(C), v(C), v(C)
        self.s1, self.s2, self.k1, self.k2, self.k3 = v(C), v(C), v(C), v(C), v(C)
        self.mean_used = None
        self.batch_now = False
        self.folded_pass = -1
        ctx.bns.append(self)
        self.s12 = v(2 * C)
        self.stats_src = None       # the StatsOp whose finalize folds this BN (fuse / then_fold)
        self.pending_stats = None   # (pass, StatsOp): the consumer's materialize pass of this pass folds it (hdu_materialize_stats)
        self.bsum = None            # [BSUM_SLOTS][2][C] slot table of hdu_bn_bwd_fused, a slice of Ctx.arena (finalize)
        self._bsum_pass = -1

    def any_trainable(self):
        return self.trainable or (self.sg is not None and self.scale_trainable)

    def needs_stats(self):
        return self.mode == "batch"

    def fold(self, xvar):
        """a/b/rstd of this pass (nothing to do when the statistics reduction or the batched launch already did it)"""
        ctx = self.ctx
        sg = self.sg.data if self.sg else None
        sb = self.sb.data if self.sb else None
        self.batch_now = self.mode == "batch" and ctx.learning_phase == 1
        if self.batch_now and self.folded_pass == ctx.pass_id:
            return None  # the statistics reduction of this pass already folded this BN (StatsOp.fused)
        if not self.batch_now and ctx.prefolded_pass == ctx.pass_id:
            self.mean_used = self.mm.data
            return None  # folded with every other inference-mode BN by the batched launch at the head of this pass
        if self.batch_now:
            mean, var = xvar.stats()
            self.mean_used = mean
            args = (mean, var, self.gamma.data, self.beta.data, self.eps, sg, sb, self.a, self.b, self.rstd,
                    self.mm.data, self.mv.data, self.momentum)
        else:
            self.mean_used = self.mm.data
            args = (self.mm.data, self.mv.data, self.gamma.data, self.beta.data, self.eps, sg, sb, self.a, self.b,
                    self.rstd, None, None, self.momentum)
        ops.bn_fold(self.C, *args)
        return None

    def takes_epilogue_sums(self, xvar):
        """will backward() run the two-launch form (slot table + apply) in this pass?  Then the data-gradient launch that produces
        dz may fill the slot table in its epilogue (hdu_conv_desc.bnb_relu bit 2) and the reduction launch goes (round 6)."""
        ctx = self.ctx
        return bool((self.batch_now or self.any_trainable()) and not (ctx.shard is not None and ctx.shard.world > 1)
                    and self.bsum is not None and xvar.root.needs_grad and ctx._zeroed_bwd_pass == ctx.pass_id
                    and self._bsum_pass != ctx.pass_id)

    def backward(self, xvar, dz_act, sums_ready=False):
        """dz: gradient w.r.t. relu(a*x+b) at x's resolution.  Writes x.grad and the parameter gradients.
        sums_ready: the launch that produced dz already added S1 / S2 to this BN's slot table (takes_epilogue_sums)."""
        ctx = self.ctx
        x = xvar.act
        need_sums = self.batch_now or self.any_trainable()
        tr_bn = self.trainable
        tr_sc = self.sg is not None and self.scale_trainable
        if ctx.shard is not None and ctx.shard.world > 1 and need_sums:
            # depth-sharded: parameter gradients from the LOCAL sums (the flat gradient all-reduce adds the ranks),
            # dx coefficients from the GLOBAL sums over all shards
            ops.bn_bwd_reduce(dz_act, x, self.a, self.b, self.relu, self.mean_used, self.rstd, self.s1, self.s2, ctx.ws)
            ops.bn_bwd_coef(self.C, x.M, False, self.s1, self.s2, self.gamma.data, self.beta.data,
                            self.sg.data if self.sg else None, self.rstd, self.k1, self.k2, self.k3,
                            self.gamma.grad if tr_bn else None, self.beta.grad if tr_bn else None,
                            self.sg.grad if tr_sc else None, self.sb.grad if tr_sc else None)
            if self.batch_now:
                self.s12[:self.C].copy_(self.s1)
                self.s12[self.C:].copy_(self.s2)
                _sh.allreduce_sum(ctx.shard, self.s12)
                ops.bn_bwd_coef(self.C, x.M * ctx.shard.world, True, self.s12[:self.C], self.s12[self.C:],
                                self.gamma.data, self.beta.data, self.sg.data if self.sg else None, self.rstd, self.k1,
                                self.k2, self.k3)
        elif (need_sums and self.bsum is not None and xvar.root.needs_grad and ctx._zeroed_bwd_pass == ctx.pass_id
              and self._bsum_pass != ctx.pass_id):
            # reduction into the (zeroed) slot table, then coefficients + parameter gradients + dx in ONE launch: no finalize
            self._bsum_pass = ctx.pass_id
            acc = xvar.grad_mode()
            drop = xvar.root.drop if (ctx.dropout_enabled and ctx.learning_phase == 1) else None
            ops.bn_bwd_fused(dz_act, x, self.a, self.b, self.relu, self.mean_used, self.rstd, self.batch_now,
                             self.gamma.data, self.beta.data, self.sg.data if self.sg else None, self.bsum, self.BSUM_SLOTS,
                             self.gamma.grad if tr_bn else None, self.beta.grad if tr_bn else None,
                             self.sg.grad if tr_sc else None, self.sb.grad if tr_sc else None, xvar.grad, acc,
                             drop[0] if drop else 1.0, drop[1] if drop else 0, ctx.seed_dev if drop else None,
                             sums_ready=sums_ready)
            return
        elif need_sums:   # reduction + coefficients + parameter gradients: two launches
            assert not sums_ready
            ops.bn_bwd_reduce_coef(dz_act, x, self.a, self.b, self.relu, self.mean_used, self.rstd, self.batch_now,
                                   self.gamma.data, self.beta.data, self.sg.data if self.sg else None, self.s1,
                                   self.s2, self.k1, self.k2, self.k3, self.gamma.grad if tr_bn else None,
                                   self.beta.grad if tr_bn else None, self.sg.grad if tr_sc else None,
                                   self.sb.grad if tr_sc else None, ctx.ws)
        else:
            ops.bn_bwd_coef(self.C, x.M, False, None, None, self.gamma.data, self.beta.data,
                            self.sg.data if self.sg else None, self.rstd, self.k1, self.k2, self.k3)
        if xvar.root.needs_grad:
            acc = xvar.grad_mode()
            drop = xvar.root.drop if (ctx.dropout_enabled and ctx.learning_phase == 1) else None
            ops.bn_bwd_apply(dz_act, x, self.a, self.b, self.relu, self.mean_used, self.k1, self.k2, self.k3,
                             xvar.grad, acc, drop[0] if drop else 1.0, drop[1] if drop else 0,
                             ctx.seed_dev if drop else None)


class ConvLayer:
    """[BN(+Scale)+ReLU] -> [UpSampling] -> [+skip] -> [ZeroPadding] -> Conv(+bias)(+Dropout) as ONE launch."""

    def __init__(self, ctx, name, x, filters, K, stride=(1, 1, 1), pad=(0, 0, 0), bn=None, up=(0, 0, 0), skip=None,
                 use_bias=True, out=None, init="glorot", trainable=True, dropout=0.0, keras_nd=2, cin_logical=None,
                 halo=0, producer=None):
        """halo > 0 (depth sharding): the conv input buffer carries `halo` extra depth planes on both sides, filled
        from the depth neighbours before the launch; the padding on the depth axis shrinks accordingly.
        producer: the ConvLayer whose output `x` is, when NOTHING else reads that output (the model builder's promise:
        a bottleneck's 1x1 -> BN -> ReLU -> 3x3).  Whenever `bn` runs on stored statistics and no gradient flows through
        it (every predict; the frozen 2D branch of the 3dpart hybrid), the producer applies bn + ReLU in its own
        epilogue and writes this conv's operand directly (hdu_conv_desc.epi_*): one full-width pass less."""
        self.ctx, self.name, self.x, self.bn, self.up, self.skip = ctx, name, x, bn, up, skip
        self.K, self.stride = K, stride
        self.halo = halo
        if halo:
            assert not ctx.fuse_prologue, "halo mode materialises the conv input"
            assert skip is None or up == (0, 0, 0), "a skip add in halo mode must be at the stored resolution"
            pad = (pad[0] - (halo << up[0]), pad[1], pad[2])
        self.pad = pad
        self.trainable = trainable
        dt = ctx.dtype
        xa = x.act
        cin_p = xa.C
        cin = cin_logical or cin_p
        cout_p = ops.cpad(filters, dt)
        De, He, We = (xa.D + 2 * halo) << up[0], xa.H << up[1], xa.W << up[2]
        Do = (De + 2 * pad[0] - K[0]) // stride[0] + 1
        Ho = (He + 2 * pad[1] - K[1]) // stride[1] + 1
        Wo = (We + 2 * pad[2] - K[2]) // stride[2] + 1
        if out is None:
            out = ctx.new_var(xa.N, Do, Ho, Wo, cout_p)
        else:
            assert (out.act.N, out.act.D, out.act.H, out.act.W, out.act.C) == (xa.N, Do, Ho, Wo, cout_p)
        self.out = out
        kshape = ((K[1], K[2]) if keras_nd == 2 else (K[1], K[2], K[0])) + (cin, filters)
        self.kernel = ctx.add_param(name, "conv_kernel", 0, kshape, (cout_p, K[0], K[1], K[2], cin_p), trainable, init)
        self.bias = ctx.add_param(name, "conv_bias", 1, (filters,), (cout_p,), trainable, "zeros") if use_bias else None
        ctx.layer_kind[name] = "conv"
        self.T = K[0] * K[1] * K[2]
        self.cin_p, self.cout_p = cin_p, cout_p
        self.dropout = dropout
        if dropout > 0:
            ctx.drop_layers += 1
            self.drop_seed = 7919 * ctx.drop_layers
            out.root.drop = (1.0 - dropout, self.drop_seed)
        need_input_grad = x.root.needs_grad
        if ctx.grad_enabled:
            out.require_grad()
        # Operand preparation.  Fused mode: the BN/ReLU/upsample/skip prologue runs inside the conv's operand gather
        # (nothing materialised).  Default: one streaming hdu_materialize pass writes the conv input so that the conv
        # is a pure async-DMA implicit GEMM -- at the narrow output widths of this net (Cout 32..192) the fused
        # prologue costs more VALU cycles than the MFMAs it feeds (profiles/).
        self.xin = None
        self.conv_up = up
        self.pw_fused = bool(ctx.fuse_pw and bn is not None and skip is None and not halo and K == (1, 1, 1)
                             and stride == (1, 1, 1) and pad == (0, 0, 0) and up == (0, 0, 0) and cin_p <= ops.PRO_CMAX)
        if self.pw_fused:
            pass                      # no operand buffer: hdu_conv_desc.pro_* on the DMA kernels
        elif halo:
            self.xin = ctx.new_var(xa.N, xa.D + 2 * halo, xa.H, xa.W, cin_p)
            plane = xa.H * xa.W * cin_p
            self.xin_interior = ops.Act(self.xin.act.buf, halo * plane, xa.N, xa.D, xa.H, xa.W, cin_p, cin_p, dt)
            self._halo_tmp = ctx.scratch("halo_tmp", 1, 2 * halo, xa.H, xa.W, cin_p)
        elif (bn is not None or skip is not None) and not ctx.fuse_prologue:
            if skip is not None:
                self.xin = ctx.new_var(xa.N, De, He, We, cin_p)
                self.conv_up = (0, 0, 0)
            else:
                self.xin = ctx.new_var(xa.N, xa.D, xa.H, xa.W, cin_p)
        st = getattr(bn, "stats_src", None) if bn is not None else None
        if (st is not None and self.xin is not None and not halo and st.producer is not None and st.absorber is None
                and x.root is st.var.root and bn.C == x.C and x.c0 == (0 if bn is st.fold_next else st.var.c0)):
            st.absorber = self
        self.need_input_grad = need_input_grad
        self.epi_consumer = None          # set by the conv that consumes self.out through a foldable BN (see `producer`)
        self.epi_producer = None
        if (producer is not None and ctx.fuse_bn_epilogue and bn is not None and self.xin is not None and not halo
                and skip is None and up == (0, 0, 0) and x is producer.out and x.root is x and producer.epi_consumer is None):
            self.epi_producer = producer
            producer.epi_consumer = self
        self.strided = stride != (1, 1, 1)
        # the consumer BN's backward runs in the epilogue of this layer's data-gradient launch when dz IS that launch's
        # output: no up-sampling in between, no skip add, no depth halo, no dropout on the BN input
        # Round 4: the bottleneck data gradients (dz[M x Cin] = dt[M x 128|192] . W^T, Cin >= 256) run on the filter-stationary
        # STREAMING kernel, whose tile epilogue takes the BN backward at no extra pass (conv_pw_bstat_kernel<.., BNB>): there the
        # fusion pays for batch-statistics BNs too -- the dz round trip and the reduction + apply passes over the O(L^2)-wide slab
        # go, the deferred part of du follows as a correction over the producer's own channels (HDU_FUSE_BN_BWD_PW=0: off).
        pw_stream = bool(ctx.fuse_bn_bwd_pw and dt == HDU_BF16 and K == (1, 1, 1) and pad == (0, 0, 0) and cout_p in (128, 192)
                         and cin_p >= 256 and xa.M >= 64)
        self.bnb_fused = bool(ctx.fuse_bn_bwd and need_input_grad and bn is not None and up == (0, 0, 0) and skip is None
                              and not halo and stride == (1, 1, 1) and not ctx.fuse_prologue and x.root.drop is None
                              and (ctx.fuse_bn_bwd_mode >= 2 or bn.mode != "batch" or pw_stream))
        self.bnb_off = None
        self._sums_kernel_ok = None       # does this layer's data gradient run on a tile kernel that can take the BN sums? (_epilogue_sums)
        self._s2 = None
        self._s2_ok = (os.environ.get("HDU_STRIDE2_DGRAD", "1") == "1" and all(v in (1, 2) for v in stride)
                       and xa.D % stride[0] == 0 and xa.H % stride[1] == 0 and xa.W % stride[2] == 0 and up == (0, 0, 0)
                       and bn is None and skip is None and not halo)
        self.need_dgrad_filter = need_input_grad and not self.strided
        self.wf_off = self.wd_off = None
        ctx.convs.append(self)
        if bn is not None:
            ctx.need_ws(xa.M, xa.C)      # statistics and / or the backward reductions over the input
        ctx.need_ws(out.act.M, cout_p)
        if need_input_grad:
            if bn is not None or up != (0, 0, 0) or halo:
                self._dxe = ctx.scratch("dxe", xa.N, De, He, We, cin_p)
            if (bn is not None or halo) and up != (0, 0, 0):
                self._dz = ctx.scratch("dz", xa.N, xa.D + 2 * halo, xa.H, xa.W, cin_p)
        ctx.fwd.append(self.forward)
        ctx.bwd.append(self.backward)

    # ---- bound after Ctx.finalize()
    def bind(self):
        ctx = self.ctx
        esz = ctx.Wc.element_size()
        if ctx.dtype == HDU_F32:
            self.wf_ptr = ctypes.c_void_p(self.kernel.data.data_ptr())  # master IS the f32 forward filter
        else:
            self.wf_ptr = ctypes.c_void_p(ctx.Wc.data_ptr() + self.wf_off * esz)
        self.wd_ptr = ctypes.c_void_p(ctx.Wc.data_ptr() + self.wd_off * esz) if self.need_dgrad_filter else None
        x, out = self.x.act, self.out.act
        pro = (self.bn.a, self.bn.b) if self.bn is not None else None
        relu = self.bn.relu if self.bn is not None else False
        skip = self.skip.act if self.skip is not None else None
        up = self.up
        if self.xin is not None:            # materialised operand: plain conv
            x, pro, relu, skip, up = self.xin.act, None, False, None, self.conv_up
        bias = self.bias.data if self.bias is not None else None
        if self.strided and self.need_input_grad and self._s2_ok and self.out.root.needs_grad:
            # stride-2 data gradient on the MFMA path: 2^d stride-1 implicit GEMMs over the parity classes of the input
            # grid (the scalar gather form took 9.7 of the 27 ms of an end2end step for 0.4 % of its FLOPs)
            xa = self.x.act
            self._s2 = ops.Stride2Dgrad(ctx.dtype, self.kernel.data, self.out.grad, (xa.N, xa.D, xa.H, xa.W), xa.C, self.K,
                                        self.stride, self.pad, shard_world=ctx.shard_world())
        self.d_f = ctx.conv_desc(x, self.wf_ptr, out, self.K, self.stride, self.pad, up, skip, pro, relu, bias)
        self.d_f_drop = None
        if self.dropout > 0:
            self.d_f_drop = ctx.conv_desc(x, self.wf_ptr, out, self.K, self.stride, self.pad, up, skip, pro, relu,
                                          bias, False, 1.0 - self.dropout, self.drop_seed, ctx.seed_dev)
        # variants that apply the consumer's BN(+Scale)+ReLU in the epilogue and write the consumer's operand buffer
        self.d_f_epi = self.d_f_drop_epi = None
        cons = self.epi_consumer
        if cons is not None:
            epi = (cons.bn.a, cons.bn.b, cons.bn.relu)
            self.d_f_epi = ctx.conv_desc(x, self.wf_ptr, cons.xin.act, self.K, self.stride, self.pad, up, skip, pro, relu, bias,
                                         epi=epi)
            if self.dropout > 0:
                self.d_f_drop_epi = ctx.conv_desc(x, self.wf_ptr, cons.xin.act, self.K, self.stride, self.pad, up, skip, pro,
                                                  relu, bias, False, 1.0 - self.dropout, self.drop_seed, ctx.seed_dev, epi=epi)
        # training-phase variants that also accumulate the output's moments for the StatsOp that follows this conv
        self.d_f_st = self.d_f_drop_st = None
        sink = getattr(self, "stats_sink", None)
        if sink is not None:
            shift = sink.var.stats_shift()
            part = ctypes.c_void_p(ctx.stats_acc.data_ptr() + 4 * sink.acc_off)

            def with_stats(d):
                if d is None:
                    return None
                c = type(d)()
                ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(d))
                c.stats_partial, c.stats_shift, c.stats_slots = part, ctypes.c_void_p(shift.data_ptr()), sink.SLOTS
                return c
            self.d_f_st, self.d_f_drop_st = with_stats(self.d_f), with_stats(self.d_f_drop)

    def prep(self):
        ctx = self.ctx
        if ctx.dtype == HDU_F32 and not self.need_dgrad_filter:
            return
        wf = None if ctx.dtype == HDU_F32 else ctx.Wc[self.wf_off:self.wf_off + self.kernel.numel]
        wd = ctx.Wc[self.wd_off:self.wd_off + self.kernel.numel] if self.need_dgrad_filter else None
        ops.weight_prep(ctx.dtype, self.kernel.data, self.cout_p, self.T, self.cin_p, wf, wd)

    def epi_active(self):
        """does the PRODUCER write this conv's operand in the current pass?  The BN must run on stored statistics and
        nothing may need the producer's raw output: no gradient through the BN, no trainable BN / Scale parameter."""
        ctx, bn = self.ctx, self.bn
        if self.epi_producer is None or (bn.mode == "batch" and ctx.learning_phase == 1):
            return False
        if ctx.learning_phase == 0 or not (self.need_input_grad or bn.any_trainable()):      # (phase 1 = a backward pass follows)
            return True
        # Round 4: a stored-statistics BN WITH a backward pass (dense_rnn_net: every dense-block BN, hybridnet.py:11-97,182-354).
        # The fused BN-backward epilogue of this conv's data gradient needs the mask and the normalised input only, and both
        # follow from z = relu(a*u + b) alone (hdu_conv_desc.bnb_relu bit 1): the producer writes z, u never exists, the
        # materialise launch goes (HDU_FUSE_BN_EPILOGUE_TRAIN=0: off).
        return ctx.fuse_bn_epilogue_train and self.bnb_fused and self.need_input_grad and ctx.bnb_fusable()

    def forward(self):
        ctx = self.ctx
        if self.epi_producer is not None and self.epi_active():
            return self._forward_conv()          # the producer's epilogue already wrote self.xin
        if self.bn is not None:
            self.bn.fold(self.x)
        if self.xin is not None:
            bn = self.bn
            up = self.up if self.skip is not None else (0, 0, 0)
            skip = self.skip.act if self.skip is not None else None
            dst = self.xin_interior if self.halo else self.xin.act
            pend = bn.pending_stats if bn is not None else None
            if pend is not None and pend[0] == ctx.pass_id and pend[1].absorber is self:
                ops.materialize_stats(self.x.act, pend[1].stats_block(bn), bn.relu, up, skip, dst)
            else:
                ops.materialize(self.x.act, bn.a if bn else None, bn.b if bn else None, bn.relu if bn else False, up, skip, dst)
            if self.halo:
                _sh.halo_exchange(ctx.shard, self.xin.act, self.halo)
        self._forward_conv()

    def _forward_conv(self):
        ctx = self.ctx
        cons = self.epi_consumer
        if cons is not None and cons.epi_active():
            cons.bn.fold(self.out)               # (no-op after the batched fold at the head of the pass)
            drop = self.d_f_drop_epi is not None and ctx.learning_phase == 1 and ctx.dropout_enabled
            ops.conv_fprop(self.d_f_drop_epi if drop else self.d_f_epi)
            return
        # epilogue statistics once the sink has a usable shift (the previous pass's mean); the first training pass
        # after build / after new weights uses the two-pass reduction (shift = a sample of the tensor itself)
        st = ctx.learning_phase == 1 and self.d_f_st is not None and self.stats_sink.primed
        if self.d_f_drop is not None and ctx.learning_phase == 1 and ctx.dropout_enabled:
            ops.conv_fprop(self.d_f_drop_st if st else self.d_f_drop)
        else:
            ops.conv_fprop(self.d_f_st if st else self.d_f)

    def backward(self):
        ctx = self.ctx
        out = self.out
        if not out.root.needs_grad:
            return
        dy = out.dy()
        x = self.x.act
        if self.trainable and (getattr(self, "in_plan", False) or (ctx._split_now and getattr(self, "in_split_plan", False))):
            if self.bias is not None:      # (the filter gradient itself: deferred to the end of the pass, Ctx.run_backward)
                ops.colsum(dy, self.bias.grad, ctx.ws)
        elif self.trainable:
            if self.xin is not None:
                d = ctx.conv_desc(self.xin.act, self.wf_ptr, dy, self.K, self.stride, self.pad, self.conv_up)
            else:
                d = ctx.conv_desc(x, self.wf_ptr, dy, self.K, self.stride, self.pad, self.up,
                                  self.skip.act if self.skip is not None else None,
                                  (self.bn.a, self.bn.b) if self.bn is not None else None,
                                  self.bn.relu if self.bn is not None else False)
            ops.conv_wgrad(d, self.kernel.grad)
            if self.bias is not None:
                ops.colsum(dy, self.bias.grad, ctx.ws)
        if not self.need_input_grad:
            return
        K, pad = self.K, self.pad
        if self.halo:
            return self._backward_halo(dy)
        if self.bnb_fused and ctx.fuse_bn_bwd_now:
            return self._backward_fused_bn(dy)
        De, He, We = x.D << self.up[0], x.H << self.up[1], x.W << self.up[2]
        direct = self.bn is None and self.up == (0, 0, 0)
        sums_ready = False
        skip_first = self.skip is not None and self.skip.root.needs_grad and not self.skip.root.written \
            and self.skip.c0 == 0
        # where does d(x_eff) go?
        if direct:
            tgt, acc = self.x.grad, self.x.grad_mode()
        elif skip_first:
            tgt, acc = self.skip.grad, self.skip.grad_mode()
        else:
            tgt, acc = self._dxe(), False
        if self.strided:
            tgt_act = ops.Act(tgt.buf, tgt.off, x.N, x.D, x.H, x.W, x.C, tgt.ld, tgt.dtype)
            if self._s2 is not None:
                self._s2.run(tgt_act, accumulate=acc)
            else:
                d = ctx.conv_desc(tgt_act, self.wf_ptr, dy, K, self.stride, pad, accumulate=acc)
                ops.conv_dgrad_strided(d)
        else:
            d = ctx.conv_desc(dy, self.wd_ptr, ops.Act(tgt.buf, tgt.off, x.N, De, He, We, x.C, tgt.ld, tgt.dtype), K,
                              (1, 1, 1), (K[0] - 1 - pad[0], K[1] - 1 - pad[1], K[2] - 1 - pad[2]), accumulate=acc)
            sums_ready = self._epilogue_sums(d)
            ops.conv_fprop(d)
        if self.skip is not None and self.skip.root.needs_grad and not skip_first:
            # generic fallback: add d(x_eff) into an already-written skip gradient
            ops.upsample_bwd(tgt, self.skip.grad, (0, 0, 0), accumulate=self.skip.grad_mode())
        if direct:
            return
        dz = tgt
        if self.up != (0, 0, 0):
            if self.bn is None:
                ops.upsample_bwd(tgt, self.x.grad, self.up, accumulate=self.x.grad_mode())
                return
            dz = self._dz()
            ops.upsample_bwd(tgt, dz, self.up)
        if self.bn is not None:
            self.bn.backward(self.x, dz, sums_ready=sums_ready)
        else:
            ops.upsample_bwd(dz, self.x.grad, (0, 0, 0), accumulate=self.x.grad_mode())

    def _epilogue_sums(self, d):
        """Round 6: the backward sums of this conv's input BN (S1 = sum g, S2 = sum g * uhat) ride in the
        epilogue of the data-gradient launch `d` that produces dz -- raw dz is stored (hdu_conv_desc.bnb_relu bit 2), the BN's apply
        launch reads the slot table (hdu_bn_bwd_apply_sums) and the reduce_rows launch between them is gone.  Taken when dz IS the
        launch's output over the BN input's own grid (no up-sampling / skip / halo / dropout on the BN input), the launch would run on
        an im2col tile kernel anyway (the halo-tile kernels have no such epilogue: decided once from the kernel the plain descriptor
        picks) and the BN takes its two-launch form in this pass.  HDU_BNB_SUMS_EPILOGUE=0: off.  Returns True when taken."""
        ctx, bn = self.ctx, self.bn
        if not (ctx.bnb_sums_epilogue and bn is not None and self.up == (0, 0, 0) and self.skip is None and not self.halo
                and self.x.root.drop is None and ctx.bnb_fusable() and bn.BSUM_SLOTS <= 32):
            return False
        if self._sums_kernel_ok is None:
            name = ops.conv_kernel_name(d, 0)
            self._sums_kernel_ok = name.startswith("conv_igemm_ring_kernel") or name.startswith("conv_igemm_dma_kernel")
        if not self._sums_kernel_ok or not bn.takes_epilogue_sums(self.x):
            return False
        x = self.x.act
        d.bnb_u, d.bnb_ldu = x.ptr, x.ld
        d.bnb_a, d.bnb_b = bn.a.data_ptr(), bn.b.data_ptr()
        d.bnb_relu = (1 if bn.relu else 0) | 4
        d.bnb_mean, d.bnb_rstd = bn.mean_used.data_ptr(), bn.rstd.data_ptr()
        d.bnb_partial, d.bnb_slots = bn.bsum.data_ptr(), bn.BSUM_SLOTS
        return True


def _conv_backward_fused_bn(self, dy):
    """data gradient with the BN(+Scale)+ReLU backward of `self.bn` in its epilogue: x.grad (+)= a * g directly, S1 / S2
    into slot rows; hdu_bn_bwd_finalize turns the sums into the parameter gradients and adds this BN's share of the
    deferred part of du to the accumulators of the tensor it normalises (applied by Var.dy())."""
    ctx, bn, xv = self.ctx, self.bn, self.x
    x = xv.act
    K, pad = self.K, self.pad
    acc = xv.grad_mode()
    tgt = xv.grad
    d = ctx.conv_desc(dy, self.wd_ptr, ops.Act(tgt.buf, tgt.off, x.N, x.D, x.H, x.W, x.C, tgt.ld, tgt.dtype), K, (1, 1, 1),
                      (K[0] - 1 - pad[0], K[1] - 1 - pad[1], K[2] - 1 - pad[2]), accumulate=acc)
    d.bnb_u, d.bnb_ldu = x.ptr, x.ld
    d.bnb_a, d.bnb_b, d.bnb_relu = bn.a.data_ptr(), bn.b.data_ptr(), 1 if bn.relu else 0
    if self.epi_producer is not None and self.epi_active():
        # the producer's epilogue wrote z = relu(a*u + b) into this conv's operand buffer and u was never stored
        d.bnb_u, d.bnb_ldu = self.xin.act.ptr, self.xin.act.ld
        d.bnb_relu |= 2
    need_sums = bn.batch_now or bn.any_trainable()
    part = None
    if need_sums:
        n = self.BNB_SLOTS * 2 * bn.C
        part = ctx.bnb_acc[self.bnb_off:self.bnb_off + n]
        d.bnb_mean, d.bnb_rstd = bn.mean_used.data_ptr(), bn.rstd.data_ptr()
        d.bnb_partial, d.bnb_slots = part.data_ptr(), self.BNB_SLOTS
    ops.conv_fprop(d)
    if need_sums:
        tr_bn = bn.trainable
        tr_sc = bn.sg is not None and bn.scale_trainable
        c3 = c4 = None
        if bn.batch_now:
            r = xv.root
            ld = r.act.ld
            base = r.corr_off + xv.c0
            c3, c4 = ctx.corr_acc[base:base + bn.C], ctx.corr_acc[base + ld:base + ld + bn.C]
        if not bn.batch_now and ctx.batch_wgrad and ctx.defer_bnb_finalize:
            # inference-mode BN: the sums feed parameter gradients only -- finalized with every other such layer by ONE launch
            # at the end of the backward pass (run_backward), like the deferred filter gradients
            ctx._bnb_deferred.append((id(self), (part, self.BNB_SLOTS, bn.C, bn.gamma.data, bn.beta.data,
                                                 bn.sg.data if bn.sg else None, bn.gamma.grad if tr_bn else None,
                                                 bn.beta.grad if tr_bn else None, bn.sg.grad if tr_sc else None,
                                                 bn.sb.grad if tr_sc else None)))
            return
        fin_args = (part, self.BNB_SLOTS, x.M, bn.C, bn.gamma.data, bn.beta.data, bn.sg.data if bn.sg else None, bn.mean_used, bn.rstd,
                    bn.gamma.grad if tr_bn else None, bn.beta.grad if tr_bn else None,
                    bn.sg.grad if tr_sc else None, bn.sb.grad if tr_sc else None, c3, c4)
        if bn.batch_now and ctx.merge_fin_correct and self.BNB_SLOTS <= 32 and bn.C % 4 == 0:
            ctx.flush_pending_finalize()
            ctx._pending_fin = dict(root=xv.root, c0=xv.c0, C=bn.C, M=x.M, args=fin_args)      # runs with the next reader's correction (Var.dy)
            return
        ops.bn_bwd_finalize(part, self.BNB_SLOTS, x.M, bn.C, bn.batch_now, bn.gamma.data, bn.beta.data,
                            bn.sg.data if bn.sg else None, bn.mean_used, bn.rstd,
                            bn.gamma.grad if tr_bn else None, bn.beta.grad if tr_bn else None,
                            bn.sg.grad if tr_sc else None, bn.sb.grad if tr_sc else None, c3, c4)


ConvLayer._backward_fused_bn = _conv_backward_fused_bn
ConvLayer.BNB_SLOTS = int(os.environ.get("HDU_BNB_SLOTS", "16"))      # slot rows of the fused BN backward in a data-gradient epilogue (<= 32).  Round 6 sweep (profiles/r06_experiment_knob_sweep.txt): 16 beats the 32 of rounds 3-5 by 0.1-0.35 % on all three workloads (half the finalize reads), 8 and 4 lose it again to atomic contention


def _conv_backward_halo(self, dy):
    """data gradient of a depth-sharded conv: d(input incl. halo planes) -> [upsample gradient] -> halo gradients
    go back to the owning neighbours -> BN backward on the local planes"""
    ctx = self.ctx
    x = self.x.act
    K, pad, h = self.K, self.pad, self.halo
    De, He, We = (x.D + 2 * h) << self.up[0], x.H << self.up[1], x.W << self.up[2]
    tgt = self._dxe()
    if self.strided:      # the 7x7x7 stride-2 stem of a depth-sharded end-to-end hybrid: gradient w.r.t. (CT, 250*logits2d)
        d = ctx.conv_desc(ops.Act(tgt.buf, tgt.off, x.N, De, He, We, x.C, tgt.ld, tgt.dtype), self.wf_ptr, dy, K,
                          self.stride, pad)
        ops.conv_dgrad_strided(d)
    else:
        d = ctx.conv_desc(dy, self.wd_ptr, ops.Act(tgt.buf, tgt.off, x.N, De, He, We, x.C, tgt.ld, tgt.dtype), K, (1, 1, 1),
                          (K[0] - 1 - pad[0], K[1] - 1 - pad[1], K[2] - 1 - pad[2]), halo_out=(2 * h) << self.up[0])
        ops.conv_fprop(d)
    dz = tgt
    if self.up != (0, 0, 0):
        dz = self._dz()
        ops.upsample_bwd(tgt, dz, self.up)
    _sh.halo_reduce(ctx.shard, dz, h, self._halo_tmp().buf)
    plane = x.H * x.W * dz.ld
    interior = ops.Act(dz.buf, dz.off + h * plane, x.N, x.D, x.H, x.W, x.C, dz.ld, dz.dtype)
    if self.skip is not None and self.skip.root.needs_grad:      # d(x_eff) is also the gradient of the added skip
        ops.upsample_bwd(interior, self.skip.grad, (0, 0, 0), accumulate=self.skip.grad_mode())
    if self.bn is not None:
        self.bn.backward(self.x, interior)
    else:
        ops.upsample_bwd(interior, self.x.grad, (0, 0, 0), accumulate=self.x.grad_mode())


ConvLayer._backward_halo = _conv_backward_halo


class StatsOp:
    """tf.nn.moments of a freshly written tensor / slab, once, shared by every consumer BN.  When the tensor has a
    single batch-stat consumer BN over exactly these channels, `fuse(bn)` folds it in the same two launches."""

    SLOTS = int(os.environ.get("HDU_STATS_SLOTS", "32"))     # slot rows a conv epilogue spreads its float atomics over (workgroup % SLOTS; <= 32)

    def __init__(self, ctx, var):
        self.ctx, self.var = ctx, var
        self.fused = None
        self.fold_next = None
        self.absorber = None        # the ConvLayer whose materialize pass folds the BN of this segment (no finalize launch)
        world = ctx.shard.world if ctx.shard is not None else 1
        self.sync_buf = ctx.fvec(world * (1 + 2 * var.C)) if world > 1 else None      # hdu_stats_sync_floats
        ctx.need_ws(var.act.M, var.C)
        var.stats()
        # produced by the conv that was just built?  then its epilogue takes the moments (no reduction pass)
        self.producer = None
        prod = ctx.convs[-1] if ctx.convs else None
        sharded = ctx.shard is not None and ctx.shard.world > 1
        if (ctx.epilogue_stats and not sharded and prod is not None and ctx.fwd and ctx.fwd[-1] == prod.forward
                and prod.out.root is var.root and prod.out.c0 == var.c0 and prod.out.C == var.C
                and getattr(prod, "stats_sink", None) is None):
            self.producer = prod
            prod.stats_sink = self
            self.acc_off = None
            self.primed = False
            ctx.stats_sinks.append(self)
        ctx.fwd.append(self.forward)

    def fuse(self, bn):
        assert bn.C == self.var.C and bn.mode == "batch"
        self.fused = bn
        bn.stats_src = self
        return self

    def then_fold(self, bn):
        """`bn` (batch statistics) reads the slab [0, end of this segment): the finalize launch of this segment's
        epilogue statistics also folds it (hdu_bn_stats_finalize_fold_next) -- one launch less per dense layer."""
        if self.fused is None and bn.mode == "batch" and self.ctx.fold_next and bn.C == self.var.c0 + self.var.C:
            self.fold_next = bn
            bn.stats_src = self
        return self

    def stats_block(self, bn):
        """argument of ops.materialize_stats for the consumer of `bn` (= self.fused or self.fold_next)"""
        ctx = self.ctx
        n = self.SLOTS * 2 * self.var.C
        fold = (bn.gamma.data, bn.beta.data, bn.eps, bn.sg.data if bn.sg else None, bn.sb.data if bn.sb else None,
                bn.a, bn.b, bn.rstd, bn.mm.data, bn.mv.data, bn.momentum)
        part = ctx.stats_acc[self.acc_off:self.acc_off + n]
        if bn is self.fused:
            mean, var = self.var.stats()
            return (part, self.SLOTS, self.var.act.M, self.var.C, 0, self.var.stats_shift(), mean, var, fold)
        r = self.var.root
        return (part, self.SLOTS, self.var.act.M, self.var.C, self.var.c0, r.shift[:bn.C], r.mean[:bn.C], r.var[:bn.C], fold)

    def forward(self):
        ctx = self.ctx
        if ctx.learning_phase != 1:
            return
        mean, var = self.var.stats()
        bn = self.fused
        if ctx.shard is not None and ctx.shard.world > 1:
            # local moments -> global moments over all depth shards (equal shard sizes), then the consumers fold
            ops.bn_stats(self.var.act, mean, var, ctx.ws)
            _sh.sync_stats(ctx.shard, mean, var, self.var.act.M, self.sync_buf)
            return
        fold = None
        if bn is not None:
            fold = (bn.gamma.data, bn.beta.data, bn.eps, bn.sg.data if bn.sg else None, bn.sb.data if bn.sb else None,
                    bn.a, bn.b, bn.rstd, bn.mm.data, bn.mv.data, bn.momentum)
        if self.producer is not None and not self.primed:
            self.primed = True              # this pass: ordinary reduction below; its mean is the next pass's shift
            if bn is None:
                ops.bn_stats(self.var.act, mean, var, ctx.ws)
                return
            ops.bn_stats_fold(self.var.act, mean, var, *fold, ctx.ws)
        elif self.producer is not None:
            # the conv epilogue left sum(y - shift), sum((y - shift)^2) in the slot rows; shift = last step's mean
            n = self.SLOTS * 2 * self.var.C
            nb = self.fold_next
            tgt = nb if nb is not None else bn
            if tgt is not None and self.absorber is not None and ctx.absorb_stats:
                # no finalize launch: the pass that applies `tgt` (its consumer's hdu_materialize_stats) folds it from the sums
                tgt.batch_now = True
                tgt.mean_used = self.var.root.mean[:tgt.C] if nb is not None else mean
                tgt.folded_pass = ctx.pass_id
                tgt.pending_stats = (ctx.pass_id, self)
                return
            if nb is not None:
                r = self.var.root
                f2 = (nb.gamma.data, nb.beta.data, nb.eps, nb.sg.data if nb.sg else None, nb.sb.data if nb.sb else None,
                      nb.a, nb.b, nb.rstd, nb.mm.data, nb.mv.data, nb.momentum)
                ops.bn_stats_finalize_fold_next(ctx.stats_acc[self.acc_off:self.acc_off + n], self.SLOTS, self.var.act.M,
                                                self.var.C, self.var.c0, nb.C, r.shift[:nb.C], r.mean[:nb.C], r.var[:nb.C], f2)
                nb.batch_now = True
                nb.mean_used = r.mean[:nb.C]
                nb.folded_pass = ctx.pass_id
                return
            ops.bn_stats_finalize(ctx.stats_acc[self.acc_off:self.acc_off + n], self.SLOTS, self.var.act.M, self.var.C,
                                  self.var.stats_shift(), mean, var, fold)
        elif bn is None:
            ops.bn_stats(self.var.act, mean, var, ctx.ws)
            return
        else:
            ops.bn_stats_fold(self.var.act, mean, var, *fold, ctx.ws)
        if bn is not None:
            bn.mean_used = mean
            bn.folded_pass = ctx.pass_id


class MaterializeLayer:
    """z = relu(BN(+Scale)(x)) written out (where the activation feeds a pool / a skip / the HFF add)."""

    def __init__(self, ctx, x, bn, halo=0):
        self.ctx, self.x, self.bn, self.halo = ctx, x, bn, halo
        a = x.act
        self.out = ctx.new_var(a.N, a.D + 2 * halo, a.H, a.W, a.C)
        if ctx.grad_enabled:
            self.out.require_grad()
        if halo:
            self._halo_tmp = ctx.scratch("halo_tmp", 1, 2 * halo, a.H, a.W, a.C)
        ctx.need_ws(a.M, a.C)
        ctx.fwd.append(self.forward)
        ctx.bwd.append(self.backward)

    def _interior(self, act):
        a, h = self.x.act, self.halo
        return ops.Act(act.buf, act.off + h * a.H * a.W * act.ld, a.N, a.D, a.H, a.W, a.C, act.ld, act.dtype)

    def forward(self):
        self.bn.fold(self.x)
        if self.halo:
            ops.affine_act(self.x.act, self.bn.a, self.bn.b, self.bn.relu, self._interior(self.out.act))
            _sh.halo_exchange(self.ctx.shard, self.out.act, self.halo)
        else:
            ops.affine_act(self.x.act, self.bn.a, self.bn.b, self.bn.relu, self.out.act)

    def backward(self):
        if not self.out.root.needs_grad:
            return
        if self.halo:
            _sh.halo_reduce(self.ctx.shard, self.out.grad, self.halo, self._halo_tmp().buf)
            self.bn.backward(self.x, self._interior(self.out.grad))
        else:
            self.bn.backward(self.x, self.out.dy())


class MaxPoolLayer:
    """ZeroPadding(1) + MaxPooling 3x3(x3) stride 2 (denseunet.py:169-170, denseunet3d.py:135-136)."""

    def __init__(self, ctx, x, out=None, pad_d=1):
        """pad_d=0: x already carries the neighbouring depth planes as halo (depth sharding)"""
        self.ctx, self.x, self.pad_d = ctx, x, pad_d
        a = x.act
        Do = 1 if a.D == 1 else (a.D + 2 * pad_d - 3) // 2 + 1
        dims = (a.N, Do, (a.H - 1) // 2 + 1, (a.W - 1) // 2 + 1)
        self.out = out if out is not None else ctx.new_var(*dims, a.C)
        assert (self.out.act.N, self.out.act.D, self.out.act.H, self.out.act.W) == dims
        if ctx.grad_enabled:
            self.out.require_grad()
        oa = self.out.act
        self.argmax = torch.zeros(oa.M * a.C, dtype=torch.uint8, device=ctx.dev) if self.x.root.needs_grad else None
        ctx.fwd.append(lambda: ops.maxpool_fwd(self.x.act, self.out.act, self.argmax, self.pad_d))
        ctx.bwd.append(self.backward)

    def backward(self):
        if self.out.root.needs_grad and self.x.root.needs_grad:
            ops.maxpool_bwd(self.argmax, self.out.dy(), self.x.grad, self.x.grad_mode(), self.pad_d)


class AvgPoolLayer:
    """AveragePooling 2x2 over (H,W) (denseunet.py:290; 3D: (2,2,1), denseunet3d.py:102)."""

    def __init__(self, ctx, x, out):
        self.ctx, self.x, self.out = ctx, x, out
        if ctx.grad_enabled:
            out.require_grad()
        ctx.fwd.append(lambda: ops.avgpool_fwd(self.x.act, self.out.act))
        ctx.bwd.append(self.backward)

    def backward(self):
        if self.out.root.needs_grad and self.x.root.needs_grad:
            ops.avgpool_bwd(self.out.dy(), self.x.grad, self.x.grad_mode())


class LossLayer:
    """weighted softmax cross-entropy + its gradient (loss.py:5-46) over row ranges of the logits."""

    def __init__(self, ctx, logits, ranges, weights=(0.78, 0.65, 8.57)):
        self.ctx, self.logits, self.ranges, self.weights = ctx, logits, ranges, weights
        self.count = sum(m for _, m in ranges)
        self.labels = torch.zeros(logits.act.M, dtype=torch.uint8, device=ctx.dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=ctx.dev)      # re-pointed into Ctx.arena by finalize()
        self.class_count = torch.zeros(3, dtype=torch.float32, device=ctx.dev)
        self.global_scale = 1.0   # 1/world_size under data parallelism (loss.py:44 takes the mean over ALL towers)
        ctx.loss_layers.append(self)
        ctx.need_ws(1 << 14, 8)

    def set_labels(self, lab_u8_internal):
        """labels already in internal row order [N*D*H*W]"""
        lab = np.asarray(lab_u8_internal).reshape(-1)
        if lab.size != self.labels.numel():
            raise ValueError("labels: expected %d entries, got %d" % (self.labels.numel(), lab.size))
        if int(lab.max()) > 2 or int(lab.min()) < 0:
            raise ValueError("labels must be in {0,1,2} (loss.py:14-21)")
        self.labels.copy_(torch.from_numpy(lab.astype(np.uint8)).to(self.ctx.dev))

    def run(self, with_grad=True):
        if self.ctx._zeroed_fwd_pass != self.ctx.pass_id:      # not inside a step whose head launch cleared the arena
            i = self.ctx.loss_layers.index(self)                # this layer's own [loss sum | 3 class counts] slot only
            ops.zero_tensor(self.ctx.arena[4 * i:4 * i + 4])
        dl = None
        if with_grad:
            self.logits.root.written = True
            dl = self.logits.grad
        gs = self.global_scale / float(self.count)
        for row0, m in self.ranges:
            ops.wce_loss(self.logits.act, self.labels, row0, m, self.weights, gs, dl, self.loss_sum, self.class_count,
                         self.ctx.ws)

    def value(self):
        return float(self.loss_sum.item()) / float(self.count)
