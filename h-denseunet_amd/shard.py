"""Depth-axis sharding of ONE volume across ranks (BASELINE configs[4]; new capability, SURVEY.md section 8e -- the
reference only has batch towers, K.utils2/multi_gpu.py).

Rank r owns a contiguous range of depth planes at every resolution level (depth-major layout: a plane is one
contiguous memory range).  Every depth-coupled layer keeps `h` halo planes on both sides of its input buffer:

  forward : halo_exchange  -- my first/last interior planes -> neighbours' halo planes (global edges stay zero =
                              the reference's ZeroPadding3D / SAME padding)
  backward: halo_reduce    -- gradients that landed on my halo planes go back to the owning neighbour and are ADDED
                              to its boundary planes
  BN      : sync statistics -- all-reduce of per-channel (n*mean, n*(var+mean^2)) forward and (S1, S2) backward, so a
                              sharded step equals the single-device step.

Exchanges are point-to-point with the two depth neighbours (RCCL send/recv over xGMI; gloo in the CPU tests); torch is
the transport only: the halo-gradient add and the sync-BN packing run in libhdu kernels.
"""
import torch
import torch.distributed as dist

# collectives issued since reset_counts(): bench.py reports them per sharded step (`collectives_per_step`).  Every one of them is
# EXPOSED today: issued in program order on the compute stream (or waited for right away), nothing computes under it.
COUNTS = {"allreduce": 0, "allreduce_bytes": 0, "neighbour_exchange": 0, "neighbour_bytes": 0}


def reset_counts():
    for k in COUNTS:
        COUNTS[k] = 0


def counts():
    return dict(COUNTS)


def _count_exchange(*tensors):
    COUNTS["neighbour_exchange"] += 1
    COUNTS["neighbour_bytes"] += sum(t.numel() * t.element_size() for t in tensors if t is not None)


class ShardInfo:
    def __init__(self, rank, world, group=None, comm=None):
        """comm: an h-denseunet_amd.comm.Comm (RCCL through the C-ABI, hdu_comm_sendrecv / hdu_comm_allreduce_f32 on the
        compute stream); None = torch.distributed point-to-point / all-reduce (gloo in the CPU tests)"""
        self.rank, self.world, self.group, self.comm = rank, world, group, comm

    @property
    def lo(self):
        return self.rank - 1 if self.rank > 0 else None

    @property
    def hi(self):
        return self.rank + 1 if self.rank < self.world - 1 else None


def _planes(act, p0, n):
    """flat view of planes [p0, p0+n) of a full-width activation (C == ld)"""
    assert act.C == act.ld and act.N == 1
    plane = act.H * act.W * act.ld
    return act.buf[act.off + p0 * plane: act.off + (p0 + n) * plane]


def halo_exchange(sh, act, h):
    """act: [1][Dl+2h][H][W][C]; fills the 2 x h halo planes from the depth neighbours"""
    if sh is None or sh.world == 1:
        return
    D = act.D
    _count_exchange(_planes(act, h, h) if sh.lo is not None else None, _planes(act, D - 2 * h, h) if sh.hi is not None else None)
    if sh.comm is not None:       # one grouped RCCL exchange with both neighbours, enqueued on the compute stream
        sh.comm.sendrecv(sh.lo, _planes(act, h, h) if sh.lo is not None else None, _planes(act, 0, h) if sh.lo is not None else None,
                         sh.hi, _planes(act, D - 2 * h, h) if sh.hi is not None else None,
                         _planes(act, D - h, h) if sh.hi is not None else None)
        return
    ops_ = []
    if sh.lo is not None:
        ops_.append(dist.P2POp(dist.isend, _planes(act, h, h), sh.lo, sh.group))
        ops_.append(dist.P2POp(dist.irecv, _planes(act, 0, h), sh.lo, sh.group))
    if sh.hi is not None:
        ops_.append(dist.P2POp(dist.isend, _planes(act, D - 2 * h, h), sh.hi, sh.group))
        ops_.append(dist.P2POp(dist.irecv, _planes(act, D - h, h), sh.hi, sh.group))
    for r in dist.batch_isend_irecv(ops_):
        r.wait()


def halo_reduce(sh, act, h, tmp):
    """act holds gradients for Dl+2h planes; halo-plane gradients are returned to their owners and accumulated.
    tmp: scratch tensor of at least 2*h planes."""
    if sh is None or sh.world == 1:
        return
    D = act.D
    plane = act.H * act.W * act.ld
    ops_ = []
    r_lo = tmp[:h * plane]
    r_hi = tmp[h * plane:2 * h * plane]
    _count_exchange(_planes(act, 0, h) if sh.lo is not None else None, _planes(act, D - h, h) if sh.hi is not None else None)
    if sh.comm is not None:
        sh.comm.sendrecv(sh.lo, _planes(act, 0, h) if sh.lo is not None else None, r_lo if sh.lo is not None else None,
                         sh.hi, _planes(act, D - h, h) if sh.hi is not None else None, r_hi if sh.hi is not None else None)
    else:
        if sh.lo is not None:
            ops_.append(dist.P2POp(dist.isend, _planes(act, 0, h), sh.lo, sh.group))
            ops_.append(dist.P2POp(dist.irecv, r_lo, sh.lo, sh.group))
        if sh.hi is not None:
            ops_.append(dist.P2POp(dist.isend, _planes(act, D - h, h), sh.hi, sh.group))
            ops_.append(dist.P2POp(dist.irecv, r_hi, sh.hi, sh.group))
        for r in dist.batch_isend_irecv(ops_):
            r.wait()
    # the returned halo gradients are ADDED to my boundary planes by the library's accumulate kernel
    from . import ops
    H, W, C, ld = act.H, act.W, act.C, act.ld

    def view(buf, off, ldv):
        return ops.Act(buf, off, 1, h, H, W, C, ldv, act.dtype)
    if sh.lo is not None:
        ops.upsample_bwd(view(tmp, 0, C), view(act.buf, act.off + h * plane, ld), (0, 0, 0), accumulate=True)
    if sh.hi is not None:
        ops.upsample_bwd(view(tmp, h * plane, C), view(act.buf, act.off + (D - 2 * h) * plane, ld), (0, 0, 0), accumulate=True)


def allreduce_sum(sh, t):
    if sh is not None and sh.world > 1:
        COUNTS["allreduce"] += 1
        COUNTS["allreduce_bytes"] += t.numel() * t.element_size()
        if sh.comm is not None:
            sh.comm.allreduce_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=sh.group)
    return t


def sync_stats(sh, mean, var, n_local, buf):
    """local (mean, biased var) over n_local pixels -> statistics of the whole volume, in place.
    buf: float32 scratch of hdu_stats_sync_floats(C, world) = world * (1 + 2C) elements (include/hdu.h: every rank fills its
    slot, the SUM all-reduce gathers them, the slots are combined without an E[x^2] - E[x]^2 subtraction)."""
    if sh is None or sh.world == 1:
        return
    from . import lib as _l, ops
    C = mean.numel()
    lib = _l.get()
    n = sh.world * (1 + 2 * C)
    _l.check(lib.hdu_stats_pack(C, ops.fptr(mean), ops.fptr(var), n_local, sh.rank, sh.world, ops.fptr(buf), ops.stream()),
             "hdu_stats_pack")
    allreduce_sum(sh, buf[:n])
    _l.check(lib.hdu_stats_unpack(C, ops.fptr(buf), sh.world, ops.fptr(mean), ops.fptr(var), ops.stream()), "hdu_stats_unpack")


def exchange_ct_planes(sh, vol_h, D, plane):
    """vol_h: float32 [D+2][plane] -- the rank's D raw CT planes with one halo plane on each side.  The halos come from
    the depth neighbours (one small send/recv at the input); at the two ends of the volume the edge
    plane is replicated, which is exactly the reference's first / last 2.5D slab (denseunet3d.py:399-409)."""
    first, last = vol_h[plane:2 * plane], vol_h[D * plane:(D + 1) * plane]
    lo_halo, hi_halo = vol_h[:plane], vol_h[(D + 1) * plane:(D + 2) * plane]
    _count_exchange(first if sh.lo is not None else None, last if sh.hi is not None else None)
    if sh.comm is not None:
        if sh.lo is None:
            lo_halo.copy_(first)
        if sh.hi is None:
            hi_halo.copy_(last)
        sh.comm.sendrecv(sh.lo, first if sh.lo is not None else None, lo_halo if sh.lo is not None else None,
                         sh.hi, last if sh.hi is not None else None, hi_halo if sh.hi is not None else None)
        return
    ops_ = []
    if sh.lo is not None:
        ops_.append(dist.P2POp(dist.isend, first, sh.lo, sh.group))
        ops_.append(dist.P2POp(dist.irecv, lo_halo, sh.lo, sh.group))
    else:
        lo_halo.copy_(first)
    if sh.hi is not None:
        ops_.append(dist.P2POp(dist.isend, last, sh.hi, sh.group))
        ops_.append(dist.P2POp(dist.irecv, hi_halo, sh.hi, sh.group))
    else:
        hi_halo.copy_(last)
    if ops_:
        for r in dist.batch_isend_irecv(ops_):
            r.wait()
