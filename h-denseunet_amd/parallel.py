"""Data parallelism: one process per GPU, gradients summed with ONE flat RCCL all-reduce over xGMI.

Replaces K.utils2/multi_gpu.py:7-69 (in-graph towers: batch sliced per tower, outputs concatenated on the CPU,
gradient summation implicit in tf.gradients, BN statistics per tower).  Semantics kept: every rank runs the same
weights on its own mini-batch with LOCAL BN statistics; the loss is the mean over the merged batch (loss.py:44),
so each rank scales its gradient by 1/world_size before the sum.  All trainable parameters live in one flat float32
buffer (engine.Ctx.G), so the exchange is a single large collective (ring all-reduce is per-link bound on xGMI:
fewer, larger messages).
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and os.environ.get("HDU_FORCE_DP") != "1":
        return 0, 1
    # HDU_FORCE_DP=1: run the multi-process code path (process group, broadcast, eager all-reduce between the two
    # hipGraphs) with a world of ONE -- the only way to exercise RCCL on a single-GPU box
    os.environ.setdefault("MASTER_PORT", "29533")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def attach_data_parallel(model, expected_world=None):
    rank, world = init_process_group_from_env()
    if expected_world is not None and world != expected_world and world != 1:
        raise RuntimeError("make_parallel asked for %d GPUs but WORLD_SIZE=%d" % (expected_world, world))
    if world == 1 and os.environ.get("HDU_FORCE_DP") != "1":
        return model

    if os.environ.get("HDU_COMM") == "rccl_abi":
        # the gradient sum as an hdu_comm_allreduce_f32 call (include/hdu.h) on the compute stream instead of a
        # torch.distributed collective; the process group is only the rendezvous that carried the RCCL id
        from .comm import Comm
        model._comm = Comm.from_process_group()
        model.set_data_parallel(world, model._comm.allreduce_)
        broadcast_parameters(model)
        return model

    def allreduce(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def allreduce_async(t):
        if os.environ.get("HDU_DP_SYNC_BUCKETS") == "1":      # A/B: bucketed but on the compute stream (no overlap)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    # HDU_DP_BUCKETS: "0" (default) = one all-reduce after the backward; otherwise the parameter fractions at which the
    # backward is cut, e.g. "0.85": decoder + dense blocks 5 and 4 first, exchanged under the rest of the backward.
    # Opt-in: on the one GPU available this round the communication-stream hand-off alone cost 1.3 ms per step
    # (world 1: 293 vs 308 slices/s), about what the overlap could save on 8 GPUs; to be re-measured on a real node.
    spec = os.environ.get("HDU_DP_BUCKETS", "0")
    if spec in ("0", ""):
        model.set_data_parallel(world, allreduce)
    else:
        model.set_data_parallel(world, allreduce, allreduce_async, tuple(float(v) for v in spec.split(",")))
    broadcast_parameters(model)
    return model


def broadcast_parameters(model, src=0):
    """identical initial weights on every rank (the towers of multi_gpu.py share one set of variables)"""
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("HDU_FORCE_DP") == "1"):
        dist.broadcast(model.ctx.P, src=src)
        dist.broadcast(model.ctx.V, src=src)


def shard_depth(D, world, rank):
    """[begin, end) of the depth planes owned by `rank` when one volume is sharded on the depth axis"""
    if D % (4 * world) != 0:
        # sync-BN and the loss normalisation assume equal shards (n_global = n_local * world), and every shard must
        # satisfy the 3D net's depth rule (local depth a multiple of 4)
        raise ValueError("depth %d cannot be split evenly over %d ranks in multiples of 4 planes" % (D, world))
    per = D // world
    return rank * per, (rank + 1) * per


def depth_shard_info(backend=None):
    """ShardInfo of this process (one process per GPU / per depth shard)"""
    from .shard import ShardInfo
    rank, world = init_process_group_from_env(backend)
    comm = None
    if world > 1 and os.environ.get("HDU_COMM") == "rccl_abi":      # halo / sync-BN / gradient exchanges as hdu_comm_* calls
        from .comm import Comm
        comm = Comm.from_process_group()
    return ShardInfo(rank, world, comm=comm)


def attach_depth_shard(model):
    """gradient exchange of a depth-sharded model: every rank holds PARTIAL filter / BN gradients of the one volume
    (its own voxels), so the flat gradient buffer is summed; the loss is normalised by the global voxel count."""
    sh = model.ctx.shard
    if sh is None or sh.world == 1:
        return model

    def allreduce(t):
        from . import shard as _sh
        _sh.allreduce_sum(sh, t)        # (counted with the step's other collectives: shard.COUNTS)

    model.set_data_parallel(sh.world, allreduce)
    broadcast_parameters(model)
    return model
