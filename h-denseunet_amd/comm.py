"""RCCL through the C-ABI (include/hdu.h, hdu_comm_*): the collectives of the hot path without torch in the data path.

torch.distributed is still what LAUNCHES the processes (torch.distributed.run) and is used ONCE here, to hand rank 0's 128-byte
RCCL id to the other ranks; after that the gradient all-reduce and the depth-neighbour exchanges are hdu_comm_* calls on
torch's current stream -- ordinary stream work, so they can sit inside a captured hipGraph next to the kernels they order
against.  Opt-in (HDU_COMM=rccl_abi, or Comm.from_process_group() by hand): the default data-parallel path keeps
torch.distributed's all-reduce, which is the one the driver's multi-GPU run has exercised.
"""
import ctypes

import torch

from . import lib as _l
from . import ops


class Comm:
    def __init__(self, rank, world, id_bytes):
        if len(id_bytes) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        self.rank, self.world = rank, world
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
        _l.check(_l.get().hdu_comm_init(ctypes.byref(self._h), rank, world, buf), "hdu_comm_init")

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _l.check(_l.get().hdu_comm_unique_id(buf), "hdu_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls):
        """one communicator over the ranks of the default torch.distributed group (the id travels through its store);
        a single process without a group gets a world of one"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(0, 1, cls.unique_id())
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

    def allreduce_(self, t):
        """in-place sum of a contiguous float32 tensor over the ranks, on the current stream"""
        assert t.dtype == torch.float32 and t.is_contiguous()
        _l.check(_l.get().hdu_comm_allreduce_f32(self._h, ctypes.c_void_p(t.data_ptr()), t.numel(), ops.stream()),
                 "hdu_comm_allreduce_f32")
        return t

    def sendrecv(self, lo_rank, send_lo, recv_lo, hi_rank, send_hi, recv_hi):
        """one grouped exchange with the two depth neighbours (None = the volume's edge); equal-sized contiguous tensors"""
        ref = send_lo if send_lo is not None else send_hi
        if ref is None:
            return
        nbytes = ref.numel() * ref.element_size()
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        for t in (send_lo, recv_lo, send_hi, recv_hi):
            assert t is None or (t.is_contiguous() and t.numel() * t.element_size() == nbytes)
        _l.check(_l.get().hdu_comm_sendrecv(self._h, -1 if lo_rank is None else lo_rank, p(send_lo), p(recv_lo),
                                            -1 if hi_rank is None else hi_rank, p(send_hi), p(recv_hi), nbytes, ops.stream()),
                 "hdu_comm_sendrecv")

    def close(self):
        if self._h:
            _l.get().hdu_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown: the library may already be gone
            pass
