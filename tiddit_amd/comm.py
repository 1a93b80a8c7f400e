"""ctypes front end of the C ABI's multi-GPU exchange (``tdt_comm_*``, csrc/tdt_comm.hip): RCCL over xGMI, one process per GPU.
``dist.py`` drives the same exchange through ``torch.distributed`` (backend "nccl" = RCCL); this class is what a binder without
PyTorch uses, and what ``Comm.from_torch`` bootstraps when a process group already exists."""
import ctypes

import numpy as np

from . import _native


class Comm:
    def __init__(self, rank, world, unique_id, ctx=None):
        self.ctx = ctx or _native.default_context()
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _native.check(self.ctx.lib.tdt_comm_init(self.ctx.handle, buf, self.rank, self.world, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        """the 128-byte id rank 0 creates and every rank passes to the constructor"""
        buf = (ctypes.c_uint8 * 128)()
        _native.check(_native.load().tdt_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch(cls, ctx=None, group=None):
        """bootstrap over an initialised torch.distributed group (any backend): rank 0's id is broadcast as an object"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0], ctx)

    def allgatherv(self, d_send, d_recv, counts, elem_bytes):
        """counts: elements contributed by every rank (identical on all ranks); d_send / d_recv: device pointers.  Rank r's
        elements land at element offset sum(counts[:r]) of d_recv on every rank.  Asynchronous on the context stream."""
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        displs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
        _native.check(self.ctx.lib.tdt_allgatherv(self._h, d_send, int(counts[self.rank]), d_recv, _native.ptr(counts), _native.ptr(displs),
                                                  int(elem_bytes)))

    def allreduce_sum_f64(self, d_buf, n):
        _native.check(self.ctx.lib.tdt_allreduce_sum_f64(self._h, d_buf, int(n)))

    def close(self):
        if self._h:
            self.ctx.lib.tdt_comm_destroy(self._h)
            self._h = None
