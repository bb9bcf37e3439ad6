"""The data-parallel exchange step through the C ABI (include/stair_hip.h: stair_comm_*, stair_allreduce_grads): RCCL
called from libstair_hip.so on the caller's stream.  `Trainer(native_allreduce=True)` uses it for the one collective of a
step; the default path is torch.distributed's all_reduce, which on GPUs is the same RCCL (backend "nccl")."""
import ctypes as C

import torch

from ._lib import check, lib


class NativeComm:
    """One RCCL communicator for this process's GPU.  The 128-byte unique id is created on rank 0 and handed to the other
    ranks with torch.distributed.broadcast_object_list (any initialised backend, gloo included); world == 1 needs none."""

    def __init__(self, rank=0, world=1):
        ident = (C.c_char * 128)()
        if rank == 0:
            check(lib.stair_comm_unique_id(ident))
        if world > 1:
            import torch.distributed as dist
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        self._comm = C.c_void_p()
        check(lib.stair_comm_create(ident, rank, world, C.byref(self._comm)))
        self.rank, self.world = rank, world

    def allreduce_(self, flat):
        """In-place sum of a contiguous fp32 GPU tensor over all ranks, on the current stream."""
        if flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous():
            raise TypeError('allreduce_ takes a contiguous float32 GPU tensor')
        check(lib.stair_allreduce_grads(self._comm, C.c_void_p(flat.data_ptr()), flat.numel(),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return flat

    def ranks(self):
        """(rank, size) as the RCCL communicator reports them (ncclCommUserRank, ncclCommCount)."""
        r, n = C.c_int32(-1), C.c_int32(-1)
        check(lib.stair_comm_info(self._comm, C.byref(r), C.byref(n)))
        return r.value, n.value

    def close(self):
        comm, self._comm = getattr(self, '_comm', None), None
        if comm:
            try:
                lib.stair_comm_destroy(comm)
            except Exception:           # interpreter shutdown: the library handle may already be gone
                pass

    __del__ = close
