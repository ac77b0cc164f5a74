"""Direct RCCL calls for the data-parallel exchange steps (one process per GPU, xGMI).

Why not only `torch.distributed`: every `dist.all_reduce` goes through ProcessGroupNCCL — ≈ 60 µs
of host work per call on this stack, 584 SyncBatchNorm statistics exchanges per C3 train step —
and its watchdog thread aborts the process when a collective is recorded inside a HIP-graph
capture ("operation not permitted on an event last recorded in a capturing stream", PyTorch
2.10 / ROCm 7.0, measured r03).  With one eager launch per kernel the N > 1 step is host-bound
at ≈ 2.5× the single-GPU graph step.  RCCL itself is capture-safe: `ncclAllReduce` on the
capturing stream becomes graph nodes.  This module binds the five entry points that needs
(ctypes over the librccl.so PyTorch already loaded) and creates ONE extra communicator next to
torch's; the unique id travels over the existing torch process group.

Reference call sites this serves: tools/train.py:73-79 (process group + SyncBatchNorm
conversion), :108-111 (DistributedDataParallel gradient averaging).
"""
import ctypes
import os

import torch

_NCCL_UNIQUE_ID_BYTES = 128
# ncclDataType_t / ncclRedOp_t (nccl.h; RCCL keeps NCCL's numbering)
_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6,
           torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
_OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3, "avg": 4}


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * _NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RuntimeError("segmentron_amd.rccl: %s not found" % path)
        lib = ctypes.CDLL(path)
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId,
                                         ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy",
                     "ncclGroupStart", "ncclGroupEnd"):
            getattr(lib, name).restype = ctypes.c_int
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("segmentron_amd.rccl: %s failed (%d): %s"
                           % (what, rc, _load().ncclGetErrorString(rc).decode()))


def new_unique_id():
    """-> 128 bytes (call on ONE rank, hand the bytes to the others)."""
    uid = _UniqueId()
    _check(_load().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(ctypes.byref(uid), _NCCL_UNIQUE_ID_BYTES)


class Communicator:
    """One RCCL communicator over `world` ranks (this process = `rank`, on the current device).
    Collectives are issued on torch's CURRENT stream — also while it is being captured."""

    def __init__(self, rank, world, unique_id):
        assert len(unique_id) == _NCCL_UNIQUE_ID_BYTES
        self.rank, self.world = int(rank), int(world)
        uid = _UniqueId()
        ctypes.memmove(ctypes.byref(uid), unique_id, _NCCL_UNIQUE_ID_BYTES)
        self._comm = ctypes.c_void_p()
        _check(_load().ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank),
               "ncclCommInitRank")

    def all_reduce(self, t, op="sum"):
        """In place on `t` (contiguous device tensor), on the current stream."""
        if not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("rccl.all_reduce: contiguous HIP tensor required")
        stream = torch.cuda.current_stream(t.device).cuda_stream
        p = t.data_ptr()
        _check(_load().ncclAllReduce(p, p, t.numel(), _DTYPES[t.dtype], _OPS[op], self._comm,
                                     stream), "ncclAllReduce")
        return t

    def all_reduce_many(self, tensors, op="sum"):
        """One grouped call (RCCL fuses the launches): in place on every tensor."""
        lib = _load()
        _check(lib.ncclGroupStart(), "ncclGroupStart")
        try:
            for t in tensors:
                self.all_reduce(t, op)
        finally:
            _check(lib.ncclGroupEnd(), "ncclGroupEnd")

    def destroy(self):
        if self._comm:
            _load().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()


def communicator_from_torch_group():
    """A second communicator over the ranks of torch's default process group: rank 0 draws the
    unique id and broadcasts it through that group.  Every rank must call this."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.zeros(_NCCL_UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(new_unique_id()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(buf, src=0)
    uid = bytes(buf.cpu().numpy().tobytes())
    comm = Communicator(rank, world, uid)
    # self-check against the torch group: sum over ranks of (rank + 1)
    probe = torch.full((8,), float(rank + 1), dtype=torch.float64, device=dev)
    comm.all_reduce(probe)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2.0
    if not bool((probe == want).all()):
        comm.destroy()
        raise RuntimeError("segmentron_amd.rccl: self-check failed (%r != %r)"
                           % (probe.tolist(), want))
    return comm
