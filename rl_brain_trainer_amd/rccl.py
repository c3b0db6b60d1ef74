"""RCCL collectives enqueued on the CALLER's HIP stream (ctypes on librccl, one communicator per process group).

torch.distributed's NCCL backend runs every collective on a stream of its own: an event hop from the launch stream to that stream and one back
per call.  The data-parallel update issues one gradient all-reduce between `grad_finalize` and `adam` of EVERY optimiser step (512 per
iteration), so those two hops sit on the critical path 512 times; measured with a one-rank group on one MI355X the torch path
costs 19 us per optimiser step on top of the kernels (57.2 ms per iteration against 47.1 ms without a process group).  Here the same RCCL
kernels go straight into the stream the tile / weight-gradient / Adam kernels are on -- in order, no events -- whether that stream is being
replayed between graph segments or captured.

The communicator is bootstrapped the way torch bootstraps its own: rank 0 draws an ncclUniqueId, torch.distributed carries the 128 bytes to
the other ranks, every rank calls ncclCommInitRank.  `Dist` does this in a helper thread with a deadline, runs `self_test` on every rank and
agrees (through torch.distributed) that all ranks passed before any training collective uses it; otherwise -- an exception, a wrong sum, no answer
within the deadline -- the torch path stays.

Reference: the reference has no collectives (one host process); SURVEY.md 8(e) states the exchange steps this carries.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_NCCL_DTYPE = {torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float32: 7, torch.float64: 8}
_NCCL_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _load() -> C.CDLL:
    """the RCCL build torch itself loaded (same instance: one RCCL per process), else the ROCm one"""
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    last: Exception | None = None
    for p in cands:
        try:
            lib = C.CDLL(p)
            break
        except OSError as exc:
            last = exc
    else:
        raise OSError(f"librccl.so not found ({last})")
    vp = C.c_void_p
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetErrorString.restype = C.c_char_p
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclAllGather, lib.ncclCommDestroy):
        f.restype = C.c_int
    return lib


class RcclComm:
    def __init__(self, dist, device: torch.device) -> None:
        """collective: call on every rank of the default process group, with `device` current"""
        self.L = _load()
        self.device = device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]   # the raw 128 bytes (a c_char array read as .value stops at a NUL)
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.addressof(uid), box[0], 128)
        self._comm = C.c_void_p()
        torch.cuda.set_device(device)
        self._check(self.L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what}: {self.L.ncclGetErrorString(rc).decode()} ({rc})")

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """in place, on torch's current stream"""
        assert t.is_contiguous() and t.device == self.device
        self._check(self.L.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _NCCL_DTYPE[t.dtype], _NCCL_SUM, self._comm, self._stream()),
                    "ncclAllReduce")

    def all_gather(self, out: torch.Tensor, t: torch.Tensor) -> None:
        """out[world * t.numel()] <- every rank's t, rank-major, on torch's current stream"""
        assert t.is_contiguous() and out.is_contiguous() and out.numel() == self.world * t.numel() and out.dtype == t.dtype
        self._check(self.L.ncclAllGather(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), t.numel(), _NCCL_DTYPE[t.dtype], self._comm, self._stream()),
                    "ncclAllGather")

    def self_test(self) -> bool:
        """the three exchanges the training loop makes (float gradient sum, double statistics sum, done-byte gather) on known values"""
        w, r, dev = self.world, self.rank, self.device
        g = torch.full((4099,), float(r + 1), device=dev)
        s = torch.full((192,), float(r + 1), dtype=torch.float64, device=dev)
        b = torch.full((80,), r + 1, dtype=torch.uint8, device=dev)
        got = torch.zeros((w, 80), dtype=torch.uint8, device=dev)
        self.all_reduce_sum(g)
        self.all_reduce_sum(s)
        self.all_gather(got.view(-1), b)
        torch.cuda.synchronize(dev)
        want = float(w * (w + 1) // 2)
        rows = torch.arange(1, w + 1, dtype=torch.uint8, device=dev)[:, None].expand(-1, 80)
        return bool(torch.all(g == want).item()) and bool(torch.all(s == want).item()) and bool(torch.equal(got, rows))

    def close(self) -> None:
        if self._comm:
            self.L.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
