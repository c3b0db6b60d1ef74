"""rccl.py binds the RCCL build torch itself loads: the entry points and the enum values it relies on (no GPU, no communicator)."""
from __future__ import annotations

import ctypes as C

import torch

from rl_brain_trainer_amd import rccl


def test_rccl_library_exports_the_bound_entry_points():
    lib = rccl._load()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    assert lib.ncclGetErrorString(0) == b"no error"
    # ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank (nccl.h: #define NCCL_UNIQUE_ID_BYTES 128)
    assert C.sizeof(rccl._UniqueId) == 128


def test_rccl_datatype_codes_match_nccl_h():
    # ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7, ncclFloat64 8
    assert rccl._NCCL_DTYPE == {torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float32: 7, torch.float64: 8}
    assert rccl._NCCL_SUM == 0   # ncclRedOp_t: ncclSum 0
    import re
    from pathlib import Path

    hdr = Path("/opt/rocm/include/rccl/rccl.h")
    if hdr.exists():   # the image's own header, when present, is the authority
        text = hdr.read_text()
        for sym, val in (("ncclUint8", 1), ("ncclInt32", 2), ("ncclInt64", 4), ("ncclFloat32", 7), ("ncclFloat64", 8)):
            m = re.search(rf"\b{sym}\s*=\s*(\d+)", text)
            assert m and int(m.group(1)) == val, sym
        assert re.search(r"NCCL_UNIQUE_ID_BYTES\s+128", text)
