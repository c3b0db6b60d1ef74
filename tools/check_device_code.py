#!/usr/bin/env python3
"""Build-time guard on the gfx950 code objects inside rl_brain_trainer_amd/libkp1.so (called by __graft_entry__.build()).

Round 1 lost a GPU to a "Memory access fault ... on address (nil)" in kp1_step_kernel<double, APPROACH, false> while its reset path
still went through NON-INLINED device functions.  Every device function of the library is meant to be inlined
into its kernel; this script makes that a checked property instead of a convention:

  * no `s_swappc_b64` (device function call) anywhere in the device code;
  * every `s_setpc_b64` is a long-branch expansion (preceded by `s_getpc_b64` in the same kernel), not a return;
  * prints per-kernel scratch (private segment) sizes so a jump in spill volume is visible in the build log.

Exit status 1 on a violation.  Needs only llvm-objdump / llvm-readelf from /opt/rocm (no GPU).
"""
from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def device_code_objects(lib: Path, work: Path) -> list[Path]:
    copy = work / lib.name
    shutil.copy(lib, copy)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=work)
    return sorted(p for p in work.iterdir() if "amdgcn" in p.name)


def check(lib: Path, verbose: bool = False) -> int:
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in device_code_objects(lib, Path(tmp)):
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout
            kernel, getpc_seen = "?", False
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel, getpc_seen = m.group(1), False
                    continue
                if "s_getpc_b64" in line:
                    getpc_seen = True
                if "s_swappc_b64" in line:
                    print(f"[check_device_code] device function CALL in {kernel}: {line.strip()}", file=sys.stderr)
                    bad += 1
                if "s_setpc_b64" in line and not getpc_seen:
                    print(f"[check_device_code] s_setpc_b64 without a preceding s_getpc_b64 (a function return?) in {kernel}", file=sys.stderr)
                    bad += 1
            if verbose:
                notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
                names = re.findall(r"\.name:\s+(\S+)", notes)
                sizes = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
                for n, s in zip(names, sizes):
                    if int(s) > 0:
                        print(f"[check_device_code] scratch {int(s):5d} B/lane  {n[:100]}")
    return bad


def main() -> int:
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else Path(__file__).resolve().parent.parent / "rl_brain_trainer_amd" / "libkp1.so"
    bad = check(lib, verbose="-v" in sys.argv)
    if bad:
        print(f"[check_device_code] {bad} violation(s) in {lib}", file=sys.stderr)
        return 1
    print(f"[check_device_code] {lib.name}: no device function calls")
    return 0


if __name__ == "__main__":
    sys.exit(main())
