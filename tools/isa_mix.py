#!/usr/bin/env python3
"""Static instruction mix of one kernel of rl_brain_trainer_amd/libkp1.so (gfx950 disassembly; no GPU needed).

    python3 tools/isa_mix.py 'kp1_step_kernel<float, 0, false>' [--lib PATH] [--dump FILE]

Counts by class (VALU f32 / f64 / int, transcendental, 64-bit multiplies of the PCG64 stream, SALU, scalar and vector memory, branches) over
the whole kernel text.  A static count is an upper bound of what a wave that takes every path issues; the dynamic split (step-only vs
step + reset) comes from the SQ counter pass (tools/pmc_passes.sh).
"""
from __future__ import annotations

import argparse
import collections
import re
import shutil
import subprocess
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
ROOT = Path(__file__).resolve().parents[1]


def disassemble(lib: Path) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        copy = Path(tmp) / lib.name
        shutil.copy(lib, copy)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        out = []
        for co in sorted(Path(tmp).iterdir()):
            if "amdgcn" in co.name:
                dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout
                out.append(subprocess.run(["c++filt"], input=dis, check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_transcendental" + ("_f64" if "_f64" in op else "")
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32")):
        return "valu_int_mul"
    if op.startswith("v_"):
        if "_f64" in op:
            return "valu_f64"
        if "_f32" in op or "_f16" in op:
            return "valu_f32"
        return "valu_other"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_getpc")):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--lib", default=str(ROOT / "rl_brain_trainer_amd" / "libkp1.so"))
    ap.add_argument("--dump")
    args = ap.parse_args()
    text = disassemble(Path(args.lib))
    body, on = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            on = args.kernel in m.group(1)
            continue
        if on and line.strip():
            body.append(line)
    if not body:
        raise SystemExit(f"no kernel matching {args.kernel!r}")
    if args.dump:
        Path(args.dump).write_text("\n".join(body))
    mix = collections.Counter()
    ops = collections.Counter()
    for line in body:
        parts = line.split()
        if not parts:
            continue
        op = parts[0]
        mix[classify(op)] += 1
        ops[op] += 1
    total = sum(mix.values())
    print(f"{args.kernel}: {total} instructions (static)")
    for k, v in mix.most_common():
        print(f"  {k:28s} {v:6d}  {100.0 * v / total:5.1f} %")
    print("  top opcodes: " + ", ".join(f"{o} {c}" for o, c in ops.most_common(14)))


if __name__ == "__main__":
    main()
