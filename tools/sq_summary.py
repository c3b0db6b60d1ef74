#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/sq_passes.sh.

    python3 tools/sq_summary.py gpurun_out/<dir> profiles/r03

writes <prefix>_sq_counters.json / .csv: per kernel (bench run: mlp_tile_kernel<true>, gemm_tn_split_kernel, ...; env run:
kp1_step_kernel at 32768 stage-11 envs, step path only) the per-launch average of every counter, plus derived figures:
  per_wave                 counter / SQ_WAVES for the instruction counters (instructions one wave issues)
  mfma_busy_frac           SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES scaled to the same unit): see NOTE below
  issue_cycle_split        ACTIVE_INST_ANY / WAIT_INST_ANY / WAIT_ANY as fractions of WAVE_CYCLES (disjoint, guide "rocprofv3 PMC slots")
Counter units (guide, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES counts cycles summed over shader engines/XCDs as rocprofv3 reports it.
"""
import collections
import csv
import glob
import json
import re
import sys

KEEP = ("mlp_tile_kernel", "gemm_tn_split_kernel", "kp1_step_kernel", "grad_finalize_kernel", "adam_kernel", "curriculum_kernel")


def short(full: str) -> str:
    base = re.sub(r"\(anonymous namespace\)::", "", full)
    base = re.sub(r"^void ", "", base)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", base)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else base[:60]


def collect(root: str) -> dict:
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = short(r["Kernel_Name"])
                if name.startswith(KEEP):
                    out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main() -> None:
    src, prefix = sys.argv[1], sys.argv[2]
    table = {}
    for run in ("bench", "env"):
        for name, counters in collect(f"{src}/{run}").items():
            key = name + ("@32768" if run == "env" else "")
            rec = {c: sum(v) / len(v) for c, v in counters.items()}
            rec["launches_sampled"] = max(len(v) for v in counters.values())
            waves = rec.get("SQ_WAVES")
            if waves:
                rec["per_wave"] = {c: rec[c] / waves for c in rec if c.startswith("SQ_INSTS") and isinstance(rec[c], float)}
            wc = rec.get("SQ_WAVE_CYCLES")
            if wc and "SQ_ACTIVE_INST_ANY" in rec:
                rec["issue_cycle_split"] = {k: rec[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in rec}
            table[key] = rec
    with open(prefix + "_sq_counters.json", "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    cols = sorted({c for rec in table.values() for c in rec if isinstance(rec[c], float)})
    with open(prefix + "_sq_counters.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + cols)
        for k in sorted(table):
            w.writerow([k] + [f"{table[k].get(c, float('nan')):.6g}" for c in cols])
    for k in sorted(table):
        rec = table[k]
        pw = rec.get("per_wave", {})
        print(k, "waves", rec.get("SQ_WAVES"), "VALU/wave", round(pw.get("SQ_INSTS_VALU", 0)), "SALU/wave", round(pw.get("SQ_INSTS_SALU", 0)),
              "MFMA/wave", round(pw.get("SQ_INSTS_MFMA", 0)), "split", {a: round(b, 3) for a, b in rec.get("issue_cycle_split", {}).items()})


if __name__ == "__main__":
    main()
