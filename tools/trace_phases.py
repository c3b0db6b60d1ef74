#!/usr/bin/env python3
"""Per-update-phase durations of a kernel from a rocprofv3 kernel trace (bench_kernel_trace.csv): the warm-up iterations of a bench run and
the first launches of every update phase are slower than the steady state the timed iterations run in.

    python tools/trace_phases.py <bench_kernel_trace.csv> [kernel-name substring, default mlp_tile_kernel<true]
"""
import csv
import statistics as st
import sys

name = sys.argv[2] if len(sys.argv) > 2 else "mlp_tile_kernel<true"
rows = sorted((r for r in csv.DictReader(open(sys.argv[1])) if name in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
launches = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
phases = [[]]
for i, (t, d) in enumerate(launches):
    if i and t - launches[i - 1][0] > 2_000_000:   # > 2 ms without a launch of this kernel: a rollout lies in between
        phases.append([])
    phases[-1].append(d)
print(f"{name}: {len(launches)} launches in {len(phases)} phases; overall mean {st.mean(d for _, d in launches):.2f} us")
for k, p in enumerate(phases):
    if len(p) >= 256:
        print(f"phase {k:2d} ({len(p)} launches): first 64 {st.mean(p[:64]):6.2f}  64-128 {st.mean(p[64:128]):6.2f}  128-256 {st.mean(p[128:256]):6.2f}  "
              f"rest {st.mean(p[256:]):6.2f}  all {st.mean(p):6.2f} us")
    else:
        print(f"phase {k:2d} ({len(p)} launches): all {st.mean(p):6.2f} us")
