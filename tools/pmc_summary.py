"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh into profiles/.

    python tools/pmc_summary.py gpurun_out/<dir> profiles/r01 [suffix]

An optional third argument is appended to every kernel name (e.g. "@32768" for the config-3 env-kernel passes of
tools/env_kernel_bench.py) and the records are MERGED into an existing <prefix>_pmc_traffic.json instead of replacing it.

writes <prefix>_pmc_kernels.csv (per kernel: launches, FETCH_SIZE KB, WRITE_SIZE KB, L2 hits / misses, averages per launch)
and <prefix>_pmc_traffic.json (bench.py's roofline.traffic source: HBM bytes per launch).

Corrections, as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE on gfx950 reports half the bytes of wide coalesced
streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Kernels that read with 4-byte lanes
(grad_finalize, adam) are uncalibrated: their doubled FETCH_SIZE is an upper bound.
"""
import collections
import csv
import json
import re
import sys

SHORT = {
    "gemm_tn_frag_kernel": "gemm_tn_frag",
    "gemm_tn_split_kernel": "gemm_tn_split",
    "grad_finalize_kernel": "grad_finalize",
    "adam_kernel": "adam",
    "kp1_step_kernel": "kp1_step",
    "head_infer_kernel": "head_infer",
    "adv_minibatch_sums_kernel": "adv_minibatch_sums",
    "curriculum_kernel": "curriculum",
}


def short_name(full: str) -> str:
    base = re.sub(r"\(anonymous namespace\)::", "", full)
    base = re.sub(r"^void ", "", base)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", base)
    name = m.group(1) if m else base[:40]
    if name == "gemm_nt_kernel" and m and m.group(2):
        return "gemm_nt" + m.group(2).replace(" ", "")
    if name == "mlp_tile_kernel":
        return "mlp_train_tile" if m.group(2) and "true" in m.group(2) else "mlp_infer_tile"
    if name == "kp1_step_kernel" and m and m.group(2):
        return "kp1_step" + m.group(2).replace(" ", "")
    return SHORT.get(name, name)


def collect(path: str) -> dict:
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            out[short_name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main() -> None:
    src, prefix = sys.argv[1], sys.argv[2]
    suffix = sys.argv[3] if len(sys.argv) > 3 else ""
    merged = collections.defaultdict(dict)
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        try:
            d = collect(f"{src}/{tag}/pmc_counter_collection.csv")
        except OSError:
            continue
        for k, counters in d.items():
            for c, vals in counters.items():
                merged[k][c] = (len(vals), sum(vals) / len(vals))
    rows, traffic = [], {}
    for k, c in sorted(merged.items()):
        if not k.startswith(("mlp_", "gemm_", "grad_", "adam", "kp1_", "head_", "adv_", "curriculum", "gae", "bootstrap")):
            continue
        fetch = c.get("FETCH_SIZE", (0, 0.0))
        write = c.get("WRITE_SIZE", (0, 0.0))
        hit, miss = c.get("TCC_HIT_sum", (0, 0.0))[1], c.get("TCC_MISS_sum", (0, 0.0))[1]
        hbm = fetch[1] * 1024 * 2 + write[1] * 1024
        k = k + suffix
        if suffix and k.startswith("kp1_step<"):
            k = "kp1_step_kernel" + suffix            # bench.py looks the config-3 record up under this name
        rows.append([k, fetch[0], round(fetch[1], 1), round(write[1], 1), round(hit), round(miss), round(hit / (hit + miss), 3) if hit + miss else "",
                     round(hbm)])
        traffic[k] = {"hbm_bytes_per_launch": round(hbm), "fetch_size_kb_raw": fetch[1], "write_size_kb": write[1], "launches": fetch[0],
                      "correction": "FETCH_SIZE x2 (gfx950 wide-load undercount) + WRITE_SIZE"}
    if suffix:
        try:
            with open(prefix + "_pmc_traffic.json") as f:
                traffic = {**json.load(f), **traffic}
        except OSError:
            pass
    with open(prefix + "_pmc_kernels" + suffix.replace("@", "_") + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg_raw", "WRITE_SIZE_KB_avg", "TCC_HIT_avg", "TCC_MISS_avg", "L2_hit_rate", "hbm_bytes_per_launch_corrected"])
        w.writerows(rows)
    with open(prefix + "_pmc_traffic.json", "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
