"""print a rocprofv3 *_kernel_stats.csv compactly: python tools/kstats.py <csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
    print(f"{name[:64]:64s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  min {float(r['MinNs']) / 1e3:8.2f}  max {float(r['MaxNs']) / 1e3:9.2f}  {float(r['Percentage']):5.1f} %")
