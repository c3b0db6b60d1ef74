#!/bin/bash
# developer A/B of run-time switches: bash tools/ab_env.sh "KEY=VAL ..." ["KEY=VAL ..." ...]  -> one bench.py line per setting ("-" = defaults)
cd "$(dirname "$0")/.."
for kv in "$@"; do
  [ "$kv" = "-" ] && kv=""
  echo "setting [$kv]"
  env $kv timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > /tmp/ab_env.json 2>/dev/null && python3 tools/show_bench.py /tmp/ab_env.json
done
