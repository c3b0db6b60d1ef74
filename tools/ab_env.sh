#!/bin/bash
# developer A/B of run-time switches: bash tools/ab_env.sh "KEY=VAL ..." ["KEY=VAL ..." ...]  -> one bench.py line per setting ("-" = defaults)
cd "$(dirname "$0")/.."
for kv in "$@"; do
  [ "$kv" = "-" ] && kv=""
  echo "setting [$kv]"
  env $kv timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > /tmp/ab_env.json 2>/dev/null && python3 - <<'PY'
import json
d = json.loads(open('/tmp/ab_env.json').read().strip().splitlines()[-1])
print(round(d['value']), 'ms/iter', round(d['ms_per_step'], 2), 'rollout', round(d['config']['rollout_ms'], 2), 'update', round(d['config']['update_ms'], 2))
PY
done
