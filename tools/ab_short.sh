# one variant: the short bench line (developer A/B)
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null > /tmp/b.json; python3 -c "
import json; d=json.load(open('/tmp/b.json')); r=d['roofline']; c=d['config']; print(round(d['value']), round(c['rollout_ms'],2), round(c['update_ms'],2), {k: round(v,2) for k,v in r['optimizer_step_kernels_us'].items()})"
