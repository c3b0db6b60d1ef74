# one variant: rollout / update split of the short bench, three times (developer A/B)
for k in 1 2 3; do timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null > /tmp/ab_r.json && python3 tools/show_bench.py /tmp/ab_r.json | head -1; done
