"""print the headline fields and the extra blocks of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/iter", round(d["ms_per_step"], 2), "rollout", round(d["config"]["rollout_ms"], 2), "update", round(d["config"]["update_ms"], 2))
for k in ("finisher", "config3_env_kernel", "minibatch512", "config4_eval_shard", "approach_finisher_combined"):
    if k in d:
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a not in ("workload", "note")})
r = d["roofline"]
print("roofline", round(r["frac"], 3), r["optimizer_step_kernels_us"], "epochs", r.get("insitu_epochs_us"), "rocprof", r.get("rocprof"))
