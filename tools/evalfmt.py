import sys,json
for l in sys.stdin:
    d=json.loads(l[l.index("{"):]); s5=d["stage5"]; print(d["label"][:44].ljust(44), "| s5 succ", s5["success_rate"], "fr", s5["finisher_ready_hit_rate"], "pos", s5["mean_final_position_error"], "ori", s5["mean_final_orientation_error"], "| s0 fr", d["stage0"]["finisher_ready_hit_rate"], d["stage0"]["mean_final_position_error"], "ls", round(d["log_std"],2), "wall", d.get("wall_s"))
