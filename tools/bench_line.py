import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), round(d["config"]["update_ms"],2), {k:round(v,1) for k,v in d["roofline"]["optimizer_step_kernels_us"].items()}, round(d["roofline"]["optimizer_step_us"],1))
