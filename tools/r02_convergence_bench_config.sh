#!/bin/bash
# Convergence of the BENCHMARKED configuration itself (GPU box): the Approach lineage of tools/r02_convergence_large.sh with bench.py's
# workload shape -- 4096 envs x 128 steps per iteration, the 2x256 net, 8192-row minibatches, 50 iterations per phase (tools/chains_r02_n.json)
# -- then the Approach -> Finisher pipeline with the Finisher trained in bench.py's Finisher shape (4096 envs x 72 steps, 8192-row
# minibatches, 2x256) and the reference's evaluation protocol on 200 held-out episodes per stage.
#   bash tools/r02_convergence_bench_config.sh [seed] [outdir]         (defaults: seed 7, gpurun_out/r02_conv_bench)
set -e
cd "$(dirname "$0")/.."
seed=${1:-7}; out=${2:-gpurun_out/r02_conv_bench}
mkdir -p $out
sed "s/\"seed\": 7/\"seed\": $seed/" tools/chains_r02_n.json > $out/chain_s$seed.spec.json
python tools/train_chain.py $out/chain_s$seed.spec.json $out/approach_chain_s$seed.json --save $out/approach_s$seed > $out/approach_chain_s$seed.log 2>&1
grep EVAL $out/approach_chain_s$seed.log | tail -1 | cut -c1-330
python tools/pipeline_run.py $out/approach_s${seed}_phase6.zip $out/pipeline_s$seed.json --dock-envs 4096 --dock-n-steps 72 --dock-batch 8192 --dock-hidden 256 \
  --dock-steps 3e7 --dock-scratch-lr 1e-4 --dock-scratch-epochs 5 --dock-ft-steps 1e7 --dock-seed $seed --handoff-mode final_settled > $out/pipeline_s$seed.log 2>&1
python3 - $out/pipeline_s$seed.json <<'PY'
import json, sys
ev = json.load(open(sys.argv[1]))["evaluation"]["approach_plus_finisher"]
for k, v in ev.items():
    print("stage", k, "success", v["success_rate"], "final pos mm", round(1e3 * v["mean_final_position_error"], 2), "ori", round(v["mean_final_orientation_error"], 4))
PY
