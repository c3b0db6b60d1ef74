#!/bin/bash
# Round-2 convergence run (GPU box): the Approach lineage at the reference's own training scale, then the Approach -> Finisher pipeline.
#   bash tools/r02_convergence.sh [seed] [outdir]         (defaults: seed 7, gpurun_out/r02_conv)
# 1. tools/train_chain.py tools/chains_r02_h.json with KP1_ENVS=16: approach_default -> approach_dock_coarse_ready_v1 -> approach_finisher_ready_v1
#    (x3, falling lr) -> approach_finisher_ready_v2_settle (x2), every phase with the reference's 16 envs x 1024 steps, minibatch 256, 2x64 net.
# 2. tools/pipeline_run.py: handoff-state buffer from that Approach policy, Finisher trained at 12 envs x 256 steps (its config's scale), evaluation of
#    Approach alone and Approach + Finisher on 200 held-out episodes per stage (eval_workspace_expansion.py:86-211 protocol).
# The engine is bitwise reproducible, so the same seed gives the same numbers on every run.
set -e
cd "$(dirname "$0")/.."
seed=${1:-7}; out=${2:-gpurun_out/r02_conv}
mkdir -p $out
sed "s/\"seed\": 7/\"seed\": $seed/" tools/chains_r02_h.json > $out/chain_s$seed.spec.json
KP1_ENVS=16 python tools/train_chain.py $out/chain_s$seed.spec.json $out/approach_chain_s$seed.json --save $out/approach_s$seed > $out/approach_chain_s$seed.log 2>&1
grep EVAL $out/approach_chain_s$seed.log | tail -1 | cut -c1-400
python tools/pipeline_run.py $out/approach_s${seed}_phase6.zip $out/pipeline_s$seed.json --dock-envs 12 --dock-n-steps 256 --dock-batch 256 --dock-hidden 64 \
  --dock-steps 2e6 --dock-scratch-lr 1e-4 --dock-scratch-epochs 5 --dock-ft-steps 1e6 --dock-seed $seed > $out/pipeline_s$seed.log 2>&1
tail -2 $out/pipeline_s$seed.log | cut -c1-1500
