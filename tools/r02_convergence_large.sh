#!/bin/bash
# The same Approach lineage as tools/r02_convergence.sh at the HEADLINE env count (GPU box): 4096 envs x 128 steps per iteration, 2048-row
# minibatches, 2x64 net, 50 iterations per phase (tools/chains_r02_k.json: short phases -- long ones over-optimise the shaped rewards, DESIGN.md
# section 8), then the Approach -> Finisher pipeline with the Finisher trained at 4096 envs as well.  The handoff-state buffer takes the
# final settled states (build_finisher_handoff_state_buffer.py's default mode): this policy settles in the last steps of its 24-step episodes,
# so a 2-step confirmed streak before the end is rare; the evaluation itself hands over as eval_workspace_expansion.py:138-139 does.
#   bash tools/r02_convergence_large.sh [seed] [outdir]         (defaults: seed 7, gpurun_out/r02_conv_large)
set -e
cd "$(dirname "$0")/.."
seed=${1:-7}; out=${2:-gpurun_out/r02_conv_large}
mkdir -p $out
sed "s/\"seed\": 7/\"seed\": $seed/" tools/chains_r02_k.json > $out/chain_s$seed.spec.json
python tools/train_chain.py $out/chain_s$seed.spec.json $out/approach_chain_s$seed.json --save $out/approach_s$seed > $out/approach_chain_s$seed.log 2>&1
grep EVAL $out/approach_chain_s$seed.log | tail -1 | cut -c1-400
python tools/pipeline_run.py $out/approach_s${seed}_phase6.zip $out/pipeline_s$seed.json --dock-envs 4096 --dock-n-steps 36 --dock-batch 2048 --dock-hidden 64 \
  --dock-steps 3e7 --dock-scratch-lr 1e-4 --dock-scratch-epochs 5 --dock-ft-steps 1e7 --dock-seed $seed --handoff-mode final_settled > $out/pipeline_s$seed.log 2>&1
grep "^approach_plus_finisher" $out/pipeline_s$seed.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('approach_plus_finisher '):])
    for k, v in d.items(): print('stage', k, 'success', v['success_rate'], 'final pos mm', round(1e3 * v['mean_final_position_error'], 2), 'ori', round(v['mean_final_orientation_error'], 4))
"
