#!/bin/bash
# Developer A/B, part 1 (build container, no GPU): one library per -D setting, with the product's own flags, into build_ab/ (travels to the GPU box).
#   bash tools/ab_build.sh name1 "-DX=1" name2 "-DX=2 -DY" ...        -> build_ab/libkp1_<name>.so   ("base" = no extra flags)
# part 2 (GPU box): bash tools/ab_run.sh name1 name2 ... -- <command>   copies each variant over rl_brain_trainer_amd/libkp1.so and runs the command
cd "$(dirname "$0")/.."
mkdir -p build_ab
S="rl_brain_trainer_amd/csrc/kp1_env.hip rl_brain_trainer_amd/csrc/kp1_ppo.hip rl_brain_trainer_amd/csrc/kp1_mlp.hip"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-function-calls=false"
pids=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $F $flags -shared -o build_ab/libkp1_$name.so $S 2> build_ab/$name.log && echo "built $name [$flags]" || { echo "FAILED $name"; tail -5 build_ab/$name.log; } ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 3 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
