#!/bin/bash
# Developer A/B, part 2 (GPU box): bash tools/ab_run.sh name1 name2 ... -- <command...>   (variants built by tools/ab_build.sh)
cd "$GRAFT_REPO_ROOT"
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
shift
cp rl_brain_trainer_amd/libkp1.so /tmp/libkp1_keep.so
for n in "${names[@]}"; do
  cp build_ab/libkp1_$n.so rl_brain_trainer_amd/libkp1.so
  echo "== variant $n"
  "$@" || { echo "variant $n: command failed"; cp /tmp/libkp1_keep.so rl_brain_trainer_amd/libkp1.so; exit 1; }
done
cp /tmp/libkp1_keep.so rl_brain_trainer_amd/libkp1.so
