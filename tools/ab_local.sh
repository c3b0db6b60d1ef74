#!/bin/bash
# developer A/B, two halves.  HERE (no GPU):  bash tools/ab_local.sh build NAME=-DFLAG[,-DFLAG2] ...   builds tools/build/variants/libkp1_NAME.so
# (only kp1_mlp.hip is recompiled per variant).  On the GPU box:  bash tools/ab_local.sh run|bench [NAME ...]  times the MLP kernels with each
# (bench: also one bench.py run per variant, i.e. the kernels in situ).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
V=$ROOT/tools/build/variants
C=$ROOT/rl_brain_trainer_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-function-calls=false"
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $V
  for f in kp1_env kp1_ppo; do
    if [ ! -f $V/$f.o ] || [ $C/$f.hip -nt $V/$f.o ] || [ $C/kp1_device.hpp -nt $V/$f.o ]; then /opt/rocm/bin/hipcc $FLAGS -c -o $V/$f.o $C/$f.hip & fi
  done
  wait
  for spec in "$@"; do
    name=${spec%%=*}; defs=$(echo "${spec#*=}" | tr ',' ' '); [ "$name" = "$spec" ] && defs=""
    ( /opt/rocm/bin/hipcc $FLAGS $defs -c -o $V/mlp_$name.o $C/kp1_mlp.hip && /opt/rocm/bin/hipcc $FLAGS -shared -o $V/libkp1_$name.so $V/kp1_env.o $V/kp1_ppo.o $V/mlp_$name.o && echo "built $name ($defs)" ) &
  done
  wait
else
  cd $ROOT
  cp rl_brain_trainer_amd/libkp1.so /tmp/lib_keep.so
  for name in "$@"; do
    env_kv=""; lib=$name
    case "$name" in *@*) lib=${name%%@*}; env_kv=${name#*@};; esac
    cp $V/libkp1_$lib.so rl_brain_trainer_amd/libkp1.so
    echo "variant $name"; env $env_kv timeout -k 10 120 python tools/prof_mlp.py 8192 40 2>/dev/null | tail -1
    if [ "$mode" = bench ]; then   # the kernels in situ: one bench run per variant (in-situ event timing of a whole update epoch)
      env $env_kv timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > /tmp/ab_bench.json 2>/dev/null && python3 tools/bench_line.py /tmp/ab_bench.json
    fi
  done
  cp /tmp/lib_keep.so rl_brain_trainer_amd/libkp1.so
fi
