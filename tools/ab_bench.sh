# developer A/B: build the library with each set of -D flags (comma-separated inside one argument) and run prof + bench on one box
set -e
cd $GRAFT_REPO_ROOT
S="rl_brain_trainer_amd/csrc/kp1_env.hip rl_brain_trainer_amd/csrc/kp1_ppo.hip rl_brain_trainer_amd/csrc/kp1_mlp.hip"
cp rl_brain_trainer_amd/libkp1.so /tmp/lib_base.so
i=0
for v in "$@"; do
i=$((i+1))
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $(echo $v | tr ',' ' ') -shared -o /tmp/lib_$i.so $S 2>/dev/null &
done
wait
i=0
for v in base "$@"; do
cp /tmp/lib_$([ $v = base ] && echo base || echo $i).so rl_brain_trainer_amd/libkp1.so
echo "variant $v"; timeout -k 10 120 python tools/prof_mlp.py 8192 40 2>/dev/null | tail -1
timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), d['config']['update_ms'])"
i=$((i+1))
done
cp /tmp/lib_base.so rl_brain_trainer_amd/libkp1.so
