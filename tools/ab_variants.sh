# developer A/B: build the library with each set of -D flags given as arguments and time the MLP kernels on one box
set -e
cd $GRAFT_REPO_ROOT
S="rl_brain_trainer_amd/csrc/kp1_env.hip rl_brain_trainer_amd/csrc/kp1_ppo.hip rl_brain_trainer_amd/csrc/kp1_mlp.hip"
cp rl_brain_trainer_amd/libkp1.so /tmp/lib_base.so
for v in "$@"; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -shared -o /tmp/lib_$v.so $S 2>/dev/null &
done
wait
for v in base "$@"; do
cp /tmp/lib_$v.so rl_brain_trainer_amd/libkp1.so
echo "variant $v"; timeout -k 10 120 python tools/prof_mlp.py 8192 40 2>/dev/null
done
cp /tmp/lib_base.so rl_brain_trainer_amd/libkp1.so
