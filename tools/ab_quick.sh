# one variant's quick numbers: env kernel at three sizes, MLP kernels back to back, the short bench line
for n in 4096 32768; do python tools/env_kernel_bench.py --envs $n --launches 60 2>/dev/null | cut -c1-70; done
python tools/prof_mlp.py 8192 40 2>/dev/null
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null > /tmp/ab_quick.json && python3 tools/show_bench.py /tmp/ab_quick.json
