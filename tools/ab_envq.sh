for n in 4096 32768 524288; do python tools/env_kernel_bench.py --envs $n --launches 100 2>/dev/null | cut -c1-70; done
bash tools/ab_short.sh | cut -c1-40
