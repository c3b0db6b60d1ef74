for n in 4096 32768 524288; do python tools/env_kernel_bench.py --envs $n --launches 200 | cut -c1-70; done
