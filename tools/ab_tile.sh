timeout -k 10 300 python -m pytest tests/test_ppo_kernels_gpu.py -m gpu -x -q -k "fused_tile or loss_grad or optimiser_step or advantage_modes" 2>&1 | tail -2
python tools/prof_mlp.py 8192 40 2>/dev/null
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null > /tmp/ab_quick.json && python3 tools/show_bench.py /tmp/ab_quick.json | cut -c1-250
