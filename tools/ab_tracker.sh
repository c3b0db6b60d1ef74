# one variant: the tracker tests, then rocprofv3 kernel stats of the short bench (the curriculum kernel's row) and the bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_ppo_kernels_gpu.py -m gpu -x -q -k "curriculum" 2>&1 | tail -1
rm -rf /tmp/trk && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trk -o t -- python3 bench.py --no-extras --no-cpu-baseline > /tmp/trk.json 2>/dev/null
python3 tools/kstats.py "$(find /tmp/trk -name t_kernel_stats.csv | tail -1)" 14 | grep -i "curriculum\|env_step\|forward_env"
python3 tools/show_bench.py /tmp/trk.json | cut -c1-100
