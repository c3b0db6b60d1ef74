#!/bin/bash
# Round-end evidence on the GPU box, in the order the numbers depend on each other:
#   GPU tests + smoke -> rocprofv3 kernel stats of the bench command (bench.py quotes them as roofline.rocprof) -> config-3 env-kernel stats
#   -> PMC passes (bench kernels, then the env kernel at 32768 envs) summarised into profiles/<tag>_pmc_traffic.json (bench.py's `traffic`)
#   -> the full bench line.
# Usage: bash tools/round_artifacts.sh <tag>      outputs under gpurun_out/<tag>/, summaries also copied to profiles/<tag>_* on the box
# (profiles/ does not travel back: everything is mirrored under gpurun_out/<tag>/ and copied into profiles/ by hand afterwards).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1 && echo "gpu tests ok: $(tail -1 $out/gpu_tests.log)" || { echo "gpu tests FAILED"; tail -20 $out/gpu_tests.log; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 && echo "smoke ok: $(tail -1 $out/smoke.log | cut -c1-200)" || { echo "smoke FAILED"; tail -20 $out/smoke.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-extras --no-cpu-baseline > $out/stats.log 2>&1 || { echo "rocprof stats FAILED"; tail -5 $out/stats.log; exit 1; }
cp "$(find $out/stats -name bench_kernel_stats.csv | sort | tail -1)" profiles/${tag}_bench_kernel_stats.csv && cp profiles/${tag}_bench_kernel_stats.csv $out/ && echo "stats ok"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3 -o env -- python3 tools/env_kernel_bench.py > $out/env_stats.log 2>&1 || { echo "env rocprof FAILED"; exit 1; }
cp "$(find $out/cfg3 -name env_kernel_stats.csv | sort | tail -1)" $out/${tag}_config3_env_kernel_stats.csv && echo "config-3 env kernel stats ok"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$t -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $out/pmc.$t.log 2>&1 || { echo "pmc pass $c FAILED"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_env/$t -o pmc -- python3 tools/env_kernel_bench.py --launches 100 > $out/pmc_env.$t.log 2>&1 || { echo "env pmc pass $c FAILED"; exit 1; }
  echo "pmc pass $c ok"
done
bash tools/sq_passes.sh $tag/sq > $out/sq.log 2>&1 && python3 tools/sq_summary.py $out/sq profiles/$tag > $out/sq_summary.log 2>&1 && cp profiles/${tag}_sq_counters.* $out/ && rm -rf $out/sq && echo "sq counters ok" || { echo "sq passes FAILED"; tail -5 $out/sq.log; }
python tools/pmc_summary.py $out/pmc profiles/$tag > $out/pmc_summary.log 2>&1 && python tools/pmc_summary.py $out/pmc_env profiles/$tag @32768 >> $out/pmc_summary.log 2>&1 && cp profiles/${tag}_pmc_* $out/ && echo "pmc summary ok"
rm -rf $out/stats $out/cfg3 $out/pmc $out/pmc_env   # raw rocprof outputs: only the summaries travel back (gpurun_out/ is capped at 64 MiB)
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err && echo "bench ok" && python3 tools/bench_line.py $out/bench.json
