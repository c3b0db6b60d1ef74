#!/bin/bash
# Round-end evidence on the GPU box: full GPU test log, bench line, rocprofv3 kernel stats, PMC passes, in-kernel timelines.
# Usage: bash tools/round_artifacts.sh <tag>   (outputs under gpurun_out/<tag>*)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/$tag/gpu_tests.log 2>&1 && echo "gpu tests ok: $(tail -1 gpurun_out/$tag/gpu_tests.log)"
timeout -k 10 400 python bench.py > gpurun_out/$tag/bench.log 2>&1 && echo "bench ok"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/stats.log 2>&1 && echo "stats ok"
bash tools/pmc_passes.sh $tag/pmc
timeout -k 10 200 python tools/nt_timeline.py 8192 fused > gpurun_out/$tag/tile_timeline.log 2>&1 && echo "tile timeline ok"
timeout -k 10 200 python tools/nt_timeline.py 8192 tnfrag > gpurun_out/$tag/tnfrag_timeline.log 2>&1 && echo "tnfrag timeline ok"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.log 2>&1 && echo "smoke ok"
