#!/bin/bash
# HBM-traffic counter passes for the roofline block (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"): FETCH_SIZE and
# WRITE_SIZE need separate passes (TCC has 4 slots, they cost 3 + 2); L2 hit/miss in a third.  No trace domains besides
# --kernel-trace are combined with --pmc.  Usage (GPU box): bash tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args]
set -e
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$tag -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out.$tag.log 2>&1
  echo "pass $c done"
done
