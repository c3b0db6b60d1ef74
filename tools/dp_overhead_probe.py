"""Developer measurement (one GPU): what the data-parallel code path costs per optimiser step with a ONE-rank RCCL group, and how much of it is the
collective itself: bench.py under KP1_DIST_FORCE_SINGLE=1 with the launch-stream all-reduce / all-gather replaced by nothing / a device copy
(legal at one rank: the sum over one rank is the identity).   python tools/dp_overhead_probe.py [--stub]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KP1_DIST_FORCE_SINGLE"] = "1"
stub = "--stub" in sys.argv
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline"]
from rl_brain_trainer_amd import rccl  # noqa: E402

if stub:
    rccl.RcclComm.all_reduce_sum = lambda self, t: None
    rccl.RcclComm.all_gather = lambda self, out, t: out.copy_(t.view(-1))
import bench  # noqa: E402

bench.main()
