"""pytest configuration: markers and shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no compute).
`-m gpu`: parity tests proper; they call the HIP path through the C ABI on a real MI355X.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


# Order of a `-m gpu -x` run: parity of the hot path first (env / FK / reset / reward -> route env -> PPO kernels -> evaluator / coverage),
# then the trainers' CLIs, then the distributed paths, and the subprocess-spawning bench test LAST, so that a box-dependent failure in
# the outer layers can never hide the parity suite (round 2's driver run stopped at its first collected test).
_ORDER = (
    "test_oracle_golden", "test_route_oracle_golden", "test_host_logic",                     # CPU: oracle pinned, host logic
    "test_env_parity_gpu", "test_route_env_gpu",                                              # a1-a9, a15
    "test_ppo_kernels_gpu", "test_small_batch_gpu", "test_route_ppo_gpu",                     # a12
    "test_route_curriculum", "test_finisher_tools",                                           # a10, f3, f4
    "test_eval_checkpoint_gpu", "test_workspace_coverage", "test_sb3_zip",                    # a13, a14, f1
    "test_train_cli_gpu",                                                                     # f2
    "test_distributed_cpu", "test_distributed_gpu",                                           # e
    "test_bench_spawn_gpu",                                                                   # d (spawns bench.py subprocesses)
)


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}
    mid = rank["test_train_cli_gpu"]          # files not listed run just before the CLI / distributed / bench tail
    items.sort(key=lambda it: rank.get(Path(str(it.fspath)).stem, mid - 0.5))   # list.sort is stable: order inside a file is kept


def load_golden_config(name: str):
    """Resolved config dict the reference trainers/evaluators see (tests/golden/configs/*.json)."""
    from rl_brain_trainer_amd import config as kcfg

    cfg = json.loads((GOLDEN / "configs" / f"{name}.json").read_text())
    return kcfg.to_env_config(cfg, handoff_base_dirs=(GOLDEN,))


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN
