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


def load_golden_config(name: str):
    """Resolved config dict the reference trainers/evaluators see (tests/golden/configs/*.json)."""
    from rl_brain_trainer_amd import config as kcfg

    cfg = json.loads((GOLDEN / "configs" / f"{name}.json").read_text())
    return kcfg.to_env_config(cfg, handoff_base_dirs=(GOLDEN,))


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN
