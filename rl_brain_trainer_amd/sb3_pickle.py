"""Pickle payloads of the six non-JSON members of an SB3 ``data`` entry, written WITHOUT gymnasium / stable-baselines3.

``stable_baselines3.common.save_util.json_to_data`` restores ``policy_class``, ``observation_space``, ``action_space``, ``lr_schedule``,
``clip_range`` and ``rollout_buffer_class`` with ``cloudpickle.loads(base64.b64decode(entry[":serialized:"]))`` (the loaders the demo uses:
kinematic_phase1/eval/eval_deterministic.py:66-79, v5/phase3a_controlled_sim.py:94-103 -> ``PPO.load``).  cloudpickle stores importable
classes and their instances exactly as ``pickle`` does -- a GLOBAL reference ``module\\nqualname`` plus the instance ``__dict__`` -- so the
payloads can be produced from stand-in classes that carry the same attribute state, pickled with protocol 2 (textual GLOBAL opcodes), with
the stand-ins' module path then rewritten to the real one.  Nothing of gymnasium / SB3 is imported, installed or executed here.

State written (pins: gymnasium==1.2.3, stable-baselines3==2.8.0, final_codes_docker/Dockerfile.demo:31-32):
  gymnasium.spaces.box.Box      dtype, _shape, low, high, low_repr, high_repr, bounded_below, bounded_above, _np_random = None
                                (Space.__setstate__ updates __dict__; Box.__setstate__ only fills missing *_repr)
  gymnasium.spaces.dict.Dict    spaces (key-sorted dict, as Dict.__init__ sorts a plain dict), _shape = None, dtype = None, _np_random = None
  stable_baselines3.common.utils.FloatSchedule     value_schedule = ConstantSchedule(val)   (what FloatSchedule(float) builds)
  classes by reference          stable_baselines3.common.policies.MultiInputActorCriticPolicy, stable_baselines3.common.buffers.DictRolloutBuffer

PARITY UNPINNED: neither library is importable in this image and the reference tree holds no archive to compare with, so the attribute sets
above restate the published sources from memory of their layout; ``checkpoint.sb3_loadable`` therefore stays False for archives written here
until one has been loaded by the real ``PPO.load`` (tools/finish_sb3_zip.py remains the verified-by-construction route).
"""
from __future__ import annotations

import base64
import pickle
from typing import Any

import numpy as np

_HERE = __name__   # the stand-ins are importable as <this module>.<name>, which is what pickle checks when it writes a GLOBAL


class _BoxStandIn:
    pass


class _DictStandIn:
    pass


class _FloatScheduleStandIn:
    pass


class _ConstantScheduleStandIn:
    pass


class _PolicyClassStandIn:
    pass


class _RolloutBufferClassStandIn:
    pass


# stand-in qualname -> (real module, real qualname)
_REAL = {
    "_BoxStandIn": ("gymnasium.spaces.box", "Box"),
    "_DictStandIn": ("gymnasium.spaces.dict", "Dict"),
    "_FloatScheduleStandIn": ("stable_baselines3.common.utils", "FloatSchedule"),
    "_ConstantScheduleStandIn": ("stable_baselines3.common.utils", "ConstantSchedule"),
    "_PolicyClassStandIn": ("stable_baselines3.common.policies", "MultiInputActorCriticPolicy"),
    "_RolloutBufferClassStandIn": ("stable_baselines3.common.buffers", "DictRolloutBuffer"),
}


def _dumps(obj: Any) -> bytes:
    """protocol-2 pickle of `obj` with every stand-in GLOBAL (``c<module>\\n<name>\\n``) rewritten to the class it stands for"""
    raw = pickle.dumps(obj, protocol=2)
    for stand_in, (module, name) in _REAL.items():
        raw = raw.replace(f"c{_HERE}\n{stand_in}\n".encode(), f"c{module}\n{name}\n".encode())
    assert _HERE.encode() not in raw, "a stand-in reference survived the rewrite"
    return raw


def box(low: float, high: float, n: int) -> _BoxStandIn:
    b = _BoxStandIn()
    lo, hi = np.full((n,), low, dtype=np.float32), np.full((n,), high, dtype=np.float32)
    b.__dict__.update({"dtype": np.dtype(np.float32), "_shape": (int(n),), "low": lo, "high": hi, "low_repr": str(float(low)), "high_repr": str(float(high)),
                       "bounded_below": np.full((n,), True), "bounded_above": np.full((n,), True), "_np_random": None})
    return b


def dict_space(boxes: dict[str, _BoxStandIn]) -> _DictStandIn:
    d = _DictStandIn()
    d.__dict__.update({"spaces": {k: boxes[k] for k in sorted(boxes)}, "_shape": None, "dtype": None, "_np_random": None})
    return d


def float_schedule(value: float) -> _FloatScheduleStandIn:
    c = _ConstantScheduleStandIn()
    c.__dict__.update({"val": float(value)})
    s = _FloatScheduleStandIn()
    s.__dict__.update({"value_schedule": c})
    return s


def serialized(obj: Any) -> str:
    """the ":serialized:" string of an SB3 data entry (base64 of the pickle stream)"""
    return base64.b64encode(_dumps(obj)).decode()


POLICY_CLASS = _PolicyClassStandIn
ROLLOUT_BUFFER_CLASS = _RolloutBufferClassStandIn


def describe(payload_b64: str) -> list[tuple[str, str]]:
    """(module, name) of every GLOBAL a payload refers to, read from the opcode stream WITHOUT unpickling it (tests; `pickletools.genops`)"""
    import pickletools

    out = []
    for op, arg, _pos in pickletools.genops(base64.b64decode(payload_b64)):
        if op.name == "GLOBAL":
            module, name = arg.split(" ")
            out.append((module, name))
    return out
