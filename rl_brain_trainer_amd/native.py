"""ctypes binding of the C ABI in include/kp1.h (librl kp1: rl_brain_trainer_amd/libkp1.so).

The product path has no CPU fallback: importing this module never touches ``oracle/`` and
``load()`` raises if the HIP library has not been built (run ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C rl_brain_trainer_amd``).
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

from . import config as kcfg

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libkp1.so"

KP1_OK = 0
REAL_F32, REAL_F64 = 0, 1
DONE_TERMINATED, DONE_TRUNCATED, DONE_SUCCESS, DONE_INVALID = 1, 2, 4, 8
FLAG_PRE_NEAR_HIT, FLAG_NEAR_HIT, FLAG_SUCCESS = 1, 2, 4


class EvalBuffers(C.Structure):
    """include/kp1.h kp1_eval_buffers: device pointers of the batched evaluator's per-episode accumulators"""
    _fields_ = [(n, C.c_void_p) for n in ("metrics", "counters", "flags", "state", "hand_metrics", "hand_step", "hand_success", "hand_state", "n_alive")]


class Kp1Error(RuntimeError):
    pass


_lib: C.CDLL | None = None


def declared_symbols() -> list[str]:
    """Every function include/kp1.h and include/kp1_ppo.h declare (used by the CPU symbol-export test)."""
    names: list[str] = []
    for header in sorted((kcfg.repo_root() / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
        names += re.findall(r"\b(kp1_[a-z0-9_]+)\s*\(", text)
    seen, out = set(), []
    for n in names:
        if n not in seen and not n.endswith("_t"):
            seen.add(n)
            out.append(n)
    return out


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise Kp1Error(
            f"{LIB_PATH} is missing: the HIP extension has not been built. There is no CPU fallback; "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    L = C.CDLL(str(LIB_PATH))
    vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
    L.kp1_last_error.restype = C.c_char_p
    L.kp1_abi_version.restype = C.c_int
    L.kp1_config_size.restype = u64
    L.kp1_config_default.argtypes = [C.POINTER(kcfg.Kp1Config)]
    L.kp1_create.argtypes = [C.POINTER(kcfg.Kp1Config), i32, i32, i32, u64, u64, vp, C.POINTER(vp)]
    L.kp1_destroy.argtypes = [vp]
    L.kp1_set_stream.argtypes = [vp, vp]
    L.kp1_num_envs.argtypes = [vp]
    L.kp1_set_stage.argtypes = [vp, i32]
    L.kp1_get_stage.argtypes = [vp, C.POINTER(i32)]
    L.kp1_set_mode.argtypes = [vp, i32]
    L.kp1_update_config.argtypes = [vp, C.POINTER(kcfg.Kp1Config)]
    L.kp1_set_handoff_states.argtypes = [vp, vp, i32]
    L.kp1_seed.argtypes = [vp, u64, u64]
    L.kp1_reset.argtypes = [vp, vp, C.POINTER(kcfg.ResetOpts), vp]
    L.kp1_step.argtypes = [vp, vp, vp, vp, vp, vp, i32]
    L.kp1_observe.argtypes = [vp, vp]
    L.kp1_get_info.argtypes = [vp, C.POINTER(kcfg.InfoView)]
    L.kp1_eval_accumulate.argtypes = [vp, C.POINTER(EvalBuffers), vp, vp, vp, i32, vp, i32, vp]
    L.kp1_get_reward_components.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
    L.kp1_enable_reward_components.argtypes = [vp, i32]
    L.kp1_component_name.argtypes = [i32, i32]
    L.kp1_component_name.restype = C.c_char_p
    L.kp1_num_components.argtypes = [i32]
    L.kp1_get_state.argtypes = [vp] + [vp] * 5
    L.kp1_set_state.argtypes = [vp] + [vp] * 5 + [i32]
    L.kp1_rng_get.argtypes = [vp, vp]
    L.kp1_rng_set.argtypes = [vp, vp]
    L.kp1_state_snapshot.argtypes = [vp]
    L.kp1_state_restore.argtypes = [vp]
    L.kp1_fk_pose6.argtypes = [i32, i32, vp, vp, i64, vp]
    L.kp1_pose_error.argtypes = [i32, i32, vp, vp, vp, vp, vp, i64, vp]
    L.kp1_joint_utils.argtypes = [i32, i32, C.POINTER(kcfg.Kp1Config), vp, vp, vp, vp, vp, vp, i64, vp]
    L.kp1_rng_seed_state.argtypes = [u64, C.POINTER(kcfg.RngState)]
    f32 = C.c_float
    L.kp1_gae_scan.argtypes = [i32, vp, vp, vp, vp, f32, f32, vp, vp, i32, i32, vp]
    L.kp1_bootstrap_truncated.argtypes = [i32, vp, vp, vp, f32, i64, vp]
    L.kp1_adv_minibatch_sums.argtypes = [i32, vp, vp, i64, i64, vp, vp]
    L.kp1_adv_minibatch_stats.argtypes = [i32, vp, i64, vp, vp]
    L.kp1_random_permutation.argtypes = [i32, i64, vp, vp, vp]
    L.kp1_curriculum_create.argtypes = [i32, C.c_double, i32, i32, i32, i32, C.POINTER(vp)]
    L.kp1_curriculum_destroy.argtypes = [i32, vp]
    L.kp1_curriculum_observe.argtypes = [i32, vp, vp, i32, i32, vp]
    L.kp1_curriculum_observe_chunk.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    L.kp1_curriculum_read.argtypes = [i32, vp, vp, vp]
    L.kp1_bind_stage_ptr.argtypes = [vp, vp]
    L.kp1_set_obs_stride.argtypes = [vp, i32]
    if L.kp1_config_size() != C.sizeof(kcfg.Kp1Config):
        raise Kp1Error(f"kp1_config layout mismatch: library {L.kp1_config_size()} vs binding {C.sizeof(kcfg.Kp1Config)}")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != KP1_OK:
        msg = load().kp1_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(msg)
        raise Kp1Error(f"kp1 error {rc}: {msg}")


def component_names(mode: int) -> list[str]:
    L = load()
    return [L.kp1_component_name(mode, i).decode() for i in range(L.kp1_num_components(mode))]
