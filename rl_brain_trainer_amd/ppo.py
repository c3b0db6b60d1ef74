"""PPO rollout-and-update loop on one MI355X (or one rank of a data-parallel job).

Restates stable_baselines3.PPO("MultiInputPolicy") as the reference drives it
(kinematic_phase1/train_workspace_expansion.py:175-232, training/train_dock_policy.py:86-102):

* policy: CombinedExtractor (56 floats, alphabetical key order) -> separate tanh MLPs ``pi`` / ``vf``
  (SB3 default 2x64; BASELINE config 2 asks 2x256) -> ``action_net`` (7) / ``value_net`` (1),
  state-independent ``log_std`` (init 0), orthogonal init (gain sqrt2 / 0.01 / 1);
* rollout: sample a ~ N(mean, exp(log_std)), store the unclipped action, clip to [-1, 1] for the env,
  bootstrap truncated episodes with gamma * V(terminal_observation), GAE(lambda);
* update: n_epochs x shuffled minibatches, per-minibatch advantage normalisation, clipped surrogate +
  0.5 * MSE value loss - ent_coef * entropy, global grad-norm clip 0.5, Adam(eps=1e-5).

SB3 itself is not in the reference tree nor in this image: these semantics are "parity unpinned"
 and are tested against plain-PyTorch references of each op.

Everything stays on the device: the env kernel writes straight into the rollout buffer, the curriculum
tracker is a device kernel, and the only host synchronisation per iteration is the logging read-back.
Under torch.distributed (backend "nccl" = RCCL) env ranges are sharded by rank and gradients are summed
with one flat all-reduce per optimiser step.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch

from . import config as kcfg
from . import native
from .curriculum import PointCurriculum
from .vec_env import ArmKinematicVecEnv

OBS_DIM = kcfg.OBS_DIM
ACT_DIM = kcfg.NJ


@dataclass
class PPOConfig:
    """algorithms.ppo block of the YAML (ppo_default.yaml) + SB3 defaults for what the YAML leaves out."""

    learning_rate: float = 3e-4
    n_steps: int = 2048
    batch_size: int = 256
    n_epochs: int = 10
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    adam_eps: float = 1e-5
    seed: int = 0
    hidden: int = 64            # SB3 default net_arch = dict(pi=[64, 64], vf=[64, 64])
    normalize_advantage: bool = True
    total_timesteps: int = 100_000

    @classmethod
    def from_algo_kwargs(cls, kwargs: dict[str, Any], **overrides: Any) -> "PPOConfig":
        known = {f for f in cls.__dataclass_fields__}
        data = {k: v for k, v in kwargs.items() if k in known}
        unknown = set(kwargs) - known
        if unknown:
            raise TypeError(f"PPO.__init__() got unexpected keyword arguments {sorted(unknown)}")
        data.update(overrides)
        return cls(**data)


def param_spec(hidden: int, obs_dim: int = OBS_DIM) -> list[tuple[str, tuple[int, ...]]]:
    """SB3 MultiInputActorCriticPolicy.state_dict() keys/shapes for net_arch pi=vf=[hidden, hidden]; ``obs_dim`` = width of the
    flattened Dict observation (56, or 80 with the route keys of route/route_observation.py:14-61)."""
    H = hidden
    return [
        ("log_std", (ACT_DIM,)),
        ("mlp_extractor.policy_net.0.weight", (H, obs_dim)), ("mlp_extractor.policy_net.0.bias", (H,)),
        ("mlp_extractor.policy_net.2.weight", (H, H)), ("mlp_extractor.policy_net.2.bias", (H,)),
        ("mlp_extractor.value_net.0.weight", (H, obs_dim)), ("mlp_extractor.value_net.0.bias", (H,)),
        ("mlp_extractor.value_net.2.weight", (H, H)), ("mlp_extractor.value_net.2.bias", (H,)),
        ("action_net.weight", (ACT_DIM, H)), ("action_net.bias", (ACT_DIM,)),
        ("value_net.weight", (1, H)), ("value_net.bias", (1,)),
    ]


class ActorCritic:
    """Flat fp32 parameter buffer with SB3-named views."""

    def __init__(self, hidden: int, device: torch.device, seed: int = 0, obs_dim: int = OBS_DIM) -> None:
        self.hidden = hidden
        self.obs_dim = int(obs_dim)
        self.device = device
        self.spec = param_spec(hidden, self.obs_dim)
        self.numel = sum(math.prod(s) for _, s in self.spec)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views: dict[str, torch.Tensor] = {}
        off = 0
        for name, shape in self.spec:
            n = math.prod(shape)
            self.views[name] = self.flat[off:off + n].view(shape)
            off += n
        self._init(seed)

    def _init(self, seed: int) -> None:
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        gains = {"mlp_extractor": math.sqrt(2.0), "action_net": 0.01, "value_net": 1.0}
        for name, shape in self.spec:
            if name.endswith("weight"):
                w = torch.empty(shape, dtype=torch.float32)
                torch.nn.init.orthogonal_(w, gain=gains[name.split(".")[0]], generator=g)
                self.views[name].copy_(w)
            else:
                self.views[name].zero_()  # biases and log_std start at 0

    def state_dict(self) -> dict[str, torch.Tensor]:
        return {k: v.detach().clone().cpu() for k, v in self.views.items()}

    def load_state_dict(self, sd: dict[str, torch.Tensor]) -> None:
        for name, shape in self.spec:
            t = sd[name]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != {tuple(shape)}")
            self.views[name].copy_(t.to(self.device, torch.float32))


def mlp_forward(P: dict[str, torch.Tensor], obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """mean[B,7], value[B] of the SB3 MlpExtractor + heads (plain torch; also the reference for the HIP kernels)."""
    hp = torch.tanh(torch.addmm(P["mlp_extractor.policy_net.0.bias"], obs, P["mlp_extractor.policy_net.0.weight"].t()))
    hp = torch.tanh(torch.addmm(P["mlp_extractor.policy_net.2.bias"], hp, P["mlp_extractor.policy_net.2.weight"].t()))
    mean = torch.addmm(P["action_net.bias"], hp, P["action_net.weight"].t())
    hv = torch.tanh(torch.addmm(P["mlp_extractor.value_net.0.bias"], obs, P["mlp_extractor.value_net.0.weight"].t()))
    hv = torch.tanh(torch.addmm(P["mlp_extractor.value_net.2.bias"], hv, P["mlp_extractor.value_net.2.weight"].t()))
    value = torch.addmm(P["value_net.bias"], hv, P["value_net.weight"].t()).squeeze(-1)
    return mean, value


LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def gaussian_log_prob(actions: torch.Tensor, mean: torch.Tensor, log_std: torch.Tensor) -> torch.Tensor:
    z = (actions - mean) * torch.exp(-log_std)
    return (-0.5 * z * z - log_std - LOG_SQRT_2PI).sum(-1)


class GraphSegments:
    """A launch sequence captured as a CHAIN of hipGraphs with eager calls between them (the data-parallel collectives when they are not
    captured themselves).  During capture `cut(fn)` ends the running segment, records `fn` and opens the next one; `replay()` walks the
    chain in capture order on the current stream.  All segments share one memory pool (tensors allocated inside one segment and read by a
    later one, or by a recorded call, keep their addresses; replaying strictly in capture order is what makes sharing the pool safe).
    `on_begin` runs inside every freshly opened capture (the env re-reads the capture stream there)."""

    def __init__(self, device: torch.device, on_begin=None) -> None:
        self.device = device
        self.plan: list[Any] = []
        self.n_graphs = 0
        self._pool = torch.cuda.graph_pool_handle()
        self._on_begin = on_begin
        self._cur = None

    def begin(self) -> None:
        g = torch.cuda.CUDAGraph()
        ctx = torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local")
        ctx.__enter__()
        self._cur = (g, ctx)
        if self._on_begin is not None:
            self._on_begin()

    def _end(self) -> None:
        g, ctx = self._cur
        self._cur = None
        ctx.__exit__(None, None, None)
        self.plan.append(g)
        self.n_graphs += 1

    def cut(self, fn) -> None:
        self._end()
        self.plan.append(fn)
        self.begin()

    def finish(self) -> None:
        self._end()

    def abort(self) -> None:
        if self._cur is not None:
            g, ctx = self._cur
            self._cur = None
            try:
                ctx.__exit__(None, None, None)
            except Exception:  # noqa: BLE001 -- already unwinding from the error that made the capture fail
                pass

    def replay(self) -> None:
        for item in self.plan:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()


class Dist:
    """torch.distributed glue (backend nccl = RCCL on ROCm, gloo on CPU tests).  world_size 1 = no-ops.

    How the collectives meet the hipGraphs of the rollout and of the update epoch (``graph_mode``):

    * ``"segmented"`` (the default for world_size > 1): the compute between two collectives is captured as one hipGraph segment
      (GraphSegments) and the collectives are launched eagerly between the replays, on the same stream.  Needs nothing from the backend
      (works with gloo), keeps every kernel sequence out of the host's launch path, and costs two graph launches + one collective enqueue
      per optimiser step (~40 us of host time against ~85 us of kernels: the host stays ahead).
    * ``"captured"`` (opt-in, ``KP1_DIST_GRAPHS=1``, nccl only): the RCCL collectives are captured INSIDE the graphs.  No multi-GPU run of
      this path has been recorded yet (the builder has one GPU), so it is not the default; when selected it is probed once per process
      (capture + replay + check of a 4-float all-reduce and a 16-byte all-gather, the verdict all-reduced so every rank takes the same
      path) and a watchdog turns a hang of the probe, of the first real rollout replay or of the first real epoch replay into a process
      exit with a message (never a re-exec); PPO additionally checks after those first replays that all ranks hold identical parameters.
    * single process: plain captured graphs."""

    def __init__(self) -> None:
        import torch.distributed as dist

        self.dist = dist
        # KP1_DIST_FORCE_SINGLE=1 (test hook): a one-rank process group takes the data-parallel code path too, so that the RCCL calls, their
        # stream ordering against the graph segments and the byte all-gather run on a box with one GPU (tests/test_distributed_gpu.py)
        force = os.environ.get("KP1_DIST_FORCE_SINGLE", "0") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.world_size = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self.backend = str(dist.get_backend()) if self.enabled else ""
        self._graphs_ok: bool | None = None
        self._segments: GraphSegments | None = None     # set while a segmented capture records: collectives become cut points
        self.rccl = None                                # rccl.RcclComm once use_stream_collectives() has agreed on it
        self.collectives = "torch.distributed" if self.enabled else "none"

    def use_stream_collectives(self, device: torch.device) -> bool:
        """nccl backend: put the training collectives on the LAUNCH stream (rccl.RcclComm: same RCCL, no hop to torch's NCCL stream and back per
        call).  Collective: call on every rank.  Every rank self-tests its communicator and the verdicts are combined through torch.distributed,
        so all ranks take the same path; any failure (library, bootstrap, a wrong sum) leaves the torch.distributed path in place.
        KP1_RCCL_DIRECT=0 switches it off."""
        if not self.enabled or self.backend != "nccl" or self.rccl is not None or os.environ.get("KP1_RCCL_DIRECT", "1") == "0":
            return self.rccl is not None
        import sys
        import threading

        # The bootstrap runs in a helper thread with a deadline: a rank whose ncclCommInitRank does not return (a peer that never arrives) reports
        # "not available" like any other failure and every rank stays on torch.distributed -- a slow path instead of a dead run.
        box: dict[str, Any] = {}

        def boot() -> None:
            try:
                from . import rccl

                torch.cuda.set_device(device)          # the current device is per thread
                box["comm"] = rccl.RcclComm(self.dist, device)
                box["ok"] = box["comm"].self_test()
            except Exception as exc:  # noqa: BLE001 -- whatever goes wrong here, the torch.distributed path is complete by itself
                box["exc"] = exc

        worker = threading.Thread(target=boot, name="kp1-rccl-bootstrap", daemon=True)
        worker.start()
        worker.join(float(os.environ.get("KP1_RCCL_BOOT_SECONDS", "120")))
        comm, ok = box.get("comm"), bool(box.get("ok")) and not worker.is_alive()
        if not ok:
            why = "no answer within the deadline" if worker.is_alive() else (f"{type(box['exc']).__name__}: {box['exc']}" if "exc" in box else "self-test mismatch")
            print(f"[kp1] rank {self.rank}: RCCL on the launch stream not available ({why}); collectives stay on torch.distributed", flush=True, file=sys.stderr)
        verdict = torch.tensor([1.0 if ok else 0.0], device=device)
        self.dist.all_reduce(verdict, op=self.dist.ReduceOp.MIN)
        if verdict.item() > 0.5:
            self.rccl = comm
            self.collectives = "rccl on the launch stream"
        elif comm is not None and not worker.is_alive():
            comm.close()
        return self.rccl is not None

    def close(self) -> None:
        """destroy the launch-stream RCCL communicator (collective in effect: call on every rank, after the last training collective has completed)"""
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None
            self.collectives = "torch.distributed" if self.enabled else "none"

    def _collective(self, fn) -> None:
        if self._segments is not None:
            self._segments.cut(fn)
        else:
            fn()

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            if self.rccl is not None and t.is_cuda and t.is_contiguous():
                self._collective(lambda: self.rccl.all_reduce_sum(t))
            else:
                self._collective(lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM))
        return t

    def all_gather_bytes(self, t: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return t
        out = torch.empty(self.world_size * t.numel(), dtype=t.dtype, device=t.device)  # rank-major = global env order
        self.dist.all_gather_into_tensor(out, t.contiguous().view(-1))
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """out[world, ...] <- every rank's t (rank-major), into a caller-owned buffer (graph replays need a fixed address)."""
        if self.enabled:
            dst, src = out.view(-1), t.contiguous().view(-1)
            if self.rccl is not None and src.is_cuda:
                self._collective(lambda: self.rccl.all_gather(dst, src))
            else:
                self._collective(lambda: self.dist.all_gather_into_tensor(dst, src))
        else:
            out.view(-1).copy_(t.reshape(-1))
        return out

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.enabled:
            self.dist.broadcast(t, src)
        return t

    def graphs_ok(self, device: torch.device) -> bool:
        """True when collectives of this process group can live inside a captured hipGraph (single process: trivially)."""
        if not self.enabled:
            return True
        if self._graphs_ok is None:
            self._graphs_ok = self._probe_graph_collectives(device)
        return self._graphs_ok

    def graph_mode(self, device: torch.device) -> str:
        """"captured" (collectives inside the graphs; always so for a single process) or "segmented" (graph segments, eager collectives)"""
        return "captured" if self.graphs_ok(device) else "segmented"

    def watchdog(self, seconds: float, what: str):
        """arm a timer that ends the process (exit code 3, a message, no re-exec) if `what` has not finished; returns the timer to cancel"""
        import os
        import sys
        import threading

        def _hung() -> None:
            print(f"[kp1] {what} did not complete within {seconds:.0f} s: re-run with KP1_DIST_GRAPHS=0 KP1_RCCL_DIRECT=0 (segmented graphs, torch.distributed collectives)",
                  file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(seconds, _hung)
        dog.daemon = True
        dog.start()
        return dog

    def ranks_agree(self, flat: torch.Tensor) -> bool:
        """every rank holds bit-identical `flat` (digest min == max over ranks); a collective itself, call it on all ranks"""
        if not self.enabled:
            return True
        d = flat.double()
        digest = torch.stack([d.sum(), d.abs().sum(), d[::97].sum()])
        lo, hi = digest.clone(), digest.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi)) and bool(torch.isfinite(digest).all())

    def _probe_graph_collectives(self, device: torch.device) -> bool:
        import os
        import sys
        import threading

        ok = self.backend == "nccl" and os.environ.get("KP1_DIST_GRAPHS", "0") == "1"   # opt-in: see the class docstring
        verdict = torch.tensor([1.0 if ok else 0.0], device=device)
        if ok:
            buf = torch.full((4,), float(self.rank + 1), device=device)
            expect = float(self.world_size * (self.world_size + 1) // 2)
            # the two collectives the captured graphs contain: the flat gradient all-reduce (update epoch) and the done-byte all-gather (rollout)
            mine = torch.full((16,), self.rank + 1, dtype=torch.uint8, device=device)
            gathered = torch.zeros((self.world_size, 16), dtype=torch.uint8, device=device)
            self.dist.all_reduce(buf.clone())          # communicator fully set up before any capture
            self.dist.all_gather_into_tensor(gathered.clone().view(-1), mine)
            torch.cuda.synchronize(device)

            def _hung() -> None:
                print("[kp1] a captured RCCL all-reduce did not complete within 120 s: re-run with KP1_DIST_GRAPHS=0", file=sys.stderr, flush=True)
                os._exit(3)

            dog = threading.Timer(120.0, _hung)
            dog.daemon = True
            dog.start()
            g, good = None, True
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.dist.all_reduce(buf)
                    self.dist.all_gather_into_tensor(gathered.view(-1), mine)
            except Exception as exc:  # noqa: BLE001 -- capture of a collective is refused in many ways depending on the build
                print(f"[kp1] rank {self.rank}: RCCL inside hipGraph capture not available ({type(exc).__name__}: {exc}); eager collectives", file=sys.stderr)
                good = False
            # a rank whose capture failed never replays: agree (eagerly) that EVERY rank captured before any rank replays
            captured = torch.tensor([1.0 if good else 0.0], device=device)
            self.dist.all_reduce(captured, op=self.dist.ReduceOp.MIN)
            good = bool(captured.item() > 0.5)
            if good:
                buf.fill_(float(self.rank + 1))
                gathered.zero_()
                g.replay()
                torch.cuda.synchronize(device)
                want = torch.arange(1, self.world_size + 1, dtype=torch.uint8, device=device)[:, None].expand(-1, 16)
                good = bool(torch.all(buf == expect).item()) and bool(torch.equal(gathered, want))
            dog.cancel()
            verdict.fill_(1.0 if good else 0.0)
        self.dist.all_reduce(verdict, op=self.dist.ReduceOp.MIN)   # every rank takes the same path
        return bool(verdict.item() > 0.5)


class PPO:
    def __init__(self, env: ArmKinematicVecEnv, cfg: PPOConfig, *, curriculum: PointCurriculum | None = None,
                 dist: Dist | None = None, backend: str = "hip", use_graphs: bool = True) -> None:
        self.env = env
        self.cfg = cfg
        self.device = env.device
        self.dist = dist or Dist()
        self.backend = backend
        self.L = native.load()
        self.n_envs = env.n_envs
        self.obs_dim = int(getattr(env, "obs_dim", OBS_DIM))   # 80 for the route envs with include_route_keys
        self.policy = ActorCritic(cfg.hidden, self.device, seed=cfg.seed, obs_dim=self.obs_dim)
        self.dist.broadcast(self.policy.flat)
        self.adam_m = torch.zeros_like(self.policy.flat)
        self.adam_v = torch.zeros_like(self.policy.flat)
        self.adam_t = 0
        self.n_train_calls = 0         # train() calls so far: SB3's _n_updates = n_train_calls * n_epochs
        self.actor_extra_steps = 0    # optimiser steps only the actor tensors took (teacher-anchor side updates)
        self._epoch_warm = False       # one eager epoch has run (kernel attributes set, code objects loaded) before the epoch graph is captured
        self.curriculum = curriculum
        if curriculum is not None:
            curriculum.attach(env)
        T, N = cfg.n_steps, self.n_envs
        dev = self.device
        # hip backend: observation rows are written with pitch 64 / 128 (zero padded) so the MFMA GEMMs read them directly
        self.obs_w = (64 if self.obs_dim <= 64 else 128) if backend == "hip" else self.obs_dim
        if backend == "hip":
            env.set_obs_stride(self.obs_w)
        self.obs_buf = torch.zeros((T + 1, N, self.obs_w), dtype=torch.float32, device=dev)
        self.term_obs_buf = torch.zeros((T, N, self.obs_w), dtype=torch.float32, device=dev)
        self.act_buf = torch.zeros((T, N, ACT_DIM), dtype=torch.float32, device=dev)
        self.clip_act = torch.zeros((N, ACT_DIM), dtype=torch.float32, device=dev)
        self.logp_buf = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.val_buf = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.rew_buf = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.done_buf = torch.zeros((T, N), dtype=torch.uint8, device=dev)
        self.adv_buf = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.ret_buf = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.gen = torch.Generator(device=dev).manual_seed(int(cfg.seed) + 7919 * self.dist.rank)
        self._perm_rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(cfg.seed), self.dist.rank, 0x6B7031])))   # shuffle keys
        self.num_timesteps = 0
        self._needs_reset = True
        self._last_stats_dev: tuple[torch.Tensor, int] | None = None    # (loss sums of the last train() on the device, number of updates)
        self._last_stats_host: dict[str, float] = {}
        if env.dtype != torch.float32:
            raise ValueError("PPO drives the production f32 env")
        self._mlp = None
        if backend == "hip":
            from . import mlp as _mlp

            local_bs = max(cfg.batch_size // self.dist.world_size, 1)
            self._mlp = _mlp.MlpKernels(cfg.hidden, self.device, max_batch=max(N, local_bs, 8192), obs_dim=self.obs_dim)
            self._mlp.pack(self.policy.flat)
            if os.environ.get("KP1_BF16X3_WGRAD", "0") == "1" and cfg.hidden == 256:
                # round-3 EXPERIMENT (off by default, never used by bench.py's value): weight-gradient GEMMs on bf16 x 3 operands
                self._mlp.set_bf16x3_wgrad(True)
            self.grad = torch.zeros_like(self.policy.flat)
            self.stats_dev = torch.zeros(4, dtype=torch.float32, device=dev)
            self.noise = torch.zeros((N, ACT_DIM), dtype=torch.float32, device=dev)
        elif backend != "torch":
            raise ValueError("backend must be 'hip' or 'torch'")
        # hipGraph replay of the rollout (T x 3 launches) and of one update epoch.  Data parallel: graph segments with eager collectives
        # between them, or (opt-in) the RCCL collectives captured inside the graphs: Dist.graph_mode.
        self.use_graphs = bool(use_graphs and backend == "hip")
        if backend == "hip":
            self.dist.use_stream_collectives(self.device)     # nccl: the training collectives go on the launch stream (rccl.py); else a no-op
        self.graph_mode = self.dist.graph_mode(self.device) if self.use_graphs else "none"
        self._first_replay_checked = {"rollout": not (self.dist.enabled and self.graph_mode == "captured"),
                                      "epoch": not (self.dist.enabled and self.graph_mode == "captured")}
        self._rollout_graph = None
        self._epoch_graph = None
        self._epoch_policy = os.environ.get("KP1_DP_EPOCH", "auto")     # segmented mode only: "auto" (measured), "graph", "eager"
        self._epoch_auto = None
        self.epoch_form = "one graph" if self.graph_mode == "captured" else ("eager launches" if self.graph_mode == "none" or self._epoch_policy == "eager" else "graph segments")
        self._rollout_graph_key = None     # (env.launch_args_version, curriculum attached?) the rollout graph was captured with
        self._epoch_graph_key = None       # hyper-parameters baked into the epoch graph's kernel arguments
        self._kernels_warm = False         # the rollout kernels have run once (code objects loaded, attributes set)
        # data parallel: done bytes are exchanged once per `done_chunk` env steps (one all-gather of chunk x N bytes instead of one
        # latency-bound collective per 45 us env step); the device tracker replays the chunk in the reference's order
        self.done_chunk = 1
        if self.dist.enabled:
            import os as _os

            self.done_chunk = math.gcd(T, max(int(_os.environ.get("KP1_DONE_EXCHANGE_STEPS", "16")), 1))
            self._done_gather = torch.zeros((self.dist.world_size, self.done_chunk, N), dtype=torch.uint8, device=dev)
        # one launch per rollout step for policy forward + env step where the env is the plain fp32 vectorised env and the tile kernels run
        # (KP1_FUSED_ROLLOUT=0: the two-launch form, kept as the A/B and test reference)
        self._fused_env_step = bool(backend == "hip" and type(env) is ArmKinematicVecEnv and env.dtype == torch.float32 and cfg.hidden == 256
                                    and self.obs_dim <= 64 and not getattr(env, "_reward_components_on", False)
                                    and os.environ.get("KP1_FUSED_ROLLOUT", "1") != "0")
        # optional host hook after every env step (done bits of that step, device tensor): what SB3 callbacks' _on_step sees.
        # Setting it makes the rollout eager (a host hook cannot live inside a hipGraph replay).
        self.step_callback = None
        if backend == "hip":
            # exploration noise of a whole rollout is drawn in ONE call, graphs or not: the eager and the replayed rollout then consume the
            # generator identically and stay bit-identical (tests/test_distributed_gpu.py compares them)
            self.noise_all = torch.zeros((T, N, ACT_DIM), dtype=torch.float32, device=dev)
        if self.use_graphs:
            self.perm = torch.zeros(T * N, dtype=torch.int64, device=dev)

    # Minibatch shuffles.  At the benchmarked size (524288 samples) a sort-based torch.randperm is 0.16 ms of full-chip kernels, eight times per
    # iteration (2.5 %); drawing them on a side stream under the rollout only moved that time (the sort kernels fill the chip and the
    # rollout's launches queue behind them: rollout +1.3 ms for update -1.7 ms).  From PERM_CIPHER_MIN samples on, the permutation is a keyed
    # bijection evaluated per element (kp1_random_permutation: one elementwise launch); below it torch.randperm stays -- it is cheap there.
    PERM_CIPHER_MIN = (1 << 17) if os.environ.get("KP1_PERM_CIPHER", "1") != "0" else (1 << 62)   # KP1_PERM_CIPHER=0: developer A/B switch

    def _draw_perm(self, total: int, out: torch.Tensor | None = None) -> torch.Tensor:
        """indices of one epoch's minibatches: a random permutation of [0, total) on the device (SB3: np.random.permutation in RolloutBuffer.get)"""
        if self._mlp is None or total < self.PERM_CIPHER_MIN:
            if out is None:
                return torch.randperm(total, device=self.device, generator=self.gen)
            return torch.randperm(total, device=self.device, generator=self.gen, out=out)
        if out is None:
            out = torch.empty(total, dtype=torch.int64, device=self.device)
        keys = self._perm_rng.integers(0, 1 << 32, size=8, dtype=np.uint64).astype(np.uint32)
        native.check(self.L.kp1_random_permutation(self.device.index or 0, total, keys.ctypes.data_as(C.c_void_p), C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def invalidate_graphs(self) -> None:
        """Drop the captured hipGraphs: a capture freezes host-side scalars into kernel arguments (env stage / mode / config pointers; learning
        rate, clip range, ent / vf coefficients, minibatch geometry), so after changing any of them the next rollout / epoch is re-captured.
        Called automatically when ``env.launch_args_version`` or the PPOConfig fields the epoch graph bakes in have changed."""
        self._rollout_graph = None
        self._epoch_graph = None

    def _epoch_key(self) -> tuple:
        c = self.cfg
        return (c.learning_rate, c.clip_range, c.ent_coef, c.vf_coef, c.max_grad_norm, c.adam_eps, c.batch_size, c.n_steps, c.normalize_advantage,
                self.dist.world_size)

    def _curriculum_observe(self, t: int) -> None:
        """after env step t: feed the done bytes to the device tracker (single process: every step; data parallel: once per chunk)"""
        if self.curriculum is None:
            return
        if not self.dist.enabled:
            self.curriculum.observe(self.done_buf[t], self.n_envs)
            return
        c = self.done_chunk
        if (t + 1) % c == 0:
            self.dist.all_gather_into(self._done_gather, self.done_buf[t + 1 - c:t + 1])
            self.curriculum.observe_chunk(self._done_gather, self.n_envs, c, self.dist.world_size)

    # ------------------------------------------------------------------ PPO.load
    def load_checkpoint(self, path: str, *, restore_optimizer: bool = True, restore_timesteps: bool = False,
                        restore_hyperparameters: bool = False) -> dict[str, Any]:
        """``PPO.load(path, env=...)`` for an SB3 zip (the reference's resume paths, train_workspace_expansion.py:187-197,
        train_route_curriculum.py:129-139): policy weights, and the Adam state of ``policy.optimizer.pth`` when present -- first / second
        moments per tensor and torch's per-tensor step counts (a common count, plus the extra steps of the actor tensors when a
        teacher-anchor run wrote the file).  ``restore_hyperparameters``: a loaded SB3 model keeps the algorithm constants it was saved
        with (gamma, gae_lambda, ent_coef, vf_coef, max_grad_norm, n_epochs, normalize_advantage, and clip_range when the zip stores it
        as a plain value); the trainers then re-apply only the YAML's learning rate.  The rollout geometry (n_envs, n_steps, minibatch)
        stays the engine's: it is sized for the GPU, not for the 12-16 CPU envs a reference checkpoint was collected with.  Must be called
        before the first update (the constants are baked into the captured graphs).  Returns what was restored."""
        from . import checkpoint

        self.policy.load_state_dict(checkpoint.load_policy_state_dict(path))
        restored: dict[str, Any] = {"policy": True, "optimizer": False}
        opt = checkpoint.load_optimizer_state_dict(path) if restore_optimizer else None
        if opt and opt.get("state"):
            steps, off = [], 0
            for i, (name, shape) in enumerate(self.policy.spec):
                n = math.prod(shape)
                st = opt["state"].get(i)
                if st is None or tuple(st["exp_avg"].shape) != tuple(shape):
                    raise ValueError(f"optimizer state of parameter {i} ({name}) does not match the policy")
                self.adam_m[off:off + n].copy_(st["exp_avg"].to(self.device, torch.float32).reshape(-1))
                self.adam_v[off:off + n].copy_(st["exp_avg_sq"].to(self.device, torch.float32).reshape(-1))
                steps.append((name, int(float(st["step"]))))
                off += n
            actor = [s_ for n_, s_ in steps if n_.startswith(("mlp_extractor.policy_net", "action_net"))]
            rest = [s_ for n_, s_ in steps if not n_.startswith(("mlp_extractor.policy_net", "action_net"))]
            if len(set(actor)) != 1 or len(set(rest)) != 1 or actor[0] < rest[0]:
                raise ValueError(f"unsupported per-tensor Adam step pattern {steps}")
            self.adam_t, self.actor_extra_steps = rest[0], actor[0] - rest[0]
            restored.update({"optimizer": True, "adam_steps": self.adam_t, "actor_extra_steps": self.actor_extra_steps})
        if self._mlp is not None:
            self._mlp.pack(self.policy.flat)
            self._mlp.set_step_count(self.adam_t)
            self._mlp.set_actor_extra_steps(self.actor_extra_steps)
        if restore_timesteps or restore_hyperparameters:
            data = checkpoint.load_data(path)
            if restore_timesteps:
                self.num_timesteps = int(data.get("num_timesteps", 0))
                restored["num_timesteps"] = self.num_timesteps
                saved_epochs = data.get("n_epochs")
                if isinstance(data.get("_n_updates"), int) and isinstance(saved_epochs, int) and saved_epochs > 0:
                    self.n_train_calls = int(data["_n_updates"]) // saved_epochs   # SB3: _n_updates += n_epochs per train() call
            if restore_hyperparameters:
                if self._epoch_graph is not None or self.adam_t != restored.get("adam_steps", self.adam_t):
                    raise RuntimeError("restore_hyperparameters must happen before the first update")
                taken = {}
                for key, cast in (("gamma", float), ("gae_lambda", float), ("ent_coef", float), ("vf_coef", float), ("max_grad_norm", float),
                                  ("n_epochs", int), ("normalize_advantage", bool)):
                    v = data.get(key)
                    if isinstance(v, (int, float, bool)) and not (isinstance(v, bool) and cast is not bool):
                        setattr(self.cfg, key, cast(v))
                        taken[key] = cast(v)
                clip = data.get("clip_range")
                if isinstance(clip, dict) and isinstance(clip.get("value"), (int, float)):   # this engine's writer; SB3 pickles the schedule
                    self.cfg.clip_range = float(clip["value"])
                    taken["clip_range"] = self.cfg.clip_range
                restored["hyperparameters"] = taken
        return restored

    # ------------------------------------------------------------------ policy evaluation
    def _forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """mean[n,7], value[n] for obs [n, 56 or 64]"""
        if self._mlp is not None:
            return self._mlp.mean_value(obs.contiguous())
        return mlp_forward(self.policy.views, obs[:, :self.obs_dim])

    def predict(self, obs: torch.Tensor, deterministic: bool = True) -> torch.Tensor:
        """model.predict(obs, deterministic): mean (or a sample) clipped to the action space (eval_three_stage.py:25-27)."""
        with torch.no_grad():
            mean, _ = self._forward(obs)
            if not deterministic:
                mean = mean + torch.exp(self.policy.views["log_std"]) * torch.randn(mean.shape, device=mean.device, generator=self.gen)
            return mean.clamp(-1.0, 1.0)

    def predict_unclipped(self, obs: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self._forward(obs)[0]

    def _bootstrap_truncated(self, dense: bool = False) -> None:
        """SB3 collect_rollouts time-limit bootstrap: rewards[t, i] += gamma * V(terminal_observation) where step t of env i was
        truncated and not terminated.  ``dense`` = critic over all T * N terminal observations + kp1_bootstrap_truncated (the
        torch backend's path, kept as the reference for the compacted one)."""
        cfg, env = self.cfg, self.env
        T, N = cfg.n_steps, self.n_envs
        stream = torch.cuda.current_stream(self.device).cuda_stream
        dev = self.device.index or 0
        if self._mlp is not None and not dense:
            # Only truncated steps need the critic.  An env truncates at most T // max_episode_steps + 1 times per rollout, so the
            # truncated positions fit a fixed-size index list (torch.nonzero_static: no device->host size query) and the value net
            # runs on N * (that bound) terminal observations instead of all T * N.
            max_steps = max(int(env.config.c.termination.max_episode_steps), 1)
            cap = min(N * (T // max_steps + 1), T * N)
            trunc = (self.done_buf.view(-1) & 3) == 2
            idx = torch.nonzero_static(trunc, size=cap, fill_value=0).view(-1)
            valid = torch.arange(cap, device=self.device) < trunc.sum()
            sel = self.term_obs_buf.view(T * N, self.obs_w).index_select(0, idx)
            tv = torch.empty(cap, dtype=torch.float32, device=self.device)
            for s0 in range(0, cap, self._mlp.max_batch):
                e0 = min(s0 + self._mlp.max_batch, cap)
                self._mlp.forward(sel[s0:e0], value=tv[s0:e0])
            self.rew_buf.view(-1).index_add_(0, idx, torch.where(valid, cfg.gamma * tv, torch.zeros_like(tv)))
        else:
            _, tv = self._forward(self.term_obs_buf.view(T * N, self.obs_w))
            tv = tv.contiguous()
            native.check(self.L.kp1_bootstrap_truncated(dev, C.c_void_p(self.rew_buf.data_ptr()), C.c_void_p(tv.data_ptr()),
                                                        C.c_void_p(self.done_buf.data_ptr()), cfg.gamma, T * N, C.c_void_p(stream)))

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def collect_rollouts(self) -> None:
        cfg, env = self.cfg, self.env
        T, N = cfg.n_steps, self.n_envs
        if self._needs_reset:
            self.obs_buf[0].copy_(env.reset())
            self._needs_reset = False
        else:
            self.obs_buf[0].copy_(self.obs_buf[T])
        log_std = self.policy.views["log_std"]
        std = torch.exp(log_std)
        world = self.dist.world_size
        graph_rollout = self.use_graphs and self.step_callback is None
        if self._mlp is not None:
            self.noise_all.normal_(generator=self.gen)
        if graph_rollout:
            key = (getattr(env, "launch_args_version", 0), self.curriculum is not None, cfg.gamma, cfg.gae_lambda)   # what a capture freezes
            if self._rollout_graph is None or self._rollout_graph_key != key:
                self._capture_rollout()
                self._rollout_graph_key = key
            self._replay_checked("rollout", self._rollout_graph)
        for t in range(0 if not graph_rollout else T, T):
            if self._mlp is not None:
                self._rollout_step_hip(t)      # the launch sequence the captured rollout replays
                if self.step_callback is not None:
                    self.step_callback(self.done_buf[t])
                continue
            else:
                mean, value = self._forward(self.obs_buf[t])
                noise = torch.randn((N, ACT_DIM), dtype=torch.float32, device=self.device, generator=self.gen)
                action = torch.addcmul(mean, std, noise)
                self.act_buf[t].copy_(action)
                self.logp_buf[t].copy_((-0.5 * noise * noise - log_std - LOG_SQRT_2PI).sum(-1))
                self.val_buf[t].copy_(value)
                torch.clamp(action, -1.0, 1.0, out=self.clip_act)
            env.step_into(self.clip_act, self.obs_buf[t + 1], self.rew_buf[t], self.done_buf[t], self.term_obs_buf[t], True)
            self._curriculum_observe(t)
            if self.step_callback is not None:
                self.step_callback(self.done_buf[t])
        if not graph_rollout:
            self._kernels_warm = True
        self.num_timesteps += T * N * world
        if not graph_rollout:
            self._post_rollout()     # (the captured rollout ends with it)

    def _post_rollout(self) -> None:
        """what SB3's collect_rollouts does after the last env step: time-limit bootstrap, value of the last observation, GAE.  Static shapes
        and no host synchronisation, so the captured rollout graph carries it as its tail (a dozen small launches whose host latency
        was 0.5 ms per iteration)."""
        cfg = self.cfg
        T, N = cfg.n_steps, self.n_envs
        self._bootstrap_truncated()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        dev = self.device.index or 0
        if self._mlp is not None:
            last_v = torch.empty(N, dtype=torch.float32, device=self.device)
            self._mlp.forward(self.obs_buf[T], value=last_v)   # value net only
        else:
            _, last_v = self._forward(self.obs_buf[T])
            last_v = last_v.contiguous()
        native.check(self.L.kp1_gae_scan(dev, C.c_void_p(self.rew_buf.data_ptr()), C.c_void_p(self.val_buf.data_ptr()),
                                         C.c_void_p(self.done_buf.data_ptr()), C.c_void_p(last_v.data_ptr()), cfg.gamma, cfg.gae_lambda,
                                         C.c_void_p(self.adv_buf.data_ptr()), C.c_void_p(self.ret_buf.data_ptr()), T, N, C.c_void_p(stream)))

    def _rollout_step_hip(self, t: int) -> None:
        # (running the tracker on a side stream under the next policy forward was measured: the fork / join inside the graph
        # cost more than the 4 us kernel it hid -- 7.4 ms vs 5.9 ms per 128-step rollout)
        self._policy_env_step(t)
        self._curriculum_observe(t)

    def _policy_env_step(self, t: int) -> None:
        if self._fused_env_step:
            # policy forward + sampling + env step (auto-reset included) in ONE launch: the tile's policy workgroup steps its 32 envs itself
            self._mlp.forward_env_step(self.env, self.obs_buf[t], noise=self.noise_all[t], value=self.val_buf[t], action=self.act_buf[t],
                                       log_prob=self.logp_buf[t], next_obs=self.obs_buf[t + 1], reward=self.rew_buf[t], done=self.done_buf[t],
                                       terminal_obs=self.term_obs_buf[t])
        else:
            self._mlp.forward(self.obs_buf[t], noise=self.noise_all[t], value=self.val_buf[t], action=self.act_buf[t],
                              clipped=self.clip_act, log_prob=self.logp_buf[t])
            self.env.step_into(self.clip_act, self.obs_buf[t + 1], self.rew_buf[t], self.done_buf[t], self.term_obs_buf[t], True)

    def _capture_rollout(self) -> None:
        """Record the T-step rollout (policy forward, env step, curriculum tracker; data parallel: the done-byte all-gather of every chunk)
        once; every later rollout is one replay.  The kernels must have run once before a capture (code objects loaded, attributes set).  At
        the very start that is a warm-up step between a device snapshot of the env state and its restore; a RE-capture in the middle of
        training (an env setter or a hyper-parameter changed, a step callback was removed) finds them warm from the rollouts already done.
        Either way the running episodes and random streams are untouched."""
        T = self.cfg.n_steps
        if not self._kernels_warm:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            can_snapshot = hasattr(self.env, "snapshot")
            with torch.cuda.stream(side):
                self.env.use_current_stream()
                if can_snapshot:
                    self.env.snapshot()      # the warm-up step below must not move the episodes or the random streams
                self._policy_env_step(0)
                if self.curriculum is not None:
                    self.curriculum.observe(self.done_buf[0].zero_(), 0)   # also loads the module the chunk form of the tracker lives in
                self._post_rollout()                                        # the graph's tail, once eagerly (its buffers are rewritten by the real rollout)
                self.env.use_current_stream()
                if can_snapshot:
                    self.env.restore()       # a captured rollout therefore continues exactly like an eager one (tests/test_distributed_gpu.py)
                else:
                    # wrappers without a device snapshot (the route env keeps state of its own): start the episodes again instead, at the
                    # price of one extra reset() draw per env stream
                    self.obs_buf[0].copy_(self.env.reset())
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._kernels_warm = True
        torch.cuda.synchronize(self.device)

        def body() -> None:
            for t in range(T):
                self._rollout_step_hip(t)
            self._post_rollout()

        self._rollout_graph = self._capture(body, on_begin=self.env.use_current_stream)
        self.env.use_current_stream()

    def _capture_epoch(self, obs, act, old_logp, adv, ret, total: int, local_bs: int) -> None:
        """one update epoch = advantage statistics of all minibatches (+ ONE all-reduce of them), then per minibatch: tile kernel, weight
        gradients, finalize, [flat gradient all-reduce, sum of squares], Adam -- captured with the collectives inside"""
        torch.cuda.synchronize(self.device)

        def body() -> None:
            mb_stats = self._epoch_adv_stats(adv, self.perm, total, local_bs)
            for i, start in enumerate(range(0, total, local_bs)):
                self._hip_minibatch_step(obs, self.perm[start:start + local_bs], act, old_logp, adv, ret, device_step=True,
                                         adv_stats=None if mb_stats is None else mb_stats[i])

        self._epoch_graph = self._capture(body)

    def _capture(self, body, on_begin=None):
        """record `body` once: ONE hipGraph (single process, or collectives captured inside), or a GraphSegments chain cut at every collective"""
        if self.dist.enabled and self.graph_mode == "segmented":
            seg = GraphSegments(self.device, on_begin=on_begin)
            self.dist._segments = seg
            try:
                seg.begin()
                body()
                seg.finish()
            except BaseException:
                seg.abort()
                raise
            finally:
                self.dist._segments = None
            return seg
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local" if self.dist.enabled else "global"):
            if on_begin is not None:
                on_begin()
            body()
        return g

    def _replay_checked(self, what: str, graph) -> None:
        """replay; the FIRST replay of a graph that has RCCL collectives captured inside runs under a watchdog and is followed by a check
        that every rank still holds the same parameters (that path has no recorded multi-GPU run: a hang or a divergence must end the job
        with a message instead of a silent wrong result)"""
        if self._first_replay_checked[what]:
            graph.replay()
            return
        dog = self.dist.watchdog(300.0, f"the first replay of the {what} graph with captured RCCL collectives")
        graph.replay()
        torch.cuda.synchronize(self.device)
        ok = self.dist.ranks_agree(self.policy.flat)
        dog.cancel()
        self._first_replay_checked[what] = True
        if not ok:
            raise RuntimeError(f"ranks diverged after the first replay of the {what} graph with captured RCCL collectives; re-run with KP1_DIST_GRAPHS=0")

    # ------------------------------------------------------------------ update
    def train(self) -> None:
        cfg = self.cfg
        T, N = cfg.n_steps, self.n_envs
        total = T * N
        obs = self.obs_buf[:T].view(total, self.obs_w)
        act = self.act_buf.view(total, ACT_DIM)
        old_logp = self.logp_buf.view(total)
        adv = self.adv_buf.view(total)
        ret = self.ret_buf.view(total)
        world = self.dist.world_size
        local_bs = max(cfg.batch_size // world, 1)
        stats = torch.zeros(4, device=self.device)
        n_updates = 0
        self.n_train_calls += 1
        if self._mlp is not None:
            self.stats_dev.zero_()
        for _epoch in range(cfg.n_epochs):
            if self.use_graphs:
                self._draw_perm(total, out=self.perm)     # the epoch graph reads the fixed address self.perm
                if self._epoch_graph is not None and self._epoch_graph_key != self._epoch_key():
                    self._epoch_graph = None     # a hyper-parameter baked into the captured kernel arguments changed: capture again
                if self._epoch_graph is None:
                    # one eager epoch first (warm-up + it is a real epoch), then capture for the following ones
                    if not self._epoch_warm:
                        self._epoch_warm = True
                        mb_stats = self._epoch_adv_stats(adv, self.perm, total, local_bs)
                        for i, start in enumerate(range(0, total, local_bs)):
                            self._hip_minibatch_step(obs, self.perm[start:start + local_bs], act, old_logp, adv, ret, device_step=True,
                                                     adv_stats=None if mb_stats is None else mb_stats[i])
                            n_updates += 1
                            self.adam_t += 1
                        continue
                    self._capture_epoch(obs, act, old_logp, adv, ret, total, local_bs)
                    self._epoch_graph_key = self._epoch_key()
                    self._epoch_auto = {"phase": 0} if (self.graph_mode == "segmented" and self._epoch_policy == "auto") else None
                    self._replay_checked("epoch", self._epoch_graph)      # the first replay carries one-time costs: never the timed one
                    n_updates += (total + local_bs - 1) // local_bs
                    self.adam_t += (total + local_bs - 1) // local_bs
                    continue
                if self.graph_mode == "segmented" and self._segmented_epoch_eager():
                    # data parallel, segments measured slower than eager launches on this box (same bits either way)
                    mb_stats = self._epoch_adv_stats(adv, self.perm, total, local_bs)
                    for i, start in enumerate(range(0, total, local_bs)):
                        self._hip_minibatch_step(obs, self.perm[start:start + local_bs], act, old_logp, adv, ret, device_step=True,
                                                 adv_stats=None if mb_stats is None else mb_stats[i])
                else:
                    self._replay_checked("epoch", self._epoch_graph)
                self._segmented_epoch_timed()
                n_updates += (total + local_bs - 1) // local_bs
                self.adam_t += (total + local_bs - 1) // local_bs
                continue
            perm = self._draw_perm(total)
            mb_stats = self._epoch_adv_stats(adv, perm, total, local_bs) if self._mlp is not None else None
            if self._mlp is not None:
                # the eager loop steps Adam from the device-resident step count too, exactly as the captured epoch does (bias corrections
                # computed by the same device arithmetic): eager and replayed training stay bit-identical.  The count is set from the host
                # mirror first, so loss_grad calls made outside train() cannot have moved it.
                self._mlp.set_step_count(self.adam_t)
            for i, start in enumerate(range(0, total, local_bs)):
                idx = perm[start:start + local_bs]
                if self._mlp is not None:
                    self._hip_minibatch_step(obs, idx, act, old_logp, adv, ret, device_step=True, adv_stats=None if mb_stats is None else mb_stats[i])
                    self.adam_t += 1
                else:
                    stats += self._minibatch_step(obs[idx], act[idx], old_logp[idx], adv[idx], ret[idx])
                n_updates += 1
        if self._mlp is not None:
            stats = self.stats_dev.clone()
        # read back lazily (last_stats): a .tolist() here would make every iteration wait for its own update before the host can enqueue
        # the next rollout
        self._last_stats_dev = (stats, n_updates)

    # Data parallel, graph_mode "segmented": an update epoch is 64 graph segments of four kernels with a collective between them.  A hipGraph
    # launch boundary costs more on the GPU timeline than a kernel boundary of an eager launch (one-rank RCCL group on one MI355X, collectives
    # stubbed out: 48.9 ms per update as segments, 45.6 ms as eager launches, 43.7 ms as ONE graph without a process group), while a slow or
    # busy host favours the segments.  Both forms leave the same bits (tests/test_distributed_gpu.py), so the choice is measured: after the
    # capture, one replayed epoch and one eager epoch are timed with events, the times are max-reduced over the ranks, and the faster form
    # runs from then on (KP1_DP_EPOCH=graph|eager pins it).
    def _segmented_epoch_eager(self) -> bool:
        if self._epoch_policy != "auto":
            return self._epoch_policy == "eager"
        a = self._epoch_auto
        if a is None:
            return False
        if a["phase"] in (0, 1):          # phase 0: time a replay; phase 1: time an eager epoch
            a["ev"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            a["ev"][0].record()
            return a["phase"] == 1
        return bool(a["eager"])

    def _segmented_epoch_timed(self) -> None:
        a = self._epoch_auto
        if a is None or self._epoch_policy != "auto" or a["phase"] > 1:
            return
        a["ev"][1].record()
        a["t_graph" if a["phase"] == 0 else "t_eager"] = a["ev"]
        a["phase"] += 1
        if a["phase"] == 2:
            torch.cuda.synchronize(self.device)
            t = torch.tensor([a["t_graph"][0].elapsed_time(a["t_graph"][1]), a["t_eager"][0].elapsed_time(a["t_eager"][1])], dtype=torch.float64, device=self.device)
            if self.dist.enabled:
                self.dist.dist.all_reduce(t, op=self.dist.dist.ReduceOp.MAX)      # every rank takes the same form
            a["ms"] = [float(x) for x in t.tolist()]
            a["eager"] = a["ms"][1] < a["ms"][0]
            self.epoch_form = "eager launches" if a["eager"] else "graph segments"

    @property
    def last_stats(self) -> dict[str, float]:
        """mean policy loss / value loss / entropy / approx_kl over the minibatches of the last train() call, and their number (SB3 logger keys
        train/policy_gradient_loss, train/value_loss, train/entropy_loss, train/approx_kl); reading it synchronises with that update"""
        if self._last_stats_dev is not None:
            stats, n_updates = self._last_stats_dev
            self._last_stats_host = dict(zip(("policy_loss", "value_loss", "entropy", "approx_kl"), (stats / max(n_updates, 1)).tolist()))
            self._last_stats_host["n_updates"] = n_updates
            self._last_stats_dev = None
        return self._last_stats_host

    def _epoch_adv_stats(self, adv, perm, total: int, local_bs: int):
        """(mean, 1/(std + 1e-8)) of the advantages of every minibatch of this epoch, f32 [n_minibatches, 2] on the device.

        SB3 normalises per minibatch (ppo.py train()).  One kernel sums all minibatches, and with data parallelism ONE
        all-reduce per epoch makes the statistics those of the global minibatch (all ranks' shards together), so the
        update does not depend on how the rollout is sharded -- instead of a 3-float collective before every optimiser step."""
        if not self.cfg.normalize_advantage:
            return None
        n_mb = (total + local_bs - 1) // local_bs
        sums = torch.empty((n_mb, 3), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        native.check(self.L.kp1_adv_minibatch_sums(self.device.index or 0, C.c_void_p(adv.data_ptr()), C.c_void_p(perm.data_ptr()), total, local_bs,
                                                   C.c_void_p(sums.data_ptr()), C.c_void_p(stream)))
        self.dist.all_reduce_sum(sums)
        out = torch.empty((n_mb, 2), dtype=torch.float32, device=self.device)
        native.check(self.L.kp1_adv_minibatch_stats(self.device.index or 0, C.c_void_p(sums.data_ptr()), n_mb, C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(stream)))
        return out

    def _hip_minibatch_step(self, obs, idx, act, old_logp, adv, ret, device_step: bool = False, adv_stats=None) -> None:
        """one optimiser step, all on the device: gathered fwd + loss + bwd (MFMA), flat grad all-reduce, clip + Adam + repack"""
        cfg = self.cfg
        n = int(idx.numel())
        world = self.dist.world_size
        if adv_stats is None and cfg.normalize_advantage and self.dist.enabled:
            mean, inv_std = global_advantage_stats(adv[idx], self.dist)
            adv_stats = torch.stack([mean, inv_std]).float()
        self._mlp.loss_grad(obs, idx, n, act, old_logp, adv, ret, clip_range=cfg.clip_range, ent_coef=cfg.ent_coef / world, vf_coef=cfg.vf_coef,
                            inv_count=1.0 / (n * world), grad_out=self.grad, stats_out=self.stats_dev, adv_stats=adv_stats,
                            normalize=cfg.normalize_advantage)
        self.dist.all_reduce_sum(self.grad)
        if not device_step:
            self.adam_t += 1
        # device_step: the Adam step count lives on the device (incremented by loss_grad's finalize kernel) so that the
        # launch sequence can be replayed from a hipGraph; the host mirror self.adam_t is advanced by the caller
        self._mlp.adam_step(self.policy.flat, self.grad, self.adam_m, self.adam_v, lr=cfg.learning_rate, eps=cfg.adam_eps,
                            max_grad_norm=cfg.max_grad_norm, step=0 if device_step else self.adam_t, fused_norm=not self.dist.enabled)

    def _normalize_adv(self, adv: torch.Tensor) -> torch.Tensor:
        if not self.cfg.normalize_advantage or adv.numel() * self.dist.world_size <= 1:
            return adv
        if not self.dist.enabled:
            return (adv - adv.mean()) / (adv.std() + 1e-8)
        # GPU-count invariant normalisation: global (sum, sum^2, count)
        s = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), device=adv.device)])
        self.dist.all_reduce_sum(s)
        mean = s[0] / s[2]
        var = (s[1] - s[2] * mean * mean) / (s[2] - 1.0)
        return (adv - mean) / (var.clamp_min(0).sqrt() + 1e-8)

    def _minibatch_step(self, obs, act, old_logp, adv, ret) -> torch.Tensor:
        cfg = self.cfg
        adv = self._normalize_adv(adv)
        grad, terms = self._torch_loss_and_grad(obs[:, :self.obs_dim], act, old_logp, adv, ret)
        self.dist.all_reduce_sum(grad)
        self._clip_and_adam(grad)
        return terms

    def _torch_loss_and_grad(self, obs, act, old_logp, adv, ret) -> tuple[torch.Tensor, torch.Tensor]:
        cfg = self.cfg
        flat = self.policy.flat.detach().requires_grad_(True)
        P, off = {}, 0
        for name, shape in self.policy.spec:
            n = math.prod(shape)
            P[name] = flat[off:off + n].view(shape)
            off += n
        mean, value = mlp_forward(P, obs)
        log_std = P["log_std"]
        logp = gaussian_log_prob(act, mean, log_std)
        ratio = torch.exp(logp - old_logp)
        denom = float(obs.shape[0] * self.dist.world_size)
        pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - cfg.clip_range, 1 + cfg.clip_range)).sum() / denom
        vl = ((ret - value) ** 2).sum() / denom
        ent = (0.5 + LOG_SQRT_2PI + log_std).sum()  # per-sample entropy is constant: mean == value
        loss = pl + cfg.vf_coef * vl - cfg.ent_coef * ent / self.dist.world_size
        (grad,) = torch.autograd.grad(loss, flat)
        with torch.no_grad():
            kl = ((ratio - 1) - (logp - old_logp)).mean()
            terms = torch.stack([pl.detach(), vl.detach(), ent.detach(), kl])
        return grad, terms

    def _clip_and_adam(self, grad: torch.Tensor) -> None:
        cfg = self.cfg
        with torch.no_grad():
            norm = torch.linalg.vector_norm(grad)
            scale = torch.clamp(cfg.max_grad_norm / (norm + 1e-6), max=1.0)  # clip_grad_norm_
            grad = grad * scale
            self.adam_t += 1
            b1, b2 = 0.9, 0.999
            self.adam_m.mul_(b1).add_(grad, alpha=1 - b1)
            self.adam_v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
            # torch.optim.Adam counts steps per tensor: the actor tensors are `actor_extra_steps` ahead after teacher-anchor side updates
            steps = torch.full_like(self.policy.flat, float(self.adam_t))
            if self.actor_extra_steps:
                off = 0
                for name, shape in self.policy.spec:
                    n = math.prod(shape)
                    if name.startswith(("mlp_extractor.policy_net", "action_net")):
                        steps[off:off + n] += float(self.actor_extra_steps)
                    off += n
            bc1 = 1 - b1 ** steps
            bc2 = 1 - b2 ** steps
            denom = (self.adam_v.sqrt() / bc2.sqrt()).add_(cfg.adam_eps)
            self.policy.flat.sub_(cfg.learning_rate * self.adam_m / (bc1 * denom))

    # ------------------------------------------------------------------ driver
    def learn(self, total_timesteps: int | None = None, log_every: int = 0) -> "PPO":
        total = int(total_timesteps if total_timesteps is not None else self.cfg.total_timesteps)
        start_steps = self.num_timesteps
        it = 0
        t0 = time.time()
        while self.num_timesteps - start_steps < total:
            self.collect_rollouts()
            self.train()
            it += 1
            if log_every and it % log_every == 0 and self.dist.rank == 0:
                dt = time.time() - t0
                stage = self.curriculum.read().stage_index if self.curriculum is not None else -1
                print(f"[ppo] it={it} steps={self.num_timesteps} fps={(self.num_timesteps - start_steps) / dt:,.0f} stage={stage} "
                      f"rew={self.rew_buf.mean().item():.4f} {self.last_stats}", flush=True)
        return self


class InferencePolicy:
    """``PPO.load(path)`` + ``model.predict(obs, deterministic=True)`` for evaluators and the demo loaders
    (eval_deterministic.py:66-79, eval_three_stage.py:25-27): mean action clipped to the action space."""

    def __init__(self, state_dict: dict[str, torch.Tensor], device: torch.device | int = 0, max_batch: int = 8192) -> None:
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        hidden, obs_dim = (int(v) for v in state_dict["mlp_extractor.policy_net.0.weight"].shape)
        self.obs_dim = obs_dim
        self.policy = ActorCritic(hidden, self.device, obs_dim=obs_dim)
        self.policy.load_state_dict(state_dict)
        from . import mlp as _mlp

        # 2x64 (SB3's default, what reference-trained archives hold), 2x128 and 2x256 all evaluate on the MFMA kernels; any other width is
        # refused by kp1_mlp_create_ex -- there is no torch fallback in the product path
        self._mlp = _mlp.MlpKernels(hidden, self.device, max_batch=max_batch, obs_dim=obs_dim)
        self._mlp.pack(self.policy.flat)

    @classmethod
    def load(cls, path: str, device: torch.device | int = 0, max_batch: int = 8192) -> "InferencePolicy":
        from . import checkpoint

        return cls(checkpoint.load_policy_state_dict(path), device=device, max_batch=max_batch)

    @torch.no_grad()
    def predict(self, obs: torch.Tensor, deterministic: bool = True) -> torch.Tensor:
        if not deterministic:
            raise NotImplementedError("evaluators use deterministic=True")
        mean, _ = self._mlp.mean_value(obs.contiguous())
        return mean.clamp(-1.0, 1.0)

    __call__ = predict


def ppo_loss_and_grad_torch(flat: torch.Tensor, spec, obs, act, old_logp, adv, ret, *, clip_range: float, ent_coef: float, vf_coef: float,
                            world_size: int = 1) -> torch.Tensor:
    """SB3's PPO loss on plain torch (any device), gradient w.r.t. the flat parameter vector, scaled so that summing the
    result over ``world_size`` equal shards gives the single-process gradient.  Used by the CPU/gloo data-parallel tests."""
    flat = flat.detach().clone().requires_grad_(True)
    P, off = {}, 0
    for name, shape in spec:
        n = math.prod(shape)
        P[name] = flat[off:off + n].view(shape)
        off += n
    mean, value = mlp_forward(P, obs)
    logp = gaussian_log_prob(act, mean, P["log_std"])
    ratio = torch.exp(logp - old_logp)
    denom = float(obs.shape[0] * world_size)
    pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)).sum() / denom
    vl = ((ret - value) ** 2).sum() / denom
    ent = (0.5 + LOG_SQRT_2PI + P["log_std"]).sum()
    loss = pl + vf_coef * vl - ent_coef * ent / world_size
    (grad,) = torch.autograd.grad(loss, flat)
    return grad


def global_advantage_stats(adv: torch.Tensor, dist: "Dist") -> tuple[torch.Tensor, torch.Tensor]:
    """(mean, 1/(std + 1e-8)) of the advantages of ALL ranks (unbiased std), via one all-reduce of (sum, sum^2, count):
    makes the normalisation, and hence the update, independent of how the minibatch is sharded over GPUs."""
    s = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), device=adv.device, dtype=adv.dtype)])
    dist.all_reduce_sum(s)
    mean = s[0] / s[2]
    var = ((s[1] - s[2] * mean * mean) / (s[2] - 1.0)).clamp_min(0)
    return mean, 1.0 / (var.sqrt() + 1e-8)


def smoke() -> dict[str, Any]:
    """tiny rollout + update on cuda:0 for __graft_entry__.smoke()"""
    cfg_dict = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml")
    env_cfg = kcfg.to_env_config(cfg_dict)
    env = ArmKinematicVecEnv(env_cfg, 256, seed=806)
    env.set_curriculum_stage(5)
    ppo = PPO(env, PPOConfig(n_steps=16, batch_size=1024, n_epochs=2, hidden=256, learning_rate=6e-6, gamma=0.995, clip_range=0.1, ent_coef=3e-4, seed=806))
    before = ppo.policy.flat.clone()
    ppo.collect_rollouts()
    ppo.train()
    torch.cuda.synchronize()
    delta = (ppo.policy.flat - before).abs().max().item()
    assert math.isfinite(delta) and delta > 0.0, "PPO update did not change the parameters"
    assert torch.isfinite(ppo.adv_buf).all()
    env.close()
    return {"ppo_param_delta": delta, "ppo_backend": ppo.backend}
