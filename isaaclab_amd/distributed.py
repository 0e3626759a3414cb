"""Multi-GPU layout of the env step: contiguous env-id shards, one process per GPU, no data-path collective.

Every environment is independent (state, terms, actuators and rays never cross envs), so rank r of W simply owns envs
``[r * n, (r + 1) * n)``.  The single exchange is the episode-statistics vector the managers log when envs reset
(``reward_manager.py:115-121``, ``termination_manager.py:142-144``, ``command_manager.py:136-142``): the fused MDP
kernel accumulates per-GPU partial sums into ``log_sums`` and ONE all-reduce(sum) of that packed vector (RCCL on GPUs,
gloo in CPU tests) makes every rank hold the statistics of the whole env set.  The reference itself reduces nothing
across ranks (each rank logs its own shard); the reduced values equal what a single process computes on the
concatenated env set -- counts exactly, sums up to fp32 re-association.
"""

from __future__ import annotations

from typing import Sequence

import torch


def env_shard(total_envs: int, rank: int, world_size: int) -> range:
    """Contiguous env-id range owned by ``rank`` (``app_launcher.py:515-521`` maps one process to one GPU)."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} for world size {world_size}")
    if total_envs % world_size != 0:
        raise ValueError(f"{total_envs} envs do not split evenly over {world_size} ranks")
    per = total_envs // world_size
    return range(rank * per, (rank + 1) * per)


def all_reduce_log_sums(log_sums: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the packed statistics vector over all ranks, in place, on the tensor's own device / current stream.

    A no-op when ``torch.distributed`` is not initialised (single GPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(log_sums, op=dist.ReduceOp.SUM, group=group)
    return log_sums


class OverlappedLogReducer:
    """The same all-reduce, taken off the step's critical path.

    A 17-float all-reduce is pure latency (tens of microseconds over xGMI), a tenth of an env step; nothing in the step
    needs its result, so it runs beside the next steps.  Two statistics vectors alternate: step k accumulates into
    ``buffer(k)``, its all-reduce is started asynchronously right after the step, and the totals are read when the
    buffer comes round again, two steps later (``begin_step`` waits for that all-reduce first: a stream wait on RCCL,
    a host wait on gloo).  With no process group (single GPU) the totals are the local sums, with the same delay.

        red = OverlappedLogReducer(n, device)
        for k in ...:
            buf, totals = red.begin_step()   # totals: global statistics of step k - 2 (None before), or skip them
            mdp.bind(log_sums=buf); env.step(...)
            red.end_step()
        red.finish()
    """

    def __init__(self, numel: int, device, group=None):
        import torch.distributed as dist

        self._dist = dist if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 else None
        self._group = group
        self._bufs = [torch.zeros(numel, device=device), torch.zeros(numel, device=device)]
        self._works = [None, None]
        self._filled = [False, False]
        self._k = 0

    def buffer(self, i: int) -> torch.Tensor:
        """Statistics vector ``i`` (0 or 1); step k accumulates into ``buffer(k & 1)`` (launches recorded in a HIP graph
        are recorded once per vector)."""
        return self._bufs[i]

    @property
    def index(self) -> int:
        """Which vector the step that begins next accumulates into."""
        return self._k & 1

    def begin_step(self, want_totals: bool = False, zero: bool = True):
        """``zero=False``: the caller clears the vector itself (a step recorded in a HIP graph does it inside the graph)."""
        i = self._k & 1
        if self._works[i] is not None:
            self._works[i].wait()
            self._works[i] = None
        totals = self._bufs[i].clone() if want_totals and self._filled[i] else None
        if zero:
            self._bufs[i].zero_()
        return self._bufs[i], totals

    def end_step(self):
        i = self._k & 1
        if self._dist is not None:
            self._works[i] = self._dist.all_reduce(self._bufs[i], op=self._dist.ReduceOp.SUM, group=self._group, async_op=True)
        self._filled[i] = True
        self._k += 1

    def finish(self) -> list[torch.Tensor]:
        """Wait for the all-reduces still in flight; returns the totals not handed out yet, oldest first."""
        out = []
        for i in ((self._k & 1), (self._k + 1) & 1):
            if self._works[i] is not None:
                self._works[i].wait()
                self._works[i] = None
            if self._filled[i]:
                out.append(self._bufs[i].clone())
                self._filled[i] = False
        return out


def episode_log(log_sums: torch.Tensor, reward_names: Sequence[str], termination_names: Sequence[str],
                max_episode_length_s: float, metric_names: Sequence[str] = ("error_vel_xy", "error_vel_yaw"),
                command_name: str = "base_velocity") -> dict:
    """Unpack the (reduced) vector into the ``extras["log"]`` entries the managers' ``reset`` methods return.

    Layout: ``[R reward episode sums | T termination counts | 2 command metric sums | number of reset envs]``.
    ``Episode_Reward/<term>`` = mean over the reset envs / max_episode_length_s (``reward_manager.py:115-121``);
    ``Episode_Termination/<term>`` = count (``termination_manager.py:142-144``); ``Metrics/<command>/<metric>`` = mean
    (``command_manager.py:136-142``).  Reads the vector on the host (one small device-to-host copy)."""
    v = log_sums.detach().cpu().double()
    nr, nt = len(reward_names), len(termination_names)
    if v.numel() != nr + nt + 3:
        raise ValueError(f"log vector has {v.numel()} entries, expected {nr + nt + 3}")
    count = float(v[-1])
    out = {}
    for i, name in enumerate(reward_names):
        out[f"Episode_Reward/{name}"] = float(v[i]) / count / max_episode_length_s if count > 0 else float("nan")
    for i, name in enumerate(termination_names):
        out[f"Episode_Termination/{name}"] = int(round(float(v[nr + i])))
    for i, name in enumerate(metric_names):
        out[f"Metrics/{command_name}/{name}"] = float(v[nr + nt + i]) / count if count > 0 else float("nan")
    out["num_resets"] = int(round(count))
    return out
