"""Oracle for the manager arithmetic (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/managers/reward_manager.py:100-157``, ``termination_manager.py:127-174`` and
``observation_manager.py:260-335``: what the managers do with the term values, independent of which terms run.
"""

from __future__ import annotations

import torch

from . import mathx as M


def reward_compute(term_values: list, weights: list, dt: float, episode_sums: torch.Tensor, step_reward: torch.Tensor):
    """reward_manager.py:128-157.

    ``term_values[i]`` is the raw term output (N,) or ``None`` for a zero-weight term (skipped, :145-146).
    ``episode_sums`` (T, N) and ``step_reward`` (N, T) are updated in place; returns ``reward_buf`` (N,).
    """
    n = step_reward.shape[0]
    reward_buf = torch.zeros(n)
    for i, (val, w) in enumerate(zip(term_values, weights)):
        if w == 0.0:
            continue
        value = val * w * dt  # tensor * float * float: each scalar is rounded to fp32 before the multiply
        reward_buf = reward_buf + value
        episode_sums[i] += value
        step_reward[:, i] = value / dt
    return reward_buf


def reward_reset(episode_sums: torch.Tensor, reset_mask, max_episode_length_s: float):
    """reward_manager.py:100-126: per-term mean of the episodic sums over the reset envs / episode length [s]."""
    out = []
    for i in range(episode_sums.shape[0]):
        if reset_mask.any():
            out.append(torch.mean(episode_sums[i][reset_mask]) / max_episode_length_s)
        else:
            out.append(torch.tensor(float("nan")))
        episode_sums[i][reset_mask] = 0.0
    return torch.stack(out) if out else torch.zeros(0)


def termination_compute(term_values: list, time_out_flags: list):
    """termination_manager.py:151-174: returns (dones, terminated, truncated, term_dones (T,N))."""
    n = term_values[0].shape[0]
    truncated = torch.zeros(n, dtype=torch.bool)
    terminated = torch.zeros(n, dtype=torch.bool)
    for v, is_to in zip(term_values, time_out_flags):
        if is_to:
            truncated |= v
        else:
            terminated |= v
    return truncated | terminated, terminated, truncated, torch.stack(term_values)


def termination_reset_counts(term_dones, reset_mask):
    """termination_manager.py:142-144: count_nonzero of each term over the reset envs."""
    return torch.stack([torch.count_nonzero(t[reset_mask]) for t in term_dones])


def observation_concat(terms: list):
    """observation_manager.py:331-335."""
    return torch.cat(terms, dim=-1)
