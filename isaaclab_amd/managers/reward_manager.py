"""``RewardManager`` (interface of ``isaaclab/managers/reward_manager.py``) on the fused MDP kernel: the whole term
list is one launch per group of terms that read the same scene entities (usually one)."""

from __future__ import annotations

import torch

from ..envs import fused_mdp as FM
from ..envs.mdp import _fused as F
from ._base import iter_terms, resolve_term
from .manager_term_cfg import RewardTermCfg


class RewardManager:
    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._term_names, self._term_cfgs, self._specs = [], [], []
        for name, term in iter_terms(cfg, RewardTermCfg):
            if not isinstance(term.weight, (float, int)):  # reward_manager.py:233-237
                raise TypeError(f"Weight for the term '{name}' is not of type float or int. Received: '{type(term.weight)}'.")
            self._term_names.append(name)
            self._term_cfgs.append(term)
            self._specs.append(resolve_term(env, name, term))
        n, dev = env.num_envs, env.device
        self._reward_buf = torch.zeros(n, dtype=torch.float, device=dev)
        self._episode_sums = {name: torch.zeros(n, dtype=torch.float, device=dev) for name in self._term_names}
        self._step_reward = torch.zeros(n, len(self._term_names), dtype=torch.float, device=dev)
        self._plans = None

    # -- properties
    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    def _build(self):
        self._plans = []
        for bindings, idx in F.group_specs(self._specs):
            specs = [self._specs[i] for i in idx]
            weights = [float(self._term_cfgs[i].weight) for i in idx]
            mdp = F.make_plan(self._env, "reward", specs, weights=weights, step_dt=self._env.step_dt, bindings=bindings)
            self._plans.append((bindings, idx, specs, mdp))

    # -- operations
    def reset(self, env_ids=None) -> dict[str, torch.Tensor]:
        """reward_manager.py:100-126."""
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for key in self._episode_sums:
            avg = torch.mean(self._episode_sums[key][env_ids])
            extras["Episode_Reward/" + key] = avg / self._env.max_episode_length_s
            self._episode_sums[key][env_ids] = 0.0
        return extras

    def compute(self, dt: float) -> torch.Tensor:
        """reward_manager.py:128-157."""
        if self._plans is None:
            self._build()
        self._reward_buf.zero_()
        for bindings, idx, specs, mdp in self._plans:
            mdp.desc.step_dt = dt
            for k, i in enumerate(idx):  # the plan accumulates into this manager's per-term sums
                mdp.episode_sums[k].copy_(self._episode_sums[self._term_names[i]])
            F.run_plan(mdp, self._env, "reward", specs, bindings)
            self._reward_buf += mdp.reward_buf
            for k, i in enumerate(idx):
                if self._term_cfgs[i].weight == 0.0:
                    continue  # :145-146
                self._episode_sums[self._term_names[i]].copy_(mdp.episode_sums[k])
                self._step_reward[:, i] = mdp.step_reward[:, k]
        return self._reward_buf

    def set_term_cfg(self, term_name: str, cfg: RewardTermCfg):
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        i = self._term_names.index(term_name)
        self._term_cfgs[i] = cfg
        self._specs[i] = resolve_term(self._env, term_name, cfg)
        self._plans = None  # weights / parameters live in the plan

    def get_term_cfg(self, term_name: str) -> RewardTermCfg:
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        return self._term_cfgs[self._term_names.index(term_name)]
