"""``TerminationManager`` (interface of ``isaaclab/managers/termination_manager.py``) on the fused MDP kernel."""

from __future__ import annotations

import torch

from ..envs.mdp import _fused as F
from ._base import iter_terms, resolve_term
from .manager_term_cfg import TerminationTermCfg


class TerminationManager:
    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._term_names, self._term_cfgs, self._specs = [], [], []
        for name, term in iter_terms(cfg, TerminationTermCfg):
            self._term_names.append(name)
            self._term_cfgs.append(term)
            self._specs.append(resolve_term(env, name, term))
        n, dev = env.num_envs, env.device
        self._term_dones = {name: torch.zeros(n, device=dev, dtype=torch.bool) for name in self._term_names}
        self._truncated_buf = torch.zeros(n, device=dev, dtype=torch.bool)
        self._terminated_buf = torch.zeros_like(self._truncated_buf)
        self._all = torch.zeros(max(len(self._term_names), 1), n, device=dev, dtype=torch.bool)
        self._plans = None

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    @property
    def dones(self) -> torch.Tensor:
        return self._truncated_buf | self._terminated_buf

    @property
    def time_outs(self) -> torch.Tensor:
        return self._truncated_buf

    @property
    def terminated(self) -> torch.Tensor:
        return self._terminated_buf

    def term_dones_tensor(self) -> torch.Tensor:
        """(num_terms, N) bool: the rows of ``_term_dones`` in term order."""
        return self._all

    def reset(self, env_ids=None) -> dict[str, torch.Tensor]:
        """termination_manager.py:127-149."""
        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for key in self._term_dones:
            extras["Episode_Termination/" + key] = torch.count_nonzero(self._term_dones[key][env_ids]).item()
        return extras

    def compute(self) -> torch.Tensor:
        """termination_manager.py:151-174."""
        if self._plans is None:
            self._plans = []
            for bindings, idx in F.group_specs(self._specs):
                specs = [self._specs[i] for i in idx]
                flags = [self._term_cfgs[i].time_out for i in idx]
                mdp = F.make_plan(self._env, "termination", specs, time_out_flags=flags, bindings=bindings)
                self._plans.append((bindings, idx, specs, mdp))
        self._truncated_buf[:] = False
        self._terminated_buf[:] = False
        for bindings, idx, specs, mdp in self._plans:
            F.run_plan(mdp, self._env, "termination", specs, bindings)
            self._truncated_buf |= mdp.time_outs
            self._terminated_buf |= mdp.terminated
            for k, i in enumerate(idx):
                self._term_dones[self._term_names[i]][:] = mdp.term_dones[k]
                self._all[i] = mdp.term_dones[k]
        return self._truncated_buf | self._terminated_buf

    def get_term(self, name: str) -> torch.Tensor:
        return self._term_dones[name]

    def get_term_cfg(self, term_name: str) -> TerminationTermCfg:
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        return self._term_cfgs[self._term_names.index(term_name)]

    def set_term_cfg(self, term_name: str, cfg: TerminationTermCfg):
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        i = self._term_names.index(term_name)
        self._term_cfgs[i] = cfg
        self._specs[i] = resolve_term(self._env, term_name, cfg)
        self._plans = None
