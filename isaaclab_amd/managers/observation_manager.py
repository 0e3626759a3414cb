"""``ObservationManager`` (interface of ``isaaclab/managers/observation_manager.py``) on the fused MDP kernel: a group
is one launch that writes the concatenated, noise-corrupted, clipped and scaled row directly."""

from __future__ import annotations

import torch

from ..envs.mdp import _fused as F
from ._base import iter_terms, resolve_term
from .manager_term_cfg import ObservationGroupCfg, ObservationTermCfg


class ObservationManager:
    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._group_obs_term_names, self._group_obs_term_cfgs, self._group_specs, self._group_cfg = {}, {}, {}, {}
        groups = cfg.items() if isinstance(cfg, dict) else vars(cfg).items()
        for gname, gcfg in groups:
            if gcfg is None or gname.startswith("_"):
                continue
            names, cfgs, specs = [], [], []
            for name, term in iter_terms(gcfg, ObservationTermCfg):
                if term.history_length:
                    raise NotImplementedError("observation history buffers are not part of the fused kernel")
                if term.noise is not None and getattr(term.noise, "operation", "add") != "add":
                    raise NotImplementedError("only additive uniform noise is compiled into the fused kernel")
                names.append(name), cfgs.append(term), specs.append(resolve_term(env, name, term))
            self._group_obs_term_names[gname], self._group_obs_term_cfgs[gname], self._group_specs[gname] = names, cfgs, specs
            self._group_cfg[gname] = gcfg
        self._plans = {}
        self._obs_buffer = None

    @property
    def active_terms(self) -> dict[str, list[str]]:
        return self._group_obs_term_names

    @property
    def group_obs_dim(self) -> dict[str, tuple[int, ...]]:
        return {g: (self._plan(g)[3].obs_dim,) for g in self._group_specs}

    def _flag(self, group, name, default):
        g = self._group_cfg[group]
        return g.get(name, default) if isinstance(g, dict) else getattr(g, name, default)

    def _plan(self, group):
        plan = self._plans.get(group)
        if plan is None:
            specs, cfgs = self._group_specs[group], self._group_obs_term_cfgs[group]
            if not self._flag(group, "concatenate_terms", True):
                raise NotImplementedError("groups with concatenate_terms=False are not compiled into the fused kernel")
            groups = F.group_specs(specs)
            if len(groups) != 1:
                raise NotImplementedError("an observation group must read one articulation / sensor set")
            corrupt = self._flag(group, "enable_corruption", False)
            posts = [dict(noise=(c.noise.n_min, c.noise.n_max) if (c.noise is not None and corrupt) else None, clip=c.clip,
                          scale=c.scale) for c in cfgs]
            mdp = F.make_plan(self._env, "observation", specs, posts=posts, bindings=groups[0][0])
            plan = self._plans[group] = (groups[0][0], None, specs, mdp)
        return plan

    def reset(self, env_ids=None) -> dict[str, float]:
        return {}

    def compute(self) -> dict[str, torch.Tensor]:
        """observation_manager.py:240-258."""
        self._obs_buffer = {g: self.compute_group(g) for g in self._group_specs}
        return self._obs_buffer

    def compute_group(self, group_name: str, noise_u: torch.Tensor | None = None) -> torch.Tensor:
        """observation_manager.py:260-335.  ``noise_u`` (no counterpart in the reference) injects the U[0,1) draws."""
        if group_name not in self._group_specs:
            raise ValueError(f"Unable to find the group '{group_name}' in the observation manager."
                             f" Available groups are: {list(self._group_specs)}")
        bindings, _, specs, mdp = self._plan(group_name)
        mdp.bind(noise_u=noise_u)
        F.run_plan(mdp, self._env, "observation", specs, bindings)
        return mdp.obs
