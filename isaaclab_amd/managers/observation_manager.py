"""``ObservationManager`` (interface of ``isaaclab/managers/observation_manager.py``) on the fused MDP kernel: a group
is one launch that writes the concatenated, noise-corrupted, clipped and scaled row directly.  A term with custom modifiers
(observation_manager.py:308-318), or with a noise other than additive uniform noise with scalar bounds, leaves the fused
launch raw; its modifiers, noise, clip and scale then run on its columns of the row, one launch each, in the reference's order.  A term with a history (:319-327) appends its columns to a ring and the
ordered history goes into the term's columns of the group's final row from the same launch."""

from __future__ import annotations

import inspect

import torch

from ..envs.mdp import _fused as F
from ..utils import modifiers
from ..utils.buffers import CircularBuffer
from ..utils.noise import ConstantNoiseCfg, constant_noise
from ..utils.modifiers import modifier as _mod
from ._base import iter_terms, resolve_term
from .manager_term_cfg import ObservationGroupCfg, ObservationTermCfg, UniformNoiseCfg


class _CompositeObservationPlan:
    """The row of an observation group whose terms read different articulations / sensor sets
    (observation_manager.py:260-335 concatenates whatever its terms return): one fused plan per set of bindings, each
    launched on its own buffers, its terms' columns then copied to their places in the group's row.  Offers what the
    manager uses of a single plan: ``obs``, ``obs_dim``, ``observations`` / ``_obs_dim`` (the terms in the group's order
    and their widths), ``bind(noise_u=...)``."""

    class _Term:
        def __init__(self, width):
            self.width = width

    def __init__(self, env, specs, posts, groups):
        self._env = env
        self._parts = []
        where = [None] * len(specs)   # term -> (part, first column in the part's row, width)
        for bindings, idx in groups:
            sub_specs = [specs[i] for i in idx]
            mdp = F.make_plan(env, "observation", sub_specs, posts=[posts[i] for i in idx], bindings=bindings)
            col = 0
            for i, spec in zip(idx, mdp.observations):
                width = mdp._obs_dim(spec)
                where[i] = (len(self._parts), col, width)
                col += width
            self._parts.append((bindings, sub_specs, mdp))
        self.observations = [self._Term(w[2]) for w in where]
        self.obs_dim = sum(w[2] for w in where)
        self.obs = torch.zeros(env.num_envs, self.obs_dim, device=env.device)
        self._moves, col = [], 0      # (part, source column, row column, width) per term, in the group's order
        for part, src, width in where:
            self._moves.append((part, src, col, width))
            col += width

    @staticmethod
    def _obs_dim(term) -> int:
        return term.width

    def bind(self, noise_u=None):
        """``noise_u``: (num_envs, obs_dim) draws for the group's row, dealt to the plans by column, or None."""
        for k, (_, _, mdp) in enumerate(self._parts):
            if noise_u is None:
                mdp.bind(noise_u=None)
            else:
                cols = [noise_u[:, dst:dst + w] for part, _, dst, w in self._moves if part == k]
                mdp.bind(noise_u=torch.cat(cols, dim=1).contiguous())

    def run(self):
        for bindings, sub_specs, mdp in self._parts:
            F.run_plan(mdp, self._env, "observation", sub_specs, bindings)
        for part, src, dst, width in self._moves:
            self.obs[:, dst:dst + width].copy_(self._parts[part][2].obs[:, src:src + width])


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
                group_history = self._flag_of(gcfg, "history_length", None)
                if group_history is not None:      # observation_manager.py:403-406
                    term.history_length, term.flatten_history_dim = group_history, self._flag_of(gcfg, "flatten_history_dim", True)
                if term.history_length > 0 and not term.flatten_history_dim and self._flag_of(gcfg, "concatenate_terms", True):
                    # the reference concatenates (N, H, d) blocks along d, which only works for equal H: not supported here
                    raise NotImplementedError("un-flattened observation histories need a group with concatenate_terms=False")
                if term.noise is not None and not callable(getattr(term.noise, "func", None)):
                    raise TypeError(f"Noise configuration of observation term '{name}' has no callable 'func': {term.noise}")
                for mod_cfg in term.modifiers or ():            # observation_manager.py:443-470
                    if not isinstance(mod_cfg, modifiers.ModifierCfg):
                        raise TypeError(f"Modifier configuration '{mod_cfg}' of observation term '{name}' is not of"
                                        f" required type ModifierCfg, Received: '{type(mod_cfg)}'")
                    if inspect.isclass(mod_cfg.func) and not issubclass(mod_cfg.func, modifiers.ModifierBase):
                        raise TypeError(f"Modifier function '{mod_cfg.func}' for observation term '{name}'"
                                        f" is not a subclass of 'ModifierBase'. Received: '{type(mod_cfg.func)}'.")
                    if not callable(mod_cfg.func):
                        raise AttributeError(f"Modifier '{mod_cfg}' of observation term '{name}' is not callable."
                                             f" Received: {mod_cfg.func}")
                names.append(name), cfgs.append(term), specs.append(resolve_term(env, name, term))
            self._group_obs_term_names[gname], self._group_obs_term_cfgs[gname], self._group_specs[gname] = names, cfgs, specs
            self._group_cfg[gname] = gcfg
        self._plans = {}
        self._post: dict[str, list] = {}
        self._group_obs_term_history_buffer: dict[str, dict[str, CircularBuffer]] = {}
        self._history_rows: dict[str, tuple] = {}
        self._group_obs_class_modifiers: list[modifiers.ModifierBase] = []
        self._post_noise_draws: dict[str, torch.Tensor] = {}   # "group/term" -> draws to replay (no counterpart in the reference)
        self._obs_buffer = None

    @property
    def active_terms(self) -> dict[str, list[str]]:
        return self._group_obs_term_names

    @property
    def group_obs_dim(self) -> dict[str, tuple[int, ...]]:
        dims = {}
        for g in self._group_specs:
            mdp = self._plan(g)[3]
            rows = self._history_rows.get(g)
            if self._flag(g, "concatenate_terms", True):
                dims[g] = (mdp.obs_dim if rows is None else rows[0].shape[1],)
            else:   # observation_manager.py:100-101: the terms' own shapes
                dims[g] = [tuple(t.shape[1:]) for t in self._split(g, mdp.obs if rows is None else rows[0]).values()]
        return dims

    @staticmethod
    def _fused_noise(noise) -> bool:
        """Additive uniform noise with scalar bounds is drawn inside the fused launch; every other noise runs after it."""
        return (isinstance(noise, UniformNoiseCfg) and noise.operation == "add" and not isinstance(noise.n_min, torch.Tensor)
                and not isinstance(noise.n_max, torch.Tensor))

    @classmethod
    def _needs_post(cls, term) -> bool:
        # a per-column scale (observation_manager.py:424-440) is applied after the fused launch as well
        return (bool(term.modifiers) or (term.noise is not None and not cls._fused_noise(term.noise))
                or (term.scale is not None and not isinstance(term.scale, (int, float))))

    @staticmethod
    def _flag_of(g, name, default):
        return g.get(name, default) if isinstance(g, dict) else getattr(g, name, default)

    def _flag(self, group, name, default):
        return self._flag_of(self._group_cfg[group], name, default)

    def _plan(self, group):
        plan = self._plans.get(group)
        if plan is None:
            specs, cfgs = self._group_specs[group], self._group_obs_term_cfgs[group]
            groups = F.group_specs(specs)
            corrupt = self._flag(group, "enable_corruption", False)
            posts = [dict(noise=None, clip=None, scale=None) if self._needs_post(c) else
                     dict(noise=(c.noise.n_min, c.noise.n_max) if (c.noise is not None and corrupt) else None, clip=c.clip,
                          scale=c.scale) for c in cfgs]
            if len(groups) == 1:
                mdp = F.make_plan(self._env, "observation", specs, posts=posts, bindings=groups[0][0])
            else:   # terms of several articulations / sensor sets (a robot and an object, say): one fused plan per set
                mdp = _CompositeObservationPlan(self._env, specs, posts, groups)
            plan = self._plans[group] = (groups[0][0], None, specs, mdp)
            post, col = [], 0
            for name, c, spec in zip(self._group_obs_term_names[group], cfgs, mdp.observations):
                width = mdp._obs_dim(spec)
                if c.scale is not None:   # observation_manager.py:425-440: checked once, when the manager is built
                    if not isinstance(c.scale, (float, int, tuple, torch.Tensor)):
                        raise TypeError(f"Scale for observation term '{name}' in group '{group}' is not of type float, int or tuple."
                                        f" Received: '{type(c.scale)}'.")
                    if isinstance(c.scale, (tuple, torch.Tensor)):
                        if len(c.scale) != width:
                            raise ValueError(f"Scale for observation term '{name}' in group '{group}' does not match the dimensions"
                                             f" of the observation. Expected: {width} but received: {len(c.scale)}.")
                        c.scale = torch.as_tensor(c.scale, dtype=torch.float, device=self._env.device)
                if self._needs_post(c):
                    for mod_cfg in c.modifiers or ():   # class modifiers are built once with the term's shape (:449-458)
                        if inspect.isclass(mod_cfg.func):
                            mod_cfg.func = mod_cfg.func(cfg=mod_cfg, data_dim=(self._env.num_envs, width), device=self._env.device)
                            self._group_obs_class_modifiers.append(mod_cfg.func)
                    post.append((col, col + width, c, name if corrupt else None))
                col += width
            self._post[group] = post
            if any(c.history_length > 0 for c in cfgs):   # observation_manager.py:412-420
                n, dev = self._env.num_envs, self._env.device
                rings, layout, col, out_col = {}, [], 0, 0
                for name, c, spec in zip(self._group_obs_term_names[group], cfgs, mdp.observations):
                    width = mdp._obs_dim(spec)
                    out_width = width * max(c.history_length, 1)
                    if c.history_length > 0:
                        rings[name] = CircularBuffer(max_len=c.history_length, batch_size=n, device=dev)
                    layout.append((name, col, col + width, out_col, out_col + out_width))
                    col, out_col = col + width, out_col + out_width
                self._group_obs_term_history_buffer[group] = rings
                self._history_rows[group] = (torch.zeros(n, out_col, device=dev), layout)
        return plan

    def _split(self, group: str, row: torch.Tensor) -> dict[str, torch.Tensor]:
        """The terms of a group with ``concatenate_terms=False`` as views of the group's row (observation_manager.py:332-335);
        a history that is not flattened comes out as (num_envs, history_length, width)."""
        mdp = self._plans[group][3]
        out, col = {}, 0
        for name, c, spec in zip(self._group_obs_term_names[group], self._group_obs_term_cfgs[group], mdp.observations):
            width = mdp._obs_dim(spec)
            span = width * max(c.history_length, 1)
            block = row[:, col:col + span]
            out[name] = block.unflatten(1, (c.history_length, width)) if (c.history_length > 0 and not c.flatten_history_dim) else block
            col += span
        return out

    def reset(self, env_ids=None) -> dict[str, float]:
        """observation_manager.py:222-236: class modifiers drop the history of the reset envs."""
        for group in self._group_specs:
            if any(c.modifiers or c.history_length > 0 for c in self._group_obs_term_cfgs[group]):
                self._plan(group)
        for mod in self._group_obs_class_modifiers:
            mod.reset(env_ids=env_ids)
        for rings in self._group_obs_term_history_buffer.values():   # :227-230
            for ring in rings.values():
                ring.reset(batch_ids=env_ids)
        return {}

    def compute(self) -> dict[str, torch.Tensor]:
        """observation_manager.py:240-258."""
        self._obs_buffer = {g: self.compute_group(g) for g in self._group_specs}
        return self._obs_buffer

    def compute_group(self, group_name: str, noise_u: torch.Tensor | None = None) -> torch.Tensor | dict[str, torch.Tensor]:
        """observation_manager.py:260-335.  ``noise_u`` (no counterpart in the reference) injects the U[0,1) draws."""
        if group_name not in self._group_specs:
            raise ValueError(f"Unable to find the group '{group_name}' in the observation manager."
                             f" Available groups are: {list(self._group_specs)}")
        bindings, _, specs, mdp = self._plan(group_name)
        mdp.bind(noise_u=noise_u)
        if isinstance(mdp, _CompositeObservationPlan):
            mdp.run()
        else:
            F.run_plan(mdp, self._env, "observation", specs, bindings)
        for c0, c1, cfg, noisy in self._post[group_name]:
            cols = mdp.obs[:, c0:c1]
            obs = cols
            for m in cfg.modifiers or ():
                obs = m.func(obs, **m.params)
            if cfg.noise is not None and noisy is not None:      # :313-314, only in groups with enable_corruption (:398-400)
                draws = self._post_noise_draws.get(f"{group_name}/{noisy}")
                obs = cfg.noise.func(obs, cfg.noise) if draws is None else cfg.noise.func(obs, cfg.noise, draws)
            if cfg.clip:          # in place, as the reference's clip_ / mul_ (an Integrator's state is the tensor it returned)
                _mod.apply(_mod.CLIP, obs, out=obs, p0=float(cfg.clip[0]), p1=float(cfg.clip[1]), has_lo=1, has_hi=1)
            if cfg.scale is not None and isinstance(cfg.scale, (int, float)):
                _mod.apply(_mod.SCALE, obs, out=obs, p0=float(cfg.scale))
            elif cfg.scale is not None:   # one factor per column (the tensor was built when the group was planned)
                obs = constant_noise(obs, ConstantNoiseCfg(bias=cfg.scale, operation="scale"))
            if obs.data_ptr() != cols.data_ptr():
                _mod.apply(_mod.COPY, obs, out=cols)
        rows = self._history_rows.get(group_name)
        split = not self._flag(group_name, "concatenate_terms", True)
        if rows is None:
            return self._split(group_name, mdp.obs) if split else mdp.obs
        final, layout = rows
        rings = self._group_obs_term_history_buffer[group_name]
        for name, c0, c1, o0, o1 in layout:
            if name in rings:
                rings[name].append(mdp.obs[:, c0:c1], ordered_out=final[:, o0:o1])
            else:
                _mod.apply(_mod.COPY, mdp.obs[:, c0:c1], out=final[:, o0:o1])
        return self._split(group_name, final) if split else final
