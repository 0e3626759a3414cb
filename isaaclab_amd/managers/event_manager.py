"""``EventManager`` (interface of ``isaaclab/managers/event_manager.py``): mode dispatch of event terms.

The manager itself is bookkeeping on a few (N,) tensors -- per-env interval timers, per-env "last triggered at step"
counters -- kept on the device with the reference's dtypes; the terms it calls are the HIP event kernels of
:mod:`isaaclab_amd.envs.mdp.events` (or any callable ``func(env, env_ids, **params)``).  Like the reference it picks the
due envs with ``nonzero()``, i.e. one host synchronisation per interval / throttled reset term and call.
"""

from __future__ import annotations

import inspect
import warnings
from collections.abc import Sequence
from dataclasses import dataclass

import torch

from .manager_term_cfg import ManagerTermBaseCfg
from .scene_entity_cfg import SceneEntityCfg


@dataclass
class EventTermCfg(ManagerTermBaseCfg):
    """manager_term_cfg.py:240-296."""

    mode: str = None
    interval_range_s: tuple[float, float] | None = None
    is_global_time: bool = False
    min_step_count_between_reset: int = 0


class ManagerTermBase:
    """manager_base.py:26-113: a term implemented as a class -- constructed once with ``(cfg, env)``, then called like a
    function; ``reset(env_ids)`` is invoked by the owning manager's ``reset``."""

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env

    @property
    def num_envs(self) -> int:
        return self._env.num_envs

    @property
    def device(self):
        return self._env.device

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        pass

    def __call__(self, *args, **kwargs):
        raise NotImplementedError


class EventManager:
    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self.num_envs, self.device = env.num_envs, torch.device(env.device)
        self._mode_term_names: dict[str, list[str]] = {}
        self._mode_term_cfgs: dict[str, list[EventTermCfg]] = {}
        self._mode_class_term_cfgs: dict[str, list[EventTermCfg]] = {}
        self._interval_term_time_left: list[torch.Tensor] = []
        self._reset_term_last_triggered_step_id: list[torch.Tensor] = []
        self._reset_term_last_triggered_once: list[torch.Tensor] = []
        items = cfg.items() if isinstance(cfg, dict) else vars(cfg).items()
        for name, term in items:  # event_manager.py:320-399
            if term is None:
                continue
            if not isinstance(term, EventTermCfg):
                raise TypeError(f"Configuration for the term '{name}' is not of type EventTermCfg. Received: '{type(term)}'.")
            if term.mode != "reset" and term.min_step_count_between_reset != 0:
                warnings.warn(f"Event term '{name}' has 'min_step_count_between_reset' set to a non-zero value but the mode is not"
                              " 'reset'. Ignoring the 'min_step_count_between_reset' value.")
            self._resolve_common_term_cfg(name, term)
            if term.mode not in self._mode_term_names:
                self._mode_term_names[term.mode], self._mode_term_cfgs[term.mode] = [], []
                self._mode_class_term_cfgs[term.mode] = []
            self._mode_term_names[term.mode].append(name)
            self._mode_term_cfgs[term.mode].append(term)
            if isinstance(term.func, ManagerTermBase):
                self._mode_class_term_cfgs[term.mode].append(term)
            if term.mode == "interval":
                if term.interval_range_s is None:
                    raise ValueError(f"Event term '{name}' has mode 'interval' but 'interval_range_s' is not specified.")
                lower, upper = term.interval_range_s
                if term.is_global_time:
                    self._interval_term_time_left.append(self._rand(1, "cpu") * (upper - lower) + lower)
                else:
                    self._interval_term_time_left.append(self._rand(self.num_envs, self.device) * (upper - lower) + lower)
            elif term.mode == "reset":
                if term.min_step_count_between_reset < 0:
                    raise ValueError(f"Event term '{name}' has mode 'reset' but 'min_step_count_between_reset' is negative: "
                                     f"{term.min_step_count_between_reset}. Please provide a non-negative value.")
                self._reset_term_last_triggered_step_id.append(torch.zeros(self.num_envs, device=self.device, dtype=torch.int32))
                self._reset_term_last_triggered_once.append(torch.zeros(self.num_envs, device=self.device, dtype=torch.bool))

    def _rand(self, count: int, device) -> torch.Tensor:
        """Unit-interval draws for the interval timers (``torch.rand`` in the reference); tests replace it to replay."""
        return torch.rand(count, device=device)

    def _resolve_common_term_cfg(self, name: str, term: EventTermCfg):
        """manager_base.py:243-298 for event terms: callable check, class terms constructed, entities resolved, and the
        term must accept ``(env, env_ids, **params)``."""
        if inspect.isclass(term.func):
            if not issubclass(term.func, ManagerTermBase):
                raise TypeError(f"Configuration for the term '{name}' is not of type ManagerTermBase. Received: '{type(term.func)}'.")
        elif not callable(term.func):
            raise AttributeError(f"The term '{name}' is not callable. Received: {term.func}")
        for value in term.params.values():
            if isinstance(value, SceneEntityCfg):
                value.resolve(self._env.scene)
        target = term.func.__call__ if inspect.isclass(term.func) else term.func
        try:
            params = list(inspect.signature(target).parameters.values())
        except (TypeError, ValueError):
            params = None
        if params is not None:
            if inspect.isclass(term.func):
                params = params[1:]  # self
            named = [p for p in params[2:] if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY) and not p.name.startswith("_")]
            has_kwargs = any(p.kind == p.VAR_KEYWORD for p in params)
            required = {p.name for p in named if p.default is inspect.Parameter.empty}
            known = {p.name for p in named}
            given = set(term.params)
            if len(params) < 2 or not required <= given or (not has_kwargs and not given <= known):
                raise ValueError(f"The term '{name}' expects mandatory parameters: {sorted(required)} and optional parameters: "
                                 f"{sorted(known - required)}, but received: {sorted(given)}.")
        if inspect.isclass(term.func):
            term.func = term.func(cfg=term, env=self._env)

    def __str__(self) -> str:
        msg = f"<EventManager> contains {len(self._mode_term_names)} active terms.\n"
        for mode, names in self._mode_term_names.items():
            msg += f"Active Event Terms in Mode: '{mode}': " + ", ".join(names) + "\n"
        return msg

    @property
    def active_terms(self) -> dict[str, list[str]]:
        return self._mode_term_names

    @property
    def available_modes(self) -> list[str]:
        return list(self._mode_term_names.keys())

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        """event_manager.py:117-145.  As there, the interval timers that restart with the episode are those of the
        *class* terms of the interval mode, addressed by their position in that list."""
        for mode_cfg in self._mode_class_term_cfgs.values():
            for term in mode_cfg:
                term.func.reset(env_ids=env_ids)
        num_envs = self.num_envs if env_ids is None else len(env_ids)
        for index, term in enumerate(self._mode_class_term_cfgs.get("interval", [])):
            if not term.is_global_time:
                lower, upper = term.interval_range_s
                self._interval_term_time_left[index][env_ids] = self._rand(num_envs, self.device) * (upper - lower) + lower
        return {}

    def apply(self, mode: str, env_ids: Sequence[int] | None = None, dt: float | None = None,
              global_env_step_count: int | None = None):
        """event_manager.py:147-262."""
        if mode not in self._mode_term_names:
            warnings.warn(f"Event mode '{mode}' is not defined. Skipping event.")
            return
        if mode == "interval" and dt is None:
            raise ValueError(f"Event mode '{mode}' requires the time-step of the environment.")
        if mode == "interval" and env_ids is not None:
            raise ValueError(f"Event mode '{mode}' does not require environment indices. This is an undefined behavior as the "
                             "environment indices are computed based on the time left for each environment.")
        if mode == "reset" and global_env_step_count is None:
            raise ValueError(f"Event mode '{mode}' requires the total number of environment steps to be provided.")
        for index, term in enumerate(self._mode_term_cfgs[mode]):
            if mode == "interval":
                time_left = self._interval_term_time_left[index]
                time_left -= dt
                lower, upper = term.interval_range_s
                if term.is_global_time:
                    if time_left < 1e-6:
                        time_left[:] = self._rand(1, "cpu") * (upper - lower) + lower
                        term.func(self._env, None, **term.params)
                else:
                    valid_env_ids = (time_left < 1e-6).nonzero().flatten()
                    if len(valid_env_ids) > 0:
                        time_left[valid_env_ids] = self._rand(len(valid_env_ids), self.device) * (upper - lower) + lower
                        term.func(self._env, valid_env_ids, **term.params)
            elif mode == "reset":
                min_step_count = term.min_step_count_between_reset
                if env_ids is None:
                    env_ids = slice(None)
                last_step = self._reset_term_last_triggered_step_id[index]
                once = self._reset_term_last_triggered_once[index]
                if min_step_count == 0:
                    last_step[env_ids] = global_env_step_count
                    once[env_ids] = True
                    term.func(self._env, env_ids, **term.params)
                else:
                    last_triggered_step = last_step[env_ids]
                    valid_trigger = (global_env_step_count - last_triggered_step) >= min_step_count
                    valid_trigger |= (last_triggered_step == 0) & ~once[env_ids]
                    if isinstance(env_ids, slice):
                        valid_env_ids = valid_trigger.nonzero().flatten()
                    else:
                        valid_env_ids = torch.as_tensor(env_ids, device=self.device)[valid_trigger]
                    if len(valid_env_ids) > 0:
                        once[valid_env_ids] = True
                        last_step[valid_env_ids] = global_env_step_count
                        term.func(self._env, valid_env_ids, **term.params)
            else:
                term.func(self._env, env_ids, **term.params)

    def set_term_cfg(self, term_name: str, cfg: EventTermCfg):
        for mode, terms in self._mode_term_names.items():
            if term_name in terms:
                self._mode_term_cfgs[mode][terms.index(term_name)] = cfg
                return
        raise ValueError(f"Event term '{term_name}' not found.")

    def get_term_cfg(self, term_name: str) -> EventTermCfg:
        for mode, terms in self._mode_term_names.items():
            if term_name in terms:
                return self._mode_term_cfgs[mode][terms.index(term_name)]
        raise ValueError(f"Event term '{term_name}' not found.")
