"""``ActionTerm`` / ``ActionManager`` (interface of ``isaaclab/managers/action_manager.py``)."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch


@dataclass
class ActionTermCfg:
    """manager_term_cfg.py (ActionTermCfg)."""

    class_type: type = None
    asset_name: str = "robot"
    debug_vis: bool = False
    clip: dict | None = None


class ActionTerm:
    """action_manager.py:29-161: ``process_actions`` once per env step, ``apply_actions`` once per physics substep."""

    def __init__(self, cfg: ActionTermCfg, env):
        self.cfg, self._env = cfg, env
        self.num_envs, self.device = env.num_envs, torch.device(env.device)
        self._asset = env.scene[cfg.asset_name]

    @property
    def action_dim(self) -> int:
        raise NotImplementedError

    @property
    def raw_actions(self) -> torch.Tensor:
        raise NotImplementedError

    @property
    def processed_actions(self) -> torch.Tensor:
        raise NotImplementedError

    def process_actions(self, actions: torch.Tensor):
        raise NotImplementedError

    def apply_actions(self):
        raise NotImplementedError

    def reset(self, env_ids: Sequence[int] | None = None) -> None:
        pass


def _is_action_term(term) -> bool:
    return all(hasattr(term, a) for a in ("action_dim", "process_actions", "apply_actions"))


class ActionManager:
    """action_manager.py:164-393.

    With a single term (every task configuration of the five benchmarks) the manager's own bookkeeping
    (``prev_action <- action <- raw``) rides in the term's launch; with several terms it is two copies followed by one
    launch per term on that term's slice of the action.
    """

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._term_names: list[str] = []
        self._terms: dict[str, ActionTerm] = {}
        items = cfg.items() if isinstance(cfg, dict) else vars(cfg).items()
        for name, term_cfg in items:
            if term_cfg is None:
                continue
            if not hasattr(term_cfg, "class_type"):
                raise TypeError(f"Configuration for the term '{name}' is not of type ActionTermCfg. Received: '{type(term_cfg)}'.")
            term = term_cfg.class_type(term_cfg, env)
            if not _is_action_term(term):
                raise TypeError(f"Returned object for the term '{name}' is not of type ActionType.")
            self._term_names.append(name)
            self._terms[name] = term
        n, dev = env.num_envs, torch.device(env.device)
        self._action = torch.zeros(n, self.total_action_dim, device=dev)
        self._prev_action = torch.zeros_like(self._action)

    def __str__(self) -> str:
        return f"<ActionManager> contains {len(self._term_names)} active terms (shape: {self.total_action_dim}): " + ", ".join(
            f"{name} ({self._terms[name].action_dim})" for name in self._term_names)

    @property
    def total_action_dim(self) -> int:
        return sum(self.action_term_dim)

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    @property
    def action_term_dim(self) -> list[int]:
        return [term.action_dim for term in self._terms.values()]

    @property
    def action(self) -> torch.Tensor:
        return self._action

    @property
    def prev_action(self) -> torch.Tensor:
        return self._prev_action

    def reset(self, env_ids: Sequence[int] | None = None) -> dict:
        if env_ids is None:
            env_ids = slice(None)
        self._prev_action[env_ids] = 0.0
        self._action[env_ids] = 0.0
        for term in self._terms.values():
            term.reset(env_ids=env_ids)
        return {}

    def process_action(self, action: torch.Tensor):
        if self.total_action_dim != action.shape[1]:  # action_manager.py:328-329
            raise ValueError(f"Invalid action shape, expected: {self.total_action_dim}, received: {action.shape[1]}.")
        action = action.to(self._action.device)
        terms = list(self._terms.values())
        if len(terms) == 1 and hasattr(terms[0], "_process_with_manager"):
            terms[0]._process_with_manager(action, self._action, self._prev_action)
            return
        self._prev_action[:] = self._action
        self._action[:] = action
        idx = 0
        for term in terms:
            term.process_actions(action[:, idx: idx + term.action_dim])
            idx += term.action_dim

    def apply_action(self) -> None:
        for term in self._terms.values():
            term.apply_actions()

    def get_term(self, name: str) -> ActionTerm:
        return self._terms[name]
