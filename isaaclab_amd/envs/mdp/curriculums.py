"""Curriculum terms (``isaaclab_tasks/manager_based/locomotion/velocity/mdp/curriculums.py``) and the
``CurriculumManager`` that calls them (``isaaclab/managers/curriculum_manager.py``)."""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import torch

from ...managers.manager_term_cfg import ManagerTermBaseCfg
from ...managers.scene_entity_cfg import SceneEntityCfg


def terrain_levels_vel(env, env_ids: Sequence[int], asset_cfg: SceneEntityCfg = SceneEntityCfg("robot"), *, _random_levels=None):
    """curriculums.py:26-55: envs that walked further than half a sub-terrain move a level up, envs that covered less than
    half of what their command asked for move one down; returns the mean terrain level.

    With an :class:`isaaclab_amd.terrains.TerrainImporter` the whole term is one launch; any other terrain object gets
    the reference's sequence (move flags from torch ops, then its own ``update_env_origins``).
    """
    from ...terrains import TerrainImporter

    asset = env.scene[asset_cfg.name]
    data = getattr(asset, "data", asset)
    terrain = env.scene.terrain
    command = env.command_manager.get_command("base_velocity")
    if isinstance(terrain, TerrainImporter):
        if terrain.env_origins is not env.scene.env_origins:
            raise ValueError("env.scene.env_origins must be the terrain importer's env_origins tensor")
        terrain.step_curriculum(env_ids, data.root_state_w, command, env.max_episode_length_s, _random_levels=_random_levels)
    else:
        distance = torch.norm(data.root_state_w[env_ids, :2] - env.scene.env_origins[env_ids, :2], dim=1)
        move_up = distance > terrain.cfg.terrain_generator.size[0] / 2
        move_down = distance < torch.norm(command[env_ids, :2], dim=1) * env.max_episode_length_s * 0.5
        move_down *= ~move_up
        terrain.update_env_origins(env_ids, move_up, move_down)
    return torch.mean(terrain.terrain_levels.float())


def modify_reward_weight(env, env_ids: Sequence[int], term_name: str, weight: float, num_steps: int):
    """``isaaclab/envs/mdp/curriculums.py:20-36``: after ``num_steps`` common steps the reward term gets a new weight (the
    fused reward plan is rebuilt once, when the weight really changes)."""
    if env.common_step_counter > num_steps:
        term_cfg = env.reward_manager.get_term_cfg(term_name)
        if term_cfg.weight != weight:
            term_cfg.weight = weight
            env.reward_manager.set_term_cfg(term_name, term_cfg)


@dataclass
class CurriculumTermCfg(ManagerTermBaseCfg):
    """manager_term_cfg.py:127-137."""


class CurriculumManager:
    """curriculum_manager.py:22-185: terms ``func(env, env_ids, **params) -> state``; ``reset`` logs the last states."""

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._term_names, self._term_cfgs = [], []
        self._curriculum_state = {}
        items = cfg.items() if isinstance(cfg, dict) else vars(cfg).items()
        for name, term in items:
            if term is None:
                continue
            if not isinstance(term, CurriculumTermCfg):
                raise TypeError(f"Configuration for the term '{name}' is not of type CurriculumTermCfg. Received: '{type(term)}'.")
            if not callable(term.func):
                raise AttributeError(f"The term '{name}' is not callable. Received: {term.func}")
            for value in term.params.values():
                if isinstance(value, SceneEntityCfg):
                    value.resolve(env.scene)
            self._term_names.append(name)
            self._term_cfgs.append(term)
            self._curriculum_state[name] = None

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        extras = {}
        for name, state in self._curriculum_state.items():
            if state is None:
                continue
            if isinstance(state, dict):
                for key, value in state.items():
                    extras[f"Curriculum/{name}/{key}"] = value.item() if isinstance(value, torch.Tensor) else value
            else:
                extras[f"Curriculum/{name}"] = state.item() if isinstance(state, torch.Tensor) else state
        return extras

    def compute(self, env_ids: Sequence[int] | None = None):
        if env_ids is None:
            env_ids = slice(None)
        for name, term in zip(self._term_names, self._term_cfgs):
            self._curriculum_state[name] = term.func(self._env, env_ids, **term.params)
