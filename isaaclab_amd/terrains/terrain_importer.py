"""Env-origin side of ``TerrainImporter`` (``isaaclab/terrains/terrain_importer.py:283-365``): where each env lives on
the terrain, and the curriculum's level changes.  Importing / generating the terrain prims themselves is the
simulator's business; the collision mesh reaches the ray caster through :func:`isaaclab_amd.utils.mesh.convert_to_mesh`.
"""

from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch

from .. import _abi
from ..utils import rng as _rng



@dataclass
class TerrainImporterCfg:
    """The fields of terrain_importer_cfg.py that the origin logic reads."""

    num_envs: int = 1
    env_spacing: float | None = None
    max_init_terrain_level: int | None = None
    terrain_generator: object = None  # anything with ``.size`` (sub-terrain width, length in metres)


class TerrainImporter:
    def __init__(self, cfg: TerrainImporterCfg, device, seed: int | None = None):
        # seed None: the seed the environment installed (utils/rng.py), read at every draw
        self.cfg, self.device, self._seed = cfg, torch.device(device), seed
        self.terrain_origins = None
        self.env_origins = None
        self.terrain_levels = self.terrain_types = None
        self.max_terrain_level = 0

    @property
    def seed(self) -> int:
        return _rng.seed() if self._seed is None else int(self._seed)

    def configure_env_origins(self, origins=None):
        """terrain_importer.py:283-305: sub-terrain origins (num_rows, num_cols, 3) -> curriculum placement, else a grid."""
        if origins is not None:
            if isinstance(origins, np.ndarray):
                origins = torch.from_numpy(origins)
            self.terrain_origins = origins.to(self.device, dtype=torch.float).contiguous()
            self.env_origins = self._compute_env_origins_curriculum(self.cfg.num_envs, self.terrain_origins)
        else:
            self.terrain_origins = None
            if self.cfg.env_spacing is None:
                raise ValueError("Environment spacing must be specified for configuring grid-like origins.")
            self.env_origins = self._compute_env_origins_grid(self.cfg.num_envs, self.cfg.env_spacing)

    def _launch(self, env_ids, n_ids, root_state_w, command, half_size, max_episode_length_s, move_up, move_down, random_levels):
        rows, cols = self.terrain_origins.shape[:2]
        _abi.check(_abi.curriculum_terrain_levels(
            env_ids.data_ptr(), n_ids, _abi.ptr(root_state_w), _abi.ptr(command), 0 if command is None else command.shape[1],
            half_size, max_episode_length_s, None if move_up is None else move_up.data_ptr(),
            None if move_down is None else move_down.data_ptr(), self.terrain_levels.data_ptr(), self.terrain_types.data_ptr(),
            _abi.ptr(self.terrain_origins), rows, cols, None if random_levels is None else random_levels.data_ptr(), self.seed,
            _rng.next_draw("terrain"), _abi.ptr(self.env_origins), _abi.stream()), "ilx_curriculum_terrain_levels")

    def _ids(self, env_ids) -> torch.Tensor:
        if env_ids is None or (isinstance(env_ids, slice) and env_ids == slice(None)):
            return torch.arange(self.cfg.num_envs, dtype=torch.long, device=self.device)
        return torch.as_tensor(env_ids, dtype=torch.long, device=self.device).contiguous()

    def update_env_origins(self, env_ids, move_up: torch.Tensor, move_down: torch.Tensor, *, _random_levels=None):
        """terrain_importer.py:307-322.  ``_random_levels`` (one int64 per listed env) replays the reference's
        ``randint_like`` draws in tests; otherwise the replacement levels come from Philox."""
        if self.terrain_origins is None:
            return
        ids = self._ids(env_ids)
        up = move_up.to(self.device, torch.bool).contiguous()
        down = move_down.to(self.device, torch.bool).contiguous()
        rl = None if _random_levels is None else _random_levels.to(self.device, torch.long).contiguous()
        self._launch(ids, ids.numel(), None, None, 0.0, 0.0, up, down, rl)

    def step_curriculum(self, env_ids, root_state_w, command, max_episode_length_s: float, *, _random_levels=None):
        """curriculums.py:44-53 and update_env_origins in one launch: move flags from the distance walked, then the
        level / origin update."""
        if self.terrain_origins is None:
            return
        ids = self._ids(env_ids)
        rl = None if _random_levels is None else _random_levels.to(self.device, torch.long).contiguous()
        half = self.cfg.terrain_generator.size[0] / 2
        self._launch(ids, ids.numel(), root_state_w, command, half, max_episode_length_s, None, None, rl)

    def _compute_env_origins_curriculum(self, num_envs: int, origins: torch.Tensor) -> torch.Tensor:
        """terrain_importer.py:328-349."""
        num_rows, num_cols = origins.shape[:2]
        if self.cfg.max_init_terrain_level is None:
            max_init_level = num_rows - 1
        else:
            max_init_level = min(self.cfg.max_init_terrain_level, num_rows - 1)
        self.max_terrain_level = num_rows
        self.terrain_levels = torch.randint(0, max_init_level + 1, (num_envs,), device=self.device)
        self.terrain_types = torch.div(torch.arange(num_envs, device=self.device), (num_envs / num_cols),
                                       rounding_mode="floor").to(torch.long)
        env_origins = torch.zeros(num_envs, 3, device=self.device)
        env_origins[:] = origins[self.terrain_levels, self.terrain_types]
        return env_origins

    def _compute_env_origins_grid(self, num_envs: int, env_spacing: float) -> torch.Tensor:
        """terrain_importer.py:351-364."""
        env_origins = torch.zeros(num_envs, 3, device=self.device)
        num_rows = np.ceil(num_envs / int(np.sqrt(num_envs)))
        num_cols = np.ceil(num_envs / num_rows)
        ii, jj = torch.meshgrid(torch.arange(num_rows, device=self.device), torch.arange(num_cols, device=self.device), indexing="ij")
        env_origins[:, 0] = -(ii.flatten()[:num_envs] - (num_rows - 1) / 2) * env_spacing
        env_origins[:, 1] = (jj.flatten()[:num_envs] - (num_cols - 1) / 2) * env_spacing
        env_origins[:, 2] = 0.0
        return env_origins


def generator_cfg(size=(8.0, 8.0)):
    """A stand-in for ``TerrainGeneratorCfg`` carrying the sub-terrain size."""
    return SimpleNamespace(size=tuple(size))
