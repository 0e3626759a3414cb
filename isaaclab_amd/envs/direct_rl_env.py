"""``DirectRLEnv`` (interface of ``isaaclab/envs/direct_rl_env.py``): the reference's step and reset sequences around the
hooks a task implements (``_pre_physics_step``, ``_apply_action``, ``_get_dones``, ``_get_rewards``, ``_get_observations``,
``_reset_idx``), with the action / observation noise models and the event manager on this package's HIP kernels.

The simulator is the caller's: ``physics_step(env)`` stands where the reference calls ``sim.step()``.
(:class:`isaaclab_amd.envs.inhand_step.InHandManipulationStep` is the Shadow-hand task's hooks fused into two launches.)
"""

from __future__ import annotations

import math
from collections.abc import Callable, Sequence
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from ..managers import EventManager
from ..utils import math as math_utils
from ..utils import rng


@dataclass
class DirectRLEnvCfg:
    """direct_rl_env_cfg.py: the fields the step reads."""

    decimation: int = 1
    sim_dt: float = 1.0 / 60.0
    episode_length_s: float = 10.0
    is_finite_horizon: bool = False
    seed: int | None = None
    events: object = None
    action_noise_model: object = None          # NoiseModelCfg
    observation_noise_model: object = None     # NoiseModelCfg


class DirectRLEnv:
    def __init__(self, cfg: DirectRLEnvCfg, scene, physics_step: Callable | None = None, device=None):
        self.cfg, self.scene = cfg, scene
        self.num_envs = scene.num_envs
        self.device = torch.device(device if device is not None else scene.device)
        self._physics_step = physics_step
        self.sim = SimpleNamespace(get_physics_dt=lambda: self.cfg.sim_dt, device=self.device)
        self.seed = 0
        if cfg.seed is not None:
            self.seed = self.seed_all(cfg.seed)
        n, dev = self.num_envs, self.device
        self._sim_step_counter = 0
        self.common_step_counter = 0
        self.episode_length_buf = torch.zeros(n, dtype=torch.long, device=dev)
        self.reset_terminated = torch.zeros(n, dtype=torch.bool, device=dev)
        self.reset_time_outs = torch.zeros(n, dtype=torch.bool, device=dev)
        self.reset_buf = torch.zeros(n, dtype=torch.bool, device=dev)
        self.extras: dict = {}
        self.event_manager = None
        if cfg.events:                                               # direct_rl_env.py:146-150
            self.event_manager = EventManager(cfg.events, self)
            if "startup" in self.event_manager.available_modes:
                self.event_manager.apply(mode="startup")
        self._action_noise_model = self._observation_noise_model = None
        if cfg.action_noise_model:                                   # :183-191
            self._action_noise_model = cfg.action_noise_model.class_type(cfg.action_noise_model, num_envs=n, device=dev)
        if cfg.observation_noise_model:
            self._observation_noise_model = cfg.observation_noise_model.class_type(cfg.observation_noise_model, num_envs=n, device=dev)

    # -- properties (direct_rl_env.py:215-245)
    @property
    def physics_dt(self) -> float:
        return self.cfg.sim_dt

    @property
    def step_dt(self) -> float:
        return self.cfg.sim_dt * self.cfg.decimation

    @property
    def max_episode_length_s(self) -> float:
        return self.cfg.episode_length_s

    @property
    def max_episode_length(self) -> int:
        return math.ceil(self.max_episode_length_s / self.step_dt)

    @staticmethod
    def seed_all(seed: int) -> int:
        """direct_rl_env.py:388-404: torch's generators and every device-side generator (one seed; their shared draw
        counter restarts)."""
        torch.manual_seed(int(seed))
        return rng.seed_all(seed)

    # -- operations
    def reset(self, seed: int | None = None):
        """direct_rl_env.py:252-290."""
        if seed is not None:
            self.seed = self.seed_all(seed)
        self._reset_idx(torch.arange(self.num_envs, dtype=torch.int64, device=self.device))
        self._write_data_to_sim()
        return self._get_observations(), self.extras

    def step(self, action: torch.Tensor):
        """direct_rl_env.py:292-379."""
        action = action.to(self.device)
        if self._action_noise_model is not None:
            action = self._action_noise_model.apply(action)
        self._pre_physics_step(action)
        for _ in range(self.cfg.decimation):
            self._sim_step_counter += 1
            self._apply_action()
            self._write_data_to_sim()
            if self._physics_step is not None:
                self._physics_step(self)
            self.scene.update(dt=self.physics_dt)
        self.episode_length_buf += 1
        self.common_step_counter += 1
        self.reset_terminated[:], self.reset_time_outs[:] = self._get_dones()
        self.reset_buf = self.reset_terminated | self.reset_time_outs
        self.reward_buf = self._get_rewards()
        reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        if len(reset_env_ids) > 0:
            self._reset_idx(reset_env_ids)
            self._write_data_to_sim()
        if self.event_manager is not None and "interval" in self.event_manager.available_modes:
            self.event_manager.apply(mode="interval", dt=self.step_dt)
        self.obs_buf = self._get_observations()
        if self._observation_noise_model is not None:                # the state space of the critic gets no noise
            self.obs_buf["policy"] = self._observation_noise_model.apply(self.obs_buf["policy"])
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def _write_data_to_sim(self):
        self.scene.write_data_to_sim()

    # -- hooks (direct_rl_env.py:556-640)
    def _reset_idx(self, env_ids: Sequence[int]):
        """direct_rl_env.py:556-580; a task extends it (state writes, its own buffers)."""
        self.scene.reset(env_ids)
        if self.event_manager is not None and "reset" in self.event_manager.available_modes:
            self.event_manager.apply(mode="reset", env_ids=env_ids, global_env_step_count=self._sim_step_counter // self.cfg.decimation)
        if self._action_noise_model is not None:
            self._action_noise_model.reset(env_ids)
        if self._observation_noise_model is not None:
            self._observation_noise_model.reset(env_ids)
        self.episode_length_buf[env_ids] = 0

    def _pre_physics_step(self, actions: torch.Tensor):
        raise NotImplementedError(f"Please implement the '_pre_physics_step' method for {self.__class__.__name__}.")

    def _apply_action(self):
        raise NotImplementedError(f"Please implement the '_apply_action' method for {self.__class__.__name__}.")

    def _get_observations(self) -> dict:
        raise NotImplementedError(f"Please implement the '_get_observations' method for {self.__class__.__name__}.")

    def _get_rewards(self) -> torch.Tensor:
        raise NotImplementedError(f"Please implement the '_get_rewards' method for {self.__class__.__name__}.")

    def _get_dones(self) -> tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError(f"Please implement the '_get_dones' method for {self.__class__.__name__}.")
