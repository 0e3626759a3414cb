"""``ManagerBasedRLEnv`` (interface of ``isaaclab/envs/manager_based_env.py`` + ``manager_based_rl_env.py``): the reference's
step and reset sequences on this package's manager classes.

The simulator is the caller's: ``physics_step(env)`` stands where the reference calls ``sim.step()`` and stores the next
state in the scene's entities (``data.set_sim_state``, the sensors' ``set_*`` hand-overs).  Everything around it -- action
terms, actuator models, sensor updates, terminations, rewards, resets, curriculum, commands, events, observations -- runs
in the order of ``manager_based_rl_env.py:152-241`` and ``:315-372``, each manager on its HIP kernels.
(:class:`isaaclab_amd.envs.locomotion_step.LocomotionVelocityStep` is the same step with the five managers compiled into
one launch.)
"""

from __future__ import annotations

import math
from collections.abc import Callable, Sequence
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from ..managers import ActionManager, CommandManager, EventManager, ObservationManager, RewardManager, TerminationManager
from .mdp.curriculums import CurriculumManager
from ..utils import math as math_utils
from ..utils import rng


@dataclass
class ManagerBasedRLEnvCfg:
    """manager_based_env_cfg.py + manager_based_rl_env_cfg.py: the fields the step reads."""

    decimation: int = 1
    sim_dt: float = 1.0 / 60.0            # sim.dt
    episode_length_s: float = 10.0
    is_finite_horizon: bool = False
    seed: int | None = None
    observations: object = None
    actions: object = None
    events: object = None
    rewards: object = None
    terminations: object = None
    curriculum: object = None
    commands: object = None


class ManagerBasedRLEnv:
    def __init__(self, cfg: ManagerBasedRLEnvCfg, scene, physics_step: Callable | None = None, device=None):
        self.cfg, self.scene = cfg, scene
        self.num_envs = scene.num_envs
        self.device = torch.device(device if device is not None else scene.device)
        self._physics_step = physics_step
        self.sim = SimpleNamespace(get_physics_dt=lambda: self.cfg.sim_dt, device=self.device)
        self.seed = self.seed_all(cfg.seed) if cfg.seed is not None else 0
        self._sim_step_counter = 0
        self.common_step_counter = 0
        self.episode_length_buf = torch.zeros(self.num_envs, dtype=torch.long, device=self.device)
        self.extras: dict = {}
        self.load_managers()
        if "startup" in self.event_manager.available_modes:      # manager_based_env.py:156-158
            self.event_manager.apply(mode="startup")

    # -- properties (manager_based_env.py:180-215, manager_based_rl_env.py:120-131)
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
        """manager_based_env.py:412-430: torch's generators and every device-side generator (events, commands,
        observation noise, samplers, terrain curriculum): one seed, and their shared draw counter restarts."""
        torch.manual_seed(seed)
        return rng.seed_all(seed)

    def load_managers(self):
        """manager_based_rl_env.py:137-160: commands first (terms of the other managers read them), then actions,
        observations and events, then terminations, rewards, curriculum."""
        c = self.cfg
        self.command_manager = CommandManager(c.commands if c.commands is not None else {}, self)
        self.action_manager = ActionManager(c.actions, self)
        self.observation_manager = ObservationManager(c.observations, self)
        self.event_manager = EventManager(c.events if c.events is not None else {}, self)
        self.termination_manager = TerminationManager(c.terminations, self)
        self.reward_manager = RewardManager(c.rewards, self)
        self.curriculum_manager = CurriculumManager(c.curriculum if c.curriculum is not None else {}, self)

    # -- operations
    def reset(self, seed: int | None = None, env_ids: Sequence[int] | None = None):
        """manager_based_env.py:262-296."""
        if env_ids is None:
            env_ids = torch.arange(self.num_envs, dtype=torch.int64, device=self.device)
        if seed is not None:
            self.seed = self.seed_all(seed)
        self._reset_idx(env_ids)
        self._write_and_update()
        self.obs_buf = self.observation_manager.compute()
        return self.obs_buf, self.extras

    def step(self, action: torch.Tensor):
        """manager_based_rl_env.py:152-241."""
        self.action_manager.process_action(action.to(self.device))
        for _ in range(self.cfg.decimation):
            self._sim_step_counter += 1
            self.action_manager.apply_action()
            self._write_data_to_sim()
            if self._physics_step is not None:
                self._physics_step(self)
            self.scene.update(dt=self.physics_dt)
        self.episode_length_buf += 1
        self.common_step_counter += 1
        self.reset_buf = self.termination_manager.compute()
        self.reset_terminated = self.termination_manager.terminated
        self.reset_time_outs = self.termination_manager.time_outs
        self.reward_buf = self.reward_manager.compute(dt=self.step_dt)
        reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        if len(reset_env_ids) > 0:
            self._reset_idx(reset_env_ids)
            self._write_and_update()
        self.command_manager.compute(dt=self.step_dt)
        if "interval" in self.event_manager.available_modes:
            self.event_manager.apply(mode="interval", dt=self.step_dt)
        self.obs_buf = self.observation_manager.compute()
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    # -- helpers
    def _write_data_to_sim(self):
        """interactive_scene.py:470-482."""
        self.scene.write_data_to_sim()

    def _write_and_update(self):
        """manager_based_rl_env.py:211-214 / manager_based_env.py:284-286: what follows a reset."""
        self._write_data_to_sim()

    def _reset_idx(self, env_ids):
        """manager_based_rl_env.py:315-372."""
        self.curriculum_manager.compute(env_ids=env_ids)
        self.scene.reset(env_ids)
        if "reset" in self.event_manager.available_modes:
            self.event_manager.apply(mode="reset", env_ids=env_ids, global_env_step_count=self._sim_step_counter // self.cfg.decimation)
        log = self.extras["log"] = dict()
        log.update(self.observation_manager.reset(env_ids))
        log.update(self.action_manager.reset(env_ids))
        log.update(self.reward_manager.reset(env_ids))
        log.update(self.curriculum_manager.reset(env_ids))
        log.update(self.command_manager.reset(env_ids))
        log.update(self.event_manager.reset(env_ids))
        log.update(self.termination_manager.reset(env_ids))
        self.episode_length_buf[env_ids] = 0
