"""GPU: the generic ``DirectRLEnv`` (direct_rl_env.py:292-379, 556-580): order of the hooks, counters, resets, and the action /
observation noise models on a small task whose hooks are written with the package's device functions."""

import pytest
import torch

from conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu


def test_direct_env_sequence_and_noise_models():
    from isaaclab_amd.assets import Articulation, ArticulationData
    from isaaclab_amd.envs.direct_rl_env import DirectRLEnv, DirectRLEnvCfg
    from isaaclab_amd.scene import InteractiveScene
    from isaaclab_amd.utils import noise as N

    dev = torch.device("cuda:0")
    n = 64
    data = ArticulationData(n, 2, 1, dev, joint_names=["a", "b"], body_names=["base"])
    scene = InteractiveScene(n, dev)
    scene.articulations["robot"] = Articulation(data)
    calls = []

    class Task(DirectRLEnv):
        def _pre_physics_step(self, actions):
            calls.append("pre")
            self.actions = actions.clone()

        def _apply_action(self):
            calls.append("apply")
            self.scene["robot"].set_joint_effort_target(self.actions)

        def _get_dones(self):
            calls.append("dones")
            return data.joint_pos[:, 0] > 0.5, self.episode_length_buf >= self.max_episode_length

        def _get_rewards(self):
            calls.append("rewards")
            return -data.joint_pos[:, 0].abs()

        def _get_observations(self):
            calls.append("obs")
            return {"policy": torch.cat([data.joint_pos, data.joint_vel], dim=1)}

        def _reset_idx(self, env_ids):
            calls.append("reset")
            super()._reset_idx(env_ids)
            data.joint_pos[env_ids] = 0.0

    cfg = DirectRLEnvCfg(decimation=2, sim_dt=0.01, episode_length_s=0.06, seed=5,
                         action_noise_model=N.NoiseModelCfg(noise_cfg=N.ConstantNoiseCfg(bias=0.25)),
                         observation_noise_model=N.NoiseModelWithAdditiveBiasCfg(noise_cfg=N.ConstantNoiseCfg(bias=0.0),
                                                                                 bias_noise_cfg=N.ConstantNoiseCfg(bias=0.5, operation="abs")))
    g = torch.Generator().manual_seed(0)

    def physics(env):
        calls.append("physics")
        data.joint_pos = torch.rand(n, 2, generator=g).to(dev)
        data.joint_vel = torch.randn(n, 2, generator=g).to(dev)

    env = Task(cfg, scene, physics, dev)
    assert env.max_episode_length == 3 and env.step_dt == pytest.approx(0.02)
    obs, _ = env.reset()
    calls.clear()
    raw = torch.rand(n, 2, generator=g)
    obs, rew, terminated, time_outs, _ = env.step(raw)
    assert calls[:7] == ["pre", "apply", "physics", "apply", "physics", "dones", "rewards"] and calls[-1] == "obs"
    assert_exact(env.actions, raw + 0.25, "action noise model")
    assert_exact(data.joint_effort_target, raw + 0.25, "targets from the noisy action")
    assert bool(terminated.any()) and "reset" in calls
    assert_exact(data.joint_pos[terminated][:, 0], torch.zeros(int(terminated.sum())), "the task's reset hook ran for the done envs")
    assert_exact(env.episode_length_buf, (~terminated).long().cpu(), "episode counters")
    # the observation noise model: bias 0.5 (drawn at reset with the 'abs' constant noise) on every column
    want = torch.cat([data.joint_pos, data.joint_vel], dim=1) + 0.5
    assert_close(obs["policy"], want, atol=1e-6, what="observation with the model's bias")
    for _ in range(2):
        obs, rew, terminated, time_outs, _ = env.step(raw)
    assert bool(time_outs.any())                # three steps of 0.02 s reach the 0.06 s episode
    assert env.common_step_counter == 3 and env._sim_step_counter == 6
    with pytest.raises(NotImplementedError, match="_pre_physics_step"):
        DirectRLEnv(DirectRLEnvCfg(), scene, None, dev).step(raw)
