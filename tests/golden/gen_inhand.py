"""Golden vectors for the in-hand manipulation direct env (Shadow hand cfg), produced by the REFERENCE's own
module-level functions and ``InHandManipulationEnv`` methods called on a stand-in ``self``.
Runs only where /root/reference is mounted; writes tests/golden/inhand.npz."""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refload as R  # noqa: E402

from isaaclab_amd import synth  # noqa: E402

R.bootstrap()
for name in ("isaaclab.sim", "isaaclab.sim.spawners", "isaaclab.sim.spawners.from_files", "isaaclab.markers", "isaaclab.envs",
             "isaaclab.assets"):
    R.placeholder(name)
env_mod = R.load("isaaclab_tasks.direct.inhand_manipulation.inhand_manipulation_env",
                 f"{R.T}/direct/inhand_manipulation/inhand_manipulation_env.py")
Env = env_mod.InHandManipulationEnv
out = {}


def put(k, v):
    out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def run(tag, n, seed, moving_average, max_consecutive_success):
    s = synth.inhand_state(n, seed)
    for k, v in s.items():
        put(f"{tag}_in_{k}", v if torch.is_tensor(v) else torch.tensor(v))
    cfg = types.SimpleNamespace(dist_reward_scale=-10.0, rot_reward_scale=1.0, rot_eps=0.1, action_penalty_scale=-0.0002,
                                reach_goal_bonus=250, fall_penalty=0, fall_dist=0.24, vel_obs_scale=0.2, success_tolerance=0.1,
                                max_consecutive_success=max_consecutive_success, av_factor=0.1, act_moving_average=moving_average,
                                asymmetric_obs=False, obs_type="full")
    targets = {}

    class Hand:
        data = types.SimpleNamespace(body_pos_w=s["body_state"][..., 0:3].clone(), body_quat_w=s["body_state"][..., 3:7].clone(),
                                     body_vel_w=s["body_state"][..., 7:13].clone(), joint_pos=s["joint_pos"], joint_vel=s["joint_vel"])

        def set_joint_position_target(self, t, joint_ids=None):
            targets["last"] = t.clone()

    obj = types.SimpleNamespace(data=types.SimpleNamespace(
        root_pos_w=s["object_state"][:, 0:3], root_quat_w=s["object_state"][:, 3:7], root_vel_w=s["object_state"][:, 7:13],
        root_lin_vel_w=s["object_state"][:, 7:10], root_ang_vel_w=s["object_state"][:, 10:13]))
    me = types.SimpleNamespace(
        cfg=cfg, num_envs=n, device="cpu", hand=Hand(), object=obj, scene=types.SimpleNamespace(env_origins=s["env_origins"]),
        actuated_dof_indices=s["actuated"], finger_bodies=s["fingers"], num_fingertips=len(s["fingers"]),
        hand_dof_lower_limits=s["lower"], hand_dof_upper_limits=s["upper"], cur_targets=s["cur_targets"].clone(),
        prev_targets=s["prev_targets"].clone(), in_hand_pos=s["in_hand_pos"], goal_rot=s["goal_rot"].clone(),
        successes=s["successes"].clone(), consecutive_successes=s["consecutive_successes"].clone(),
        reset_goal_buf=s["reset_goal_buf"].clone(), episode_length_buf=s["episode_length_buf"].clone(), max_episode_length=600,
        actions=s["actions"].clone(), extras={}, goal_pos=torch.zeros(n, 3),
        goal_markers=types.SimpleNamespace(visualize=lambda *a, **k: None),
        x_unit_tensor=torch.tensor([1.0, 0, 0]).repeat(n, 1), y_unit_tensor=torch.tensor([0, 1.0, 0]).repeat(n, 1))
    me._compute_intermediate_values = lambda: Env._compute_intermediate_values(me)
    me._reset_target_pose = lambda ids: Env._reset_target_pose(me, ids)
    me.compute_full_observations = lambda: Env.compute_full_observations(me)
    g = torch.Generator().manual_seed(seed + 1)
    goal_samples = torch.rand(n, 2, generator=g) * 2 - 1
    reset_goal_samples = torch.rand(n, 2, generator=g) * 2 - 1
    put(f"{tag}_goal_samples", goal_samples)
    put(f"{tag}_reset_goal_samples", reset_goal_samples)
    feed = {"next": None}
    env_mod.sample_uniform = lambda lo, hi, size, device=None: feed["next"]
    # ---- DirectRLEnv.step order (direct_rl_env.py:330-386), simulator calls left out
    for _ in range(2):
        Env._apply_action(me)
    put(f"{tag}_cur_targets", me.cur_targets)
    put(f"{tag}_prev_targets", me.prev_targets)
    put(f"{tag}_joint_target", targets["last"])
    me.episode_length_buf = me.episode_length_buf + 1
    terminated, time_out = Env._get_dones(me)
    me.reset_buf = terminated | time_out
    put(f"{tag}_terminated", terminated)
    put(f"{tag}_time_outs", time_out)
    goal_ids_before = None

    def reset_target_pose(ids):
        feed["next"] = (goal_samples if feed["phase"] == "goal" else reset_goal_samples)[ids]
        Env._reset_target_pose(me, ids)

    me._reset_target_pose = reset_target_pose
    feed["phase"] = "goal"
    reward = Env._get_rewards(me)
    put(f"{tag}_reward", reward)
    put(f"{tag}_successes_after_reward", me.successes)
    put(f"{tag}_consecutive_successes", me.consecutive_successes)
    put(f"{tag}_log_consecutive_successes", me.extras["log"]["consecutive_successes"])
    feed["phase"] = "reset"
    ids = me.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    if len(ids) > 0:   # the parts of _reset_idx that do not write into the simulator (:200-247, direct_rl_env.py:_reset_idx)
        me.episode_length_buf[ids] = 0
        reset_target_pose(ids)
        me.successes[ids] = 0
    Env._compute_intermediate_values(me)
    put(f"{tag}_goal_rot", me.goal_rot)
    put(f"{tag}_reset_goal_buf", me.reset_goal_buf)
    put(f"{tag}_successes", me.successes)
    put(f"{tag}_episode_length_buf", me.episode_length_buf)
    put(f"{tag}_obs", Env._get_observations(me)["policy"])
    print(tag, "resets", int(me.reset_buf.sum()), "fallen", int(terminated.sum()), "goal resets", int((reward > 100).sum()))


run("shadow", 512, 5001, 1.0, 0)
run("shadow_ema", 512, 5002, 0.8, 50)
np.savez_compressed(os.path.join(HERE, "inhand.npz"), **out)
print("wrote inhand.npz with", len(out), "arrays")
