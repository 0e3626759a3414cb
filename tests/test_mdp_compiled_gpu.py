"""GPU: the fused MDP kernels compiled for the term lists of the reference's task configurations against the
plan-interpreting kernel on the same plan and the same inputs -- every output bit for bit over three steps with resets
(the logging sums: the same addends in another order, so to 1e-5).  The plan-interpreting kernel itself is held to the
reference managers' fixtures in test_mdp_gpu.py, and both kernels run the end-to-end steps of test_step_gpu.py /
test_config_steps_gpu.py."""

import math

import pytest
import torch

from conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu


def _config(name):
    from isaaclab_amd.tasks import cartpole as CP
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.tasks import reach as RE

    if name == "anymal_c_rough":
        return dict(dims=dict(num_joints=12, num_bodies=17, num_sensor_bodies=17, history_len=3, num_rays=187, action_dim=12, command_dim=3),
                    terms=LV.anymal_c_rough_terms(), command=LV.VELOCITY_COMMAND, step_dt=LV.STEP_DT, max_len=LV.MAX_EPISODE_LENGTH)
    if name == "h1_rough":
        return dict(dims=dict(num_joints=19, num_bodies=20, num_sensor_bodies=20, history_len=3, num_rays=187, action_dim=19, command_dim=3),
                    terms=LV.h1_rough_terms(), command=LV.H1_VELOCITY_COMMAND, step_dt=LV.STEP_DT, max_len=LV.MAX_EPISODE_LENGTH)
    if name == "cartpole":
        return dict(dims=dict(num_joints=2, num_bodies=3, action_dim=1, command_dim=0), terms=CP.cartpole_terms(), command=None,
                    step_dt=CP.STEP_DT, max_len=CP.MAX_EPISODE_LENGTH)
    joints, bodies, action = {"reach_j9_a7": (9, 11, 7), "reach_j9_a6": (9, 11, 6), "reach_j8_a7": (8, 10, 7), "reach_j8_a6": (8, 10, 6)}[name]
    return dict(dims=dict(num_joints=joints, num_bodies=bodies, action_dim=action, command_dim=7), terms=RE.reach_terms(),
                command=dict(RE.POSE_COMMAND, body_index=8), step_dt=RE.STEP_DT, max_len=RE.MAX_EPISODE_LENGTH)


def _inputs(cfg, n, seed):
    """Seeded state of every array a term of the five configurations reads; some envs are about to time out, some stand
    on their base."""
    g = torch.Generator().manual_seed(seed)
    d = cfg["dims"]
    j, b, cb = d["num_joints"], d.get("num_bodies", 1), d.get("num_sensor_bodies", 0)
    h, r, a, cd = d.get("history_len", 0), d.get("num_rays", 0), d["action_dim"], d["command_dim"]
    rn = lambda *s: torch.randn(*s, generator=g)
    ru = lambda *s: torch.rand(*s, generator=g)
    quat = rn(n, 4)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    t = dict(root_state_w=torch.cat([rn(n, 3), quat, rn(n, 6)], -1), joint_pos=2.5 * rn(n, j), joint_vel=2 * rn(n, j), joint_acc=50 * rn(n, j),
             applied_torque=20 * rn(n, j), computed_torque=20 * rn(n, j), default_joint_pos=0.3 * rn(n, j), default_joint_vel=torch.zeros(n, j),
             soft_joint_pos_limits=torch.stack([-1.5 + 0.2 * rn(n, j), 1.5 + 0.2 * rn(n, j)], -1), soft_joint_vel_limits=7.5 + ru(n, j),
             body_state_w=rn(n, b, 13), action=2 * ru(n, a) - 1, prev_action=2 * ru(n, a) - 1)
    t["body_state_w"][..., 3:7] /= t["body_state_w"][..., 3:7].norm(dim=-1, keepdim=True)
    if cb:
        forces = rn(n, h, cb, 3) * (ru(n, h, cb, 1) < 0.3) * 30.0
        t.update(net_forces_w_history=forces, current_air_time=ru(n, cb) * (ru(n, cb) < 0.5), last_air_time=ru(n, cb),
                 current_contact_time=ru(n, cb) * 0.03 * (ru(n, cb) < 0.7), last_contact_time=ru(n, cb))
    if r:
        t.update(height_scan=2 * rn(n, r), ray_hits_w=rn(n, r, 3), sensor_pos_w=rn(n, 3))
    state = dict(episode_length_buf=torch.randint(0, cfg["max_len"] - 3, (n,), generator=g), episode_sums=None)
    state["episode_length_buf"][::7] = cfg["max_len"] - 2     # time out on the second step
    if cfg["command"] is not None:
        cmd = rn(n, cd)
        if cd == 7:
            cmd[:, 3:] /= cmd[:, 3:].norm(dim=-1, keepdim=True)
        state.update(command=cmd, time_left=ru(n) * 0.05, command_counter=torch.randint(0, 3, (n,), generator=g),
                     heading_target=3 * rn(n), is_heading_env=ru(n) < 0.8, is_standing_env=ru(n) < 0.1, command_metrics=ru(2, n))
    return t, state, g


def _build(cfg, n, dev, tensors, state, compiled):
    from isaaclab_amd.envs import fused_mdp as FM

    d = cfg["dims"]
    m = FM.FusedMdp(n, dev, step_dt=cfg["step_dt"], max_episode_length=cfg["max_len"], command=cfg["command"], seed=5, **d, **cfg["terms"])
    m.select_kernel(compiled)
    m.bind(**{k: v.to(dev).contiguous().clone() for k, v in tensors.items()})
    for k, v in state.items():
        if v is not None:
            getattr(m, k).copy_(v)
    m.episode_sums.copy_(torch.arange(m.episode_sums.numel(), dtype=torch.float32).reshape(m.episode_sums.shape) * 1e-3)
    m.step_reward.fill_(-7.0)   # entries of zero-weight terms must keep their values
    return m


OUTPUTS = ("obs", "reward_buf", "terminated", "time_outs", "reset_buf", "term_dones", "episode_sums", "step_reward", "episode_length_buf",
           "action", "prev_action")
COMMAND_OUTPUTS = ("command", "time_left", "command_counter", "command_metrics", "heading_target", "is_heading_env", "is_standing_env")


@pytest.mark.parametrize("name, n", [("anymal_c_rough", 1000), ("anymal_c_rough", 37), ("anymal_c_rough", 4096), ("h1_rough", 1000),
                                     ("h1_rough", 130), ("cartpole", 1000), ("cartpole", 5), ("reach_j9_a7", 1000), ("reach_j9_a6", 333),
                                     ("reach_j8_a7", 64), ("reach_j8_a6", 257)])
@pytest.mark.parametrize("inject", [False, True])
def test_compiled_kernel_equals_plan_interpreting_kernel(dev, name, n, inject):
    cfg = _config(name)
    tensors, state, g = _inputs(cfg, n, seed=sum(map(ord, name)) + n)
    fast, slow = (_build(cfg, n, dev, tensors, state, compiled) for compiled in (True, False))
    assert fast.kernel == name, f"the plan of {name} is served by '{fast.kernel}'"
    assert slow.kernel == "generic"
    total_done = 0
    for step in range(3):
        if inject:   # the reference's draws handed in, as the parity tests do
            noise = torch.rand(n, fast.obs_dim, generator=g).to(dev)
            samples, reset_samples = torch.rand(n, 7, generator=g).to(dev), torch.rand(n, 7, generator=g).to(dev)
            for m in (fast, slow):
                m.bind(noise_u=noise, command_samples=samples, reset_command_samples=reset_samples)
        if step == 1 and "net_forces_w_history" in tensors:   # new contacts, the same for both
            f = (torch.randn(tensors["net_forces_w_history"].shape, generator=g) * 40.0).to(dev)
            fast.net_forces_w_history.copy_(f), slow.net_forces_w_history.copy_(f)
        if step == 2:   # a new action on top of the restarted histories
            a = (2 * torch.rand(fast.action.shape, generator=g) - 1).to(dev)
            for m in (fast, slow):
                m.prev_action.copy_(m.action)
                m.action.copy_(a)
        for m in (fast, slow):
            m.log_sums.zero_()
            m.step()
        torch.cuda.synchronize()
        for key in OUTPUTS + (COMMAND_OUTPUTS if cfg["command"] is not None else ()):
            assert_exact(getattr(fast, key), getattr(slow, key).cpu(), f"{name} n={n} step {step}: {key}")
        assert_close(fast.log_sums, slow.log_sums.cpu(), rtol=1e-5, atol=1e-5, what=f"{name} step {step}: log_sums")
        total_done += int(slow.reset_buf.sum())
    assert total_done > 0 or n < 8, "no env was reset: the reset path was not exercised"


def test_other_term_lists_and_partial_steps_stay_with_the_plan_interpreting_kernel(dev):
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.tasks import cartpole as CP

    terms = CP.cartpole_terms()
    terms["rewards"] = terms["rewards"][:-1]   # one term fewer: no compiled kernel
    m = FM.FusedMdp(8, dev, num_joints=2, num_bodies=3, action_dim=1, command_dim=0, step_dt=CP.STEP_DT,
                    max_episode_length=CP.MAX_EPISODE_LENGTH, command=None, **terms)
    assert m.kernel == "generic"
    m = FM.FusedMdp(8, dev, num_joints=3, num_bodies=3, action_dim=1, command_dim=0, step_dt=CP.STEP_DT,
                    max_episode_length=CP.MAX_EPISODE_LENGTH, command=None, **CP.cartpole_terms())
    assert m.kernel == "generic"   # another joint count
