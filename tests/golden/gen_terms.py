"""Golden fixtures for the MDP term functions, the velocity / pose command terms and the managers, produced
by calling the reference's own functions and manager classes on a duck-typed environment."""

from __future__ import annotations

import math
import sys
import types

import torch

import _refload as R
from gen_golden import gen, save

SECTIONS = {}
NS = types.SimpleNamespace


def section(fn):
    SECTIONS[fn.__name__] = fn
    return fn


def rand_quat(n, g, tilt=0.15):
    yaw = (torch.rand(n, generator=g) * 2 - 1) * math.pi
    rp = torch.randn(n, 2, generator=g) * tilt
    m = R.bootstrap()["math"]
    return m.quat_from_euler_xyz(rp[:, 0], rp[:, 1], yaw)


def make_state(g, n, j, b, h=3, a=None):
    """Synthetic articulation + contact state with values on both sides of every threshold."""
    m = R.bootstrap()["math"]
    a = j if a is None else a
    s = NS()
    s.root_pos_w = torch.randn(n, 3, generator=g) * 3.0 + torch.tensor([0.0, 0.0, 0.6])
    s.root_quat_w = rand_quat(n, g)
    s.root_lin_vel_w = torch.randn(n, 3, generator=g) * 0.5
    s.root_ang_vel_w = torch.randn(n, 3, generator=g) * 0.5
    s.root_state_w = torch.cat([s.root_pos_w, s.root_quat_w, s.root_lin_vel_w, s.root_ang_vel_w], dim=-1)
    s.root_lin_vel_b = m.quat_rotate_inverse(s.root_quat_w, s.root_lin_vel_w)
    s.root_ang_vel_b = m.quat_rotate_inverse(s.root_quat_w, s.root_ang_vel_w)
    grav = torch.tensor([0.0, 0.0, -1.0]).repeat(n, 1)
    s.projected_gravity_b = m.quat_rotate_inverse(s.root_quat_w, grav)
    fwd = m.quat_apply(s.root_quat_w, torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1))
    s.heading_w = torch.atan2(fwd[:, 1], fwd[:, 0])
    s.default_joint_pos = (torch.randn(j, generator=g) * 0.4).repeat(n, 1)
    s.default_joint_vel = torch.zeros(n, j)
    s.joint_pos = s.default_joint_pos + torch.randn(n, j, generator=g) * 0.4
    s.joint_vel = torch.randn(n, j, generator=g) * 2.0
    s.joint_acc = torch.randn(n, j, generator=g) * 40.0
    s.computed_torque = torch.randn(n, j, generator=g) * 40.0
    s.applied_torque = s.computed_torque.clip(-60.0, 60.0)
    lim = torch.stack([s.default_joint_pos - 0.7, s.default_joint_pos + 0.7], dim=-1)
    s.soft_joint_pos_limits = lim
    s.soft_joint_vel_limits = torch.full((n, j), 3.0)
    s.body_state_w = torch.randn(n, b, 13, generator=g)
    s.body_pos_w, s.body_quat_w = s.body_state_w[..., :3], s.body_state_w[..., 3:7]
    s.body_lin_vel_w = s.body_state_w[..., 7:10]
    s.body_lin_acc_w = torch.randn(n, b, 3, generator=g) * 5.0
    # contact history: half the entries zero, magnitudes around the 1 N threshold and far above it
    on = torch.rand(n, h, b, generator=g) < 0.5
    mag = torch.where(torch.rand(n, h, b, generator=g) < 0.3, torch.rand(n, h, b, generator=g) * 2.0,
                      torch.rand(n, h, b, generator=g) * 60.0)
    f = torch.randn(n, h, b, 3, generator=g)
    s.net_forces_w_history = f / f.norm(dim=-1, keepdim=True) * (mag * on).unsqueeze(-1)
    s.net_forces_w_history[0, 0, 0] = torch.tensor([0.6, 0.8, 0.0])      # |F| == 1.0 exactly: not > threshold
    s.net_forces_w_history[0, 1:, 0] = 0.0
    s.net_forces_w_history[1, 0, 0] = torch.tensor([0.6, 0.8, 1e-3])
    s.current_contact_time = torch.where(torch.rand(n, b, generator=g) < 0.5, torch.zeros(n, b),
                                         torch.round(torch.rand(n, b, generator=g) * 8) * 0.005)
    s.current_contact_time[2, :] = 0.02          # exactly dt: inside the first-contact window (dt + 1e-8)
    s.current_contact_time[3, :] = 0.025
    s.current_air_time = torch.where(s.current_contact_time > 0, torch.zeros(n, b), torch.rand(n, b, generator=g) * 0.6)
    s.last_air_time = torch.rand(n, b, generator=g) * 0.8
    s.last_contact_time = torch.rand(n, b, generator=g) * 0.8
    s.action = torch.rand(n, a, generator=g) * 2 - 1
    s.prev_action = torch.rand(n, a, generator=g) * 2 - 1
    s.command = torch.rand(n, 3, generator=g) * 2 - 1
    s.command[:8] *= 0.05                         # small commands: feet_air_time gate (|cmd_xy| > 0.1)
    s.command[8] = torch.tensor([0.06, 0.08, 0.3])  # |cmd_xy| == 0.1 up to rounding
    s.episode_length_buf = torch.randint(0, 1003, (n,), generator=g)
    s.episode_length_buf[:4] = torch.tensor([999, 1000, 1001, 0])
    return s


def make_env(s, n, step_dt=0.02, max_len=1000, sensor_mod=None):
    data = s
    robot = NS(data=data)
    sensor = None
    if sensor_mod is not None:
        sensor = object.__new__(sensor_mod.ContactSensor)
        sensor._initialize_handle = sensor._invalidate_initialize_handle = sensor._debug_vis_handle = None
        sensor.cfg = NS(track_air_time=True, history_length=3)
        sensor._data = NS(net_forces_w_history=s.net_forces_w_history, current_contact_time=s.current_contact_time,
                          current_air_time=s.current_air_time, last_air_time=s.last_air_time,
                          last_contact_time=s.last_contact_time)
        sensor._is_outdated = torch.zeros(n, dtype=torch.bool)
        sensor._update_outdated_buffers = lambda: None

    class Scene(dict):
        pass

    scene = Scene(robot=robot)
    scene.sensors = {"contact_forces": sensor}
    scene.env_origins = torch.zeros(n, 3)
    env = NS(num_envs=n, device="cpu", step_dt=step_dt, max_episode_length=max_len, max_episode_length_s=max_len * step_dt,
             episode_length_buf=s.episode_length_buf, scene=scene,
             command_manager=NS(get_command=lambda name: s.command),
             action_manager=NS(action=s.action, prev_action=s.prev_action))
    return env


def _sensor_mod():
    import gen_sensors

    return gen_sensors._load_sensor_modules()["cs"]


STATE_KEYS = ["root_state_w", "root_lin_vel_b", "root_ang_vel_b", "projected_gravity_b", "heading_w", "joint_pos",
              "joint_vel", "joint_acc", "computed_torque", "applied_torque", "default_joint_pos", "default_joint_vel",
              "soft_joint_pos_limits", "soft_joint_vel_limits", "body_state_w", "body_lin_acc_w", "net_forces_w_history",
              "current_contact_time", "current_air_time", "last_air_time", "last_contact_time", "action", "prev_action",
              "command", "episode_length_buf"]


def dump_state(s, prefix=""):
    return {prefix + k: getattr(s, k) for k in STATE_KEYS}


@section
def terms_locomotion():
    """All reward / termination / observation terms used by the rough-terrain velocity tasks
    (ANYmal-C: J=12, B=17; H1: J=19, B=20) plus the remaining library terms of the three mdp files."""
    m = R.bootstrap()
    rew, term, obs, vel = m["rewards"], m["terminations"], m["observations"], m["vel_rewards"]
    SE = R.SceneEntity
    cs = _sensor_mod()
    out = {}
    for tag, seed, n, j, b, feet, thigh, base, some_j in (
        ("anymal", 6001, 256, 12, 17, [13, 14, 15, 16], [5, 6, 7, 8], [0], [0, 3, 7]),
        ("h1", 6002, 192, 19, 20, [18, 19], [3, 4], [2], [2, 5, 11, 18]),
    ):
        g = gen(seed)
        s = make_state(g, n, j, b)
        env = make_env(s, n, sensor_mod=cs)
        robot = SE("robot")
        o = {}
        o["lin_vel_z_l2"] = rew.lin_vel_z_l2(env, robot)
        o["ang_vel_xy_l2"] = rew.ang_vel_xy_l2(env, robot)
        o["flat_orientation_l2"] = rew.flat_orientation_l2(env, robot)
        o["base_height_l2"] = rew.base_height_l2(env, 0.55, robot)
        o["body_lin_acc_l2"] = rew.body_lin_acc_l2(env, SE("robot", body_ids=feet))
        o["joint_torques_l2"] = rew.joint_torques_l2(env, robot)
        o["joint_torques_l2_ids"] = rew.joint_torques_l2(env, SE("robot", joint_ids=some_j))
        o["joint_vel_l1"] = rew.joint_vel_l1(env, robot)
        o["joint_vel_l2"] = rew.joint_vel_l2(env, robot)
        o["joint_acc_l2"] = rew.joint_acc_l2(env, robot)
        o["joint_deviation_l1"] = rew.joint_deviation_l1(env, SE("robot", joint_ids=some_j))
        o["joint_pos_limits"] = rew.joint_pos_limits(env, robot)
        o["joint_vel_limits"] = rew.joint_vel_limits(env, 0.9, robot)
        o["applied_torque_limits"] = rew.applied_torque_limits(env, robot)
        o["action_rate_l2"] = rew.action_rate_l2(env)
        o["action_l2"] = rew.action_l2(env)
        o["undesired_contacts"] = rew.undesired_contacts(env, 1.0, SE("contact_forces", body_ids=thigh))
        o["contact_forces"] = rew.contact_forces(env, 5.0, SE("contact_forces", body_ids=feet))
        o["track_lin_vel_xy_exp"] = rew.track_lin_vel_xy_exp(env, math.sqrt(0.25), "base_velocity", robot)
        o["track_ang_vel_z_exp"] = rew.track_ang_vel_z_exp(env, math.sqrt(0.25), "base_velocity", robot)
        o["feet_air_time"] = vel.feet_air_time(env, "base_velocity", SE("contact_forces", body_ids=feet), 0.5)
        o["feet_air_time_positive_biped"] = vel.feet_air_time_positive_biped(env, "base_velocity", 0.4,
                                                                             SE("contact_forces", body_ids=feet))
        o["feet_slide"] = vel.feet_slide(env, SE("contact_forces", body_ids=feet), SE("robot", body_ids=feet))
        o["track_lin_vel_xy_yaw_frame_exp"] = vel.track_lin_vel_xy_yaw_frame_exp(env, 0.5, "base_velocity", robot)
        o["track_ang_vel_z_world_exp"] = vel.track_ang_vel_z_world_exp(env, "base_velocity", 0.5, robot)
        # terminations
        o["time_out"] = term.time_out(env)
        o["illegal_contact"] = term.illegal_contact(env, 1.0, SE("contact_forces", body_ids=base))
        o["bad_orientation"] = term.bad_orientation(env, 0.3, robot)
        o["root_height_below_minimum"] = term.root_height_below_minimum(env, 0.5, robot)
        o["joint_pos_out_of_manual_limit"] = term.joint_pos_out_of_manual_limit(env, (-0.9, 0.9), SE("robot", joint_ids=some_j))
        o["joint_vel_out_of_limit"] = term.joint_vel_out_of_limit(env, robot)
        o["joint_vel_out_of_manual_limit"] = term.joint_vel_out_of_manual_limit(env, 4.0, robot)
        o["joint_effort_out_of_limit"] = term.joint_effort_out_of_limit(env, robot)
        # observations
        o["obs_joint_pos_rel"] = obs.joint_pos_rel(env, robot)
        o["obs_joint_vel_rel"] = obs.joint_vel_rel(env, robot)
        o["obs_joint_pos_limit_normalized"] = obs.joint_pos_limit_normalized(env, robot)
        o["obs_last_action"] = obs.last_action(env)
        out.update(dump_state(s, tag + "_"))
        out.update({f"{tag}_{k}": v for k, v in o.items()})
        out[f"{tag}_ids_feet"] = torch.tensor(feet)
        out[f"{tag}_ids_thigh"] = torch.tensor(thigh)
        out[f"{tag}_ids_base"] = torch.tensor(base)
        out[f"{tag}_ids_some_j"] = torch.tensor(some_j)
    # quirk on record: joint_pos_out_of_limit indexes a 1-D result with two indices -> IndexError in the reference
    try:
        term.joint_pos_out_of_limit(env, robot)
        out["joint_pos_out_of_limit_raises"] = torch.tensor(0)
    except IndexError:
        out["joint_pos_out_of_limit_raises"] = torch.tensor(1)
    save("terms_locomotion", **out)


@section
def terms_manipulation():
    """Cartpole and Reach terms."""
    m = R.bootstrap()
    rew, term, obs = m["rewards"], m["terminations"], m["observations"]
    SE = R.SceneEntity
    out = {}
    # cartpole (cartpole_env_cfg.py:116-149): slider_to_cart = joint 0, cart_to_pole = joint 1
    g = gen(6101)
    n = 256
    s = make_state(g, n, 2, 3, a=1)
    s.joint_pos = torch.stack([(torch.rand(n, generator=g) * 2 - 1) * 3.2, (torch.rand(n, generator=g) * 2 - 1) * 7.0], -1)
    s.joint_pos[0] = torch.tensor([3.0, math.pi])
    s.joint_pos[1] = torch.tensor([-3.0, -math.pi])
    s.joint_pos[2] = torch.tensor([3.0000002, 3 * math.pi])
    s.terminated = torch.rand(n, generator=g) < 0.1
    env = make_env(s, n, step_dt=1 / 60, max_len=300)
    env.termination_manager = NS(terminated=s.terminated)
    cp = m["cartpole_rewards"]
    out.update(dump_state(s, "cartpole_"))
    out["cartpole_terminated"] = s.terminated
    out["cartpole_is_alive"] = rew.is_alive(env)
    out["cartpole_is_terminated"] = rew.is_terminated(env)
    out["cartpole_pole_pos"] = cp.joint_pos_target_l2(env, 0.0, SE("robot", joint_ids=[1]))
    out["cartpole_cart_vel"] = rew.joint_vel_l1(env, SE("robot", joint_ids=[0]))
    out["cartpole_pole_vel"] = rew.joint_vel_l1(env, SE("robot", joint_ids=[1]))
    out["cartpole_out_of_bounds"] = term.joint_pos_out_of_manual_limit(env, (-3.0, 3.0), SE("robot", joint_ids=[0]))
    out["cartpole_obs_joint_pos_rel"] = obs.joint_pos_rel(env, SE("robot"))
    out["cartpole_obs_joint_vel_rel"] = obs.joint_vel_rel(env, SE("robot"))
    # reach (reach_env_cfg.py:134-156): ee body index 8 of 11, command = pose in the root frame
    g = gen(6102)
    n = 256
    s = make_state(g, n, 9, 11, a=7)
    s.body_state_w[:, :, 3:7] = s.body_state_w[:, :, 3:7] / s.body_state_w[:, :, 3:7].norm(dim=-1, keepdim=True)
    cq = torch.randn(n, 4, generator=g)
    s.command = torch.cat([torch.rand(n, 3, generator=g) * 0.5, cq / cq.norm(dim=-1, keepdim=True)], dim=-1)
    env = make_env(s, n, step_dt=1 / 30, max_len=360)
    re = m["reach_rewards"]
    ee = SE("robot", body_ids=[8])
    out.update(dump_state(s, "reach_"))
    out["reach_position_command_error"] = re.position_command_error(env, "ee_pose", ee)
    out["reach_position_command_error_tanh"] = re.position_command_error_tanh(env, 0.1, "ee_pose", ee)
    out["reach_orientation_command_error"] = re.orientation_command_error(env, "ee_pose", ee)
    out["reach_action_rate_l2"] = rew.action_rate_l2(env)
    out["reach_joint_vel_l2"] = rew.joint_vel_l2(env, SE("robot"))
    save("terms_manipulation", **out)


@section
def velocity_command():
    """CommandTerm.compute / reset with the real UniformVelocityCommand hooks, RNG seeded so that the drawn values can
    be replayed (they are stored as the `samples` the kernels consume)."""
    m = R.bootstrap()
    R.placeholder("isaaclab.managers.action_manager")
    cm = R.load("isaaclab.managers.command_manager", f"{R.L}/managers/command_manager.py")
    sys.modules["isaaclab.managers"].CommandTerm = cm.CommandTerm
    R.placeholder("isaaclab.envs.mdp.commands")
    vc = R.load("isaaclab.envs.mdp.commands.velocity_command", f"{R.L}/envs/mdp/commands/velocity_command.py")
    g = gen(6201)
    n = 200
    ranges = NS(lin_vel_x=(-1.0, 1.0), lin_vel_y=(-1.0, 1.0), ang_vel_z=(-1.0, 1.0), heading=(-math.pi, math.pi))
    cfg = NS(resampling_time_range=(10.0, 10.0), rel_standing_envs=0.02, rel_heading_envs=1.0, heading_command=True,
             heading_control_stiffness=0.5, ranges=ranges, debug_vis=False)
    out = {}
    for tag, cfg_i in (("rough", cfg),
                       ("mixed", NS(resampling_time_range=(0.05, 0.3), rel_standing_envs=0.3, rel_heading_envs=0.5,
                                    heading_command=True, heading_control_stiffness=0.5, ranges=ranges, debug_vis=False))):
        term = object.__new__(vc.UniformVelocityCommand)
        term._debug_vis_handle = None
        term.cfg = cfg_i
        term._env = NS(step_dt=0.02, num_envs=n, device="cpu")
        s = make_state(g, n, 12, 17)
        term.robot = NS(data=s)
        term.vel_command_b = torch.rand(n, 3, generator=g) * 2 - 1
        term.heading_target = (torch.rand(n, generator=g) * 2 - 1) * math.pi
        term.is_heading_env = torch.rand(n, generator=g) < 0.7
        term.is_standing_env = torch.rand(n, generator=g) < 0.1
        term.metrics = dict(error_vel_xy=torch.rand(n, generator=g), error_vel_yaw=torch.rand(n, generator=g))
        term.time_left = torch.rand(n, generator=g) * 0.1 - 0.02   # about 40 % expire this step (dt = 0.02)
        term.command_counter = torch.randint(0, 3, (n,), generator=g)
        # `num_envs` / `device` are properties of ManagerTermBase reading self._env
        pre = dict(vel_command_b=term.vel_command_b.clone(), heading_target=term.heading_target.clone(),
                   is_heading_env=term.is_heading_env.clone(), is_standing_env=term.is_standing_env.clone(),
                   metric_xy=term.metrics["error_vel_xy"].clone(), metric_yaw=term.metrics["error_vel_yaw"].clone(),
                   time_left=term.time_left.clone(), command_counter=term.command_counter.clone(),
                   root_lin_vel_b=s.root_lin_vel_b, root_ang_vel_b=s.root_ang_vel_b, heading_w=s.heading_w)
        steps = []
        for k in range(3):
            seed = 700 + k
            # replay: which envs resample, and what do they draw (same seed, same call order)
            ids = ((term.time_left - 0.02) <= 0.0).nonzero().flatten()
            torch.manual_seed(seed)
            clone = object.__new__(vc.UniformVelocityCommand)
            clone._debug_vis_handle = None
            clone.cfg, clone._env = cfg_i, term._env
            for a in ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter"):
                setattr(clone, a, getattr(term, a).clone())
            samples = torch.zeros(n, 7)
            if len(ids) > 0:
                # capture raw uniform(0,1) draws for the two probability tests by re-drawing with the same stream
                tl = clone.time_left[ids].uniform_(*cfg_i.resampling_time_range)
                r = torch.empty(len(ids))
                vx = r.uniform_(*ranges.lin_vel_x).clone()
                vy = r.uniform_(*ranges.lin_vel_y).clone()
                wz = r.uniform_(*ranges.ang_vel_z).clone()
                hd = r.uniform_(*ranges.heading).clone()
                uh = r.uniform_(0.0, 1.0).clone()
                us = r.uniform_(0.0, 1.0).clone()
                samples[ids] = torch.stack([tl, vx, vy, wz, hd, uh, us], dim=-1)
            torch.manual_seed(seed)
            term.compute(0.02)
            steps.append(dict(samples=samples, vel_command_b=term.vel_command_b.clone(),
                              heading_target=term.heading_target.clone(), is_heading_env=term.is_heading_env.clone(),
                              is_standing_env=term.is_standing_env.clone(), metric_xy=term.metrics["error_vel_xy"].clone(),
                              metric_yaw=term.metrics["error_vel_yaw"].clone(), time_left=term.time_left.clone(),
                              command_counter=term.command_counter.clone(), resampled=torch.zeros(n, dtype=torch.bool).index_fill_(0, ids, True)))
        for k, v in pre.items():
            out[f"{tag}_pre_{k}"] = v
        for k in steps[0]:
            out[f"{tag}_{k}"] = torch.stack([st[k] for st in steps])
        # reset of a subset (command_manager.py:120-149)
        ids = torch.tensor([0, 5, 17, 100])
        seed = 800
        torch.manual_seed(seed)
        tl = torch.zeros(len(ids)).uniform_(*cfg_i.resampling_time_range)
        r = torch.empty(len(ids))
        draws = [tl] + [r.uniform_(*rg).clone() for rg in (ranges.lin_vel_x, ranges.lin_vel_y, ranges.ang_vel_z, ranges.heading)]
        draws += [r.uniform_(0.0, 1.0).clone(), r.uniform_(0.0, 1.0).clone()]
        samples = torch.zeros(n, 7)
        samples[ids] = torch.stack(draws, dim=-1)
        torch.manual_seed(seed)
        extras = term.reset(ids)
        out[f"{tag}_reset_ids"] = ids
        out[f"{tag}_reset_samples"] = samples
        out[f"{tag}_reset_extras"] = torch.tensor([extras["error_vel_xy"], extras["error_vel_yaw"]], dtype=torch.float64)
        for a in ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter"):
            out[f"{tag}_reset_{a}"] = getattr(term, a).clone()
        out[f"{tag}_reset_metric_xy"] = term.metrics["error_vel_xy"].clone()
        out[f"{tag}_cfg"] = torch.tensor([*cfg_i.resampling_time_range, cfg_i.rel_standing_envs, cfg_i.rel_heading_envs,
                                          cfg_i.heading_control_stiffness], dtype=torch.float64)
    save("velocity_command", **out)


@section
def managers_anymal():
    """The real Reward / Termination / Observation managers on the ANYmal-C rough term tables
    (velocity_env_cfg.py:119-143, 226-265), three steps with resets in between."""
    m = R.bootstrap()
    rew, term, obs, vel = m["rewards"], m["terminations"], m["observations"], m["vel_rewards"]
    tc = m["manager_term_cfg"]
    ncfg = m["noise_cfg"]
    SE = R.SceneEntity
    cs = _sensor_mod()
    g = gen(6301)
    n, j, b, r = 160, 12, 17, 187
    feet, thigh, base = [13, 14, 15, 16], [5, 6, 7, 8], [0]

    def se(name, **k):
        return SE(name, **k)

    rewards = dict(
        track_lin_vel_xy_exp=tc.RewardTermCfg(func=rew.track_lin_vel_xy_exp, weight=1.0,
                                              params={"command_name": "base_velocity", "std": math.sqrt(0.25)}),
        track_ang_vel_z_exp=tc.RewardTermCfg(func=rew.track_ang_vel_z_exp, weight=0.5,
                                             params={"command_name": "base_velocity", "std": math.sqrt(0.25)}),
        lin_vel_z_l2=tc.RewardTermCfg(func=rew.lin_vel_z_l2, weight=-2.0),
        ang_vel_xy_l2=tc.RewardTermCfg(func=rew.ang_vel_xy_l2, weight=-0.05),
        dof_torques_l2=tc.RewardTermCfg(func=rew.joint_torques_l2, weight=-1.0e-5),
        dof_acc_l2=tc.RewardTermCfg(func=rew.joint_acc_l2, weight=-2.5e-7),
        action_rate_l2=tc.RewardTermCfg(func=rew.action_rate_l2, weight=-0.01),
        feet_air_time=tc.RewardTermCfg(func=vel.feet_air_time, weight=0.125,
                                       params={"sensor_cfg": se("contact_forces", body_ids=feet),
                                               "command_name": "base_velocity", "threshold": 0.5}),
        undesired_contacts=tc.RewardTermCfg(func=rew.undesired_contacts, weight=-1.0,
                                            params={"sensor_cfg": se("contact_forces", body_ids=thigh), "threshold": 1.0}),
        flat_orientation_l2=tc.RewardTermCfg(func=rew.flat_orientation_l2, weight=0.0),
        dof_pos_limits=tc.RewardTermCfg(func=rew.joint_pos_limits, weight=0.0),
    )
    terminations = dict(
        time_out=tc.TerminationTermCfg(func=term.time_out, time_out=True),
        base_contact=tc.TerminationTermCfg(func=term.illegal_contact,
                                           params={"sensor_cfg": se("contact_forces", body_ids=base), "threshold": 1.0}),
    )
    U = ncfg.UniformNoiseCfg

    class PolicyCfg(tc.ObservationGroupCfg):
        pass

    pol = tc.ObservationGroupCfg()
    pol.enable_corruption = True
    pol.concatenate_terms = True
    policy_terms = dict(
        base_lin_vel=tc.ObservationTermCfg(func=obs.base_lin_vel, noise=U(n_min=-0.1, n_max=0.1)),
        base_ang_vel=tc.ObservationTermCfg(func=obs.base_ang_vel, noise=U(n_min=-0.2, n_max=0.2)),
        projected_gravity=tc.ObservationTermCfg(func=obs.projected_gravity, noise=U(n_min=-0.05, n_max=0.05)),
        velocity_commands=tc.ObservationTermCfg(func=obs.generated_commands, params={"command_name": "base_velocity"}),
        joint_pos=tc.ObservationTermCfg(func=obs.joint_pos_rel, noise=U(n_min=-0.01, n_max=0.01)),
        joint_vel=tc.ObservationTermCfg(func=obs.joint_vel_rel, noise=U(n_min=-1.5, n_max=1.5)),
        actions=tc.ObservationTermCfg(func=obs.last_action),
        height_scan=tc.ObservationTermCfg(func=obs.height_scan, params={"sensor_cfg": se("height_scanner")},
                                          noise=U(n_min=-0.1, n_max=0.1), clip=(-1.0, 1.0)),
    )
    for k, v in policy_terms.items():
        setattr(pol, k, v)
    obs_cfg = NS(policy=pol)

    out = {}
    steps = 3
    env = None
    managers = None
    for k in range(steps):
        s = make_state(g, n, j, b)
        ray_hits = torch.randn(n, r, 3, generator=g)
        ray_hits[..., 2] = s.root_pos_w[:, 2:3] - 0.6 + torch.randn(n, r, generator=g) * 0.3
        ray_hits[0, :5] = float("inf")  # missed rays
        sensor_pos = s.root_pos_w.clone()
        if env is None:
            env = make_env(s, n, sensor_mod=cs)
            env.scene.sensors["height_scanner"] = NS(data=NS(pos_w=sensor_pos, ray_hits_w=ray_hits))
            env.scene["robot"].data = s
            env._cmd = s.command
            env.command_manager = NS(get_command=lambda name: env._cmd)
            env.termination_manager = None
            rm = m["reward_manager"].RewardManager(rewards, env)
            tm = m["termination_manager"].TerminationManager(terminations, env)
            om = m["observation_manager"].ObservationManager(obs_cfg, env)
            managers = (rm, tm, om)
        rm, tm, om = managers
        # swap in this step's state
        env.scene["robot"].data = s
        env.scene.sensors["height_scanner"].data = NS(pos_w=sensor_pos, ray_hits_w=ray_hits)
        sen = env.scene.sensors["contact_forces"]
        sen._data = NS(net_forces_w_history=s.net_forces_w_history, current_contact_time=s.current_contact_time,
                       current_air_time=s.current_air_time, last_air_time=s.last_air_time,
                       last_contact_time=s.last_contact_time)
        env._cmd = s.command
        env.action_manager = NS(action=s.action, prev_action=s.prev_action)
        env.episode_length_buf = s.episode_length_buf
        dones = tm.compute().clone()
        reward = rm.compute(0.02).clone()
        seed = 900 + k
        torch.manual_seed(seed)
        o = om.compute()["policy"].clone()
        # replay the rand_like draws in term order (noise_model.py:63-64)
        torch.manual_seed(seed)
        u = torch.zeros(n, 235)
        col = 0
        for name, width, noisy in (("base_lin_vel", 3, True), ("base_ang_vel", 3, True), ("projected_gravity", 3, True),
                                   ("velocity_commands", 3, False), ("joint_pos", 12, True), ("joint_vel", 12, True),
                                   ("actions", 12, False), ("height_scan", 187, True)):
            if noisy:
                u[:, col:col + width] = torch.rand(n, width)
            col += width
        rec = dump_state(s)
        rec.update(ray_hits_w=ray_hits, sensor_pos_w=sensor_pos, noise_u=u, obs=o, reward=reward, dones=dones,
                   terminated=tm.terminated.clone(), time_outs=tm.time_outs.clone(),
                   term_dones=torch.stack([tm.get_term(t).clone() for t in tm.active_terms]),
                   step_reward=rm._step_reward.clone(),
                   episode_sums=torch.stack([rm._episode_sums[t].clone() for t in rm.active_terms]))
        # reset bookkeeping for the done envs (manager_based_rl_env.py:364-388)
        ids = dones.nonzero().flatten()
        if len(ids) > 0:
            r_extras = rm.reset(ids)
            t_extras = tm.reset(ids)
            rec["reset_reward_log"] = torch.stack([r_extras["Episode_Reward/" + t] for t in rm.active_terms])
            rec["reset_term_log"] = torch.tensor([t_extras["Episode_Termination/" + t] for t in tm.active_terms])
        else:
            rec["reset_reward_log"] = torch.zeros(len(rm.active_terms))
            rec["reset_term_log"] = torch.zeros(len(tm.active_terms), dtype=torch.long)
        rec["episode_sums_after_reset"] = torch.stack([rm._episode_sums[t].clone() for t in rm.active_terms])
        for kk, v in rec.items():
            out.setdefault(kk, []).append(v)
    out = {k: torch.stack(v) for k, v in out.items()}
    out["reward_names"] = torch.tensor([ord(c) for c in ",".join(rm.active_terms)])
    out["reward_weights"] = torch.tensor([rewards[t].weight for t in rm.active_terms], dtype=torch.float64)
    save("managers_anymal", **out)
