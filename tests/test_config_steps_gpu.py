"""GPU: end-to-end non-physics env steps of Isaac-Cartpole-v0 and Isaac-Reach-Franka-v0 (joint-position and
DifferentialIK arm actions) against a composition of the oracle's pieces in the reference's step order
(manager_based_rl_env.py:152-241).  The pieces themselves are pinned to the reference by the golden fixtures."""

import math

import pytest
import torch

from conftest import assert_close, assert_exact
from oracle import actuators as A
from oracle import articulation as AD
from oracle import command as C
from oracle import ik as IK
from oracle import managers as MG
from oracle import mathx as M
from oracle import terms as T

pytestmark = pytest.mark.gpu


def test_cartpole_step(dev):
    from isaaclab_amd.envs.manager_based_step import CartpoleStep
    from isaaclab_amd.tasks import cartpole as CP

    n = 512
    g = torch.Generator().manual_seed(1001)
    env = CartpoleStep(n, dev)
    q0 = torch.stack([torch.rand(n, generator=g) * 6.4 - 3.2, (torch.rand(n, generator=g) * 2 - 1) * math.pi], -1)
    qd0 = torch.randn(n, 2, generator=g)
    env.data.joint_pos, env.data.joint_vel = q0.to(dev), qd0.to(dev)
    env.data._previous_joint_vel.copy_(qd0)
    ep = torch.randint(0, 300, (n,), generator=g)
    ep[:3] = 298
    env.mdp.episode_length_buf.copy_(ep)
    kd = torch.tensor([10.0, 0.0]).repeat(n, 1)
    q, qd, prev_qd = q0, qd0, qd0.clone()
    action, prev_action = torch.zeros(n, 1), torch.zeros(n, 1)
    sums, step_reward = torch.zeros(5, n), torch.zeros(n, 5)
    sim_time, acc_time = 0.0, -1.0
    for step in range(3):
        raw = torch.rand(n, 1, generator=g) * 2 - 1
        phys = dict(joint_pos=torch.stack([q0 + 0.3 * torch.randn(n, 2, generator=g) for _ in range(2)]),
                    joint_vel=torch.stack([torch.randn(n, 2, generator=g) for _ in range(2)]), root_state_w=torch.zeros(n, 13))
        phys["root_state_w"][:, 3] = 1.0
        env.load_physics(phys)
        obs, reward, terminated, time_outs = env.step(raw.to(dev))
        # ---- oracle composition
        prev_action, action = action, raw.clone()
        tau_ff = torch.cat([raw * 100.0, torch.zeros(n, 1)], -1)
        for k in range(2):
            comp = A.pd_torque(torch.zeros(n, 2), torch.zeros(n, 2), tau_ff, q, qd, torch.zeros(n, 2), kd)
            app = A.clip_effort(comp, torch.full((n, 2), 400.0))
            q, qd = phys["joint_pos"][k], phys["joint_vel"][k]
            sim_time += CP.PHYSICS_DT
            acc, prev_qd = AD.joint_acc(qd, prev_qd, sim_time - acc_time)
            acc_time = sim_time
        ep = ep + 1
        dones, term, trunc, term_dones = MG.termination_compute(
            [T.time_out(ep, CP.MAX_EPISODE_LENGTH), T.joint_pos_out_of_manual_limit(q, (-3.0, 3.0), [0])], [True, False])
        vals = [T.is_alive(term), T.is_terminated(term), T.joint_pos_target_l2(q, 0.0, [1]), T.joint_vel_l1(qd, [0]), T.joint_vel_l1(qd, [1])]
        ref_reward = MG.reward_compute(vals, [1.0, -2.0, -1.0, -0.01, -0.005], CP.STEP_DT, sums, step_reward)
        MG.reward_reset(sums, dones, CP.MAX_EPISODE_LENGTH * CP.STEP_DT)
        ep = torch.where(dones, torch.zeros_like(ep), ep)
        # action_manager.py:309-311: the action history of the done envs restarts at zero
        action = torch.where(dones[:, None], torch.zeros_like(action), action)
        prev_action = torch.where(dones[:, None], torch.zeros_like(prev_action), prev_action)
        ref_obs = torch.cat([q, qd], -1)   # defaults are zero
        assert_exact(terminated, term, f"terminated step {step}")
        assert_exact(time_outs, trunc, f"time_outs step {step}")
        assert_exact(env.mdp.episode_length_buf, ep, f"episode_length_buf step {step}")
        assert_exact(env.data.computed_torque, comp, f"computed torque step {step}")
        assert_exact(env.data.applied_torque, app, f"applied torque step {step}")
        assert_exact(env.data.joint_acc, acc, f"joint_acc step {step}")
        assert_exact(obs, ref_obs, f"obs step {step}")
        assert_close(reward, ref_reward, atol=1e-6, what=f"reward step {step}")
        assert_close(env.mdp.episode_sums, sums, atol=1e-6, what=f"episode sums step {step}")
        assert_exact(env.action, action, "action")
        assert_exact(env.prev_action, prev_action, "prev_action")
        assert term.any()


FRANKA = dict(robot="franka", joints=9, bodies=11, arm=7, ee=8, default=[0.0, -0.569, 0.0, -2.81, 0.0, 3.037, 0.741, 0.04, 0.04],   # franka.py:38-48
              kp=[80.0] * 7 + [2e3] * 2, kd=[4.0] * 7 + [1e2] * 2, lim=[87.0] * 4 + [12.0] * 3 + [200.0] * 2, offset=True)
# the fork's UFactory xArm6 with gripper (IsaacLab_UFactory_Xarm6_config): 6 arm + 2 finger joints, 2000 / 100 gains
XARM6 = dict(robot="xarm6", joints=8, bodies=10, arm=6, ee=6, default=[0.0, -1.22, -0.78, 0.0, 1.30, 0.0, 0.0, 0.0],
             kp=[2000.0] * 8, kd=[100.0] * 8, lim=[87.0] * 3 + [12.0] * 3 + [200.0] * 2, offset=False)


def _reach_physics(g, n, q_mean, spec=FRANKA):
    nj, nb, arm = spec["joints"], spec["bodies"], spec["arm"]
    root = torch.zeros(n, 13)
    root[:, :3] = torch.randn(n, 3, generator=g)
    yaw = torch.rand(n, generator=g) * 6.28 - 3.14
    root[:, 3], root[:, 6] = torch.cos(yaw / 2), torch.sin(yaw / 2)
    body = torch.zeros(n, nb, 13)
    body[..., :3] = root[:, None, :3] + 0.4 * torch.randn(n, nb, 3, generator=g)
    body[..., 3:7] = M.normalize(torch.randn(n, nb, 4, generator=g))
    q = torch.stack([q_mean + 0.2 * torch.randn(n, nj, generator=g) for _ in range(2)])
    jac = torch.zeros(n, nb - 1, 6, nj)
    if spec["robot"] == "xarm6":
        # the arm's real forward kinematics (ROBOT/xarm/xarm6_with_gripper.urdf:79-225): end-effector pose and geometric
        # Jacobian of link6 at the joint positions the step ends on, as the simulator's views would hold them
        from isaaclab_amd import synth

        ee_pos, ee_quat, j6 = synth.serial_arm_kinematics(q[-1][:, :arm].numpy(), root[:, :3].numpy(), root[:, 3:7].numpy())
        body[:, spec["ee"], :3], body[:, spec["ee"], 3:7] = torch.from_numpy(ee_pos), torch.from_numpy(ee_quat)
        jac[..., :arm] = 0.05 * torch.randn(n, nb - 1, 6, arm, generator=g)
        jac[:, spec["ee"] - 1, :, :arm] = torch.from_numpy(j6)
    else:
        # Jacobians of the moving links: well-conditioned 6 x arm blocks (sigma in [0.2, 1.5]) + finger columns
        U, _ = torch.linalg.qr(torch.randn(n, nb - 1, 6, 6, generator=g))
        V, _ = torch.linalg.qr(torch.randn(n, nb - 1, arm, arm, generator=g))
        S = 0.2 + 1.3 * torch.rand(n, nb - 1, 6, generator=g)
        jac[..., :arm] = U @ torch.diag_embed(S) @ V[..., :6].transpose(-1, -2)
    jac[..., arm:] = 0.1 * torch.randn(n, nb - 1, 6, nj - arm, generator=g)
    return dict(joint_pos=q, joint_vel=torch.stack([0.5 * torch.randn(n, nj, generator=g) for _ in range(2)]), root_state_w=root,
                body_state_w=body, jacobians=jac)


@pytest.mark.parametrize("arm_action", ["joint_pos", "ik_rel", "ik_abs"])
def test_reach_step(dev, arm_action):
    run_reach(dev, arm_action, 384)


@pytest.mark.parametrize("arm_action", ["joint_pos", "ik_rel"])
def test_reach_step_at_benchmark_size(dev, arm_action):
    """8,192 envs (the size the configuration is quoted at): index-width and grid-size errors show here, not at 384."""
    run_reach(dev, arm_action, 8192)


@pytest.mark.parametrize("arm_action, n", [("joint_pos", 384), ("ik_rel", 384), ("ik_abs", 384), ("joint_pos", 8192), ("ik_rel", 8192)])
def test_reach_step_xarm6(dev, arm_action, n):
    """The same task on the fork's xArm6 with gripper: 8 joints, 10 bodies, 29-wide observation (8 + 8 + 7 + 6), implicit
    actuators with gains 2000 / 100, end-effector pose and Jacobian from the arm's own forward kinematics."""
    run_reach(dev, arm_action, n, XARM6)


def run_reach(dev, arm_action, n, spec=FRANKA):
    from isaaclab_amd.envs.manager_based_step import ReachStep
    from isaaclab_amd.tasks import reach as RE

    nj, arm, ee = spec["joints"], spec["arm"], spec["ee"]
    g = torch.Generator().manual_seed(2002)
    env = ReachStep(n, dev, arm_action=arm_action, robot=spec["robot"])
    assert env.mdp.kernel == f"reach_j{nj}_a{env.action.shape[1]}", env.mdp.kernel
    default = torch.tensor(spec["default"]).repeat(n, 1)
    env.set_default_joint_pos(default)
    adim = env.action.shape[1]
    assert env.mdp.obs_dim == 2 * nj + 7 + adim
    q, qd = default + 0.1 * torch.randn(n, nj, generator=g), torch.zeros(n, nj)
    env.data.joint_pos, env.data.joint_vel = q.to(dev), qd.to(dev)
    ep = torch.randint(0, 360, (n,), generator=g)
    ep[:4], ep[4:8] = 359, 358       # time out on the first and on the second step
    env.mdp.episode_length_buf.copy_(ep)
    cmd_state = dict(cmd=torch.cat([torch.rand(n, 3, generator=g) * 0.5, M.normalize(torch.randn(n, 4, generator=g))], -1),
                     time_left=torch.rand(n, generator=g) * 0.1, counter=torch.zeros(n, dtype=torch.long))
    env.mdp.command.copy_(cmd_state["cmd"])
    env.mdp.time_left.copy_(cmd_state["time_left"])
    cfg = C.PoseCommandCfg()
    kp, kd, lim = (torch.tensor(spec[k]).repeat(n, 1) for k in ("kp", "kd", "lim"))
    prev_qd = torch.zeros(n, nj)
    action, prev_action = torch.zeros(n, adim), torch.zeros(n, adim)
    sums, step_reward = torch.zeros(5, n), torch.zeros(n, 5)
    target = torch.zeros(n, nj)
    sim_time, acc_time = 0.0, -1.0
    op, orot = (torch.tensor([RE.IK_BODY_OFFSET[0]]), torch.tensor([RE.IK_BODY_OFFSET[1]])) if spec["offset"] else (None, None)
    phys = _reach_physics(g, n, default, spec)
    for step in range(2):
        env.load_physics(phys)       # pose / Jacobian the action terms see = end of the previous step
        if arm_action == "ik_abs":   # a reachable absolute pose near the current one, in the root frame
            pb, qb = IK.frame_pose(phys["root_state_w"][:, :3], phys["root_state_w"][:, 3:7], phys["body_state_w"][:, ee, :3],
                                   phys["body_state_w"][:, ee, 3:7], op, orot)
            tp, tq = M.apply_delta_pose(pb, qb, torch.cat([0.05 * torch.randn(n, 3, generator=g), 0.2 * torch.randn(n, 3, generator=g)], -1))
            raw = torch.cat([tp, tq], -1)
        else:
            raw = torch.rand(n, adim, generator=g) * 2 - 1
        noise_u = torch.rand(n, env.mdp.obs_dim, generator=g)
        samples = torch.rand(n, 7, generator=g)
        samples[:, 0] = 4.0
        new_phys = _reach_physics(g, n, default, spec)
        step_phys = dict(phys, joint_pos=new_phys["joint_pos"], joint_vel=new_phys["joint_vel"])
        env.load_physics(step_phys)
        obs, reward, terminated, time_outs = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), samples.to(dev))
        # ---- oracle composition
        prev_action, action = action, raw.clone()
        rp, rq = phys["root_state_w"][:, :3], phys["root_state_w"][:, 3:7]
        ep_w, eq_w, jac_w = phys["body_state_w"][:, ee, :3], phys["body_state_w"][:, ee, 3:7], phys["jacobians"][:, ee - 1][:, :, :arm]
        if arm_action == "joint_pos":
            arm_target = raw * 0.5 + default[:, :arm]
        else:
            scale = 0.5 if arm_action == "ik_rel" else 1.0
            _, pdes, qdes = IK.process_actions(raw, scale, None, rp, rq, ep_w, eq_w, op, orot, "pose", arm_action == "ik_rel")
        for k in range(2):
            if arm_action != "joint_pos":
                arm_target = IK.apply_actions(rp, rq, ep_w, eq_w, jac_w, q[:, :arm], pdes, qdes, op, orot, "pose", "dls",
                                              {"lambda_val": 0.01}, f64=True)
            target = torch.cat([arm_target, target[:, arm:]], -1)
            comp = A.pd_torque(target, torch.zeros(n, nj), torch.zeros(n, nj), q, qd, kp, kd)
            app = A.clip_effort(comp, lim)
            q, qd = step_phys["joint_pos"][k], step_phys["joint_vel"][k]
            sim_time += RE.PHYSICS_DT
            acc, prev_qd = AD.joint_acc(qd, prev_qd, sim_time - acc_time)
            acc_time = sim_time
        ep = ep + 1
        rs, bs = phys["root_state_w"], phys["body_state_w"]
        dones, term, trunc, _ = MG.termination_compute([T.time_out(ep, RE.MAX_EPISODE_LENGTH)], [True])
        cmd = cmd_state["cmd"]
        vals = [T.position_command_error(cmd, rs, bs[:, ee, :3]), T.position_command_error_tanh(cmd, rs, bs[:, ee, :3], 0.1),
                T.orientation_command_error(cmd, rs, bs[:, ee, 3:7]), T.action_rate_l2(action, prev_action), T.joint_vel_l2(qd)]
        ref_reward = MG.reward_compute(vals, [-0.2, 0.1, -0.1, -0.0001, -0.0001], RE.STEP_DT, sums, step_reward)
        MG.reward_reset(sums, dones, RE.MAX_EPISODE_LENGTH * RE.STEP_DT)
        # command: reset of the done envs (resample), then compute(dt)
        if dones.any():
            c2, tl2, cnt2, _ = C.pose_compute(cfg, cmd, torch.where(dones, torch.full((n,), -1.0), cmd_state["time_left"] + RE.STEP_DT),
                                              torch.where(dones, torch.zeros(n, dtype=torch.long), cmd_state["counter"]), RE.STEP_DT, samples)
            cmd = torch.where(dones.unsqueeze(-1), c2, cmd)
            cmd_state["time_left"] = torch.where(dones, samples[:, 0], cmd_state["time_left"])
            cmd_state["counter"] = torch.where(dones, torch.ones(n, dtype=torch.long), cmd_state["counter"])
        cmd, cmd_state["time_left"], cmd_state["counter"], _ = C.pose_compute(cfg, cmd, cmd_state["time_left"], cmd_state["counter"],
                                                                             RE.STEP_DT, samples)
        cmd_state["cmd"] = cmd
        ep = torch.where(dones, torch.zeros_like(ep), ep)
        action = torch.where(dones[:, None], torch.zeros_like(action), action)            # action_manager.py:309-311
        prev_action = torch.where(dones[:, None], torch.zeros_like(prev_action), prev_action)
        parts = [(T.joint_pos_rel(q, default), (-0.01, 0.01)), (T.joint_vel_rel(qd, torch.zeros(n, nj)), (-0.01, 0.01)), (cmd, None),
                 (action, None)]
        cols, col = [], 0
        for val, noise in parts:
            w = val.shape[1]
            cols.append(T.apply_obs_pipeline(val, noise_u[:, col:col + w], noise, None))
            col += w
        ref_obs = MG.observation_concat(cols)
        assert_exact(time_outs, trunc, f"time_outs step {step}")
        assert not terminated.any()
        assert_exact(env.mdp.episode_length_buf, ep, f"episode_length_buf step {step}")
        assert_exact(env.mdp.time_left, cmd_state["time_left"], f"command time_left step {step}")
        assert_exact(env.mdp.command_counter, cmd_state["counter"], f"command counter step {step}")
        assert_close(env.mdp.command, cmd, atol=1e-6, what=f"command step {step}")
        # the IK update goes through a 6x6 solve; with the xArm's own Jacobians near-singular poses occur (the reference's
        # float32 inverse is itself 1e-4 .. 1e-3 off there), so those targets are compared where J J^T is well conditioned
        tol = 1e-6 if arm_action == "joint_pos" else 2e-5
        rows = torch.ones(n, dtype=torch.bool)
        if arm_action != "joint_pos" and spec["robot"] == "xarm6":
            rows = torch.linalg.svdvals(jac_w.double())[:, -1] > 0.05
            assert rows.float().mean() > 0.5
        assert_close(env.data.joint_pos_target[:, :arm][rows.to(dev)], arm_target[rows], rtol=1e-5, atol=tol, what=f"arm joint targets step {step}")
        big = 2e-3 * (float(kp.max()) / 80.0) if arm_action != "joint_pos" else 1e-5 * (float(kp.max()) / 80.0)
        assert_close(env.data.computed_torque[rows.to(dev)], comp[rows], rtol=1e-5, atol=big, what=f"computed torque step {step}")
        assert_exact(env.data.joint_acc, acc, f"joint_acc step {step}")
        assert_close(obs, ref_obs, atol=2e-6, what=f"obs step {step}")
        assert_close(reward, ref_reward, atol=2e-6, what=f"reward step {step}")
        assert_close(env.mdp.episode_sums, sums, atol=2e-6, what=f"episode sums step {step}")
        assert trunc.any()
        phys = dict(new_phys)


@pytest.mark.parametrize("config, robot", [("cartpole", None), ("reach", "franka"), ("reach", "xarm6")])
def test_fused_first_and_last_launches_equal_the_separate_ones(dev, config, robot):
    """The step with its joint action on the first actuator launch (ilx_actuator_pd_joint_action), the scene updates paired
    with the next substep's actuator launch and the last one inside the managers' launch (ilx_mdp_step_scene_update) against
    the same step issued as separate launches, and against the fused calls served by the plan-interpreting kernel (where
    the last update is a launch of its own again): every buffer of the step bit for bit over three steps."""
    import bench_configs as BC

    def run(paired, rides, compiled):
        cfg = BC.build(config, dev, 1000, seed=77, **({"robot": robot} if robot else {}))
        env = cfg.env
        env.paired_substeps, env.joint_action_rides = paired, rides
        env.mdp.select_kernel(compiled)
        for _ in range(3):
            cfg.step()
        torch.cuda.synchronize()
        d, m = env.data, env.mdp
        return dict(obs=m.obs, reward=m.reward_buf, terminated=m.terminated, time_outs=m.time_outs, episode_sums=m.episode_sums,
                    episode_length=m.episode_length_buf, action=env.action, prev_action=env.prev_action, joint_acc=d._joint_acc,
                    previous_joint_vel=d._previous_joint_vel, computed=d.computed_torque, applied=d.applied_torque,
                    joint_pos_target=d.joint_pos_target, joint_effort_target=d.joint_effort_target)

    fused = run(True, True, True)
    assert fused["obs"].abs().sum() > 0
    for name, other in (("separate launches", run(False, False, True)), ("plan-interpreting kernel", run(True, True, False))):
        for key, value in fused.items():
            assert_exact(value, other[key].cpu(), f"{config} {robot or ''}: {key} against the {name}")


@pytest.mark.parametrize("config", ["cartpole", "reach", "anymal", "h1", "shadow"])
def test_step_replayed_from_a_command_list_equals_the_step_issued_directly(dev, config):
    """The step's launches kept in a command list (ilx_commands_begin / _end) and re-issued by ilx_commands_replay against
    the same step issued call by call: nothing runs while recording, a replay runs exactly the step -- every buffer bit for
    bit over three steps, the generator's step count advancing in device memory in both."""
    import bench_configs as BC
    from isaaclab_amd.utils.command_list import CommandList

    def buffers(cfg):
        env = cfg.env
        if config == "shadow":
            return dict(obs=env.obs, reward=env.reward, goal_rot=env.goal_rot, successes=env.successes, episode_length=env.episode_length_buf,
                        terminated=env.reset_terminated, time_outs=env.reset_time_outs, consecutive=env.consecutive_successes,
                        cur_targets=env.cur_targets, prev_targets=env.prev_targets)
        m, d = env.mdp, env.data
        out = dict(obs=m.obs, reward=m.reward_buf, terminated=m.terminated, time_outs=m.time_outs, episode_sums=m.episode_sums,
                   episode_length=m.episode_length_buf, action=env.action, prev_action=env.prev_action, joint_acc=d._joint_acc,
                   applied=d.applied_torque)
        if config in ("anymal", "h1"):
            out.update(height_scan=env.height_scan, air_time=env.contact.data.current_air_time, scanner_clock=env.scanner._timestamp_last_update)
        return out

    def run(listed):
        cfg = BC.build(config, dev, 1000, seed=31)
        cfg.env.count_steps_on_device()
        cfg.step()                       # whatever is created at the first step is created here
        before = {k: v.clone() for k, v in buffers(cfg).items()}
        if listed:
            commands = CommandList()
            with commands.record():
                cfg.step()
            torch.cuda.synchronize()
            for k, v in buffers(cfg).items():
                assert_exact(v, before[k].cpu(), f"{config}: {k} while recording")
            assert len(commands) == sum(n for _, n in cfg.launches.values()), "one kept launch per launch of the step"
            for _ in range(3):
                commands.replay()
        else:
            for _ in range(3):
                cfg.step()
        torch.cuda.synchronize()
        return {k: v.cpu().clone() for k, v in buffers(cfg).items()}

    direct, listed = run(False), run(True)
    assert direct["obs"].abs().sum() > 0
    for k, v in direct.items():
        assert_exact(listed[k], v, f"{config}: {k} after three replays")


@pytest.mark.parametrize("config", ["cartpole", "anymal"])
def test_step_recorded_in_stretches_around_the_physics_steps(dev, config):
    """The one-chain step cut where the simulator's steps belong (SegmentedCommandList): decimation + 1 stretches whose
    replay, with a callback between them standing in for the simulator, equals the step issued directly -- bit for bit over
    three steps -- and the callback sees every substep in order."""
    import bench_configs as BC
    from isaaclab_amd.utils.command_list import SegmentedCommandList

    def final(cfg):
        m = cfg.env.mdp
        return [t.cpu().clone() for t in (m.obs, m.reward_buf, m.terminated, m.time_outs, m.episode_sums, cfg.env.action)]

    direct = BC.build(config, dev, 1000, seed=5)
    direct.env.count_steps_on_device()
    for _ in range(4):
        direct.step()
    torch.cuda.synchronize()

    cfg = BC.build(config, dev, 1000, seed=5)
    cfg.env.count_steps_on_device()
    cfg.step()
    step = SegmentedCommandList().record(cfg.env, cfg.step)
    decimation = cfg.info["decimation"]
    assert len(step.segments) == decimation + 1 and all(len(seg) >= 1 for seg in step.segments)
    assert len(step) == sum(n for _, n in cfg.launches.values())
    assert cfg.env.on_physics is None
    seen = []
    for _ in range(3):
        step.replay(lambda k: (torch.cuda.current_stream().synchronize(), seen.append(k)))   # "the simulator": waits for the launches so far
    torch.cuda.synchronize()
    assert seen == list(range(decimation)) * 3
    for k, (a, b) in enumerate(zip(final(cfg), final(direct))):
        assert_exact(a, b, f"{config}: buffer {k}")
