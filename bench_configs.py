"""The five configurations `bench.py --config` can run: for each, the step object on synthetic state, the step's
launches (name -> callable, launches per env step) and the algorithmic bytes per env of each launch.

Byte rule (same for every table): each distinct per-env input field read once and each output field written once by
a launch that has to exist; fp32 = 4 B, bool = 1 B, int64 = 8 B; per-joint constants (gains, limits, defaults), network
weights and the terrain mesh are not counted.  J = joints, B = bodies, R = rays.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

ENVS = {"cartpole": 4096, "reach": 8192, "anymal": 16384, "h1": 4096, "shadow": 8192}
WORKLOAD = {
    "cartpole": "Isaac-Cartpole-v0 non-physics step",
    "reach": "Isaac-Reach-Franka-v0 non-physics step (joint-position arm action)",
    "anymal": "Isaac-Velocity-Rough-Anymal-C-v0 non-physics step",
    "h1": "Isaac-Velocity-Rough-H1-v0 non-physics step",
    "shadow": "Isaac-Repose-Cube-Shadow-Direct-v0 non-physics step",
}


@dataclass
class Config:
    name: str
    envs: int
    step: callable                       # one env step of the hot path (all launches, in order)
    launches: dict                       # kernel name -> (callable issuing ONE launch of it, launches per env step)
    bytes_per_env: dict                  # kernel name -> algorithmic bytes per env per launch
    info: dict = field(default_factory=dict)   # goes into the JSON line's "config"
    log_sums: torch.Tensor | None = None       # the statistics vector the multi-GPU all-reduce carries (manager-based)
    bind_log_sums: callable = None

    def bytes_per_env_step(self):
        return sum(self.bytes_per_env[k] * n for k, (_, n) in self.launches.items())


def _locomotion_bytes(j, b, r, obs_dim, n_rewards, actuator, hit_records=True):
    f = 4 * j           # one fp32 per joint
    out = {
        # InteractiveScene.update in one launch -- contact: read net forces, write one history row (ring; net_forces_w
        # is its newest entry and not written apart) + head 8, read+write 4 timers, clock 18; joint_acc: read vel,
        # read+write previous vel, write acc; scanner clock 18
        "scene_update": 2 * 12 * b + 8 + 8 * 4 * b + 18 + 4 * f + 18,
        # read pose 28 + clock 9, write pose 28, ray hits R*12 (not in a refresh that leaves them to their first reader:
        # RayCaster.lean_hits, what the fused steps run), height scan R*4, clock 5
        "raycaster_refresh": 28 + 9 + 28 + (r * 12 if hit_records else 0) + r * 4 + 5,
        # read root 52, joints (pos, vel, acc, torque, default) 5f, actions 2f, contact history 3*B*12, timers 2*4B,
        # height scan 4R, sensor pos 12, command state r+w ~82, ep_len r+w 16 (noise: parity mode only, not counted);
        # write obs, reward 4, step_reward, episode sums r+w, flags 5
        "mdp_step": 52 + 5 * f + 2 * f + 36 * b + 8 * b + 4 * r + 12 + 82 + 16 + 4 * obs_dim + 4 + 4 * n_rewards + 8 * n_rewards + 5,
        # read raw, write action + previous action, read default, write target
        "joint_action_process": 5 * f,
    }
    if actuator == "actuator_net_mlp":
        # read q_des, q, qd; read+write pos-error / velocity histories (3 deep); write computed, applied
        out[actuator] = 3 * f + 4 * 3 * f + 2 * f
    elif actuator == "actuator_net_lstm":
        # read q_des, q, qd; read+write hidden and cell state (2 x 2 layers x 8 x 4 B per joint); write computed, applied
        out[actuator] = 3 * f + 2 * 2 * 64 * j + 2 * f
    else:
        # read q_des, qd_des, tau_ff, q, qd; write computed, applied
        out[actuator] = 7 * f
    return out


def _ride_on_first_actuator_launch(env, raw, launches, nb, decimation):
    """The step classes whose joint action term rides on the first substep's actuator launch (ManagerBasedStep.
    joint_action_ride): that launch as a kernel of its own in the tables; process_action and the plain first actuator
    launch are still timed, at 0 launches per step."""
    if not env.joint_action_rides or env.joint_action_ride(raw) is None:
        return
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.utils.types import ArticulationActions

    d = env.data

    def first_substep():
        ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                 joint_efforts=d.joint_effort_target, joint_indices=slice(None))
        env.actuator.compute(ca, d.joint_pos, d.joint_vel, joint_action=env.joint_action_ride(raw))

    launches["actuator_pd_joint_action"] = (first_substep, 1)
    launches["joint_action_process"] = (launches["joint_action_process"][0], 0)
    launches["actuator_pd"] = (launches["actuator_pd"][0], decimation - 1)
    nb["actuator_pd_joint_action"] = nb["actuator_pd"] + nb["joint_action_process"]
    if env.paired_substeps:
        # the recorded step is one launch per physics substep boundary: {action + actuator(0)}, {scene update(k - 1) +
        # actuator(k)} ..., {scene update(last) + managers}; the tables above time the parts on their own
        launches["scene_update"] = (launches["scene_update"][0], 0)
        launches["actuator_pd"] = (launches["actuator_pd"][0], 0)

        def update_and_actuate():
            ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                     joint_efforts=d.joint_effort_target, joint_indices=slice(None))
            env.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=env.scene.describe_update(env.physics_dt))

        launches["actuator_pd+scene_update"] = (update_and_actuate, decimation - 1)
        nb["actuator_pd+scene_update"] = nb["actuator_pd"] + nb["scene_update"]
        step_alone = launches["mdp_step"][0]
        launches["mdp_step"] = (step_alone, 0)
        launches["mdp_step+scene_update"] = (lambda: env.mdp.step(FM.PHASE_ALL, hold_step_count=True, scene_update=env.scene.describe_update(env.physics_dt)), 1)
        nb["mdp_step+scene_update"] = nb["mdp_step"] + nb["scene_update"]


def build_locomotion(device, envs, seed, robot="anymal_c", actuator="mlp", static_scan=False, root_scatter="origins"):
    from isaaclab_amd import synth
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.envs.locomotion_step import LocomotionVelocityStep
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.mesh import convert_to_mesh
    from isaaclab_amd.utils.string import resolve_matching_names_values
    from isaaclab_amd.utils.types import ArticulationActions
    from isaaclab_amd import _abi

    verts, tris, height_at = synth.rough_terrain_mesh(seed=3)
    mesh = convert_to_mesh(verts, tris, str(device))
    if robot == "anymal_c":
        joints, bodies, default_cfg = LV.ANYMAL_JOINTS, LV.ANYMAL_BODIES, LV.ANYMAL_DEFAULT_JOINT_POS
        feet = [13, 14, 15, 16]
        act_kernel = "actuator_net_" + actuator
        ws, bs = synth.actuator_mlp_weights()
        net = synth.actuator_lstm_weights() if actuator == "lstm" else dict(weights=ws, biases=bs, activation="softsign")
    else:
        joints, bodies, default_cfg = LV.H1_JOINTS, LV.H1_BODIES, LV.H1_DEFAULT_JOINT_POS
        feet = [i for i, b in enumerate(bodies) if b.endswith("ankle_link")]
        act_kernel, net = "actuator_dc_motor", None
    j, b = len(joints), len(bodies)
    idx, _, vals = resolve_matching_names_values(default_cfg, joints)
    default = torch.zeros(j)
    default[idx] = torch.tensor(vals)
    # where the robots stand: around the env origins that the reference's terrain importer deals for this terrain (env
    # blocks per terrain column, random start level; each robot within the reset event's +-0.5 m of its origin), or
    # ("uniform") anywhere on the terrain, every env somewhere else: no two envs near each other, the worst case for
    # the caches under the height scan
    origins = synth.curriculum_env_origins(envs, seed) if root_scatter == "origins" else None
    state = synth.locomotion_state(envs, seed, num_joints=j, num_bodies=b, default_joint_pos=default, feet_ids=feet,
                                   height_at=height_at, with_bodies=robot == "h1", env_origins=origins)
    env = LocomotionVelocityStep(envs, device, mesh, robot, actuator_net=net, seed=seed)
    env.load_state(state)
    env.scanner.static_workspace = static_scan
    raw = state["raw_action"].to(device)
    d = env.data

    def actuate():
        ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                 joint_efforts=d.joint_effort_target)
        env.actuator.compute(ca, d.joint_pos, d.joint_vel)

    def rays():
        env.scanner._is_outdated.fill_(True)   # every env is due, as once per env step
        env.scanner._update_outdated_buffers(height_out=env.height_scan, height_offset=env.height_scan_offset)

    def rays_overhead():
        env.scanner._is_outdated.fill_(True)

    def action():
        _abi.check(_abi.joint_action_process(_abi.ptr(raw), _abi.ptr(env.action), _abi.ptr(env.prev_action), None, 0.5,
                                             _abi.ptr(d.default_joint_pos), 0.0, None, _abi.ptr(d.joint_pos_target), env.n, j,
                                             _abi.stream()))

    def update_and_prepare():   # the step's last scene update, which also prepares the scanner's refresh (every env due)
        env.scanner._is_outdated.fill_(True)
        env._bind_scanner_pose()
        env.scene.update(LV.PHYSICS_DT, prepare_scanner=True)

    def cast():                 # the refresh behind it: the cast alone, on the records of the last preparation
        env.scanner._update_outdated_buffers(height_out=env.height_scan, height_offset=env.height_scan_offset, prepared=True)

    fused = env.fuse_scanner_preparation and not static_scan
    if fused:
        for _ in range(2):      # the first preparation on a stream packs the pattern; every env's record marked due
            update_and_prepare()
        launches = {act_kernel: (actuate, LV.DECIMATION),
                    "scene_update": (lambda: env.scene.update(LV.PHYSICS_DT), LV.DECIMATION - 1),
                    "scene_update_prepare": ((update_and_prepare, rays_overhead), 1), "raycaster_cast": (cast, 1),
                    "mdp_step": (lambda: env.mdp.step(FM.PHASE_ALL, hold_step_count=True), 1), "joint_action_process": (action, 1)}
    else:
        launches = {act_kernel: (actuate, LV.DECIMATION), "scene_update": (lambda: env.scene.update(LV.PHYSICS_DT), LV.DECIMATION),
                    "raycaster_refresh": ((rays, rays_overhead), 1),
                    "mdp_step": (lambda: env.mdp.step(FM.PHASE_ALL, hold_step_count=True), 1), "joint_action_process": (action, 1)}
    r = env.scanner.num_rays
    n_rewards = int(env.mdp.step_reward.shape[1])
    info = {"root_positions": ("env origins of the curriculum terrain (terrain_importer.py) + U(-0.5, 0.5) m" if origins is not None
                               else "uniform over the terrain"),
            "decimation": LV.DECIMATION, "joints": j, "bodies": b, "rays": r, "obs_dim": int(env.mdp.obs_dim),
            "terrain_triangles": int(env.scanner.meshes["/World/ground"].num_triangles)}
    if robot == "anymal_c":
        info["actuator"] = ("ActuatorNetMLP 6-32-32-32-1 softsign (f32 first layer, hidden layers as split-f16 MFMA)" if actuator == "mlp" else
                            "ActuatorNetLSTM LSTM(2, 8, 2 layers) + 8-32-1 softsign head")
    else:
        info["actuator"] = "DCMotor (PD + torque-speed envelope)"
    nbytes = _locomotion_bytes(j, b, r, int(env.mdp.obs_dim), n_rewards, act_kernel, hit_records=not env.scanner.lean_hits)
    if act_kernel == "actuator_net_mlp" and env.joint_action_rides:
        # process_action is done by the step's first network launch: three plain network launches and one that carries it
        def actuate_first():
            ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                     joint_efforts=d.joint_effort_target)
            env.actuator.compute(ca, d.joint_pos, d.joint_vel,
                                 joint_action=(raw, env.action, env.prev_action, env.action_scale, d.default_joint_pos, None))

        launches[act_kernel] = (actuate, LV.DECIMATION - 1)
        launches["actuator_net_mlp_joint_action"] = (actuate_first, 1)
        del launches["joint_action_process"]
        nbytes["actuator_net_mlp_joint_action"] = nbytes[act_kernel] + nbytes.pop("joint_action_process")
        if fused and env.paired_substeps:
            # the recorded chain (LocomotionVelocityStep._step_paired) issues a substep's scene update and the next substep's
            # network as ONE launch; the two on their own are what the eager step and the two-branch graph issue, and are
            # timed for comparison (0 launches per step of the chain)
            def update_and_actuate():
                ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                         joint_efforts=d.joint_effort_target)
                env.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=env.scene.describe_update(LV.PHYSICS_DT))

            launches["actuator_net_mlp+scene_update"] = (update_and_actuate, LV.DECIMATION - 1)
            nbytes["actuator_net_mlp+scene_update"] = nbytes[act_kernel] + nbytes["scene_update"]
            launches[act_kernel], launches["scene_update"] = (actuate, 0), (launches["scene_update"][0], 0)
    if act_kernel == "actuator_dc_motor" and env.joint_action_rides:
        # process_action is done by the step's first actuator launch (ilx_actuator_pd_joint_action)
        def actuate_first_dc():
            ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                     joint_efforts=d.joint_effort_target)
            env.actuator.compute(ca, d.joint_pos, d.joint_vel, joint_action=env._process_action(raw, None))

        launches[act_kernel] = (actuate, LV.DECIMATION - 1)
        launches["actuator_dc_motor_joint_action"] = (actuate_first_dc, 1)
        del launches["joint_action_process"]
        nbytes["actuator_dc_motor_joint_action"] = nbytes[act_kernel] + nbytes.pop("joint_action_process")
        if fused and env.paired_substeps:   # as for the network actuators: a substep's scene update and the next actuator launch as one
            def update_and_actuate_dc():
                ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                         joint_efforts=d.joint_effort_target)
                env.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=env.scene.describe_update(LV.PHYSICS_DT))

            launches["actuator_dc_motor+scene_update"] = (update_and_actuate_dc, LV.DECIMATION - 1)
            nbytes["actuator_dc_motor+scene_update"] = nbytes[act_kernel] + nbytes["scene_update"]
            launches[act_kernel], launches["scene_update"] = (actuate, 0), (launches["scene_update"][0], 0)
    if fused:   # the refresh's bytes dealt to the two launches: pose in / out and clocks with the update, the results with the cast
        nbytes["scene_update_prepare"] = nbytes["scene_update"] + 28 + 28 + 5
        nbytes["raycaster_cast"] = nbytes.pop("raycaster_refresh") - (28 + 28 + 5)
    cfg = Config("anymal" if robot == "anymal_c" else "h1", envs, lambda: env.step(raw), launches, nbytes, info,
                 env.mdp.log_sums, lambda buf: env.mdp.bind(log_sums=buf))
    cfg.env, cfg.state = env, state

    # The sensors' float32 clocks (sensor_base.py:199-200) decide which envs refresh.  A real run restarts them at every
    # env reset (sensor_base.py:182-195), so they stay below the episode length (20 s); this synthetic loop never
    # resets the scene, and beyond 16 s the clock's rounding makes `now - last + 1e-6 >= period` fail on some steps: the
    # scanner then refreshes every other step and the step gets cheaper.  The bench restarts the clocks ahead of every
    # timed region and counts the steps on which every env was refreshed.
    def restart_sensor_clocks():
        for sensor in (env.contact, env.scanner):
            sensor._timestamp.zero_()
            sensor._timestamp_last_update.zero_()
            sensor._is_outdated.fill_(False)

    def sensors_refreshed():
        return all(bool(torch.equal(s._timestamp, s._timestamp_last_update)) for s in (env.contact, env.scanner))

    cfg.restart_sensor_clocks, cfg.sensors_refreshed = restart_sensor_clocks, sensors_refreshed
    # replayed from a graph: the sensor chain as a parallel branch beside the actuator network (pays off for the network
    # actuators; with H1's PD actuators there is nothing to run beside)
    cfg.overlap_in_graph = robot == "anymal_c"
    return cfg


def build_cartpole(device, envs, seed):
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.envs.manager_based_step import CartpoleStep
    from isaaclab_amd.tasks import cartpole as CP

    g = torch.Generator().manual_seed(seed)
    n = envs
    env = CartpoleStep(n, device, seed=seed)
    # cart position U(-3.2, 3.2) (about 6 % beyond the +-3 m bound), pole angle U(-pi, pi)
    q0 = torch.stack([torch.rand(n, generator=g) * 6.4 - 3.2, (torch.rand(n, generator=g) * 2 - 1) * math.pi], -1)
    qd0 = torch.randn(n, 2, generator=g)
    env.data.joint_pos, env.data.joint_vel = q0.to(device), qd0.to(device)
    env.data._previous_joint_vel.copy_(qd0)
    env.mdp.episode_length_buf.copy_(torch.randint(0, CP.MAX_EPISODE_LENGTH, (n,), generator=g))
    root = torch.zeros(n, 13)
    root[:, 3] = 1.0
    env.load_physics(dict(joint_pos=torch.stack([q0 + 0.3 * torch.randn(n, 2, generator=g) for _ in range(CP.DECIMATION)]),
                          joint_vel=torch.stack([torch.randn(n, 2, generator=g) for _ in range(CP.DECIMATION)]), root_state_w=root))
    raw = (torch.rand(n, 1, generator=g) * 2 - 1).to(device)
    launches = {"joint_action_process": (lambda: env.process_action(raw), 1), "actuator_pd": (lambda: env.apply_action(first=False), CP.DECIMATION),
                "scene_update": (lambda: env.scene.update(CP.PHYSICS_DT), CP.DECIMATION),
                "mdp_step": (lambda: env.mdp.step(FM.PHASE_ALL, hold_step_count=True), 1)}
    nb = {
        "joint_action_process": 4 + 8 + 4,          # read raw; write action, previous action; write the scaled effort
        # effort target of the cart 4 + read q_des, qd_des, tau_ff, q, qd (5 x 8); write computed, applied (2 x 8)
        "actuator_pd": 4 + 7 * 8,
        "scene_update": 8 + 16 + 8,                 # joint_acc: read vel, read+write previous vel, write acc
        # read joint pos, vel 16; ep_len r+w 16; action + previous action (zeroed on reset) 8; write obs 16, reward 4,
        # step_reward 20, episode sums r+w 40, flags 5
        "mdp_step": 16 + 16 + 8 + 16 + 4 + 20 + 40 + 5,
    }
    _ride_on_first_actuator_launch(env, raw, launches, nb, CP.DECIMATION)
    info = {"decimation": CP.DECIMATION, "joints": 2, "bodies": len(CP.BODIES), "obs_dim": int(env.mdp.obs_dim),
            "actuator": "ImplicitActuator (PD estimate)"}
    cfg = Config("cartpole", n, lambda: env.step(raw), launches, nb, info, env.mdp.log_sums, lambda buf: env.mdp.bind(log_sums=buf))
    cfg.env = env
    return cfg


def build_reach(device, envs, seed, robot="franka"):
    from isaaclab_amd import synth
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.envs.manager_based_step import ReachStep
    from isaaclab_amd.tasks import reach as RE
    from isaaclab_amd.utils.string import resolve_matching_names_values

    g = torch.Generator().manual_seed(seed)
    n = envs
    env = ReachStep(n, device, arm_action="joint_pos", seed=seed, robot=robot)
    spec = RE.ROBOTS[robot]
    nj, nb_bodies, arm = len(spec["joints"]), len(spec["bodies"]), env.arm_dof
    if robot == "franka":
        default = torch.tensor([0.0, -0.569, 0.0, -2.81, 0.0, 3.037, 0.741, 0.04, 0.04]).repeat(n, 1)   # franka.py:38-48
    else:
        idx, _, vals = resolve_matching_names_values(RE.XARM6_DEFAULT_JOINT_POS, spec["joints"])
        default = torch.zeros(nj)
        default[idx] = torch.tensor(vals)
        default = default.repeat(n, 1)
    env.set_default_joint_pos(default)
    q = default + 0.3 * torch.randn(n, nj, generator=g)
    env.data.joint_pos, env.data.joint_vel = q.to(device), torch.zeros(n, nj, device=device)
    env.mdp.episode_length_buf.copy_(torch.randint(0, RE.MAX_EPISODE_LENGTH, (n,), generator=g))
    quat = torch.randn(n, 4, generator=g)
    env.mdp.command.copy_(torch.cat([torch.rand(n, 3, generator=g) * 0.5, quat / quat.norm(dim=-1, keepdim=True)], -1))
    env.mdp.time_left.copy_(torch.rand(n, generator=g) * 4.0)
    root = torch.zeros(n, 13)
    root[:, :3] = torch.randn(n, 3, generator=g)
    yaw = torch.rand(n, generator=g) * 6.28 - 3.14
    root[:, 3], root[:, 6] = torch.cos(yaw / 2), torch.sin(yaw / 2)
    body = torch.zeros(n, nb_bodies, 13)
    body[..., :3] = root[:, None, :3] + 0.4 * torch.randn(n, nb_bodies, 3, generator=g)
    bq = torch.randn(n, nb_bodies, 4, generator=g)
    body[..., 3:7] = bq / bq.norm(dim=-1, keepdim=True)
    q_steps = torch.stack([default + 0.2 * torch.randn(n, nj, generator=g) for _ in range(RE.DECIMATION)])
    if robot == "xarm6":   # the end effector where the arm's own forward kinematics puts it (urdf:79-225)
        ee_pos, ee_quat, _ = synth.serial_arm_kinematics(q_steps[-1][:, :arm].numpy(), root[:, :3].numpy(), root[:, 3:7].numpy())
        body[:, env.ee_body, :3], body[:, env.ee_body, 3:7] = torch.from_numpy(ee_pos), torch.from_numpy(ee_quat)
    env.load_physics(dict(joint_pos=q_steps, joint_vel=torch.stack([0.5 * torch.randn(n, nj, generator=g) for _ in range(RE.DECIMATION)]),
                          root_state_w=root, body_state_w=body))
    raw = (torch.rand(n, arm, generator=g) * 2 - 1).to(device)
    launches = {"joint_action_process": (lambda: env.process_action(raw), 1), "actuator_pd": (lambda: env.apply_action(first=False), RE.DECIMATION),
                "scene_update": (lambda: env.scene.update(RE.PHYSICS_DT), RE.DECIMATION),
                "mdp_step": (lambda: env.mdp.step(FM.PHASE_ALL, hold_step_count=True), 1)}
    f, fa = 4 * nj, 4 * arm
    nb = {
        "joint_action_process": 5 * fa,             # read raw, default; write action, previous action, arm target (arm joints)
        # arm target into the joint target + read q_des, qd_des, tau_ff, q, qd (5 x J); write computed, applied (2 x J)
        "actuator_pd": fa + 7 * f,
        "scene_update": f + 2 * f + f,              # joint_acc: read vel, read+write previous vel, write acc
        # read root 52, end-effector pose 28, joint pos / vel 2J, actions 2A; command r+w 56, time_left r+w 8, counter r+w
        # 16, ep_len r+w 16; write obs, reward 4, step_reward 20, episode sums r+w 40, flags 3
        "mdp_step": 52 + 28 + 2 * f + 2 * fa + 56 + 8 + 16 + 16 + 4 * int(env.mdp.obs_dim) + 4 + 20 + 40 + 3,
    }
    _ride_on_first_actuator_launch(env, raw, launches, nb, RE.DECIMATION)
    info = {"robot": robot, "decimation": RE.DECIMATION, "joints": nj, "bodies": nb_bodies, "obs_dim": int(env.mdp.obs_dim),
            "actuator": "ImplicitActuator (PD estimate)", "arm_action": "JointPositionAction scale 0.5", "mdp_kernel": env.mdp.kernel}
    cfg = Config("reach", n, lambda: env.step(raw), launches, nb, info, env.mdp.log_sums, lambda buf: env.mdp.bind(log_sums=buf))
    cfg.env = env
    return cfg


def build_shadow(device, envs, seed):
    import ctypes

    from isaaclab_amd import _abi, synth
    from isaaclab_amd.envs import inhand_step as IS

    n = envs
    s = synth.inhand_state(n, seed=seed)
    env = IS.InHandManipulationStep(IS.InHandManipulationCfg(act_moving_average=1.0, max_consecutive_success=0), n, device,
                                    num_joints=24, num_bodies=26, actuated_dof_indices=s["actuated"], finger_bodies=s["fingers"],
                                    lower_limits=s["lower"], upper_limits=s["upper"], in_hand_pos=s["in_hand_pos"],
                                    env_origins=s["env_origins"], seed=seed)
    for k in ("cur_targets", "prev_targets", "goal_rot", "successes", "consecutive_successes", "episode_length_buf"):
        getattr(env, k).copy_(s[k])
    env.set_sim_state(s["joint_pos"].to(device), s["joint_vel"].to(device), s["body_state"].to(device), s["object_state"].to(device))
    actions = s["actions"].to(device)
    env.step(actions)   # fills the descriptor the single launches below reuse

    def one(fn):
        return lambda: _abi.check(fn(ctypes.byref(env.desc), _abi.stream()))

    launches = {"inhand_apply_action": (one(IS._apply_action), env.cfg.decimation), "inhand_step": (one(IS._step), 1),
                "inhand_finish": (one(IS._finish), 1)}
    a, j, tips = 20, 24, 5
    nb = {
        # read actions 4A; read+write cur_targets and prev_targets (4J each way)
        "inhand_apply_action": 4 * a + 4 * 4 * j,
        # read joint pos / vel 8J, object state 52, fingertip states 52 each, goal_rot r+w 32, actions 4A, successes r+w 8,
        # reset_goal_buf r+w 2, ep_len r+w 16; write obs 4 x 157, reward 4, flags 3
        "inhand_step": 8 * j + 52 + 52 * tips + 32 + 4 * a + 8 + 2 + 16 + 4 * env.obs_dim + 4 + 3,
        "inhand_finish": 0,                          # two floats for the whole shard
    }
    info = {"decimation": env.cfg.decimation, "joints": j, "actuated": a, "bodies": 26, "fingertips": tips, "obs_dim": env.obs_dim,
            "actuator": "implicit (targets only)"}
    cfg = Config("shadow", n, lambda: env.step(actions), launches, nb, info)
    cfg.env = env
    return cfg


def build(name, device, envs=None, seed=0, actuator="mlp", static_scan=False, root_scatter="origins", robot="franka"):
    envs = envs or ENVS[name]
    if name == "anymal":
        return build_locomotion(device, envs, seed, "anymal_c", actuator, static_scan, root_scatter)
    if name == "h1":
        return build_locomotion(device, envs, seed, "h1", static_scan=static_scan, root_scatter=root_scatter)
    if name == "cartpole":
        return build_cartpole(device, envs, seed)
    if name == "reach":
        return build_reach(device, envs, seed, robot)
    if name == "shadow":
        return build_shadow(device, envs, seed)
    raise ValueError(f"unknown config '{name}'")
