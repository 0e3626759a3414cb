"""The non-physics half of one ``ManagerBasedRLEnv.step`` for the rough-terrain velocity tasks, on HIP kernels.

Sequence (``source/isaaclab/isaaclab/envs/manager_based_rl_env.py:152-241`` with ``sim.step`` removed):

    process_action                                  -> ilx_joint_action_process
    decimation x [ actuator model (write_data_to_sim)  -> ilx_actuator_net_mlp | ilx_actuator_pd
                   <physics: the caller supplies the next joint state / contact forces>
                   scene.update: joint_acc, contact    -> ilx_joint_acc, ilx_contact_update ]
    height scanner refresh (lazy in the reference)  -> ilx_raycaster_update
    terminations, rewards, manager resets, command, observations -> ilx_mdp_step (one launch)

All buffers have the reference's names and shapes (ArticulationData / ContactSensorData / RayCasterData / manager
buffers) and are exposed as torch tensors.
"""

from __future__ import annotations

import torch

from .. import _abi
from ..actuators import ActuatorNetLSTM, ActuatorNetLSTMCfg, ActuatorNetMLP, ActuatorNetMLPCfg, DCMotor, DCMotorCfg
from ..assets.articulation_data import ArticulationData
from ..sensors.contact_sensor import ContactSensor, ContactSensorCfg
from ..sensors.patterns import GridPatternCfg
from ..scene import InteractiveScene
from ..sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
from ..tasks import locomotion_velocity as LV
from ..utils.string import resolve_matching_names_values
from ..utils.types import ArticulationActions
from . import fused_mdp as FM


class LocomotionVelocityStep:
    def __init__(self, num_envs: int, device, mesh, robot: str = "anymal_c", actuator_net=None, seed: int = 0):
        self.n, self.device, self.robot = num_envs, torch.device(device), robot
        n, dev = num_envs, self.device
        if robot == "anymal_c":
            joints, bodies = LV.ANYMAL_JOINTS, LV.ANYMAL_BODIES
            terms = LV.anymal_c_rough_terms()
            default = LV.ANYMAL_DEFAULT_JOINT_POS
        elif robot == "h1":
            joints, bodies = LV.H1_JOINTS, LV.H1_BODIES
            terms = LV.h1_rough_terms()
            default = LV.H1_DEFAULT_JOINT_POS
        else:
            raise ValueError(f"unknown robot '{robot}'")
        self.joint_names, self.body_names = joints, bodies
        j, b = len(joints), len(bodies)
        self.data = ArticulationData(n, j, b, dev, joint_names=joints, body_names=bodies)
        idx, _, vals = resolve_matching_names_values(default, joints)
        self.data.default_joint_pos[:, idx] = torch.tensor(vals, device=dev)
        lim = torch.stack([self.data.default_joint_pos - 1.5, self.data.default_joint_pos + 1.5], dim=-1)
        mean, rng = (lim[..., 0] + lim[..., 1]) / 2, lim[..., 1] - lim[..., 0]
        factor = LV.H1_SOFT_JOINT_POS_LIMIT_FACTOR if robot == "h1" else 0.95
        self.data.soft_joint_pos_limits[..., 0] = mean - 0.5 * rng * factor  # articulation.py:1230-1237
        self.data.soft_joint_pos_limits[..., 1] = mean + 0.5 * rng * factor
        # the body the height scanner hangs on (None: the articulation root)
        self.scanner_body = bodies.index(LV.H1_SCANNER_BODY) if robot == "h1" else None
        # actuators (one group over all joints)
        if robot == "anymal_c":
            net = actuator_net
            if "lstm" in net:  # the ANYdrive 3 model of the reference asset (anymal.py:43-51)
                cfg = ActuatorNetLSTMCfg(network=net, **LV.ANYMAL_ACTUATOR)
                self.actuator = ActuatorNetLSTM(cfg, joint_names=joints, joint_ids=slice(None), num_envs=n, device=dev,
                                                stiffness=0.0, damping=0.0)
            else:
                cfg = ActuatorNetMLPCfg(network=net, **LV.ACTUATOR_NET, **LV.ANYMAL_ACTUATOR)
                self.actuator = ActuatorNetMLP(cfg, joint_names=joints, joint_ids=slice(None), num_envs=n, device=dev,
                                               stiffness=0.0, damping=0.0)
        else:
            cfg = DCMotorCfg(stiffness=LV.H1_STIFFNESS, damping=LV.H1_DAMPING, effort_limit=LV.H1_EFFORT_LIMIT,
                             velocity_limit=LV.H1_VELOCITY_LIMIT, saturation_effort=300.0)
            self.actuator = DCMotor(cfg, joint_names=joints, joint_ids=slice(None), num_envs=n, device=dev)
        # the group covers every joint: let the actuator write straight into the ArticulationData buffers
        self.actuator.computed_effort = self.data.computed_torque
        self.actuator.applied_effort = self.data.applied_torque
        self.data.soft_joint_vel_limits[:] = self.actuator.velocity_limit
        # sensors
        self.contact = ContactSensor(ContactSensorCfg(**LV.CONTACT_SENSOR), n, bodies, dev)
        self.contact.net_forces_from_history = True   # every term of these tasks reads the force history
        hs = LV.HEIGHT_SCAN
        self.scanner = RayCaster(RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, hs["offset_z"])), attach_yaw_only=True,
                                              pattern_cfg=GridPatternCfg(resolution=hs["resolution"], size=hs["size"]),
                                              update_period=hs["update_period"]), n, mesh, dev)
        self.scene = InteractiveScene(n, dev)
        self.scene.articulations["robot"] = self.data
        self.scene.sensors["contact_forces"] = self.contact
        self.scene.sensors["height_scanner"] = self.scanner
        # action manager buffers (action_manager.py:318-339) and the JointPositionAction constants (velocity_env_cfg.py:112)
        self.action = torch.zeros(n, j, device=dev)
        self.prev_action = torch.zeros(n, j, device=dev)
        self.action_scale = 0.5
        # managers: one fused plan
        self.mdp = FM.FusedMdp(n, dev, num_joints=j, num_bodies=b, num_sensor_bodies=b, history_len=LV.CONTACT_SENSOR["history_length"],
                               num_rays=self.scanner.num_rays, action_dim=j, command_dim=3, step_dt=LV.STEP_DT,
                               max_episode_length=LV.MAX_EPISODE_LENGTH,
                               command=LV.H1_VELOCITY_COMMAND if robot == "h1" else LV.VELOCITY_COMMAND, seed=seed, **terms)
        # the height_scan observation term's input, written by the scanner refresh with the term's offset
        self.height_scan = torch.zeros(n, self.scanner.num_rays, device=dev)
        self.height_scan_offset = next(float(t.aux) for t in self.mdp.observations if t.kind == "height_scan")
        self.scanner.bind_height_scan(self.height_scan, self.height_scan_offset)   # whoever triggers the refresh
        # the managers of these tasks read the height scan, not the hit records: the refresh leaves those to whoever reads
        # `scanner.data.ray_hits_w` (formed then, the same bits)
        self.scanner.lean_hits = True
        self._bind()
        # see step(): worth it when the step is replayed from a HIP graph and the actuator chain is long (network
        # actuators); issued launch by launch, the fork and join cost the host more than the overlap returns
        self.concurrent_chains = False
        # the managers add the done envs' statistics to mdp.log_sums; set, every step starts the vector at zero itself
        self.clear_log_sums = False
        self.on_physics = None      # callable(substep) or None: called where the simulator's step belongs (one-chain step)
        self._advance_count_first = False   # see count_steps_on_device
        self._side_streams: dict[int, torch.cuda.Stream] = {}   # by priority, see _fork_stream
        # the last scene update of a step also issues the preparation of the scanner's refresh (see step_sensors)
        self.fuse_scanner_preparation = True
        # the step as one chain with scene.update(k) and the actuator model of substep k + 1 as one launch (see
        # _step_paired; MLP and DC-motor actuators); concurrent_chains takes precedence
        self.paired_substeps = True
        # network actuators: process_action is done by the step's first actuator launch (ilx_actuator_net_mlp_joint_action)
        self.joint_action_rides = True

    def _bind(self):
        d, c = self.data, self.contact._data
        self.mdp.bind(root_state_w=d.root_state_w, joint_pos=d.joint_pos, joint_vel=d.joint_vel, joint_acc=d._joint_acc,
                      applied_torque=d.applied_torque, computed_torque=d.computed_torque, default_joint_pos=d.default_joint_pos,
                      default_joint_vel=d.default_joint_vel, soft_joint_pos_limits=d.soft_joint_pos_limits,
                      soft_joint_vel_limits=d.soft_joint_vel_limits, body_state_w=d.body_state_w,
                      net_forces_w_history=c.net_forces_w_history, current_air_time=c.current_air_time,
                      last_air_time=c.last_air_time, current_contact_time=c.current_contact_time,
                      last_contact_time=c.last_contact_time, ray_hits_w=self.scanner._data.hits_storage,
                      sensor_pos_w=self.scanner._data.pos_w, action=self.action, prev_action=self.prev_action,
                      height_scan=self.height_scan)

    def load_state(self, s: dict):
        """Install persistent buffers from :func:`isaaclab_amd.synth.locomotion_state` (host tensors)."""
        dev = self.device
        c = self.contact._data
        c.net_forces_w_history.copy_(s["net_forces_w_history"])
        for k in ("current_contact_time", "current_air_time", "last_air_time", "last_contact_time"):
            getattr(c, k).copy_(s[k])
        self.action.copy_(s["action"])
        self.prev_action.copy_(s["prev_action"])
        self.data._previous_joint_vel.copy_(s["previous_joint_vel"])
        m = self.mdp
        m.command.copy_(s["command"])
        m.heading_target.copy_(s["heading_target"])
        m.is_heading_env.copy_(s["is_heading_env"])
        m.is_standing_env.copy_(s["is_standing_env"])
        m.time_left.copy_(s["time_left"])
        m.command_counter.copy_(s["command_counter"])
        m.episode_length_buf.copy_(s["episode_length_buf"])
        if isinstance(self.actuator, ActuatorNetMLP):
            self.actuator._joint_pos_error_history.copy_(s["pos_error_history"])
            self.actuator._joint_vel_history.copy_(s["vel_history"])
        # clocks as after many steps: every sensor is due
        self.contact._is_outdated.fill_(False)
        self.scanner._is_outdated.fill_(False)
        self.physics = {k: s[k].to(dev).contiguous() for k in ("joint_pos", "joint_vel", "net_forces", "root_state_w")}
        if "body_state_w" in s:
            self.data.body_state_w.copy_(s["body_state_w"])

    def step(self, raw_action: torch.Tensor, noise_u=None, command_samples=None, reset_command_samples=None):
        """One env step on the synthetic physics sequence installed by :meth:`load_state`.

        With the simulator's results of all substeps at hand, the step is two chains that share no buffer until the
        managers read them: (A) process_action and the actuator model of every substep, :meth:`step_actuators`; (B)
        scene.update of every substep and the height-scanner refresh, :meth:`step_sensors`.  Each keeps the reference's
        order (manager_based_rl_env.py:172-195).  With ``concurrent_chains`` (B) is issued on a second stream between
        a fork and a join, so the memory-bound sensor updates can run beside the compute-bound actuator network; a
        caller that replays the step from HIP graphs can also record the three parts separately and launch (A) and (B)
        on two streams of its own."""
        if self.paired_substeps and isinstance(self.actuator, (ActuatorNetMLP, DCMotor)) and not self.concurrent_chains:
            return self._step_paired(raw_action, noise_u, command_samples, reset_command_samples)
        main = torch.cuda.current_stream()
        side = self._fork_stream(main) if self.concurrent_chains else main
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            self.step_sensors()
        # the step's opening (statistics vector cleared, device-side step count advanced) rides on the first launch of
        # chain (A), behind the fork: recorded in a graph, (B) waits for neither
        begin = None
        if self.clear_log_sums or self._advance_count_first:
            begin = self.mdp.begin_step_args(self.clear_log_sums, self._advance_count_first)
        self.step_actuators(raw_action, begin)
        if side is not main:
            main.wait_stream(side)
        return self.step_managers(noise_u, command_samples, reset_command_samples)

    @staticmethod
    def check_fork_priorities(main_priority: int, side_priority: int) -> None:
        """A fork / join between two streams of unequal priority, recorded into a HIP graph, stalled the graph's replay
        on this stack (a high-priority capture stream forking onto a normal one never came back): the step refuses to
        build that shape."""
        if main_priority != side_priority:
            raise _abi.IlxError(f"LocomotionVelocityStep: the step would fork from a stream of priority {main_priority} onto one of "
                                f"priority {side_priority}; both chains of a step must run at one priority")

    def _fork_stream(self, main):
        """The stream chain (B) is issued on beside ``main``: one per priority of the streams the caller steps on, made
        with that priority when first needed (outside any capture: creating a stream is not a stream operation)."""
        prio = int(main.priority)
        side = self._side_streams.get(prio)
        if side is None:
            side = self._side_streams[prio] = torch.cuda.Stream(device=self.device, priority=prio)
        self.check_fork_priorities(prio, int(side.priority))
        return side

    def _step_paired(self, raw_action, noise_u, command_samples, reset_command_samples):
        """The step as one chain in the reference's own order (manager_based_rl_env.py:172-195), with the launches that
        have no physics step between them paired: process_action; actuator(0); <physics 0>; {scene.update(0), actuator(1)}
        as one launch; <physics 1>; ... ; {scene.update(D-1), scanner preparation} as one launch; the scanner's cast; the
        managers.  Eight launches instead of twelve, and nothing runs ahead of a physics step it would depend on."""
        d, ph = self.data, self.physics
        begin = None
        if self.clear_log_sums or self._advance_count_first:
            begin = self.mdp.begin_step_args(self.clear_log_sums, self._advance_count_first)
        ride = self._process_action(raw_action, begin)
        prepared = False
        for k in range(LV.DECIMATION + 1):
            update = None
            if k > 0:   # the scene update of substep k - 1 (its physics results: ph[...][k - 1])
                d.joint_pos, d.joint_vel = ph["joint_pos"][k - 1], ph["joint_vel"][k - 1]
                self.contact.set_net_contact_forces(ph["net_forces"][k - 1])
                if k == LV.DECIMATION:
                    self._bind_scanner_pose()
                    prepared = self.scene.update(LV.PHYSICS_DT, prepare_scanner=self.fuse_scanner_preparation)
                    break
                update = self.scene.describe_update(LV.PHYSICS_DT)
                if update is None:
                    self.scene.update(LV.PHYSICS_DT)
            # the actuator model of substep k
            d.joint_pos, d.joint_vel = ph["joint_pos"][k], ph["joint_vel"][k]
            ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                     joint_efforts=d.joint_effort_target, joint_indices=slice(None))
            if k == 0 and ride is not None:
                self.actuator.compute(ca, d.joint_pos, d.joint_vel, joint_action=ride)
            else:
                self.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=update)
            if self.on_physics is not None:
                self.on_physics(k)     # <physics k>: a simulator steps here (see SegmentedCommandList)
        self.scanner._update_outdated_buffers(height_out=self.height_scan, height_offset=self.height_scan_offset, prepared=prepared)
        return self.step_managers(noise_u, command_samples, reset_command_samples)

    def count_steps_on_device(self):
        """Keep the managers' generator step count in device memory (for a step recorded in a HIP graph).  The step then
        advances the count itself, ahead of its launches at the head of chain (A), where the small add is off the
        step's critical path; the numbers drawn are those of the host-side count."""
        self.mdp.count_steps_on_device(start=-1)
        self._advance_count_first = True

    def step_sensors(self):
        """Chain (B): per substep ``scene.update`` (joint_acc, contact sensor, sensor clocks: one launch), then the
        height scanner's refresh (once per env step: update_period = step_dt; lazy in the reference)."""
        d = self.data
        prepared = False
        for k in range(LV.DECIMATION):
            # <sim.step would run here; its results for this substep are self.physics[...][k]>
            d.joint_pos, d.joint_vel = self.physics["joint_pos"][k], self.physics["joint_vel"][k]
            self.contact.set_net_contact_forces(self.physics["net_forces"][k])
            if k == LV.DECIMATION - 1:
                # the last physics step has left the pose the scanner will see: the update's launch also prepares the
                # scanner's refresh (one launch fewer between the last physics step and the ray hits)
                self._bind_scanner_pose()
                prepared = self.scene.update(LV.PHYSICS_DT, prepare_scanner=self.fuse_scanner_preparation)
            else:
                self.scene.update(LV.PHYSICS_DT)
        # the refresh also writes the height scan the observation term reads (sensor height - hit height - offset)
        self.scanner._update_outdated_buffers(height_out=self.height_scan, height_offset=self.height_scan_offset, prepared=prepared)

    def _bind_scanner_pose(self):
        d = self.data
        d.set_sim_state(root_state_w=self.physics["root_state_w"])
        if self.scanner_body is None:
            self.scanner.set_pose(d.root_state_w[:, :3], d.root_state_w[:, 3:7])
        else:  # strided views of the body's rows: the kernel reads them in place
            pose = d.body_state_w[:, self.scanner_body]
            self.scanner.set_pose(pose[:, :3], pose[:, 3:7])

    def step_actuators(self, raw_action: torch.Tensor, begin=None):
        """Chain (A): process_action (manager_based_rl_env.py:172), then the actuator model of every substep
        (write_data_to_sim).  ``begin``: ``FusedMdp.begin_step_args`` -- the launch then also opens the step."""
        d = self.data
        ride = self._process_action(raw_action, begin)
        for k in range(LV.DECIMATION):
            d.joint_pos, d.joint_vel = self.physics["joint_pos"][k], self.physics["joint_vel"][k]
            ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                     joint_efforts=d.joint_effort_target, joint_indices=slice(None))
            if k == 0 and ride is not None:
                self.actuator.compute(ca, d.joint_pos, d.joint_vel, joint_action=ride)
            else:
                self.actuator.compute(ca, d.joint_pos, d.joint_vel)

    def _process_action(self, raw_action, begin):
        """process_action (manager_based_rl_env.py:172) -- as a launch of its own, or, for the network actuators, as
        arguments for the first actuator launch, which then does it on the way: returns those, or None when launched."""
        d = self.data
        if self.joint_action_rides and isinstance(self.actuator, ActuatorNetMLP):
            return (raw_action, self.action, self.prev_action, self.action_scale, d.default_joint_pos, begin)
        if self.joint_action_rides and isinstance(self.actuator, DCMotor):   # ilx_actuator_pd_joint_action
            return dict(raw=raw_action, action=self.action, prev_action=self.prev_action, scale=self.action_scale,
                        offset=d.default_joint_pos, target="position", first_joint=0, width=d._j, begin=begin)
        args = (_abi.ptr(raw_action), _abi.ptr(self.action), _abi.ptr(self.prev_action), None, self.action_scale,
                _abi.ptr(d.default_joint_pos), 0.0, None, _abi.ptr(d.joint_pos_target), self.n, d._j)
        if begin is not None:
            _abi.check(_abi.joint_action_process_begin(*args, *begin, _abi.stream()), "ilx_joint_action_process_begin")
        else:
            _abi.check(_abi.joint_action_process(*args, _abi.stream()), "ilx_joint_action_process")
        return None

    def step_managers(self, noise_u=None, command_samples=None, reset_command_samples=None, phases: int = FM.PHASE_ALL):
        """The post-physics managers in one launch: terminations, rewards, resets, command, observations -- or, with
        ``phases``, the launch of a step issued as two: everything but the observations (needs no ray hit), then the
        observations."""
        d = self.data
        d.joint_pos, d.joint_vel = self.physics["joint_pos"][LV.DECIMATION - 1], self.physics["joint_vel"][LV.DECIMATION - 1]
        if d.root_state_w.data_ptr() != self.physics["root_state_w"].data_ptr():
            d.set_sim_state(root_state_w=self.physics["root_state_w"])
        m = self.mdp
        m.bind(root_state_w=d.root_state_w, joint_pos=d.joint_pos, joint_vel=d.joint_vel, noise_u=noise_u,
               command_samples=command_samples, reset_command_samples=reset_command_samples,
               sensor_pos_w=self.scanner._data.pos_w, ray_hits_w=self.scanner._data.hits_storage)
        m.step(phases, hold_step_count=self._advance_count_first or not (phases & FM.PHASE_OBSERVATION))
        return m.obs, m.reward_buf, m.terminated, m.time_outs
