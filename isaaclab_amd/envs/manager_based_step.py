"""Non-physics half of ``ManagerBasedRLEnv.step`` (``isaaclab/envs/manager_based_rl_env.py:152-241``) for a scene with
one articulation, as a small base class plus the Cartpole and Franka-Reach configurations.

    process_action                                   -> ilx_joint_action_process | ilx_diff_ik_set_command
    decimation x [ apply_action + actuator model     -> ilx_diff_ik_compute, ilx_actuator_pd
                   <physics: the caller supplies the next joint state>
                   scene.update                      -> ilx_scene_update ]
    terminations, rewards, manager resets, command, observations -> ilx_mdp_step (one launch)

The rough-terrain locomotion configs (contact sensor, ray caster, actuator net) are
:class:`isaaclab_amd.envs.locomotion_step.LocomotionVelocityStep`.
"""

from __future__ import annotations

import ctypes
import types

import torch

from .. import _abi
from ..actuators import ImplicitActuator, ImplicitActuatorCfg
from ..assets.articulation_data import ArticulationData
from ..controllers.differential_ik import DifferentialIKControllerCfg
from ..scene import InteractiveScene
from ..tasks import cartpole as CP
from ..tasks import reach as RE
from ..utils.string import resolve_matching_names
from ..utils.types import ArticulationActions
from . import fused_mdp as FM
from .mdp.actions import DifferentialInverseKinematicsAction, DifferentialInverseKinematicsActionCfg, OffsetCfg


class _Asset:
    """What the action terms need from an articulation (names, data, Jacobian, target setter)."""

    is_fixed_base = True

    def __init__(self, data: ArticulationData):
        self.data = data
        self.joint_names, self.body_names = data.joint_names, data.body_names
        self.num_joints, self.num_bodies = len(self.joint_names), len(self.body_names)
        self.jacobians = None

    def find_joints(self, names, preserve_order=False):
        return resolve_matching_names(names, self.joint_names, preserve_order)

    def find_bodies(self, names, preserve_order=False):
        return resolve_matching_names(names, self.body_names, preserve_order)

    def get_jacobians(self):
        return self.jacobians

    def set_joint_position_target(self, target, joint_ids=None):
        self.data.joint_pos_target[:, joint_ids if joint_ids is not None else slice(None)] = target

    def set_joint_velocity_target(self, target, joint_ids=None):
        self.data.joint_vel_target[:, joint_ids if joint_ids is not None else slice(None)] = target

    def set_joint_effort_target(self, target, joint_ids=None):
        self.data.joint_effort_target[:, joint_ids if joint_ids is not None else slice(None)] = target


class ManagerBasedStep:
    """Shared sequence; subclasses provide the action term and the actuator groups."""

    def __init__(self, num_envs: int, device, joints, bodies, terms: dict, *, action_dim: int, step_dt: float, physics_dt: float,
                 decimation: int, max_episode_length: int, command: dict | None = None, command_dim: int = 0, seed: int = 0):
        self.n, self.device = num_envs, torch.device(device)
        self.step_dt, self.physics_dt, self.decimation = step_dt, physics_dt, decimation
        n, dev = num_envs, self.device
        self.data = ArticulationData(n, len(joints), len(bodies), dev, joint_names=joints, body_names=bodies)
        self.asset = _Asset(self.data)
        self.scene = InteractiveScene(n, dev)
        self.scene.articulations["robot"] = self.data
        self.action = torch.zeros(n, action_dim, device=dev)        # action_manager.py:318-339
        self.prev_action = torch.zeros(n, action_dim, device=dev)
        self.mdp = FM.FusedMdp(n, dev, num_joints=len(joints), num_bodies=len(bodies), action_dim=action_dim,
                               command_dim=command_dim, step_dt=step_dt, max_episode_length=max_episode_length, command=command,
                               seed=seed, **terms)
        self.physics = None
        # the managers add the done envs' statistics to mdp.log_sums; set, every step starts the vector at zero itself
        self.clear_log_sums = False
        self.on_physics = None      # callable(substep) or None: called where the simulator's step belongs
        self._advance_count_first = False   # see count_steps_on_device

    def count_steps_on_device(self):
        """Keep the managers' generator step count in device memory (for a step recorded in a HIP graph); the step then
        advances the count itself, in the one launch that also clears the statistics vector.  The numbers drawn are
        those of the host-side count."""
        self.mdp.count_steps_on_device(start=-1)
        self._advance_count_first = True

    def load_physics(self, physics: dict):
        """Synthetic simulator output of one env step: ``joint_pos`` / ``joint_vel`` (decimation, N, J) per substep,
        ``root_state_w`` (N,13), ``body_state_w`` (N,B,13) and optionally ``jacobians`` at the end of the step."""
        dev = self.device
        self.physics = {k: v.to(dev).contiguous() for k, v in physics.items()}

    opens_with_joint_action = False   # process_action is one ilx_joint_action_process launch (see _joint_action)
    _begin = None
    # True: inside step() the action term's processing and the manager's bookkeeping are part of the first substep's
    # actuator launch (ilx_actuator_pd_joint_action) instead of a launch of their own plus a copy into the target buffer
    joint_action_rides = True
    _ride = None

    def joint_action_ride(self, raw):
        """The description of the action term for ``actuator.compute(..., joint_action=)``, or None when the term cannot
        ride (then process_action / apply_action run as launches of their own)."""
        return None

    def process_action(self, raw):
        raise NotImplementedError

    def _joint_action(self, raw, scale, offset, processed, width):
        """JointAction.process_actions + the ActionManager's bookkeeping as one launch; inside :meth:`step` the launch
        also opens the step (``ilx_joint_action_process_begin``)."""
        args = (_abi.ptr(raw), _abi.ptr(self.action), _abi.ptr(self.prev_action), None, scale,
                _abi.ptr(offset) if offset is not None else None, 0.0, None, _abi.ptr(processed), self.n, width)
        if self._begin is not None:
            _abi.check(_abi.joint_action_process_begin(*args, *self._begin, _abi.stream()), "ilx_joint_action_process_begin")
        else:
            _abi.check(_abi.joint_action_process(*args, _abi.stream()), "ilx_joint_action_process")

    paired_substeps = True   # see step()

    def apply_action(self, scene_update=None, first=True):
        """``first``: the step's first substep.  The processed action does not change between the substeps of a step, so
        the targets that an action term's apply_actions sets again in every substep (joint_actions.py:162-164) are
        written once."""
        raise NotImplementedError

    def step(self, raw_action: torch.Tensor, noise_u=None, command_samples=None, reset_command_samples=None):
        d, ph = self.data, self.physics
        # the step's opening (statistics vector cleared, device-side step count advanced) rides on the first launch where
        # that is the joint action's, else it is a small launch of its own
        self._begin = None
        if self.clear_log_sums or self._advance_count_first:
            if self.opens_with_joint_action:
                self._begin = self.mdp.begin_step_args(self.clear_log_sums, self._advance_count_first)
            else:
                self.mdp.begin_step(clear_log_sums=self.clear_log_sums, advance_step_count=self._advance_count_first)
        # a joint action term with consecutive joints rides on the first substep's actuator launch (joint_action_ride)
        self._ride = self.joint_action_ride(_abi.f32c(raw_action)) if self.joint_action_rides else None
        if self._ride is None:
            self.process_action(raw_action)
        elif self._begin is None and (self.clear_log_sums or self._advance_count_first):
            raise RuntimeError("a step whose joint action rides on the actuator launch opens with it")
        self._begin = None
        update = None
        for k in range(self.decimation):
            # action terms' apply_actions + write_data_to_sim (actuators); from the second substep on the launch of the
            # actuator model also carries the scene update of the substep before (no physics step lies between them)
            self.apply_action(update, first=k == 0)
            if self.on_physics is not None:
                self.on_physics(k)     # <sim.step>: a simulator steps here (see SegmentedCommandList)
            d.joint_pos, d.joint_vel = ph["joint_pos"][k], ph["joint_vel"][k]   # its results, here synthetic
            # (the last substep's update travels with the managers' launch)
            update = self.scene.describe_update(self.physics_dt) if self.paired_substeps else None
            if update is None:
                self.scene.update(self.physics_dt)
        d.set_sim_state(root_state_w=ph["root_state_w"], body_state_w=ph.get("body_state_w"))
        m = self.mdp
        m.bind(root_state_w=d.root_state_w, joint_pos=d.joint_pos, joint_vel=d.joint_vel, joint_acc=d._joint_acc,
               applied_torque=d.applied_torque, computed_torque=d.computed_torque, default_joint_pos=d.default_joint_pos,
               default_joint_vel=d.default_joint_vel, soft_joint_pos_limits=d.soft_joint_pos_limits,
               soft_joint_vel_limits=d.soft_joint_vel_limits, body_state_w=d.body_state_w, action=self.action,
               prev_action=self.prev_action, noise_u=noise_u, command_samples=command_samples,
               reset_command_samples=reset_command_samples)
        m.step(FM.PHASE_ALL, hold_step_count=self._advance_count_first, scene_update=update)
        return m.obs, m.reward_buf, m.terminated, m.time_outs

    def _actuate(self, actuators):
        """articulation.py:1399-1436 for actuator groups that together cover every joint."""
        d = self.data
        for act, ids in actuators:
            ca = ArticulationActions(joint_positions=d.joint_pos_target[:, ids].contiguous(),
                                     joint_velocities=d.joint_vel_target[:, ids].contiguous(),
                                     joint_efforts=d.joint_effort_target[:, ids].contiguous(), joint_indices=ids)
            act.compute(ca, d.joint_pos[:, ids].contiguous(), d.joint_vel[:, ids].contiguous())
            d.computed_torque[:, ids] = act.computed_effort
            d.applied_torque[:, ids] = act.applied_effort


class CartpoleStep(ManagerBasedStep):
    """Isaac-Cartpole-v0 (cartpole_env_cfg.py): JointEffortAction on the cart (scale 100), implicit actuators
    (isaaclab_assets/robots/cartpole.py:39-50), five rewards, two terminations, 4-wide observation."""

    opens_with_joint_action = True

    def __init__(self, num_envs: int, device, seed: int = 0):
        super().__init__(num_envs, device, CP.JOINTS, CP.BODIES, CP.cartpole_terms(), action_dim=1, step_dt=CP.STEP_DT,
                         physics_dt=CP.PHYSICS_DT, decimation=CP.DECIMATION, max_episode_length=CP.MAX_EPISODE_LENGTH, seed=seed)
        cfg = ImplicitActuatorCfg(effort_limit=400.0, velocity_limit=100.0, stiffness=0.0, damping={"slider_to_cart": 10.0, "cart_to_pole": 0.0})
        self.actuator = ImplicitActuator(cfg, joint_names=CP.JOINTS, joint_ids=slice(None), num_envs=num_envs, device=self.device)
        self.actuator.computed_effort, self.actuator.applied_effort = self.data.computed_torque, self.data.applied_torque
        self._effort = torch.zeros(num_envs, 1, device=self.device)

    def process_action(self, raw):  # joint_actions.py:130-139, 214-225 (JointEffortAction, no offset)
        self._joint_action(_abi.f32c(raw), CP.ACTION_SCALE, None, self._effort, 1)

    def joint_action_ride(self, raw):   # the cart's effort target = raw * 100 (joint_actions.py:214-225)
        return dict(raw=raw, action=self.action, prev_action=self.prev_action, scale=CP.ACTION_SCALE, offset=None,
                    target="effort", first_joint=0, width=1, begin=self._begin)

    def apply_action(self, scene_update=None, first=True):
        d = self.data
        ride = self._ride if first else None
        if first and ride is None:
            d.joint_effort_target[:, 0:1] = self._effort      # set_joint_effort_target(processed, joint_ids=[cart])
        ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                 joint_efforts=d.joint_effort_target, joint_indices=slice(None))
        self.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=scene_update, joint_action=ride)
        self._ride = None


class ReachStep(ManagerBasedStep):
    """Isaac-Reach-Franka-v0 and its IK variants (reach_env_cfg.py + config/franka/*): arm action = JointPositionAction
    (scale 0.5, default offset) or the DifferentialIK action term (relative / absolute pose, DLS), implicit actuators
    (isaaclab_assets/robots/franka.py:50-72), UniformPoseCommand, five rewards, time-out, 32-wide observation
    (9 + 9 + 7 + arm action width).  ``robot="xarm6"``: the same task on the fork's UFactory xArm6 with gripper (8 joints,
    10 bodies, implicit actuators 2000 / 100; 29-wide observation with the 6-wide joint-position or relative-pose action)."""

    def __init__(self, num_envs: int, device, arm_action: str = "joint_pos", seed: int = 0, robot: str = "franka"):
        if arm_action not in ("joint_pos", "ik_rel", "ik_abs"):
            raise ValueError(f"unknown arm action '{arm_action}'")
        if robot not in RE.ROBOTS:
            raise ValueError(f"unknown robot '{robot}' (known: {sorted(RE.ROBOTS)})")
        self.arm_action, self.robot = arm_action, robot
        spec = RE.ROBOTS[robot]
        joints, bodies = spec["joints"], spec["bodies"]
        arm = resolve_matching_names(spec["arm_joints"], joints)[0]
        self._arm, self.arm_dof = arm, len(arm)
        adim = {"joint_pos": self.arm_dof, "ik_rel": 6, "ik_abs": 7}[arm_action]
        self.ee_body = bodies.index(spec["ee_body"])
        cmd = dict(RE.POSE_COMMAND, body_index=self.ee_body)
        super().__init__(num_envs, device, joints, bodies, RE.reach_terms(bodies, spec["ee_body"]), action_dim=adim, step_dt=RE.STEP_DT,
                         physics_dt=RE.PHYSICS_DT, decimation=RE.DECIMATION, max_episode_length=RE.MAX_EPISODE_LENGTH,
                         command=cmd, command_dim=7, seed=seed)
        n, dev = num_envs, self.device
        # the same joints as an index that costs no host-to-device copy per use: a slice when they are consecutive
        self._arm_index = (slice(arm[0], arm[-1] + 1) if arm == list(range(arm[0], arm[-1] + 1)) else
                           torch.tensor(arm, dtype=torch.long, device=dev))
        cfg = ImplicitActuatorCfg(**spec["actuators"])
        self.actuator = ImplicitActuator(cfg, joint_names=joints, joint_ids=slice(None), num_envs=n, device=dev)
        self.actuator.computed_effort, self.actuator.applied_effort = self.data.computed_torque, self.data.applied_torque
        self._arm_target = torch.zeros(n, self.arm_dof, device=dev)
        self._arm_default = torch.zeros(n, self.arm_dof, device=dev)
        self.ik = None
        self.opens_with_joint_action = arm_action == "joint_pos"
        if arm_action != "joint_pos":
            env = types.SimpleNamespace(num_envs=n, device=dev, scene={"robot": self.asset})
            offset = None if spec["ik_body_offset"] is None else OffsetCfg(*spec["ik_body_offset"])
            ik_cfg = DifferentialInverseKinematicsActionCfg(
                joint_names=spec["arm_joints"], body_name=spec["ee_body"], body_offset=offset,
                scale=0.5 if arm_action == "ik_rel" else 1.0,   # ik_rel_env_cfg.py:29-36 / ik_abs_env_cfg.py:29-34
                controller=DifferentialIKControllerCfg("pose", arm_action == "ik_rel", "dls"))
            self.ik = DifferentialInverseKinematicsAction(ik_cfg, env)

    def set_default_joint_pos(self, default: torch.Tensor):
        self.data.default_joint_pos[:] = default.to(self.device)
        self._arm_default = self.data.default_joint_pos[:, self._arm_index].contiguous()

    def load_physics(self, physics: dict):
        super().load_physics(physics)
        # the pose the action terms see while this step's actions are processed is the one of the previous step's end
        if "body_state_w" in self.physics:
            self.data.set_sim_state(root_state_w=self.physics["root_state_w"], body_state_w=self.physics["body_state_w"])
        self.asset.jacobians = self.physics.get("jacobians")

    def process_action(self, raw):
        raw = _abi.f32c(raw)
        if self.ik is None:  # JointPositionAction: processed = raw * 0.5 + default (joint_actions.py:145-160)
            self._joint_action(raw, RE.ARM_ACTION_SCALE, self._arm_default, self._arm_target, self.arm_dof)
        else:  # ActionManager bookkeeping (action_manager.py:331-332), then the term
            self.prev_action.copy_(self.action)
            self.action.copy_(raw)
            self.ik.process_actions(raw)

    def joint_action_ride(self, raw):   # JointPositionAction on consecutive arm joints: target = raw * 0.5 + default
        if self.ik is not None or not isinstance(self._arm_index, slice):
            return None
        return dict(raw=raw, action=self.action, prev_action=self.prev_action, scale=RE.ARM_ACTION_SCALE, offset=self._arm_default,
                    target="position", first_joint=self._arm[0], width=self.arm_dof, begin=self._begin)

    def apply_action(self, scene_update=None, first=True):
        d = self.data
        ride = self._ride if first else None
        self._ride = None
        if self.ik is None:
            if first and ride is None:
                d.joint_pos_target[:, self._arm_index] = self._arm_target
        else:
            if scene_update is not None:   # the IK term reads what the update writes (joint state of the new substep): in order
                _abi.check(_abi.scene_update(ctypes.byref(scene_update), _abi.stream()), "ilx_scene_update")
                scene_update = None
            self.ik.apply_actions()
        ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                 joint_efforts=d.joint_effort_target, joint_indices=slice(None))
        self.actuator.compute(ca, d.joint_pos, d.joint_vel, scene_update=scene_update, joint_action=ride)
