"""Operational-space controller on the HIP kernels ``ilx_osc_set_command`` / ``ilx_osc_compute`` (interface of
``isaaclab/controllers/operational_space.py`` and ``operational_space_cfg.py``).

The controller keeps the reference's buffers under the reference's names (gains and selection matrices in the task and
root frames, targets, operational-space mass matrix); ``set_command`` and ``compute`` are one launch each.
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence
from ctypes import c_float, c_int32, c_int64, c_void_p
from dataclasses import dataclass

import torch

from .. import _abi

_TARGETS = {"pose_rel": (0, 6), "pose_abs": (1, 7), "wrench_abs": (2, 6)}
_IMPEDANCE = {"fixed": 0, "variable_kp": 1, "variable": 2}


@dataclass
class OperationalSpaceControllerCfg:
    """operational_space_cfg.py."""

    target_types: Sequence[str] = None
    motion_control_axes_task: Sequence[int] = (1, 1, 1, 1, 1, 1)
    contact_wrench_control_axes_task: Sequence[int] = (0, 0, 0, 0, 0, 0)
    inertial_dynamics_decoupling: bool = False
    partial_inertial_dynamics_decoupling: bool = False
    gravity_compensation: bool = False
    impedance_mode: str = "fixed"
    motion_stiffness_task: float | Sequence[float] = (100.0, 100.0, 100.0, 100.0, 100.0, 100.0)
    motion_damping_ratio_task: float | Sequence[float] = (1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
    motion_stiffness_limits_task: tuple[float, float] = (0, 1000)
    motion_damping_ratio_limits_task: tuple[float, float] = (0, 100)
    contact_wrench_stiffness_task: float | Sequence[float] | None = None
    nullspace_control: str = "none"
    nullspace_stiffness: float = 10.0
    nullspace_damping_ratio: float = 1.0

    @property
    def class_type(self):
        return OperationalSpaceController


class OscDesc(ctypes.Structure):
    """Mirror of ``ilx_osc`` (include/isaaclab_amd_osc.h)."""

    _fields_ = [
        ("num_envs", c_int64), ("num_targets", c_int32), ("target_types", c_int32 * 3), ("impedance_mode", c_int32),
        ("inertial_dynamics_decoupling", c_int32), ("partial_inertial_dynamics_decoupling", c_int32),
        ("gravity_compensation", c_int32), ("nullspace_position", c_int32),
        ("motion_control_axes_task", c_float * 6), ("contact_wrench_control_axes_task", c_float * 6),
        ("motion_damping_ratio_task", c_float * 6), ("motion_stiffness_limits_task", c_float * 2),
        ("motion_damping_ratio_limits_task", c_float * 2), ("nullspace_p_gain", c_float), ("nullspace_d_gain", c_float),
        ("task_space_target_task", c_void_p), ("motion_p_gains_task", c_void_p), ("motion_d_gains_task", c_void_p),
        ("motion_p_gains_b", c_void_p), ("motion_d_gains_b", c_void_p), ("contact_wrench_p_gains_task", c_void_p),
        ("contact_wrench_p_gains_b", c_void_p), ("selection_matrix_motion_b", c_void_p), ("selection_matrix_force_b", c_void_p),
        ("desired_ee_pose_task", c_void_p), ("desired_ee_pose_b", c_void_p), ("desired_ee_wrench_task", c_void_p),
        ("desired_ee_wrench_b", c_void_p), ("os_mass_matrix_b", c_void_p), ("ee_contact_wrench_b", c_void_p),
    ]


_set_command = _abi._sig("ilx_osc_set_command", ctypes.POINTER(OscDesc), _abi.P, _abi.c_int, _abi.P, _abi.P, _abi.P)
_compute = _abi._sig("ilx_osc_compute", ctypes.POINTER(OscDesc), _abi.c_int, _abi.c_int, _abi.P, _abi.c_int, _abi.P, _abi.P, _abi.P,
                     _abi.P, _abi.P, _abi.P, _abi.P, _abi.P, _abi.P, _abi.P)


def _six(v, device):
    t = torch.as_tensor(v, dtype=torch.float, device=device)
    return t.expand(6).clone() if t.dim() == 0 else t


class OperationalSpaceController:
    def __init__(self, cfg: OperationalSpaceControllerCfg, num_envs: int, device):
        self.cfg, self.num_envs, self._device = cfg, num_envs, torch.device(device)
        n, dev = num_envs, self._device
        self.target_list = []
        for command_type in cfg.target_types:  # operational_space.py:52-61
            if command_type not in _TARGETS:
                raise ValueError(f"Invalid control command: {command_type}.")
            self.target_list.append(_TARGETS[command_type][1])
        self.target_dim = sum(self.target_list)
        if cfg.impedance_mode not in _IMPEDANCE:
            raise ValueError(f"Invalid impedance mode: {cfg.impedance_mode}.")
        if cfg.nullspace_control not in ("none", "position"):
            raise ValueError(f"Invalid null-space control method: {cfg.nullspace_control}.")
        axes_m = torch.tensor(cfg.motion_control_axes_task, dtype=torch.float, device=dev)
        axes_f = torch.tensor(cfg.contact_wrench_control_axes_task, dtype=torch.float, device=dev)
        self._selection_matrix_motion_task = torch.diag_embed(axes_m.unsqueeze(0).repeat(n, 1))
        self._selection_matrix_force_task = torch.diag_embed(axes_f.unsqueeze(0).repeat(n, 1))
        self._selection_matrix_motion_b = torch.zeros(n, 6, 6, device=dev)
        self._selection_matrix_force_b = torch.zeros(n, 6, 6, device=dev)
        self._task_space_target_task = torch.zeros(n, self.target_dim, device=dev)
        self._has_pose = self._has_wrench = False
        self._desired_ee_pose_task = torch.zeros(n, 7, device=dev)
        self._desired_ee_pose_b = torch.zeros(n, 7, device=dev)
        self._desired_ee_wrench_task = torch.zeros(n, 6, device=dev)
        self._desired_ee_wrench_b = torch.zeros(n, 6, device=dev)
        self._os_mass_matrix_b = torch.zeros(n, 6, 6, device=dev)
        # gains (:87-121): p zeroed on uncontrolled axes, d = 2 sqrt(p) ratio
        ratio = _six(cfg.motion_damping_ratio_task, dev)
        p = axes_m * _six(cfg.motion_stiffness_task, dev)
        self._motion_p_gains_task = torch.diag_embed(p.unsqueeze(0).repeat(n, 1)).contiguous()
        self._motion_d_gains_task = torch.diag_embed((2 * p.sqrt() * ratio).unsqueeze(0).repeat(n, 1)).contiguous()
        self._motion_p_gains_b = torch.zeros(n, 6, 6, device=dev)
        self._motion_d_gains_b = torch.zeros(n, 6, 6, device=dev)
        if cfg.contact_wrench_stiffness_task is not None:
            w = axes_f * _six(cfg.contact_wrench_stiffness_task, dev)
            self._contact_wrench_p_gains_task = torch.diag_embed(w.unsqueeze(0).repeat(n, 1)).contiguous()
            self._contact_wrench_p_gains_b = torch.zeros(n, 6, 6, device=dev)
        else:
            self._contact_wrench_p_gains_task = self._contact_wrench_p_gains_b = None
        self._ee_contact_wrench_b = torch.zeros(n, 6, device=dev)
        self._nullspace_p_gain = torch.tensor(cfg.nullspace_stiffness, dtype=torch.float, device=dev)
        self._nullspace_d_gain = 2 * torch.sqrt(self._nullspace_p_gain) * torch.tensor(cfg.nullspace_damping_ratio, dtype=torch.float,
                                                                                        device=dev)
        d = self._desc = OscDesc()
        d.num_envs, d.num_targets = n, len(self.target_list)
        for i, t in enumerate(cfg.target_types):
            d.target_types[i] = _TARGETS[t][0]
        d.impedance_mode = _IMPEDANCE[cfg.impedance_mode]
        d.inertial_dynamics_decoupling = int(cfg.inertial_dynamics_decoupling)
        d.partial_inertial_dynamics_decoupling = int(cfg.partial_inertial_dynamics_decoupling)
        d.gravity_compensation = int(cfg.gravity_compensation)
        d.nullspace_position = int(cfg.nullspace_control == "position")
        d.motion_control_axes_task = (c_float * 6)(*axes_m.tolist())
        d.contact_wrench_control_axes_task = (c_float * 6)(*axes_f.tolist())
        d.motion_damping_ratio_task = (c_float * 6)(*ratio.tolist())
        d.motion_stiffness_limits_task = (c_float * 2)(*[float(v) for v in cfg.motion_stiffness_limits_task])
        d.motion_damping_ratio_limits_task = (c_float * 2)(*[float(v) for v in cfg.motion_damping_ratio_limits_task])
        d.nullspace_p_gain, d.nullspace_d_gain = float(self._nullspace_p_gain), float(self._nullspace_d_gain)
        for name in ("task_space_target_task", "motion_p_gains_task", "motion_d_gains_task", "motion_p_gains_b", "motion_d_gains_b",
                     "contact_wrench_p_gains_task", "contact_wrench_p_gains_b", "selection_matrix_motion_b", "selection_matrix_force_b",
                     "desired_ee_pose_task", "desired_ee_pose_b", "desired_ee_wrench_task", "desired_ee_wrench_b", "os_mass_matrix_b",
                     "ee_contact_wrench_b"):
            setattr(d, name, _abi.ptr(getattr(self, "_" + name)))

    # -- the reference's optional attributes: None until a target of that kind was set
    @property
    def desired_ee_pose_task(self):
        return self._desired_ee_pose_task if self._has_pose else None

    @property
    def desired_ee_pose_b(self):
        return self._desired_ee_pose_b if self._has_pose else None

    @property
    def desired_ee_wrench_task(self):
        return self._desired_ee_wrench_task if self._has_wrench else None

    @property
    def desired_ee_wrench_b(self):
        return self._desired_ee_wrench_b if self._has_wrench else None

    @property
    def action_dim(self) -> int:
        """operational_space.py:146-160."""
        return self.target_dim + {"fixed": 0, "variable_kp": 6, "variable": 12}[self.cfg.impedance_mode]

    def reset(self):
        self._has_pose = self._has_wrench = False

    def set_command(self, command: torch.Tensor, current_ee_pose_b: torch.Tensor | None = None,
                    current_task_frame_pose_b: torch.Tensor | None = None):
        """operational_space.py:173-343."""
        if command.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"Invalid command shape '{command.shape}'. Expected: '{(self.num_envs, self.action_dim)}'.")
        if "pose_rel" in self.cfg.target_types and current_ee_pose_b is None:
            raise ValueError("Current pose is required for 'pose_rel' command.")
        f = _abi.f32c
        _abi.check(_set_command(ctypes.byref(self._desc), _abi.ptr(f(command)), self.action_dim,
                                None if current_ee_pose_b is None else _abi.ptr(f(current_ee_pose_b)),
                                None if current_task_frame_pose_b is None else _abi.ptr(f(current_task_frame_pose_b)), _abi.stream()),
                   "ilx_osc_set_command")
        self._has_pose = any(t in ("pose_rel", "pose_abs") for t in self.cfg.target_types) or self._has_pose
        self._has_wrench = "wrench_abs" in self.cfg.target_types or self._has_wrench

    def compute(self, jacobian_b: torch.Tensor, current_ee_pose_b=None, current_ee_vel_b=None, current_ee_force_b=None,
                mass_matrix=None, gravity=None, current_joint_pos=None, current_joint_vel=None,
                nullspace_joint_pos_target=None) -> torch.Tensor:
        """operational_space.py:345-548; the argument checks raise the reference's messages."""
        cfg, num_dof = self.cfg, jacobian_b.shape[2]
        if self._has_pose:
            if current_ee_pose_b is None or current_ee_vel_b is None:
                raise ValueError("Current end-effector pose and velocity are required for motion control.")
            if cfg.inertial_dynamics_decoupling and mass_matrix is None:
                raise ValueError("Mass matrix is required for inertial decoupling.")
        if self._has_wrench and cfg.contact_wrench_stiffness_task is not None and current_ee_force_b is None:
            raise ValueError("Current end-effector force is required for closed-loop force control.")
        if cfg.gravity_compensation and gravity is None:
            raise ValueError("Gravity vector is required for gravity compensation.")
        if cfg.nullspace_control == "position":
            if num_dof <= 6:
                raise ValueError("Null-space control is only applicable for redundant manipulators.")
            if cfg.inertial_dynamics_decoupling and not cfg.partial_inertial_dynamics_decoupling and (mass_matrix is None or not self._has_pose):
                raise ValueError("Mass matrix inverse is required for dynamically consistent pseudo-inverse")
            if current_joint_pos is None or current_joint_vel is None:
                raise ValueError("Current joint positions and velocities are required for null-space control.")
            if nullspace_joint_pos_target is not None and nullspace_joint_pos_target.shape != current_joint_pos.shape:
                raise ValueError(f"The target nullspace joint positions shape '{nullspace_joint_pos_target.shape}' does not"
                                 f"match the current joint positions shape '{current_joint_pos.shape}'.")
        joint_efforts = torch.empty(self.num_envs, num_dof, device=self._device)
        p = lambda t: None if t is None else _abi.ptr(_abi.f32c(t))  # noqa: E731
        _abi.check(_compute(ctypes.byref(self._desc), int(self._has_pose), int(self._has_wrench), p(jacobian_b), num_dof,
                            p(current_ee_pose_b), p(current_ee_vel_b), p(current_ee_force_b), p(mass_matrix), p(gravity),
                            p(current_joint_pos), p(current_joint_vel), p(nullspace_joint_pos_target), _abi.ptr(joint_efforts),
                            _abi.stream()), "ilx_osc_compute")
        return joint_efforts
