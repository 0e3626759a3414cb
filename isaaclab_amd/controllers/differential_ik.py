"""Differential inverse kinematics controller on the HIP kernels ``ilx_diff_ik_set_command`` / ``ilx_diff_ik_compute``
(interface of ``isaaclab/controllers/differential_ik.py`` and ``differential_ik_cfg.py``)."""

from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from .. import _abi


@dataclass
class DifferentialIKControllerCfg:
    """differential_ik_cfg.py:16-80."""

    command_type: str = "pose"
    use_relative_mode: bool = False
    ik_method: str = "dls"
    ik_params: dict | None = None

    def __post_init__(self):
        if self.command_type not in ["position", "pose"]:
            raise ValueError(f"Unsupported inverse-kinematics command: {self.command_type}.")
        if self.ik_method not in ["pinv", "svd", "trans", "dls"]:
            raise ValueError(f"Unsupported inverse-kinematics method: {self.ik_method}.")
        default_ik_params = {"pinv": {"k_val": 1.0}, "svd": {"k_val": 1.0, "min_singular_value": 1e-5},
                             "trans": {"k_val": 1.0}, "dls": {"lambda_val": 0.01}}
        ik_params = default_ik_params[self.ik_method].copy()
        if self.ik_params is not None:
            ik_params.update(self.ik_params)
        self.ik_params = ik_params

    @property
    def class_type(self):
        return DifferentialIKController


class DifferentialIKController:
    def __init__(self, cfg: DifferentialIKControllerCfg, num_envs: int, device):
        self.cfg = cfg
        self.num_envs = num_envs
        self._device = torch.device(device)
        self.ee_pos_des = torch.zeros(num_envs, 3, device=self._device)
        self.ee_quat_des = torch.zeros(num_envs, 4, device=self._device)
        self._command = torch.zeros(num_envs, self.action_dim, device=self._device)
        self._type = _abi.IK_POSITION if cfg.command_type == "position" else _abi.IK_POSE

    @property
    def action_dim(self) -> int:
        """differential_ik.py:79-86."""
        if self.cfg.command_type == "position":
            return 3
        if self.cfg.command_type == "pose" and self.cfg.use_relative_mode:
            return 6
        return 7

    def reset(self, env_ids=None):
        pass

    def set_command(self, command: torch.Tensor, ee_pos: torch.Tensor | None = None, ee_quat: torch.Tensor | None = None):
        """differential_ik.py:97-147 (same argument checks, one launch)."""
        self._command[:] = command
        if self.cfg.command_type == "position":
            if ee_quat is None:
                raise ValueError("End-effector orientation can not be None for `position_*` command type!")
            if self.cfg.use_relative_mode and ee_pos is None:
                raise ValueError("End-effector position can not be None for `position_rel` command type!")
            if ee_pos is None:
                ee_pos = self.ee_pos_des  # never read in absolute mode
        elif self.cfg.use_relative_mode and (ee_pos is None or ee_quat is None):
            raise ValueError("Neither end-effector position nor orientation can be None for `pose_rel` command type!")
        frame = None
        if ee_pos is not None and ee_quat is not None:
            frame = ctypes.byref(_abi.ik_frame(_abi.f32c(ee_pos), _abi.f32c(ee_quat)))
        _abi.check(_abi.diff_ik_set_command(frame, self._type, int(self.cfg.use_relative_mode), None, None, None, None, None,
                                            _abi.ptr(self._command), _abi.ptr(self.ee_pos_des), _abi.ptr(self.ee_quat_des),
                                            self.num_envs, _abi.stream()), "ilx_diff_ik_set_command")

    def compute(self, ee_pos: torch.Tensor, ee_quat: torch.Tensor, jacobian: torch.Tensor, joint_pos: torch.Tensor) -> torch.Tensor:
        """differential_ik.py:148-174; ``jacobian`` is (N, 6, num_joints) (position commands read its first 3 rows)."""
        jac, q = _abi.f32c(jacobian), _abi.f32c(joint_pos)
        n, rows, j = jac.shape
        if rows != 6:
            raise ValueError(f"expected an (N, 6, num_joints) Jacobian, got {tuple(jac.shape)}")
        out = torch.empty_like(q)
        p = self.cfg.ik_params
        frame = _abi.ik_frame(_abi.f32c(ee_pos), _abi.f32c(ee_quat))
        _abi.check(_abi.diff_ik_compute(ctypes.byref(frame), self._type, _abi.IK_METHODS[self.cfg.ik_method],
                                        float(p.get("k_val", 1.0)), float(p.get("lambda_val", 0.0)),
                                        float(p.get("min_singular_value", 0.0)), _abi.ptr(jac), 6 * j, j, None, _abi.ptr(q), j,
                                        None, _abi.ptr(self.ee_pos_des), _abi.ptr(self.ee_quat_des), _abi.ptr(out), j, n, j,
                                        _abi.stream()), "ilx_diff_ik_compute")
        return out
