"""``ArticulationData`` backed by caller-supplied state tensors
(field names and semantics of ``isaaclab/assets/articulation/articulation_data.py``).

The reference pulls root / joint / body state from PhysX tensor views on demand; here the owner writes the
state of the current physics step with :meth:`set_sim_state` and the derived quantities come from HIP kernels.
Unlike the reference, ``root_lin_vel_b`` / ``root_ang_vel_b`` / ``projected_gravity_b`` / ``heading_w`` are
computed once per physics step by one launch and cached, instead of ~10 eager kernels on every access.
"""

from __future__ import annotations

import torch

from .. import _abi
from ..utils import math as math_utils


class ArticulationData:
    def __init__(self, num_instances: int, num_joints: int, num_bodies: int, device, gravity=(0.0, 0.0, -9.81),
                 joint_names=None, body_names=None):
        self.device = torch.device(device)
        n, j, b = num_instances, num_joints, num_bodies
        self._n, self._j, self._b = n, j, b
        self.joint_names = list(joint_names) if joint_names is not None else [f"joint_{i}" for i in range(j)]
        self.body_names = list(body_names) if body_names is not None else [f"body_{i}" for i in range(b)]
        self._sim_timestamp = 0.0
        self._joint_acc_timestamp = -1.0  # timestamped_buffer.py: an untouched buffer carries -1.0
        self._derived_timestamp = -1.0
        g = torch.tensor(gravity, dtype=torch.float64)
        g32 = torch.tensor(gravity, dtype=torch.float32)
        g32 = g32 / g32.norm().clamp(min=1e-9)  # articulation_data.py:53-59
        self._gravity = tuple(float(v) for v in g32)
        dev = self.device
        self.GRAVITY_VEC_W = g32.to(dev).repeat(n, 1)
        self.FORWARD_VEC_B = torch.tensor((1.0, 0.0, 0.0), device=dev).repeat(n, 1)
        # state (articulation_data.py:263-346, 427-443)
        self.root_state_w = torch.zeros(n, 13, device=dev)
        self.root_state_w[:, 3] = 1.0
        self.joint_pos = torch.zeros(n, j, device=dev)
        self.joint_vel = torch.zeros(n, j, device=dev)
        self._joint_acc = torch.zeros(n, j, device=dev)
        self._previous_joint_vel = torch.zeros(n, j, device=dev)
        self.body_state_w = torch.zeros(n, b, 13, device=dev)
        self.body_state_w[..., 3] = 1.0
        self.body_acc_w = torch.zeros(n, b, 6, device=dev)
        # defaults, limits, commands (articulation_data.py:118-206, 250-254)
        self.default_root_state = torch.zeros(n, 13, device=dev)
        self.default_joint_pos = torch.zeros(n, j, device=dev)
        self.default_joint_vel = torch.zeros(n, j, device=dev)
        self.joint_pos_target = torch.zeros(n, j, device=dev)
        self.joint_vel_target = torch.zeros(n, j, device=dev)
        self.joint_effort_target = torch.zeros(n, j, device=dev)
        self.computed_torque = torch.zeros(n, j, device=dev)
        self.applied_torque = torch.zeros(n, j, device=dev)
        self.joint_stiffness = torch.zeros(n, j, device=dev)
        self.joint_damping = torch.zeros(n, j, device=dev)
        self.joint_limits = torch.zeros(n, j, 2, device=dev)
        self.soft_joint_pos_limits = torch.zeros(n, j, 2, device=dev)
        self.soft_joint_vel_limits = torch.zeros(n, j, device=dev)
        self._lin_b = torch.zeros(n, 3, device=dev)
        self._ang_b = torch.zeros(n, 3, device=dev)
        self._grav_b = torch.zeros(n, 3, device=dev)
        self._heading = torch.zeros(n, device=dev)

    # ---- hand-over from the physics side --------------------------------------------------------
    def set_sim_state(self, root_transforms_xyzw=None, root_velocities=None, joint_pos=None, joint_vel=None,
                      body_state_w=None, root_state_w=None):
        """Store what the PhysX getters would return for the current step (articulation_data.py:263-346)."""
        if root_state_w is not None:
            self.root_state_w = _abi.f32c(root_state_w)
        elif root_transforms_xyzw is not None:
            tf, vel = _abi.f32c(root_transforms_xyzw), _abi.f32c(root_velocities)
            _abi.check(_abi.root_state_pack(_abi.ptr(tf), _abi.ptr(vel), _abi.ptr(self.root_state_w), self._n,
                                            _abi.stream()), "ilx_root_state_pack")
        if joint_pos is not None:
            self.joint_pos = _abi.f32c(joint_pos)
        if joint_vel is not None:
            self.joint_vel = _abi.f32c(joint_vel)
        if body_state_w is not None:
            self.body_state_w = _abi.f32c(body_state_w)
        self._derived_timestamp = -1.0

    def update(self, dt: float):
        """articulation_data.py:77-82: advance the clock and refresh the finite-difference acceleration."""
        self._sim_timestamp += dt
        self.joint_acc  # noqa: B018  (forces the update like the reference does)

    # ---- derived quantities -----------------------------------------------------------------------
    def _derive(self):
        if self._derived_timestamp != self._sim_timestamp:
            gx, gy, gz = self._gravity
            _abi.check(_abi.root_frame_derive(_abi.ptr(self.root_state_w), gx, gy, gz, _abi.ptr(self._lin_b),
                                              _abi.ptr(self._ang_b), _abi.ptr(self._grav_b), _abi.ptr(self._heading),
                                              self._n, _abi.stream()), "ilx_root_frame_derive")
            self._derived_timestamp = self._sim_timestamp

    @property
    def joint_acc(self):
        """articulation_data.py:445-455."""
        if self._joint_acc_timestamp < self._sim_timestamp:
            elapsed = self._sim_timestamp - self._joint_acc_timestamp
            _abi.check(_abi.joint_acc(_abi.ptr(self.joint_vel), _abi.ptr(self._previous_joint_vel),
                                      _abi.ptr(self._joint_acc), float(elapsed), self._n * self._j, _abi.stream()),
                       "ilx_joint_acc")
            self._joint_acc_timestamp = self._sim_timestamp
        return self._joint_acc

    @property
    def root_pos_w(self):
        return self.root_state_w[:, :3]

    @property
    def root_quat_w(self):
        return self.root_state_w[:, 3:7]

    @property
    def root_vel_w(self):
        return self.root_state_w[:, 7:13]

    @property
    def root_lin_vel_w(self):
        return self.root_state_w[:, 7:10]

    @property
    def root_ang_vel_w(self):
        return self.root_state_w[:, 10:13]

    @property
    def root_link_quat_w(self):
        return self.root_state_w[:, 3:7]

    @property
    def root_lin_vel_b(self):
        """articulation_data.py:502-509."""
        self._derive()
        return self._lin_b

    @property
    def root_ang_vel_b(self):
        """articulation_data.py:511-518."""
        self._derive()
        return self._ang_b

    @property
    def projected_gravity_b(self):
        """articulation_data.py:411-414."""
        self._derive()
        return self._grav_b

    @property
    def heading_w(self):
        """articulation_data.py:416-425."""
        self._derive()
        return self._heading

    @property
    def body_pos_w(self):
        return self.body_state_w[..., :3]

    @property
    def body_quat_w(self):
        return self.body_state_w[..., 3:7]

    @property
    def body_vel_w(self):
        return self.body_state_w[..., 7:13]

    @property
    def body_lin_vel_w(self):
        return self.body_state_w[..., 7:10]

    @property
    def body_ang_vel_w(self):
        return self.body_state_w[..., 10:13]

    @property
    def body_lin_acc_w(self):
        return self.body_acc_w[..., 0:3]

    @property
    def body_ang_acc_w(self):
        return self.body_acc_w[..., 3:6]
