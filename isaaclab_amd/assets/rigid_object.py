"""``RigidObject`` / ``RigidObjectData`` (the part of ``isaaclab/assets/rigid_object`` that lives outside PhysX): the state
buffers of a single rigid body per env and the derived quantities the terms read (``rigid_object_data.py:225-300``), on the
same HIP kernel as the articulation's root frame (``ilx_root_frame_derive``)."""

from __future__ import annotations

from collections.abc import Sequence

import torch

from .articulation_data import ArticulationData


class RigidObjectData(ArticulationData):
    """rigid_object_data.py:16-600.  One body: ``body_state_w`` is the root state seen as (N, 1, 13)."""

    def __init__(self, num_instances: int, device, gravity=(0.0, 0.0, -9.81), body_name: str = "object"):
        super().__init__(num_instances, 0, 1, device, gravity=gravity, joint_names=[], body_names=[body_name])

    def set_sim_state(self, root_transforms_xyzw=None, root_velocities=None, root_state_w=None, **unused):
        """rigid_object_data.py:114-130: what ``RigidBodyView.get_transforms()`` / ``get_velocities()`` return now."""
        super().set_sim_state(root_transforms_xyzw=root_transforms_xyzw, root_velocities=root_velocities, root_state_w=root_state_w)

    def update(self, dt: float):
        """rigid_object_data.py:72-76 (no joints: only the clock)."""
        self._sim_timestamp += dt

    @property
    def body_state_w(self):
        """rigid_object_data.py:178-190."""
        return self.root_state_w.view(-1, 1, 13)

    @body_state_w.setter
    def body_state_w(self, value):
        pass   # the articulation base class allocates one; the rigid object's is the root state itself


class RigidObject:
    """rigid_object.py: names, the state writers the reset events use, reset and update."""

    def __init__(self, data: RigidObjectData):
        self.data = data
        self._external_force_b = torch.zeros(data.root_state_w.shape[0], 1, 3, device=data.device)
        self._external_torque_b = torch.zeros_like(self._external_force_b)
        self.has_external_wrench = False

    @property
    def num_instances(self) -> int:
        return self.data.root_state_w.shape[0]

    @property
    def num_bodies(self) -> int:
        return 1

    @property
    def body_names(self) -> list[str]:
        return self.data.body_names

    @property
    def device(self):
        return self.data.device

    def find_bodies(self, name_keys, preserve_order: bool = False):
        from ..utils.string import resolve_matching_names

        return resolve_matching_names(name_keys, self.body_names, preserve_order)

    # -- state writers (rigid_object.py:150-260): into the buffers the step reads
    def write_root_state_to_sim(self, root_state, env_ids=None):
        self.data.root_state_w[env_ids if env_ids is not None else slice(None)] = root_state
        self.data._derived_timestamp = -1.0

    def write_root_pose_to_sim(self, root_pose, env_ids=None):
        self.data.root_state_w[env_ids if env_ids is not None else slice(None), :7] = root_pose
        self.data._derived_timestamp = -1.0

    def write_root_velocity_to_sim(self, root_velocity, env_ids=None):
        self.data.root_state_w[env_ids if env_ids is not None else slice(None), 7:] = root_velocity
        self.data._derived_timestamp = -1.0

    def set_external_force_and_torque(self, forces, torques, body_ids=None, env_ids=None):
        """rigid_object.py:262-320."""
        rows = env_ids if env_ids is not None else slice(None)
        if forces.any() or torques.any():
            self.has_external_wrench = True
        else:
            self.has_external_wrench = False
        self._external_force_b[rows], self._external_torque_b[rows] = forces, torques

    def reset(self, env_ids: Sequence[int] | None = None):
        """rigid_object.py:96-103."""
        rows = env_ids if env_ids is not None else slice(None)
        self._external_force_b[rows] = 0.0
        self._external_torque_b[rows] = 0.0

    def write_data_to_sim(self):
        """rigid_object.py:105-117: the external wrench is the simulator's to apply."""

    def update(self, dt: float):
        self.data.update(dt)
