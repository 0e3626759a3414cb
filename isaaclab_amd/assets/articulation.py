"""``Articulation`` (the part of ``isaaclab/assets/articulation/articulation.py`` that lives outside PhysX): names, target
buffers, the actuator models applied at ``write_data_to_sim`` (:1399-1436), reset and update.  The simulator side is the
caller's: it reads ``data.applied_torque`` / the targets and stores the next state with ``data.set_sim_state``."""

from __future__ import annotations

from collections.abc import Sequence

import torch

from ..utils.string import resolve_matching_names
from ..utils.types import ArticulationActions
from .articulation_data import ArticulationData


class Articulation:
    def __init__(self, data: ArticulationData, actuators: dict | None = None, is_fixed_base: bool = True):
        """``actuators``: name -> (actuator model, joint ids or ``slice(None)``), together covering the driven joints."""
        self.data = data
        self.actuators = dict(actuators or {})
        self.is_fixed_base = is_fixed_base
        self.jacobians = None
        self.root_physx_view = None

    # -- properties (articulation.py:120-190)
    @property
    def num_instances(self) -> int:
        return self.data.root_state_w.shape[0]

    @property
    def joint_names(self) -> list[str]:
        return self.data.joint_names

    @property
    def body_names(self) -> list[str]:
        return self.data.body_names

    @property
    def num_joints(self) -> int:
        return len(self.data.joint_names)

    @property
    def num_bodies(self) -> int:
        return len(self.data.body_names)

    @property
    def device(self):
        return self.data.root_state_w.device

    def find_joints(self, name_keys, joint_subset=None, preserve_order: bool = False):
        return resolve_matching_names(name_keys, joint_subset if joint_subset is not None else self.joint_names, preserve_order)

    def find_bodies(self, name_keys, preserve_order: bool = False):
        return resolve_matching_names(name_keys, self.body_names, preserve_order)

    def get_jacobians(self):
        return self.jacobians

    # -- targets (articulation.py:640-720)
    def set_joint_position_target(self, target, joint_ids=None, env_ids=None):
        self.data.joint_pos_target[self._index(env_ids, joint_ids)] = target

    def set_joint_velocity_target(self, target, joint_ids=None, env_ids=None):
        self.data.joint_vel_target[self._index(env_ids, joint_ids)] = target

    def set_joint_effort_target(self, target, joint_ids=None, env_ids=None):
        self.data.joint_effort_target[self._index(env_ids, joint_ids)] = target

    @staticmethod
    def _index(env_ids, joint_ids):
        """(rows, columns) index of a (num_envs, num_joints) buffer; lists on both sides select the block, as in the reference
        (articulation.py:656-660: ``env_ids[:, None]`` against ``joint_ids``)."""
        cols = slice(None) if joint_ids is None else joint_ids
        if env_ids is None or isinstance(env_ids, slice):
            return slice(None), cols
        rows = env_ids if isinstance(env_ids, torch.Tensor) else torch.as_tensor(env_ids)
        return (rows, cols) if isinstance(cols, slice) else (rows[:, None], cols)

    # -- state writers used by the reset events (articulation.py:300-420): into the buffers the step reads
    def write_root_pose_to_sim(self, root_pose, env_ids=None):
        self.data.root_state_w[env_ids if env_ids is not None else slice(None), :7] = root_pose
        self.data._derived_timestamp = -1.0   # base-frame velocities, projected gravity and heading follow the new pose

    def write_root_velocity_to_sim(self, root_velocity, env_ids=None):
        self.data.root_state_w[env_ids if env_ids is not None else slice(None), 7:] = root_velocity
        self.data._derived_timestamp = -1.0

    def write_joint_state_to_sim(self, position, velocity, joint_ids=None, env_ids=None):
        at = self._index(env_ids, joint_ids)
        self.data.joint_pos[at], self.data.joint_vel[at] = position, velocity
        self.data._previous_joint_vel[at] = velocity
        self.data._joint_acc[at] = 0.0

    # -- operations
    def reset(self, env_ids: Sequence[int] | None = None):
        """articulation.py:196-206: actuator state of the reset envs."""
        for act, _ in self.actuators.values():
            act.reset(env_ids if env_ids is not None else slice(None))

    def write_data_to_sim(self):
        """articulation.py:208-218, 1399-1436: the actuator models turn the targets into efforts."""
        d = self.data
        for act, ids in self.actuators.values():
            whole = isinstance(ids, slice)
            ca = ArticulationActions(joint_positions=d.joint_pos_target if whole else d.joint_pos_target[:, ids].contiguous(),
                                     joint_velocities=d.joint_vel_target if whole else d.joint_vel_target[:, ids].contiguous(),
                                     joint_efforts=d.joint_effort_target if whole else d.joint_effort_target[:, ids].contiguous(),
                                     joint_indices=ids)
            act.compute(ca, d.joint_pos if whole else d.joint_pos[:, ids].contiguous(), d.joint_vel if whole else d.joint_vel[:, ids].contiguous())
            d.computed_torque[:, ids] = act.computed_effort
            d.applied_torque[:, ids] = act.applied_effort

    def update(self, dt: float):
        self.data.update(dt)
