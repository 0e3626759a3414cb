"""GPU: RigidObjectData's derived quantities (rigid_object_data.py:225-300) against the articulation oracle (the same
formulas on the root state, pinned to the reference by the articulation fixture), and the RigidObject state writers."""

import pytest
import torch

from conftest import assert_close, assert_exact
from oracle import articulation as AD
from oracle import mathx as M

pytestmark = pytest.mark.gpu


def test_rigid_object_data_and_writers():
    from isaaclab_amd.assets import RigidObject, RigidObjectData
    from isaaclab_amd.scene import InteractiveScene

    dev = torch.device("cuda:0")
    n = 200
    g = torch.Generator().manual_seed(9)
    root = torch.cat([torch.randn(n, 3, generator=g), M.normalize(torch.randn(n, 4, generator=g)), torch.randn(n, 6, generator=g)], dim=1)
    data = RigidObjectData(n, dev)
    obj = RigidObject(data)
    data.set_sim_state(root_state_w=root.to(dev))
    data.update(0.01)
    q = root[:, 3:7]
    assert_close(data.root_lin_vel_b, M.quat_rotate_inverse(q, root[:, 7:10]), atol=1e-6, what="root_lin_vel_b")
    assert_close(data.root_ang_vel_b, M.quat_rotate_inverse(q, root[:, 10:13]), atol=1e-6, what="root_ang_vel_b")
    assert_close(data.projected_gravity_b, M.quat_rotate_inverse(q, torch.tensor([[0.0, 0.0, -1.0]]).repeat(n, 1)), atol=1e-6, what="projected gravity")
    fwd = M.quat_apply(q, torch.tensor([[1.0, 0.0, 0.0]]).repeat(n, 1))
    assert_close(data.heading_w, torch.atan2(fwd[:, 1], fwd[:, 0]), atol=2e-6, what="heading")
    assert data.body_state_w.shape == (n, 1, 13) and data.body_pos_w.shape == (n, 1, 3)
    assert_exact(data.body_state_w[:, 0], root, "the single body is the root")
    # writers invalidate the derived buffers
    ids = torch.tensor([0, 5, 7], device=dev)
    obj.write_root_velocity_to_sim(torch.zeros(3, 6, device=dev), env_ids=ids)
    assert float(data.root_lin_vel_b[ids].abs().max()) == 0.0
    obj.write_root_pose_to_sim(root[:3, :7].to(dev), env_ids=ids)
    assert_exact(data.root_state_w[ids, :7], root[:3, :7], "pose written")
    obj.set_external_force_and_torque(torch.ones(3, 1, 3, device=dev), torch.zeros(3, 1, 3, device=dev), env_ids=ids)
    assert obj.has_external_wrench
    scene = InteractiveScene(n, dev)
    scene.articulations["object"] = obj
    scene.reset(ids)
    scene.write_data_to_sim()
    assert float(obj._external_force_b.abs().max()) == 0.0 and obj.find_bodies("obj.*")[0] == [0]
