"""CPU: the oracle's event terms against outputs of the reference functions (tests/golden/gen_events.py)."""

from conftest import assert_close, assert_exact, load_golden
from oracle import events as E

POSE = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14), "roll": (-0.2, 0.1)}
VEL = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.1, 0.2), "roll": (-0.5, 0.5), "pitch": (-0.5, 0.5), "yaw": (-0.5, 0.5)}
PUSH = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-1.0, 0.25)}
JOINT = {"scale": ((0.5, 1.5), (-1.0, 2.0)), "offset": ((-0.6, 0.6), (-0.5, 0.5))}


def test_reset_and_push_events():
    g = load_golden("events")
    ids = g["env_ids"]
    pose, vel = E.reset_root_state_uniform(g["default_root_state"][ids], g["env_origins"][ids], POSE, VEL, g["root_u_pose"], g["root_u_vel"])
    assert_exact(pose[:, :3], g["root_pose"][:, :3], "root position")
    assert_close(pose[:, 3:], g["root_pose"][:, 3:], atol=1e-6, what="root orientation")
    assert_exact(vel, g["root_vel"], "root velocity")
    for tag, (pr, vr) in JOINT.items():
        pos, jv = E.reset_joints(g["default_joint_pos"][ids], g["default_joint_vel"][ids], g["soft_joint_pos_limits"][ids],
                                 g["soft_joint_vel_limits"][ids], pr, vr, g[f"{tag}_u_pos"], g[f"{tag}_u_vel"], tag == "scale")
        assert_exact(pos, g[f"{tag}_joint_pos"], f"{tag} joint positions")
        assert_exact(jv, g[f"{tag}_joint_vel"], f"{tag} joint velocities")
        assert (jv.abs() == g["soft_joint_vel_limits"][ids]).any(), "some joint velocities are clamped"
    lim = g["soft_joint_pos_limits"][ids]
    assert ((pos == lim[..., 0]) | (pos == lim[..., 1])).any(), "some joint positions are clamped (offset case)"
    assert_exact(E.push_by_setting_velocity(g["root_vel_w"][ids], PUSH, g["push_u"]), g["push_vel"], "pushed velocity")


def test_random_orientation_wrench_and_default_events():
    g = load_golden("events")
    ids = g["env_ids"]
    pose, vel = E.reset_root_state_with_random_orientation(g["default_root_state"][ids], g["env_origins"][ids], {"x": (-1.0, 1.0), "z": (0.0, 0.3)},
                                                           VEL, g["rand_orient_u_pos"], g["rand_orient_z"], g["rand_orient_u_vel"])
    assert_exact(pose, g["rand_orient_pose"], "pose with random orientation")
    assert_exact(vel, g["rand_orient_vel"], "velocity")
    f, t = E.apply_external_force_torque((-8.0, 12.0), (-1.5, 0.5), g["wrench_u"][0], g["wrench_u"][1])
    assert_exact(f, g["wrench_force"], "forces")
    assert_exact(t, g["wrench_torque"], "torques")
    root, q, qd = E.reset_scene_to_default(g["default_root_state"][ids], g["env_origins"][ids], g["default_joint_pos"][ids],
                                           g["default_joint_vel"][ids])
    assert_exact(root[:, :7], g["default_pose"], "default pose")
    assert_exact(root[:, 7:], g["default_vel"], "default velocity")
    assert_exact(q, g["default_joint_pos_rows"], "default joint positions")
    assert_exact(qd, g["default_joint_vel_rows"], "default joint velocities")


def test_reset_root_state_from_terrain():
    g = load_golden("events")
    pose, vel = E.reset_root_state_from_terrain(g["default_root_state"], g["terrain_valid_positions"], g["terrain_levels"], g["terrain_types"],
                                                g["terrain_picks"], {"roll": (-0.2, 0.1), "yaw": (-3.14, 3.14)}, VEL, g["terrain_u"])
    assert_exact(pose[:, :3], g["terrain_pose"][:, :3], "root position on a flat patch")
    assert_close(pose[:, 3:], g["terrain_pose"][:, 3:], atol=1e-6, what="root orientation")
    assert_exact(vel, g["terrain_vel"], "root velocity")
