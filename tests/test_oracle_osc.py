"""CPU: the operational-space-controller oracle against the fixture recorded from the reference class."""

import pytest
import torch

from conftest import assert_close, load_golden
from oracle import osc as O

BASE = dict(target_types=["pose_abs"], motion_control_axes_task=(1, 1, 1, 1, 1, 1), contact_wrench_control_axes_task=(0, 0, 0, 0, 0, 0),
            inertial_dynamics_decoupling=False, partial_inertial_dynamics_decoupling=False, gravity_compensation=False,
            impedance_mode="fixed", motion_stiffness_task=(100.0, 100.0, 100.0, 100.0, 100.0, 100.0),
            motion_damping_ratio_task=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0), motion_stiffness_limits_task=(0, 1000),
            motion_damping_ratio_limits_task=(0, 100), contact_wrench_stiffness_task=None, nullspace_control="none",
            nullspace_stiffness=10.0, nullspace_damping_ratio=1.0)
CASES = {
    "abs_fixed": dict(),
    "rel_decoupled_gravity": dict(target_types=["pose_rel"], inertial_dynamics_decoupling=True, gravity_compensation=True,
                                  motion_stiffness_task=(150.0, 120.0, 90.0, 60.0, 50.0, 40.0),
                                  motion_damping_ratio_task=(1.0, 0.9, 0.8, 1.1, 1.2, 0.7)),
    "abs_partial_nullspace": dict(inertial_dynamics_decoupling=True, partial_inertial_dynamics_decoupling=True,
                                  nullspace_control="position", nullspace_stiffness=12.0, nullspace_damping_ratio=0.8),
    "abs_full_nullspace": dict(inertial_dynamics_decoupling=True, nullspace_control="position"),
    "variable_kp_taskframe": dict(impedance_mode="variable_kp", motion_stiffness_limits_task=(10.0, 300.0)),
    "variable_taskframe": dict(target_types=["pose_rel"], impedance_mode="variable", motion_damping_ratio_limits_task=(0.2, 1.5),
                               motion_stiffness_limits_task=(10.0, 300.0)),
    "hybrid_closed_loop": dict(target_types=["pose_abs", "wrench_abs"], motion_control_axes_task=(1, 1, 0, 1, 1, 1),
                               contact_wrench_control_axes_task=(0, 0, 1, 0, 0, 0),
                               contact_wrench_stiffness_task=(0.0, 0.0, 0.1, 0.0, 0.0, 0.0), gravity_compensation=True),
    "wrench_open_loop": dict(target_types=["wrench_abs"], motion_control_axes_task=(0, 0, 0, 0, 0, 0),
                             contact_wrench_control_axes_task=(1, 1, 1, 1, 1, 1)),
}


def step_inputs(g, tag, k):
    get = lambda key: g.get(f"{tag}_{k}_{key}")  # noqa: E731
    return get


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_matches_reference(tag):
    g = load_golden("osc")
    cfg = {**BASE, **CASES[tag]}
    s = O.init(cfg, g[f"{tag}_0_jac"].shape[0])
    for k in range(2):
        get = step_inputs(g, tag, k)
        O.set_command(s, get("command"), get("ee_pose"), get("task_pose"))
        assert_close(s["p_b"].float(), get("p_gains_b"), rtol=1e-5, atol=1e-4, what=f"{tag} p gains (root frame) step {k}")
        assert_close(s["d_b"].float(), get("d_gains_b"), rtol=1e-5, atol=1e-4, what=f"{tag} d gains (root frame) step {k}")
        assert_close(s["sel_m_b"].float(), get("sel_motion_b"), atol=1e-6, what=f"{tag} motion selection step {k}")
        if get("desired_pose_b") is not None:
            assert_close(s["pose_b"], get("desired_pose_b"), atol=2e-6, what=f"{tag} desired pose step {k}")
        if get("desired_wrench_b") is not None:
            assert_close(s["wrench_b"].float(), get("desired_wrench_b"), rtol=1e-5, atol=1e-5, what=f"{tag} desired wrench step {k}")
        tau = O.compute(s, get("jac"), get("ee_pose"), get("ee_vel"), get("ee_force"), get("mass"), get("grav"), get("q"), get("qd"),
                        get("q_null"))
        # the reference inverts the mass matrices in float32: agreement is bounded by that, relative to the row's scale
        want = get("tau")
        scale = want.abs().amax(dim=-1, keepdim=True).clamp(min=1.0)
        assert float(((tau - want).abs() / scale).max()) < 2e-4, f"{tag} joint efforts step {k}"
