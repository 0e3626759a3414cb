"""CPU: the OSC action's state assembly oracle against the fixture recorded from the reference class."""

import pytest
import torch

from conftest import assert_close, load_golden
from oracle import osc_action as O

# name: (fixed base, body offset) as recorded; the end-effector is body 8 ("hand"), the arm joints are 0..6
CASES = {
    "abs_offset_nullspace": (True, ((0.02, -0.03, 0.107), (0.9, 0.1, 0.3, -0.2))),
    "rel_variable_kp": (True, None),
    "hybrid_taskframe_floating": (False, ((0.0, 0.0, 0.1), (1.0, 0.0, 0.0, 0.0))),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_osc_action_state(tag):
    g = load_golden("osc_action")
    fixed, off = CASES[tag]
    offset = None
    if off is not None:
        q = torch.tensor(off[1])
        offset = (torch.tensor(off[0]), q / q.norm())
    cols = list(range(7)) if fixed else [i + 6 for i in range(7)]
    for step in range(g[f"{tag}_raw"].shape[0]):
        pose, vel, force, jac = O.state(g[f"{tag}_root_state_w"][step], g[f"{tag}_body_state_w"][step], 8, g[f"{tag}_jacobians"][step],
                                        7 if fixed else 8, cols, offset, g[f"{tag}_ee_force_w"][step])
        assert_close(pose, g[f"{tag}_ee_pose_b"][step], atol=1e-6, what=f"{tag} ee_pose_b {step}")
        assert_close(vel, g[f"{tag}_ee_vel_b"][step], atol=1e-6, what=f"{tag} ee_vel_b {step}")
        assert_close(jac, g[f"{tag}_jacobian_b"][step], atol=1e-6, what=f"{tag} jacobian_b {step}")
        if tag == "hybrid_taskframe_floating":
            assert_close(force, g[f"{tag}_ee_force_b"][step], atol=2e-6, what=f"{tag} ee_force_b {step}")
