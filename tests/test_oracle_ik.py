"""CPU: the IK oracle against outputs of the reference's DifferentialIKController / DifferentialInverseKinematicsAction."""

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import ik as IK

CASES = {  # tag: (command_type, relative, method, params, offset)
    "rel_dls": ("pose", True, "dls", {"lambda_val": 0.01}, ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))),
    "abs_dls": ("pose", False, "dls", {"lambda_val": 0.01}, ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))),
    "rel_dls_rotoff": ("pose", True, "dls", {"lambda_val": 0.05}, ((0.02, -0.03, 0.107), (0.9238795, 0.0, 0.3826834, 0.0))),
    "rel_pinv": ("pose", True, "pinv", {"k_val": 1.0}, None),
    "rel_svd": ("pose", True, "svd", {"k_val": 1.0, "min_singular_value": 1e-5}, ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))),
    "rel_trans": ("pose", True, "trans", {"k_val": 0.7}, ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))),
    "pos_rel_dls": ("position", True, "dls", {"lambda_val": 0.01}, ((0.0, 0.0, 0.107), (1.0, 0.0, 0.0, 0.0))),
    "pos_abs_pinv": ("position", False, "pinv", {"k_val": 1.0}, None),
}
FRANKA_PARAMS = {"dls": {"lambda_val": 0.01}, "pinv": {"k_val": 1.0}, "svd": {"k_val": 1.0, "min_singular_value": 1e-5},
                 "trans": {"k_val": 1.0}}


def offsets(offset):
    if offset is None:
        return None, None
    return torch.tensor([offset[0]]), torch.tensor([offset[1]])


def test_action_term_matches_reference():
    g = load_golden("ik")
    s = {k[6:]: v for k, v in g.items() if k.startswith("state_")}
    for tag, (ctype, rel, method, params, offset) in CASES.items():
        op, orot = offsets(offset)
        processed, pdes, qdes = IK.process_actions(g[f"{tag}_raw"], 0.5, None, s["root_pos_w"], s["root_quat_w"], s["ee_pos_w"],
                                                   s["ee_quat_w"], op, orot, ctype, rel)
        assert_exact(processed, g[f"{tag}_processed"], f"{tag} processed actions")
        pos_b, quat_b = IK.frame_pose(s["root_pos_w"], s["root_quat_w"], s["ee_pos_w"], s["ee_quat_w"], op, orot)
        assert_exact(pos_b, g[f"{tag}_ee_pos_b"], f"{tag} frame position")
        assert_exact(quat_b, g[f"{tag}_ee_quat_b"], f"{tag} frame orientation")
        assert_close(pdes, g[f"{tag}_ee_pos_des"], rtol=1e-6, atol=1e-7, what=f"{tag} desired position")
        assert_close(qdes, g[f"{tag}_ee_quat_des"], rtol=1e-6, atol=1e-7, what=f"{tag} desired orientation")
        jac = IK.frame_jacobian(s["jacobian_w"], s["root_quat_w"], op, orot)
        assert_close(jac, g[f"{tag}_jacobian_b"], rtol=1e-6, atol=1e-7, what=f"{tag} frame Jacobian")
        q_des = IK.apply_actions(s["root_pos_w"], s["root_quat_w"], s["ee_pos_w"], s["ee_quat_w"], s["jacobian_w"], s["joint_pos"],
                                 g[f"{tag}_ee_pos_des"], g[f"{tag}_ee_quat_des"], op, orot, ctype, method, params)
        # same torch.linalg calls on the same inputs; the batched solves are not bit-reproducible across shapes
        assert_close(q_des, g[f"{tag}_joint_pos_des"], rtol=1e-4, atol=2e-5, what=f"{tag} joint targets")


def test_controller_franka_shape():
    g = load_golden("ik")
    for method, params in FRANKA_PARAMS.items():
        pdes, qdes = IK.set_command(g["franka_cmd"], g["franka_ee_pos"], g["franka_ee_quat"], "pose", True)
        assert_close(pdes, g[f"franka_{method}_ee_pos_des"], rtol=1e-6, atol=1e-7, what="desired position")
        assert_close(qdes, g[f"franka_{method}_ee_quat_des"], rtol=1e-6, atol=1e-7, what="desired orientation")
        assert not torch.isnan(qdes).any(), "zero-rotation commands must not leak NaN (math.py:872-910)"
        q = IK.compute(g["franka_ee_pos"], g["franka_ee_quat"], pdes, qdes, g["franka_jac"], g["franka_q"], "pose", method, params)
        assert_close(q, g[f"franka_{method}_q_des"], rtol=1e-4, atol=2e-5, what=f"{method} joint targets")
        # float64 solve: the well-conditioned fixtures (sigma_min >= 0.1) agree with the fp32 reference to 1e-5
        q64 = IK.compute_f64(g["franka_ee_pos"], g["franka_ee_quat"], pdes, qdes, g["franka_jac"], g["franka_q"], "pose", method, params)
        tol = 2e-3 if method == "dls" else 2e-5   # dls: kappa(JJ^T + 1e-4 I) reaches 2e4 -> fp32 inverse loses ~3 digits
        assert_close(q64, g[f"franka_{method}_q_des"], rtol=tol, atol=tol, what=f"{method} float64 solve vs reference")
