"""CPU: the action oracle against the fixture recorded from the reference's ActionManager / joint action terms."""

import torch

from conftest import assert_exact, load_golden
from oracle import actions as A

J = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE", "LF_KFE", "LH_KFE", "RF_KFE", "RH_KFE"]


def test_joint_position_action_default_offset():
    g = load_golden("actions")
    action, prev = torch.zeros_like(g["anymal_raw"][0]), torch.zeros_like(g["anymal_raw"][0])
    for k in range(g["anymal_raw"].shape[0]):
        (part,) = A.manager_process(action, prev, g["anymal_raw"][k], [12])
        assert_exact(A.process_actions(part, 0.5, g["anymal_default_joint_pos"]), g["anymal_processed_joint_pos"][k], f"processed {k}")
        assert_exact(action, g["anymal_action"][k], f"action {k}")
        assert_exact(prev, g["anymal_prev_action"][k], f"prev_action {k}")
        assert_exact(g["anymal_pos_target"][k], g["anymal_processed_joint_pos"][k], f"target {k}")


def test_two_terms_with_dict_scale_offset_clip():
    g = load_golden("actions")
    n = g["split_raw"].shape[1]
    # legs: [.*_HAA, .*_KFE] in asset order -> joints 0-3, 8-11; scale 2 / -3, offset 0 / 0.25, clip HAA to (-1, 1.5)
    scale = torch.cat([torch.full((n, 4), 2.0), torch.full((n, 4), -3.0)], 1)
    offset = torch.cat([torch.zeros(n, 4), torch.full((n, 4), 0.25)], 1)
    clip = torch.tensor([[-float("inf"), float("inf")]]).repeat(n, 8, 1)
    clip[:, :4] = torch.tensor([-1.0, 1.5])
    action, prev = torch.zeros(n, 12), torch.zeros(n, 12)
    for k in range(g["split_raw"].shape[0]):
        legs, hips = A.manager_process(action, prev, g["split_raw"][k], [8, 4])
        assert_exact(A.process_actions(legs, scale, offset, clip), g["split_processed_legs"][k], f"legs {k}")
        ph = A.process_actions(hips, 0.25, 0.0)
        assert_exact(ph, g["split_processed_hips"][k], f"hips {k}")
        assert_exact(g["split_effort_target"][k][:, [0, 1, 2, 3, 8, 9, 10, 11]], g["split_processed_legs"][k], f"effort target {k}")
        assert_exact(g["split_pos_target"][k][:, 4:8], ph + g["split_joint_pos"][:, 4:8], f"relative target {k}")
    raws = [g["split_raw"][-1][:, :8].clone(), g["split_raw"][-1][:, 8:].clone()]
    A.manager_reset(action, prev, raws, g["split_reset_ids"])
    assert_exact(action, g["split_reset_action"], "reset action")
    assert_exact(prev, g["split_reset_prev_action"], "reset prev_action")
    assert_exact(raws[0], g["split_reset_raw_legs"], "reset raw legs")


def test_wrong_width_raises():
    import pytest

    with pytest.raises(ValueError, match="Invalid action shape, expected: 12, received: 11"):
        A.manager_process(torch.zeros(2, 12), torch.zeros(2, 12), torch.zeros(2, 11), [8, 4])


def test_limit_scaled_and_moving_average_actions():
    g = load_golden("actions2")
    lim = g["soft_joint_pos_limits"]
    arm = [3, 4, 5]
    for k in range(3):
        assert_exact(A.to_limits(g["limits_arm_raw"][k], 0.8, None, True, lim[:, arm, 0], lim[:, arm, 1]), g["limits_arm_processed"][k],
                     f"limits_arm {k}")
    n = lim.shape[0]
    scale = torch.ones(n, 8)
    scale[:, 3:6], scale[:, 6:8] = 0.5, 2.0
    clip = torch.tensor([[-float("inf"), float("inf")]]).repeat(n, 8, 1)
    clip[:, 3:6] = torch.tensor([-0.4, 0.6])
    for k in range(3):
        assert_exact(A.to_limits(g["limits_raw_raw"][k], scale, clip, False, None, None), g["limits_raw_processed"][k], f"limits_raw {k}")
    ids = [3, 4, 5, 6, 7]
    alpha = torch.cat([torch.full((n, 3), 0.3), torch.full((n, 2), 0.9)], 1)
    prev = g["joint_pos"][:, ids].clone()
    assert_exact(prev, g["ema_prev_after_reset"], "moving average starts from the joint positions")
    for k in range(4):
        p = A.to_limits(g["ema_raw"][k], 1.0, None, True, lim[:, ids, 0], lim[:, ids, 1])
        prev = A.ema_to_limits(p, alpha, prev, lim[:, ids, 0], lim[:, ids, 1])
        assert_exact(prev, g["ema_processed"][k], f"ema {k}")


def test_binary_and_non_holonomic_actions():
    g = load_golden("actions2")
    n = g["joint_pos"].shape[0]
    open_c, close_c = torch.tensor([0.04, 0.04]), torch.tensor([0.0, 0.005])
    clip = torch.tensor([[0.01, 0.03]]).repeat(n, 1, 1)
    for k in range(3):
        raw = g["binary_pos_raw"][k]
        raw = (raw > 0.5) if k == 1 else raw      # step 1 of the fixture used boolean actions
        assert_exact(A.binary(raw, open_c, close_c, clip), g["binary_pos_processed"][k], f"binary {k}")
    proc = g["non_holonomic_processed"][-1]
    want = g["non_holonomic_vel_target"][:, :3]
    got = A.non_holonomic(g["body_quat_w"][:, 1], proc)
    assert (got - want).abs().max() < 1e-6
