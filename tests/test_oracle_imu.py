"""CPU: the IMU refresh oracle against the fixture recorded from the reference class."""

import torch

from conftest import assert_close, load_golden
from oracle import imu as O


def test_imu_refresh_sequence():
    g = load_golden("imu")
    n = g["coms"].shape[0]
    prev_lin, prev_ang = torch.zeros(n, 3), torch.zeros(n, 3)
    names = ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b")
    seen = 0
    for k in range(g["transforms"].shape[0]):
        mask = g["refreshed"][k]
        if not mask.any():
            continue
        res = O.imu_refresh(g["transforms"][k], g["velocities"][k], g["coms"], g["offset_pos"], g["offset_quat"],
                            torch.tensor([0.0, 0.0, 9.81]), prev_lin, prev_ang, 0.02)
        for name, got in zip(names, res[:6]):
            assert_close(got[mask], g[name][k][mask], rtol=1e-5, atol=2e-5, what=f"{name} step {k}")
        prev_lin, prev_ang = torch.where(mask[:, None], res[6], prev_lin), torch.where(mask[:, None], res[7], prev_ang)
        assert_close(prev_lin, g["prev_lin_vel_w"][k], atol=1e-6, what=f"prev_lin_vel_w step {k}")
        seen += 1
    assert seen >= 3
