"""GPU: the IMU sensor (HIP ``ilx_imu_update``) against the fixture recorded from the reference class."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu

# the sequence the fixture was recorded with: (forced refresh, read data after a lazy update, reset these envs first)
PLAN = [(True, False, None), (False, True, None), (False, False, None), (False, True, [3, 10, 63]), (True, False, None),
        (False, True, None), (False, True, None)]


def test_imu_sequence_matches_reference():
    from isaaclab_amd.sensors.imu import Imu, ImuCfg, ImuOffsetCfg

    dev = torch.device("cuda:0")
    g = load_golden("imu")
    n = g["coms"].shape[0]
    cfg = ImuCfg(update_period=0.03, offset=ImuOffsetCfg(pos=tuple(g["offset_pos"].tolist()), rot=tuple(g["offset_quat"].tolist())))
    s = Imu(cfg, n, dev)
    with pytest.raises(RuntimeError, match="update function must be called"):
        s.data
    for k, (force, read, reset_ids) in enumerate(PLAN):
        s.set_physics_state(g["transforms"][k].to(dev), g["velocities"][k].to(dev), g["coms"].to(dev))
        if reset_ids is not None:
            s.reset(reset_ids)
        if k == 6:
            s._timestamp_last_update[:8] -= 0.02
        s.update(0.02, force_recompute=force)
        d = s.data if read else s._data
        for name in ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b"):
            # accelerations are differences of O(1) velocities over 0.02 s: 1e-5 relative on values of O(100)
            assert_close(getattr(d, name), g[name][k], rtol=1e-5, atol=2e-5, what=f"{name} step {k}")
        assert_close(s._timestamp, g["timestamp"][k], atol=1e-7, what=f"timestamp step {k}")
        assert_close(s._timestamp_last_update, g["timestamp_last_update"][k], atol=1e-7, what=f"last update step {k}")
        assert_exact(s._timestamp_last_update == s._timestamp, g["refreshed"][k], f"which sensors refreshed at step {k}")
        assert_close(s._prev_lin_vel_w, g["prev_lin_vel_w"][k], atol=1e-6, what=f"prev_lin_vel_w step {k}")
    # empty sensor set
    Imu(cfg, 0, dev).update(0.02, force_recompute=False)
