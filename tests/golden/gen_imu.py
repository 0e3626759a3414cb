"""Golden vectors for the IMU sensor, produced by the REFERENCE class (``isaaclab/sensors/imu/imu.py`` on
``sensors/sensor_base.py``) with the PhysX rigid-body view replaced by recorded tensors.  Writes tests/golden/imu.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
M = m["math"]
R.placeholder("omni.timeline")
R.placeholder("isaacsim.core.utils.stage", get_stage_up_axis=lambda: "Z")
R.placeholder("isaaclab.sensors")
R.load("isaaclab.sensors.sensor_base", f"{R.L}/sensors/sensor_base.py")
R.placeholder("isaaclab.sensors.imu")
idata = R.load("isaaclab.sensors.imu.imu_data", f"{R.L}/sensors/imu/imu_data.py")
imu = R.load("isaaclab.sensors.imu.imu", f"{R.L}/sensors/imu/imu.py")

out = {}
g = torch.Generator().manual_seed(3307)
n, dt, steps = 64, 0.02, 7


class View:
    count = n

    def get_transforms(self):
        return self.transforms.clone()

    def get_velocities(self):
        return self.velocities.clone()

    def get_coms(self):
        return self.coms.clone()


s = object.__new__(imu.Imu)
s._initialize_handle = s._invalidate_initialize_handle = s._debug_vis_handle = None
s.cfg = NS(update_period=0.03, history_length=0, debug_vis=False, prim_path="/World/envs/env_.*/Robot/base",
           offset=NS(pos=(0.12, -0.03, 0.2), rot=tuple(M.normalize(torch.tensor([[0.9, 0.1, -0.3, 0.2]]))[0].tolist())),
           gravity_bias=(0.0, 0.0, 9.81))
s._device, s._num_envs, s._is_visualizing = "cpu", n, False
s._view = View()
s._view.coms = torch.cat([torch.randn(n, 3, generator=g) * 0.05, torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(n, 1)], dim=1)
s._data = idata.ImuData()
s._is_outdated = torch.ones(n, dtype=torch.bool)
s._timestamp = torch.zeros(n)
s._timestamp_last_update = torch.zeros(n)
s._initialize_buffers_impl()
out["coms"] = s._view.coms
out["offset_pos"], out["offset_quat"] = torch.tensor(s.cfg.offset.pos), torch.tensor(s.cfg.offset.rot)

rec = {k: [] for k in ("transforms", "velocities", "pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b", "timestamp",
                       "timestamp_last_update", "prev_lin_vel_w")}
# step k: (forced refresh, read data after a lazy update, reset these envs first)
PLAN = [(True, False, None), (False, True, None), (False, False, None), (False, True, [3, 10, 63]), (True, False, None),
        (False, True, None), (False, True, None)]
for k, (force, read, reset_ids) in enumerate(PLAN):
    q = M.normalize(torch.randn(n, 4, generator=g))
    s._view.transforms = torch.cat([torch.randn(n, 3, generator=g), M.convert_quat(q, to="xyzw")], dim=1)
    s._view.velocities = torch.randn(n, 6, generator=g)
    if reset_ids is not None:
        s.reset(reset_ids)
    if k == 6:   # sensors that updated at different times: only the first eight are due at this step
        s._timestamp_last_update[:8] -= 0.02
    s.update(dt, force_recompute=force)
    d = s.data if read else s._data
    rec["transforms"].append(s._view.transforms.clone()), rec["velocities"].append(s._view.velocities.clone())
    for name in ("pos_w", "quat_w", "lin_vel_b", "ang_vel_b", "lin_acc_b", "ang_acc_b"):
        rec[name].append(getattr(d, name).clone())
    rec["timestamp"].append(s._timestamp.clone()), rec["timestamp_last_update"].append(s._timestamp_last_update.clone())
    rec["prev_lin_vel_w"].append(s._prev_lin_vel_w.clone())
    rec.setdefault("refreshed", []).append(s._timestamp_last_update == s._timestamp)
    print(k, "outdated after:", int(s._is_outdated.sum()), "refreshed:", int(rec["refreshed"][-1].sum()))
for k, v in rec.items():
    out[k] = torch.stack(v)
np.savez_compressed(os.path.join(HERE, "imu.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote imu.npz with", len(out), "arrays")
