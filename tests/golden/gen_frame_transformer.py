"""Golden vectors for the FrameTransformer sensor, produced by the REFERENCE class
(``isaaclab/sensors/frame_transformer/frame_transformer.py``: its own ``_initialize_impl`` and ``_update_buffers_impl``) on a
scripted stage and rigid-body view.  Writes tests/golden/frame_transformer.npz."""

import os
import re
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
M = m["math"]
n = 12
BODIES = ["base", "LF_FOOT", "RF_FOOT", "LH_SHANK", "arm_link"]
STAGE = [f"/World/envs/env_{e}"for e in range(n)] + [f"/World/envs/env_{e}/Robot" for e in range(n)] + \
        [f"/World/envs/env_{e}/Robot/{b}" for e in range(n) for b in BODIES]


class Prim:
    def __init__(self, path):
        self.path = path

    def GetPath(self):
        return NS(pathString=self.path)

    def HasAPI(self, api):
        return True


class View:
    def __init__(self, patterns):
        # PhysX order here: pattern by pattern, all envs of one body together
        self.prim_paths = [p for pat in patterns for p in STAGE if re.fullmatch(pat.replace("*", ".*"), p)]

    def get_transforms(self):
        return self.transforms.clone()


views = []


def create_rigid_body_view(patterns):
    views.append(View(patterns))
    return views[-1]


sim_view = NS(set_subspace_roots=lambda p: None, create_rigid_body_view=create_rigid_body_view)
sys.modules["omni.physics.tensors.impl.api"].create_simulation_view = lambda backend: sim_view
R.placeholder("omni.timeline")
sys.modules["pxr"].UsdPhysics = NS(RigidBodyAPI="RigidBodyAPI")
R.placeholder("isaaclab.sim", SimulationContext=NS(instance=lambda: NS(device="cpu", backend="torch", get_physics_dt=lambda: 0.005)),
              find_matching_prims=lambda expr: [Prim(p) for p in STAGE if re.fullmatch(expr, p)])
sys.modules["omni.log"].verbose = lambda *a, **k: None
R.placeholder("isaaclab.sensors")
R.load("isaaclab.sensors.sensor_base", f"{R.L}/sensors/sensor_base.py")
R.placeholder("isaaclab.sensors.frame_transformer")
fdata = R.load("isaaclab.sensors.frame_transformer.frame_transformer_data", f"{R.L}/sensors/frame_transformer/frame_transformer_data.py")
ft = R.load("isaaclab.sensors.frame_transformer.frame_transformer", f"{R.L}/sensors/frame_transformer/frame_transformer.py")

out = {}
g = torch.Generator().manual_seed(4242)
E = "/World/envs/env_.*/Robot/"
ID = NS(pos=(0.0, 0.0, 0.0), rot=(1.0, 0.0, 0.0, 0.0))
q = lambda *v: tuple(M.normalize(torch.tensor([v]))[0].tolist())  # noqa: E731
CASES = {
    # plain: no offsets anywhere
    "plain": (ID, [("LF_FOOT", None, ID), ("RF_FOOT", None, ID)]),
    # offsets on the source and on some targets, two frames on one body, user-given names
    "offsets": (NS(pos=(0.1, 0.0, -0.05), rot=q(0.9, 0.1, 0.2, -0.1)),
                [("LF_FOOT", "lf_sole", NS(pos=(0.0, 0.0, -0.02), rot=ID.rot)), ("LF_FOOT", "lf_toe", NS(pos=(0.08, 0.0, 0.0), rot=q(1.0, 0.0, 0.1, 0.0))),
                 ("arm_link", "tool", NS(pos=(0.0, 0.1, 0.3), rot=q(0.7, 0.0, 0.7, 0.0))),
                 ("LH_SHANK", None, ID)]),
    # the source body is also a target, one regex matching several bodies
    "source_target": (ID, [("base", "base_again", NS(pos=(0.2, 0.0, 0.0), rot=ID.rot)), (".*_FOOT", None, ID)]),
}
for tag, (src_off, targets) in CASES.items():
    s = object.__new__(ft.FrameTransformer)
    s._initialize_handle = s._invalidate_initialize_handle = s._debug_vis_handle = None
    s._is_visualizing = False
    s._data = fdata.FrameTransformerData()
    s.cfg = NS(prim_path=E + "base", update_period=0.0, history_length=0, debug_vis=False, source_frame_offset=src_off,
               target_frames=[NS(prim_path=E + b, name=name, offset=off) for b, name, off in targets])
    s._initialize_impl()
    view = views[-1]
    rows = len(view.prim_paths)
    view.transforms = torch.cat([torch.randn(rows, 3, generator=g), M.convert_quat(M.normalize(torch.randn(rows, 4, generator=g)), to="xyzw")], dim=1)
    s._update_buffers_impl(torch.arange(n))
    out[f"{tag}_transforms"] = view.transforms.clone()
    out[f"{tag}_prim_paths"] = np.array(view.prim_paths)
    out[f"{tag}_frame_names"] = np.array(s._data.target_frame_names)
    for name in ("source_pos_w", "source_quat_w", "target_pos_w", "target_quat_w", "target_pos_source", "target_quat_source"):
        out[f"{tag}_{name}"] = getattr(s._data, name).clone()
    print(tag, "frames:", s._data.target_frame_names, "view rows:", rows)
np.savez_compressed(os.path.join(HERE, "frame_transformer.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
print("wrote frame_transformer.npz with", len(out), "arrays")
