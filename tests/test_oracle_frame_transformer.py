"""CPU: the FrameTransformer refresh oracle against the fixture recorded from the reference class."""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close
from oracle import frame_transformer as O

ID = ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0))


def _q(*v):
    t = torch.tensor(v)
    return tuple((t / t.norm()).tolist())


# the sensor configurations the fixture was recorded with: source offset, then (body regex, frame name, offset) per target
CASES = {
    "plain": (ID, [("LF_FOOT", None, ID), ("RF_FOOT", None, ID)]),
    "offsets": (((0.1, 0.0, -0.05), _q(0.9, 0.1, 0.2, -0.1)),
                [("LF_FOOT", "lf_sole", ((0.0, 0.0, -0.02), ID[1])), ("LF_FOOT", "lf_toe", ((0.08, 0.0, 0.0), _q(1.0, 0.0, 0.1, 0.0))),
                 ("arm_link", "tool", ((0.0, 0.1, 0.3), _q(0.7, 0.0, 0.7, 0.0))), ("LH_SHANK", None, ID)]),
    "source_target": (ID, [("base", "base_again", ((0.2, 0.0, 0.0), ID[1])), (".*_FOOT", None, ID)]),
}
FRAME_BODY = {"lf_sole": "LF_FOOT", "lf_toe": "LF_FOOT", "tool": "arm_link", "base_again": "base"}
OUTPUTS = ("source_pos_w", "source_quat_w", "target_pos_w", "target_quat_w", "target_pos_source", "target_quat_source")


def load_case(tag):
    with np.load(os.path.join(GOLDEN, "frame_transformer.npz"), allow_pickle=False) as z:
        g = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}
    return {k: (torch.from_numpy(v) if v.dtype.kind == "f" else [str(s) for s in v]) for k, v in g.items()}


@pytest.mark.parametrize("tag", list(CASES))
def test_frame_transformer_refresh(tag):
    g = load_case(tag)
    paths, names = g["prim_paths"], g["frame_names"]
    n = g["source_pos_w"].shape[0]
    row = {p: i for i, p in enumerate(paths)}
    source_rows = torch.tensor([row[f"/World/envs/env_{e}/Robot/base"] for e in range(n)])
    target_rows = torch.tensor([row[f"/World/envs/env_{e}/Robot/{FRAME_BODY.get(f, f)}"] for e in range(n) for f in names])
    src_off, targets = CASES[tag]
    offsets = {name if name is not None else body: off for body, name, off in targets}
    offsets.update({f: ID for f in names if f not in offsets})           # frames a regex expanded to
    t_off = None if all(offsets[f] == ID for f in names) else (torch.tensor([offsets[f][0] for f in names]), torch.tensor([offsets[f][1] for f in names]))
    s_off = None if src_off == ID else (torch.tensor(src_off[0]), torch.tensor(src_off[1]))
    got = O.refresh(g["transforms"], source_rows, target_rows, len(names), s_off, t_off)
    for name, value in zip(OUTPUTS, got):
        assert_close(value, g[name], atol=2e-6, what=f"{tag} {name}")
