"""GPU: the FrameTransformer sensor (HIP ``ilx_frame_transformer_update``) against the fixture recorded from the reference class."""

import pytest
import torch

from conftest import assert_close
from test_oracle_frame_transformer import CASES, OUTPUTS, load_case

pytestmark = pytest.mark.gpu
E = "/World/envs/env_.*/Robot/"


@pytest.mark.parametrize("tag", list(CASES))
def test_frame_transformer_matches_reference(tag):
    from isaaclab_amd.sensors.frame_transformer import FrameTransformer, FrameTransformerCfg, OffsetCfg

    dev = torch.device("cuda:0")
    g = load_case(tag)
    n = g["source_pos_w"].shape[0]
    src_off, targets = CASES[tag]
    cfg = FrameTransformerCfg(prim_path=E + "base", source_frame_offset=OffsetCfg(*src_off),
                              target_frames=[FrameTransformerCfg.FrameCfg(prim_path=E + body, name=name, offset=OffsetCfg(*off))
                                             for body, name, off in targets])
    s = FrameTransformer(cfg, n, g["prim_paths"], dev)
    with pytest.raises(RuntimeError, match="no physics state"):
        s.data
    s.set_physics_state(g["transforms"].to(dev))
    s.update(0.02)
    d = s.data
    # the frames of one body are a set in the reference (their order varies from run to run): matched by name
    assert sorted(d.target_frame_names) == sorted(g["frame_names"])
    order = [d.target_frame_names.index(f) for f in g["frame_names"]]
    for name in OUTPUTS:
        got = getattr(d, name)
        if name.startswith("target"):
            got = got[:, order]
        assert_close(got, g[name], atol=2e-6, what=f"{tag} {name}")
    # not outdated: new transforms are not read until the clock says so
    before = d.target_pos_w.clone()
    s.cfg.update_period = 0.05
    s.set_physics_state(torch.zeros_like(g["transforms"]).to(dev))
    s.update(0.02)
    assert torch.equal(s.data.target_pos_w, before)
    s.reset([0])
    assert not torch.equal(s.data.target_pos_w, before)      # one outdated env refreshes the whole sensor, as in the reference


def test_frame_transformer_rejects_unknown_frames():
    from isaaclab_amd.sensors.frame_transformer import FrameTransformer, FrameTransformerCfg

    g = load_case("plain")
    cfg = FrameTransformerCfg(prim_path=E + "base", target_frames=[FrameTransformerCfg.FrameCfg(prim_path=E + "nothing")])
    with pytest.raises(ValueError, match="No matching prims were found"):
        FrameTransformer(cfg, 12, g["prim_paths"], torch.device("cuda:0"))
