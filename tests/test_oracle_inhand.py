"""CPU: the in-hand manipulation oracle against outputs of the reference's own functions / methods."""

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import inhand as IH

VARIANTS = {"shadow": (1.0, 0), "shadow_ema": (0.8, 50)}


def load_case(g, tag):
    s = {k[len(tag) + 4:]: v for k, v in g.items() if k.startswith(f"{tag}_in_")}
    s["reset_goal_buf"] = s["reset_goal_buf"].bool()
    return s


def test_inhand_step_matches_reference():
    g = load_golden("inhand")
    for tag, (ema, max_succ) in VARIANTS.items():
        s = load_case(g, tag)
        cfg = IH.InHandCfg(act_moving_average=ema, max_consecutive_success=max_succ)
        o = IH.InHandOracle(cfg, s, s["actuated"].tolist(), s["fingers"].tolist())
        r = o.step(s["actions"], g[f"{tag}_goal_samples"], g[f"{tag}_reset_goal_samples"])
        assert_exact(o.s["cur_targets"], g[f"{tag}_cur_targets"], f"{tag} cur_targets")
        assert_exact(o.s["prev_targets"], g[f"{tag}_prev_targets"], f"{tag} prev_targets")
        assert_exact(r["terminated"], g[f"{tag}_terminated"], f"{tag} terminated")
        assert_exact(r["time_outs"], g[f"{tag}_time_outs"], f"{tag} time_outs")
        assert_close(r["reward"], g[f"{tag}_reward"], rtol=1e-6, atol=1e-6, what=f"{tag} reward")
        assert_exact(o.s["successes"], g[f"{tag}_successes"], f"{tag} successes")
        assert_exact(o.s["episode_length_buf"], g[f"{tag}_episode_length_buf"], f"{tag} episode_length_buf")
        assert_exact(o.s["reset_goal_buf"], g[f"{tag}_reset_goal_buf"].bool(), f"{tag} reset_goal_buf")
        assert_close(o.s["consecutive_successes"], g[f"{tag}_consecutive_successes"], rtol=1e-6, what=f"{tag} consecutive successes")
        assert_close(o.s["goal_rot"], g[f"{tag}_goal_rot"], rtol=1e-6, atol=1e-7, what=f"{tag} goal_rot")
        assert_close(r["obs"], g[f"{tag}_obs"], rtol=1e-6, atol=1e-6, what=f"{tag} observations")
        assert r["obs"].shape[1] == 157 and r["terminated"].sum() > 10 and r["goal_resets"].sum() > 3
