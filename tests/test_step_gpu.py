"""GPU: the complete non-physics env step of Isaac-Velocity-Rough-Anymal-C (HIP, through the C ABI) against the
oracle step, two consecutive steps, all buffers."""

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu


def build(dev, n, seed, rows=2, cols=3):
    from isaaclab_amd import synth
    from isaaclab_amd.envs.locomotion_step import LocomotionVelocityStep
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.mesh import convert_to_mesh
    from isaaclab_amd.utils.string import resolve_matching_names_values
    from oracle import raycast as RC
    from oracle.step_locomotion import AnymalRoughOracle

    verts, tris, height_at = synth.rough_terrain_mesh(seed=seed, rows=rows, cols=cols)
    mesh = convert_to_mesh(verts, tris, str(dev))
    ws, bs = synth.actuator_mlp_weights()
    idx, _, vals = resolve_matching_names_values(LV.ANYMAL_DEFAULT_JOINT_POS, LV.ANYMAL_JOINTS)
    default = torch.zeros(12)
    default[idx] = torch.tensor(vals)
    state = synth.locomotion_state(n, seed, num_joints=12, num_bodies=17, default_joint_pos=default, feet_ids=[13, 14, 15, 16],
                                   height_at=height_at, extent=(rows * 8.0, cols * 8.0))
    # force a few time-outs and base contacts so that the reset path runs
    state["episode_length_buf"][:3] = 999
    env = LocomotionVelocityStep(n, dev, mesh, "anymal_c", actuator_net=dict(weights=ws, biases=bs, activation="softsign"))
    env.load_state(state)
    oracle = AnymalRoughOracle(n, state, ws, bs, default, RC.CMesh(verts, tris))
    return env, oracle, state


def test_anymal_full_step_matches_oracle(dev):
    from oracle import command as C

    n = 512
    env, oracle, state = build(dev, n, seed=21)
    gen = torch.Generator().manual_seed(99)
    for step in range(2):
        raw = state["raw_action"] if step == 0 else torch.rand(n, 12, generator=gen) * 2 - 1
        noise_u = torch.rand(n, 235, generator=gen)
        samples = torch.rand(n, 7, generator=gen) * 2 - 1
        samples[:, 0] = 10.0
        samples[:, 5:] = torch.rand(n, 2, generator=gen)
        reset_samples = samples.flip(0).contiguous()
        ref = oracle.step(raw, noise_u, samples, reset_samples)
        obs, reward, terminated, time_outs = env.step(raw.to(dev), noise_u.to(dev), samples.to(dev), reset_samples.to(dev))
        # decisions: bit exact
        assert_exact(terminated, ref["terminated"], f"terminated step {step}")
        assert_exact(time_outs, ref["time_outs"], f"time_outs step {step}")
        assert_exact(env.mdp.term_dones, ref["term_dones"], f"term_dones step {step}")
        assert_exact(env.mdp.episode_length_buf, oracle.s["episode_length_buf"], f"episode_length_buf step {step}")
        assert_exact(env.mdp.command_counter, oracle.cmd.command_counter, f"command_counter step {step}")
        assert_exact(env.mdp.is_standing_env, oracle.cmd.is_standing_env, f"is_standing_env step {step}")
        assert_exact(env.mdp.time_left, oracle.cmd.time_left, f"time_left step {step}")
        c = env.contact._data
        assert_exact(env.contact.ordered_history(), oracle.s["net_forces_w_history"], f"contact history step {step}")
        for k in ("current_air_time", "last_air_time", "current_contact_time", "last_contact_time"):
            assert_exact(getattr(c, k), oracle.s[k], f"{k} step {step}")
        assert_exact(env.actuator._joint_pos_error_history, oracle.s["pos_error_history"], f"pos error history step {step}")
        assert_exact(env.data.joint_acc, ref["joint_acc"], f"joint_acc step {step}")
        # ray hits: same rays up to the yaw rotation of the starts (sin/cos), so positions agree to ~1e-6 m; the
        # terrain has 5-23 cm steps, a ray within that distance of a step edge may land on the other side
        hits, rhits = env.scanner._data.ray_hits_w.cpu(), ref["ray_hits_w"]
        assert_exact(torch.isinf(hits[..., 2]), torch.isinf(rhits[..., 2]), "hit/miss pattern")
        ok = ~torch.isinf(rhits[..., 2])
        dz = (hits[..., 2] - rhits[..., 2])[ok].abs()
        assert (dz > 1e-5).float().mean() < 2e-3, f"ray hit heights: {(dz > 1e-5).float().mean():.2e} of rays differ"
        # float terms: 1e-5 relative (atol for near-zero values and for the 23 cm-step rays above)
        assert_close(env.data.computed_torque, ref["computed_torque"], atol=2e-5, what=f"computed torque step {step}")
        assert_close(env.data.applied_torque, ref["applied_torque"], atol=2e-5, what=f"applied torque step {step}")
        same_rays = ((hits[..., 2] - rhits[..., 2]).abs() <= 1e-5) | ~ok
        good = same_rays.all(dim=1)
        assert good.float().mean() > 0.9
        # height-scan columns: absolute 1e-5 m (rays start 20 m up: 1 ulp of t is 2e-6 m, and the yaw of the ray
        # starts goes through sin/cos)
        assert_close(obs.cpu()[good], ref["obs"][good], atol=1.2e-5, what=f"obs step {step}")
        assert_close(obs.cpu()[:, :48], ref["obs"][:, :48], atol=2e-6, what=f"proprioceptive obs step {step}")
        assert_close(reward, ref["reward"], atol=1e-6, what=f"reward step {step}")
        assert_close(env.mdp.step_reward, ref["step_reward"], atol=5e-5, what=f"step_reward step {step}")
        assert_close(env.mdp.episode_sums, ref["episode_sums"], atol=1e-6, what=f"episode_sums step {step}")
        assert_close(env.mdp.command, oracle.cmd.vel_command_b, atol=1e-6, what=f"command step {step}")
        nd = int(ref["dones"].sum())
        assert nd > 0 and ref["terminated"].any() and ref["time_outs"].any()
        log = env.mdp.log_sums.cpu()
        assert log[-1].item() == nd
        assert_close(log[:11] / nd / 20.0, ref["reward_log"], rtol=1e-4, atol=1e-6, what="episode reward log")
        assert_exact(log[11:13].long(), ref["term_log"], "termination counts")
        env.mdp.log_sums.zero_()
