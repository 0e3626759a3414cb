"""GPU: observation modifiers (HIP ``ilx_modifier_apply``) against the fixture recorded from the reference's functions and
classes, and inside the ObservationManager against the CPU restatement."""

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_function_modifiers(dev):
    from isaaclab_amd.utils import modifiers as M

    g = load_golden("modifiers")
    x = g["x"].to(dev)
    assert_exact(M.scale(x, 1.7), g["scale"], "scale")
    assert_exact(M.bias(x, -0.35), g["bias"], "bias")
    assert_exact(M.clip(x, (-1.5, 2.0)), g["clip_both"], "clip")
    assert_exact(M.clip(x, (-0.5, None)), g["clip_lo"], "clip, lower bound only")
    assert_exact(M.clip(x, (None, 0.25)), g["clip_hi"], "clip, upper bound only")
    # a column range of a wider row is read in place
    wide = torch.zeros(x.shape[0], 20, device=dev)
    wide[:, 5:12] = x
    assert_exact(M.scale(wide[:, 5:12], 1.7), g["scale"], "scale of a column block")
    # higher-rank data
    assert_exact(M.bias(x.reshape(12, 8, 7), -0.35).reshape(96, 7), g["bias"], "bias of a 3-D tensor")
    assert M.scale(torch.zeros(0, 7, device=dev), 2.0).shape == (0, 7)
    with pytest.raises(Exception, match="GPU tensors"):
        M.scale(g["x"], 1.7)


def test_digital_filter_and_integrator(dev):
    from isaaclab_amd.utils.modifiers import DigitalFilter, DigitalFilterCfg, Integrator, IntegratorCfg

    g = load_golden("modifiers")
    steps, shape = g["filter_in"].to(dev), tuple(g["x"].shape)
    ids = g["filter_reset_ids"]
    f = DigitalFilter(DigitalFilterCfg(A=g["filter_A"].tolist(), B=g["filter_B"].tolist()), shape, dev)
    for k in range(steps.shape[0]):
        if k == 5:
            f.reset(ids.tolist())
        if k == 7:
            f.reset()
        assert_close(f(steps[k]), g["filter_out"][k], rtol=1e-6, atol=1e-6, what=f"filter step {k}")
    assert_close(f.x_n, g["filter_x_n"], atol=1e-6, what="input history")
    assert_close(f.y_n, g["filter_y_n"], atol=1e-6, what="output history")
    for tag, a, b in (("delay", [0.0], [0.0, 1.0]), ("average", [0.0], [0.25] * 4), ("lowpass", [-0.8], [0.2])):
        f = DigitalFilter(DigitalFilterCfg(A=a, B=b), shape, dev)
        for k in range(6):
            assert_close(f(steps[k]), g[f"{tag}_out"][k], rtol=1e-6, atol=1e-6, what=f"{tag} step {k}")
    it = Integrator(IntegratorCfg(dt=0.02), shape, dev)
    for k in range(steps.shape[0]):
        if k == 4:
            it.reset(ids.to(dev))
        y = it(steps[k])
        assert y.data_ptr() == it.integral.data_ptr()          # the state itself is returned, as in the reference
        assert_exact(y, g["integrator_out"][k], f"integrator step {k}")
    assert_exact(it.y_prev, g["integrator_y_prev"], "integrator previous sample")
    with pytest.raises(ValueError, match="must not be None"):
        DigitalFilter(DigitalFilterCfg(A=None, B=[1.0]), shape, dev)
    with pytest.raises(ValueError, match="coefficients per side"):
        DigitalFilter(DigitalFilterCfg(A=[0.0], B=[0.1] * 17), shape, dev)


def test_observation_manager_applies_modifiers(dev):
    """Terms with modifiers: raw value from the fused launch, then modifiers, clip and scale on the term's columns
    (observation_manager.py:308-318), including the reference's in-place scaling of the Integrator's state."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from isaaclab_amd.utils import modifiers as M
    from oracle import modifiers as O
    from test_terms_gpu import make_env

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    env = make_env(s0, dev)
    A, B = [0.6, -0.2], [0.5, 0.3, 0.2]
    plain = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}), joint_pos=ObsTerm(mdp.joint_pos_rel, {}), joint_vel=ObsTerm(mdp.joint_vel_rel, {}))
    modified = dict(
        base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, modifiers=[M.IntegratorCfg(dt=0.02)], scale=0.9),
        joint_pos=ObsTerm(mdp.joint_pos_rel, {}, clip=(-0.2, 0.2), scale=2.0),
        joint_vel=ObsTerm(mdp.joint_vel_rel, {}, modifiers=[M.ModifierCfg(func=M.bias, params=dict(value=0.5)), M.DigitalFilterCfg(A=A, B=B)],
                          clip=(-1.0, 1.5), scale=0.25))
    om = ObservationManager(dict(plain=plain, modified=modified), env)
    assert om.group_obs_dim["modified"] == (27,)
    filt, integ = O.DigitalFilter(A, B, (n, 12)), O.Integrator(0.02, (n, 3))
    for step in range(steps):
        s = {k: v[step] for k, v in g.items() if v.shape[:1] == (steps,)}
        env.scene["robot"].data = make_env(s, dev).scene["robot"].data
        if step == 2:
            ids = [0, 3, n - 1]
            om.reset(ids)
            filt.reset(ids), integ.reset(ids)
        raw = om.compute_group("plain").cpu()
        got = om.compute_group("modified")
        integ.integral = integ(raw[:, :3]) * 0.9      # the scale lands in the state (mul_ on the tensor the modifier returned)
        want = torch.cat([integ.integral, raw[:, 3:15].clip(-0.2, 0.2) * 2.0, filt(raw[:, 15:] + 0.5).clip(-1.0, 1.5) * 0.25], dim=1)
        assert_close(got, want, rtol=1e-6, atol=1e-6, what=f"modified group step {step}")
    with pytest.raises(TypeError, match="required type ModifierCfg"):
        ObservationManager(dict(policy=dict(t=ObsTerm(mdp.base_lin_vel, {}, modifiers=[dict(func=M.scale)]))), env)
    with pytest.raises(TypeError, match="not a subclass of 'ModifierBase'"):
        ObservationManager(dict(policy=dict(t=ObsTerm(mdp.base_lin_vel, {}, modifiers=[M.ModifierCfg(func=dict)]))), env)


def test_observation_manager_noise_outside_the_fused_launch(dev):
    """Gaussian noise, multiplicative noise and noise behind modifiers (observation_manager.py:310-318) against the noise
    restatement with the same draws; without ``enable_corruption`` the noise is dropped (:398-400)."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from isaaclab_amd.utils import modifiers as M
    from isaaclab_amd.utils import noise as N
    from oracle import noise as O
    from test_terms_gpu import make_env

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    n = s0["joint_pos"].shape[0]
    env = make_env(s0, dev)

    def terms():
        return dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}, noise=N.UniformNoiseCfg(0.8, 1.2, operation="scale")),
                    joint_pos=ObsTerm(mdp.joint_pos_rel, {}, modifiers=[M.ModifierCfg(func=M.bias, params=dict(value=0.1))],
                                      noise=N.UniformNoiseCfg(-0.05, 0.05), scale=2.0),
                    joint_vel=ObsTerm(mdp.joint_vel_rel, {}, noise=N.GaussianNoiseCfg(mean=0.01, std=0.3), clip=(-1.0, 1.0)))
    plain = dict(base_lin_vel=ObsTerm(mdp.base_lin_vel, {}), joint_pos=ObsTerm(mdp.joint_pos_rel, {}), joint_vel=ObsTerm(mdp.joint_vel_rel, {}))
    om = ObservationManager(dict(plain=plain, noisy=dict(enable_corruption=True, **terms()), quiet=dict(**terms())), env)
    gen = torch.Generator().manual_seed(3)
    draws = dict(base_lin_vel=torch.rand(n, 3, generator=gen), joint_pos=torch.rand(n, 12, generator=gen), joint_vel=torch.randn(n, 12, generator=gen))
    om._post_noise_draws = {f"noisy/{k}": v.to(dev) for k, v in draws.items()}
    raw = om.compute_group("plain").cpu()
    lin, pos, vel = raw[:, :3], raw[:, 3:15], raw[:, 15:]
    want = torch.cat([O.uniform(lin, 0.8, 1.2, "scale", draws["base_lin_vel"]),
                      O.uniform(pos + 0.1, -0.05, 0.05, "add", draws["joint_pos"]) * 2.0,
                      O.gaussian(vel, 0.01, 0.3, "add", draws["joint_vel"]).clip(-1.0, 1.0)], dim=1)
    assert_close(om.compute_group("noisy"), want, rtol=1e-6, atol=1e-6, what="noise behind the fused launch")
    want_quiet = torch.cat([lin, (pos + 0.1) * 2.0, vel.clip(-1.0, 1.0)], dim=1)
    assert_close(om.compute_group("quiet"), want_quiet, rtol=1e-6, atol=1e-6, what="group without corruption")
    # generated draws: the noise is really there and differs from call to call
    om._post_noise_draws = {}
    a, b = om.compute_group("noisy").clone(), om.compute_group("noisy").clone()
    assert not torch.equal(a, b) and float((a[:, :3] - b[:, :3]).abs().max()) < 0.5 * float(lin.abs().max()) + 1e-6


def test_observation_manager_per_column_scale(dev):
    """observation_manager.py:424-440, 317-318: a tuple scale multiplies column by column."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from test_terms_gpu import make_env

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    s0 = {k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}
    env = make_env(s0, dev)
    om = ObservationManager(dict(plain=dict(v=ObsTerm(mdp.base_lin_vel, {}), q=ObsTerm(mdp.joint_pos_rel, {})),
                                 scaled=dict(v=ObsTerm(mdp.base_lin_vel, {}, clip=(-0.5, 0.5), scale=(2.0, -1.0, 0.5)), q=ObsTerm(mdp.joint_pos_rel, {}, scale=3.0))), env)
    raw = om.compute_group("plain").cpu()
    want = torch.cat([raw[:, :3].clip(-0.5, 0.5) * torch.tensor([2.0, -1.0, 0.5]), raw[:, 3:] * 3.0], dim=1)
    assert_close(om.compute_group("scaled"), want, rtol=1e-6, atol=1e-6, what="per-column scale")
    # observation_manager.py:425-437: checked when the manager is built (or at the first use of a lazily planned group)
    with pytest.raises(ValueError, match="does not match the dimensions of the observation. Expected: 3 but received: 2"):
        ObservationManager(dict(p=dict(v=ObsTerm(mdp.base_lin_vel, {}, scale=(1.0, 2.0)))), env).compute_group("p")
    with pytest.raises(TypeError, match="is not of type float, int or tuple"):
        ObservationManager(dict(p=dict(v=ObsTerm(mdp.base_lin_vel, {}, scale="2"))), env).compute_group("p")


def test_fused_noise_follows_the_environment_seed(dev):
    """Observation noise drawn inside the fused launch: two corrupted groups of one manager draw different noise for
    the same env and column, a re-seeded run replays, another seed draws other values."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import ObservationManager, ObservationTermCfg as ObsTerm
    from isaaclab_amd.utils import noise as N
    from isaaclab_amd.utils import rng
    from test_terms_gpu import make_env

    g = load_golden("managers_anymal")
    steps = g["reward"].shape[0]
    env = make_env({k: v[0] for k, v in g.items() if v.shape[:1] == (steps,)}, dev)

    def group():
        return dict(enable_corruption=True, v=ObsTerm(mdp.base_lin_vel, {}, noise=N.UniformNoiseCfg(-0.1, 0.1)),
                    q=ObsTerm(mdp.joint_pos_rel, {}, noise=N.UniformNoiseCfg(-0.01, 0.01)))

    def run(seed):
        rng.seed_all(seed)
        om = ObservationManager(dict(policy=group(), critic=group()), env)
        return [(om.compute_group("policy").clone(), om.compute_group("critic").clone()) for _ in range(2)]

    a, b, c = run(11), run(11), run(12)
    assert not torch.equal(a[0][0], a[0][1]), "two groups, same env and column: different noise"
    assert not torch.equal(a[0][0], a[1][0]), "a new step draws anew"
    for (p0, c0), (p1, c1) in zip(a, b):
        assert torch.equal(p0, p1) and torch.equal(c0, c1), "same seed replays"
    assert not torch.equal(a[0][0], c[0][0]), "another seed draws other values"
