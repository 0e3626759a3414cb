"""GPU: actuator / articulation-data / contact-sensor classes (HIP through the C ABI) against the oracle and
the reference-generated fixtures."""

import types

import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import actuators as OA

pytestmark = pytest.mark.gpu


def _names(j):
    return [f"joint_{i}" for i in range(j)]


def test_pd_actuators(dev):
    from isaaclab_amd.actuators import DCMotor, DCMotorCfg, IdealPDActuator, IdealPDActuatorCfg, ImplicitActuator
    from isaaclab_amd.utils.types import ArticulationActions

    g = load_golden("actuators_pd")
    n, j = g["ideal_q"].shape[1:]
    cfg_pd = IdealPDActuatorCfg(stiffness=25.0, damping=0.5, effort_limit=23.5, velocity_limit=30.0)
    cfg_dc = DCMotorCfg(stiffness={".*": 40.0}, damping={".*": 5.0}, effort_limit=80.0, velocity_limit=7.5,
                        saturation_effort=120.0)
    for tag, cls, cfg in (("ideal", IdealPDActuator, cfg_pd), ("implicit", ImplicitActuator, cfg_pd),
                          ("dcmotor", DCMotor, cfg_dc)):
        act = cls(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        assert_exact(act.stiffness, g[f"{tag}_stiffness"], f"{tag} stiffness buffer")
        for s in range(g[f"{tag}_q"].shape[0]):
            ca = ArticulationActions(joint_positions=g[f"{tag}_q_des"][s].to(dev), joint_velocities=g[f"{tag}_qd_des"][s].to(dev),
                                     joint_efforts=g[f"{tag}_tau_ff"][s].to(dev), joint_indices=slice(None))
            out = act.compute(ca, g[f"{tag}_q"][s].to(dev), g[f"{tag}_qd"][s].to(dev))
            assert_exact(act.computed_effort, g[f"{tag}_computed"][s], f"{tag} computed step {s}")
            assert_exact(act.applied_effort, g[f"{tag}_applied"][s], f"{tag} applied step {s}")
            if tag == "implicit":
                assert out.joint_positions is not None
            else:
                assert out.joint_positions is None and out.joint_efforts is act.applied_effort


def test_pd_with_the_joint_action_on_its_launch_equals_the_separate_launches(dev):
    """ilx_actuator_pd_joint_action (the step's first actuator launch with the action term on it) against
    ilx_joint_action_process + the copy into the target buffer + ilx_actuator_pd: every output bit for bit, for a position
    and an effort target, both models, an action term on a run of joints inside the articulation."""
    from isaaclab_amd import _abi
    from isaaclab_amd.actuators import DCMotor, DCMotorCfg, IdealPDActuator, IdealPDActuatorCfg
    from isaaclab_amd.utils.types import ArticulationActions

    n, j, first, width = 777, 9, 2, 5
    gen = torch.Generator().manual_seed(5)
    rnd = lambda *shape: torch.randn(*shape, generator=gen).to(dev)
    cfgs = ((IdealPDActuator, IdealPDActuatorCfg(stiffness=25.0, damping=0.5, effort_limit=23.5, velocity_limit=30.0)),
            (DCMotor, DCMotorCfg(stiffness=40.0, damping=5.0, effort_limit=80.0, velocity_limit=7.5, saturation_effort=120.0)))
    for cls, cfg in cfgs:
        for target in ("position", "effort"):
            raw, offset = rnd(n, width), (rnd(n, width) if target == "position" else None)
            q, qd, qd_des = rnd(n, j), rnd(n, j), rnd(n, j)
            q_des0, tau0, action0 = rnd(n, j), rnd(n, j), rnd(n, width)
            outs = []
            for fused in (False, True):
                act = cls(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
                q_des, tau_ff = q_des0.clone(), tau0.clone()
                action, prev = action0.clone(), torch.zeros(n, width, device=dev)
                log, count = torch.ones(5, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
                ca = ArticulationActions(joint_positions=q_des, joint_velocities=qd_des, joint_efforts=tau_ff, joint_indices=slice(None))
                if fused:
                    act.compute(ca, q, qd, joint_action=dict(raw=raw, action=action, prev_action=prev, scale=0.5, offset=offset, target=target,
                                                             first_joint=first, width=width, begin=(_abi.ptr(log), 5, count.data_ptr())))
                else:
                    processed = torch.empty(n, width, device=dev)
                    _abi.check(_abi.joint_action_process_begin(_abi.ptr(raw), _abi.ptr(action), _abi.ptr(prev), None, 0.5,
                                                               None if offset is None else _abi.ptr(offset), 0.0, None, _abi.ptr(processed),
                                                               n, width, _abi.ptr(log), 5, count.data_ptr(), _abi.stream()), "begin")
                    (q_des if target == "position" else tau_ff)[:, first:first + width] = processed
                    act.compute(ca, q, qd)
                outs.append((q_des, tau_ff, action, prev, log, count.float(), act.computed_effort.clone(), act.applied_effort.clone()))
            for a, b, name in zip(outs[0], outs[1], ("q_des", "tau_ff", "action", "prev_action", "log_sums", "step count", "computed", "applied")):
                assert_exact(b, a.cpu(), f"{cls.__name__} {target}: {name}")


def test_pd_regex_parameters(dev):
    from isaaclab_amd.actuators import DCMotor, DCMotorCfg

    g = load_golden("actuators_pd")
    names = ["left_hip_yaw", "left_knee", "torso", "left_ankle", "right_knee"]
    cfg = DCMotorCfg(stiffness={".*_hip_yaw": 150.0, ".*_knee": 200.0, "torso": 200.0, ".*_ankle": 20.0},
                     damping={".*_hip_yaw": 5.0, ".*_knee": 5.0, "torso": 5.0, ".*_ankle": 4.0},
                     effort_limit={".*_hip_yaw": 300.0, ".*_knee": 300.0, "torso": 300.0, ".*_ankle": 100.0},
                     velocity_limit=100.0, saturation_effort=300.0)
    act = DCMotor(cfg, joint_names=names, joint_ids=slice(None), num_envs=4, device=dev)
    assert_exact(act.stiffness, g["h1_names_stiffness"], "stiffness")
    assert_exact(act.damping, g["h1_names_damping"], "damping")
    assert_exact(act.effort_limit, g["h1_names_effort_limit"], "effort_limit")
    with pytest.raises(ValueError):
        DCMotor(DCMotorCfg(stiffness={"nope": 1.0}, damping=0.0, velocity_limit=1.0), joint_names=names,
                joint_ids=slice(None), num_envs=4, device=dev)


def test_delayed_and_remotized_pd(dev):
    """DelayedPDActuator / RemotizedPDActuator against the reference classes' fixture (9 substeps, a partial reset
    before step 5; the time lags the reference drew are installed, everything else is the class's own bookkeeping)."""
    from isaaclab_amd.actuators import DelayedPDActuator, DelayedPDActuatorCfg, RemotizedPDActuator, RemotizedPDActuatorCfg
    from isaaclab_amd.utils.types import ArticulationActions

    g = load_golden("actuators_delayed")
    cases = (("delayed", DelayedPDActuator, DelayedPDActuatorCfg(stiffness=60.0, damping=1.5, effort_limit=45.0, velocity_limit=20.0,
                                                                   min_delay=0, max_delay=4)),
             ("remotized", RemotizedPDActuator, RemotizedPDActuatorCfg(stiffness=60.0, damping=1.5, min_delay=1, max_delay=3,
                                                                       joint_parameter_lookup=g["lookup"].tolist())))
    for tag, cls, cfg in cases:
        n, j = g[f"{tag}_q"].shape[1:]
        act = cls(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        for s in range(g[f"{tag}_q"].shape[0]):
            if s in (0, 5):
                ids = slice(None) if s == 0 else [1, 3, 8]
                act.reset(ids)
                lags = act.positions_delay_buffer.time_lags
                assert int(lags.min()) >= cfg.min_delay and int(lags.max()) <= cfg.max_delay
                for b in (act.positions_delay_buffer, act.velocities_delay_buffer, act.efforts_delay_buffer):
                    b.set_time_lag(g[f"{tag}_lags"][s][ids].to(dev), ids)   # the lags the reference drew
            ca = ArticulationActions(joint_positions=g[f"{tag}_q_des"][s].to(dev), joint_velocities=g[f"{tag}_qd_des"][s].to(dev),
                                     joint_efforts=g[f"{tag}_tau_ff"][s].to(dev), joint_indices=slice(None))
            out = act.compute(ca, g[f"{tag}_q"][s].to(dev), g[f"{tag}_qd"][s].to(dev))
            assert_exact(act.computed_effort, g[f"{tag}_computed"][s], f"{tag} computed step {s}")
            assert_exact(act.applied_effort, g[f"{tag}_applied"][s], f"{tag} applied step {s}")
            assert out.joint_positions is None and out.joint_efforts is act.applied_effort
    with pytest.raises(ValueError):
        act.positions_delay_buffer.set_time_lag(9)
    with pytest.raises(TypeError):
        act.positions_delay_buffer.set_time_lag(torch.zeros(n))


@pytest.mark.parametrize("generic", [False, True])
def test_actuator_net_mlp(dev, generic):
    from isaaclab_amd.actuators import ActuatorNetMLP, ActuatorNetMLPCfg
    from isaaclab_amd.utils.types import ArticulationActions

    g = load_golden("actuators_mlp")
    for tag, actname, order, idx, lim, vl, sat in (("go1", "softsign", "pos_vel", [0, 1, 2], 23.7, 30.0, 23.7),
                                                   ("alt", "elu", "vel_pos", [0, 2, 4], 80.0, 7.5, 120.0)):
        nl = int(g[f"{tag}_nlayers"])
        net = dict(weights=[g[f"{tag}_w{i}"] for i in range(nl)], biases=[g[f"{tag}_b{i}"] for i in range(nl)],
                   activation=actname)
        n, j = g[f"{tag}_q"].shape[1:]
        cfg = ActuatorNetMLPCfg(network=net, pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0, input_order=order,
                                input_idx=idx, effort_limit=lim, velocity_limit=vl, saturation_effort=sat)
        act = ActuatorNetMLP(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        act._flags = 1 if generic else 0
        for s in range(g[f"{tag}_q"].shape[0]):
            if s == 5:
                act.reset([1, 3])
            ca = ArticulationActions(joint_positions=g[f"{tag}_q_des"][s].to(dev))
            act.compute(ca, g[f"{tag}_q"][s].to(dev), g[f"{tag}_qd"][s].to(dev))
            assert_exact(act._joint_pos_error_history, g[f"{tag}_pe_hist"][s], f"{tag} pos-error history step {s}")
            assert_exact(act._joint_vel_history, g[f"{tag}_v_hist"][s], f"{tag} velocity history step {s}")
            # torques are O(1..20): 1e-5 relative with an absolute floor of 2e-5 for near-zero outputs
            assert_close(act.computed_effort, g[f"{tag}_computed"][s], atol=2e-5, what=f"{tag} computed step {s}")
            assert_close(act.applied_effort, g[f"{tag}_applied"][s], atol=2e-5, what=f"{tag} applied step {s}")


def test_actuator_net_mlp_anymal_shape_vs_oracle(dev):
    """16k-env shape (196,608 rows), MFMA path vs the oracle; also a row count that is not a multiple of 32."""
    from isaaclab_amd.actuators import ActuatorNetMLP, ActuatorNetMLPCfg
    from isaaclab_amd.utils.types import ArticulationActions

    gen = torch.Generator().manual_seed(7)
    dims = [6, 32, 32, 32, 1]
    ws = [torch.randn(dims[i + 1], dims[i], generator=gen) / dims[i] ** 0.5 for i in range(4)]
    bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(4)]
    for n in (16384, 37):
        j = 12
        cfg = ActuatorNetMLPCfg(network=dict(weights=ws, biases=bs, activation="softsign"), pos_scale=-1.0, vel_scale=1.0,
                                torque_scale=1.0, input_order="pos_vel", input_idx=[0, 1, 2], effort_limit=80.0,
                                velocity_limit=7.5, saturation_effort=120.0)
        act = ActuatorNetMLP(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        pe, vh = torch.zeros(n, 3, j), torch.zeros(n, 3, j)
        lim, vl = torch.full((n, j), 80.0), torch.full((n, j), 7.5)
        for s in range(3):
            q_des = torch.randn(n, j, generator=gen)
            q = q_des + torch.randn(n, j, generator=gen) * 0.3
            qd = torch.randn(n, j, generator=gen) * 3.0
            comp, app, pe, vh = OA.actuator_net_mlp(q_des, q, qd, pe, vh, ws, bs, "softsign", [0, 1, 2], "pos_vel", -1.0,
                                                    1.0, 1.0, lim, vl, 120.0)
            act.compute(ArticulationActions(joint_positions=q_des.to(dev)), q.to(dev), qd.to(dev))
            assert_exact(act._joint_pos_error_history, pe, f"n={n} history step {s}")
            assert_close(act.computed_effort, comp, atol=2e-5, what=f"n={n} computed step {s}")
            assert_close(act.applied_effort, app, atol=2e-5, what=f"n={n} applied step {s}")


def test_actuator_net_mlp_split_and_f32_products(dev):
    """The default evaluation (hidden layers on the f16 matrix cores, factors split into head and remainder) against
    ILX_MLP_EXACT_F32 (every product in f32 on the f32 MFMA) and the generic kernel (plain fmaf chains, exact division in
    softsign): all three within 1e-5 of the output scale, also with large hidden weights."""
    from isaaclab_amd.actuators import ActuatorNetMLP, ActuatorNetMLPCfg
    from isaaclab_amd.utils.types import ArticulationActions

    gen = torch.Generator().manual_seed(17)
    dims = [6, 32, 32, 32, 1]
    n, j = 4096, 12
    for gain in (1.0, 6.0):   # gain 6: pre-activations of tens, softsign saturated, large products
        ws = [gain * torch.randn(dims[i + 1], dims[i], generator=gen) / dims[i] ** 0.5 for i in range(4)]
        bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(4)]
        out = {}
        for flags in (0, 1, 2):
            cfg = ActuatorNetMLPCfg(network=dict(weights=ws, biases=bs, activation="softsign"), pos_scale=-1.0, vel_scale=1.0,
                                    torque_scale=1.0, input_order="pos_vel", input_idx=[0, 1, 2], effort_limit=1e6,
                                    velocity_limit=7.5, saturation_effort=1e6)
            act = ActuatorNetMLP(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
            act._flags = flags
            g2 = torch.Generator().manual_seed(3)
            for _ in range(3):
                q_des = torch.randn(n, j, generator=g2)
                act.compute(ArticulationActions(joint_positions=q_des.to(dev)), (q_des + 0.3 * torch.randn(n, j, generator=g2)).to(dev),
                            (3.0 * torch.randn(n, j, generator=g2)).to(dev))
            out[flags] = act.computed_effort.clone()
        scale = float(out[1].abs().max())
        for flags, what in ((2, "f32 MFMA kernel"), (0, "split evaluation")):
            err = float((out[flags] - out[1]).abs().max())
            assert err <= 1e-5 * scale, f"gain {gain}: {what} off the generic kernel by {err:.3g} (outputs up to {scale:.3g})"
        assert not torch.equal(out[0], out[2]), "the default path is the split one"


def _mlp_float64(act, ws, bs, activation):
    """The network of an ActuatorNetMLP on the histories it holds (actuator_net.py:158-178), every operation in float64."""
    fn = {"softsign": torch.nn.functional.softsign, "elu": torch.nn.functional.elu, "relu": torch.relu, "tanh": torch.tanh}[activation]
    idx = list(act.cfg.input_idx)
    pe = act._joint_pos_error_history.double().cpu()[:, idx, :]      # (n, K, j)
    ve = act._joint_vel_history.double().cpu()[:, idx, :]
    n, k, j = pe.shape
    x = torch.cat([pe.permute(0, 2, 1).reshape(n * j, k) * act.cfg.pos_scale, ve.permute(0, 2, 1).reshape(n * j, k) * act.cfg.vel_scale], -1)
    for i, (w, b) in enumerate(zip(ws, bs)):
        x = x @ w.double().T + b.double()
        if i + 1 < len(ws):
            x = fn(x)
    return (x.reshape(n, j) * act.cfg.torque_scale)


def _run_mlp(dev, ws, bs, activation, flags, n=2048, j=12, steps=3, seed=3, amp=(1.0, 0.3, 3.0)):
    from isaaclab_amd.actuators import ActuatorNetMLP, ActuatorNetMLPCfg
    from isaaclab_amd.utils.types import ArticulationActions

    cfg = ActuatorNetMLPCfg(network=dict(weights=ws, biases=bs, activation=activation), pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0,
                            input_order="pos_vel", input_idx=[0, 1, 2], effort_limit=1e30, velocity_limit=7.5, saturation_effort=1e30)
    act = ActuatorNetMLP(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
    act._flags = flags
    g2 = torch.Generator().manual_seed(seed)
    for _ in range(steps):
        q_des = amp[0] * torch.randn(n, j, generator=g2)
        act.compute(ArticulationActions(joint_positions=q_des.to(dev)), (q_des + amp[1] * torch.randn(n, j, generator=g2)).to(dev),
                    (amp[2] * torch.randn(n, j, generator=g2)).to(dev))
    return act


def _random_net(gen, gain=1.0, hidden_gain=None, dims=(6, 32, 32, 32, 1)):
    hidden_gain = gain if hidden_gain is None else hidden_gain
    ws = [(hidden_gain if 0 < i < len(dims) - 2 else gain) * torch.randn(dims[i + 1], dims[i], generator=gen) / dims[i] ** 0.5
          for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(len(dims) - 1)]
    return ws, bs


def test_actuator_net_mlp_element_wise_against_float64(dev):
    """Every output of the default evaluation (split-f16 hidden layers) and of ILX_MLP_EXACT_F32 against the float64
    evaluation of the same network on the same histories: |err| <= 1e-5 |ref| + 2e-5, element by element, for the headline
    net's shape at two gains (gain 6: saturated softsign, products of tens) and for a tanh net."""
    gen = torch.Generator().manual_seed(29)
    for activation, gain in (("softsign", 1.0), ("softsign", 6.0), ("tanh", 1.5)):
        ws, bs = _random_net(gen, gain)
        for flags in (0, 2):
            act = _run_mlp(dev, ws, bs, activation, flags)
            ref = _mlp_float64(act, ws, bs, activation)
            err = (act.computed_effort.double().cpu() - ref).abs()
            bad = err > 1e-5 * ref.abs() + 2e-5
            assert not bool(bad.any()), (f"{activation} gain {gain} flags {flags}: {int(bad.sum())} outputs beyond 1e-5 |ref| + 2e-5; "
                                         f"largest error {float(err.max()):.3g} at |ref| {float(ref.abs().flatten()[err.argmax()]):.3g}")


def test_actuator_net_mlp_unbounded_activations_stay_in_f32(dev):
    """ELU / ReLU networks whose hidden pre-activations reach 1e5 (beyond the f16 range of 65 504): never the split-f16
    kernel -- the default evaluation is the f32 one bit for bit, and matches float64 element-wise to float32's own accuracy."""
    gen = torch.Generator().manual_seed(31)
    for activation in ("elu", "relu"):
        ws, bs = _random_net(gen, gain=40.0)     # first-layer outputs of hundreds, hidden pre-activations of 1e4 .. 1e6
        out = {}
        for flags in (0, 2):
            act = _run_mlp(dev, ws, bs, activation, flags, amp=(2.0, 1.0, 10.0))
            out[flags] = act.computed_effort.clone()
        ref = _mlp_float64(act, ws, bs, activation)
        assert float(ref.abs().max()) > 1e5, "the case is meant to leave the f16 range"
        assert torch.equal(out[0], out[2]), f"{activation}: the default path must be the f32 one"
        # sums of terms of 1e6 .. 1e7 in float32: the absolute part of the bound scales with them (cancellation)
        err = (out[0].double().cpu() - ref).abs()
        scale = float(ref.abs().max())
        assert bool((err <= 1e-5 * ref.abs() + 1e-6 * scale).all()), f"{activation}: largest error {float(err.max()):.3g} (outputs up to {scale:.3g})"


def test_actuator_net_mlp_small_factors(dev):
    """Hidden weights of magnitude 1e-4 .. 1e-3 and with them hidden activations of that size (f16 heads near and below
    the f16 normal range, remainders subnormal): the split evaluation stays within 1e-5 |ref| + 2e-5 of float64, and in fact
    within 1e-6 absolute."""
    gen = torch.Generator().manual_seed(37)
    ws, bs = _random_net(gen, gain=1.0, hidden_gain=3e-3)    # hidden weights ~ N(0, 5e-4)
    bs = [b * (1e-3 if 0 < i < 3 else 1.0) for i, b in enumerate(bs)]
    assert 1e-4 < float(ws[1].abs().median()) < 1e-3
    act = _run_mlp(dev, ws, bs, "softsign", 0)
    ref = _mlp_float64(act, ws, bs, "softsign")
    err = (act.computed_effort.double().cpu() - ref).abs()
    assert bool((err <= 1e-5 * ref.abs() + 2e-5).all()) and float(err.max()) < 1e-6, f"largest error {float(err.max()):.3g}"


def _lstm_net(g, tag, act):
    hidden, layers, nhead = (int(v) for v in g[f"{tag}_shape"])
    lstm = {f"{nm}_l{l}": g[f"{tag}_{nm}_l{l}"] for l in range(layers) for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    return dict(lstm=lstm, weights=[g[f"{tag}_head_w{i}"] for i in range(nhead)],
                biases=[g[f"{tag}_head_b{i}"] for i in range(nhead)], activation=act)


def test_actuator_net_lstm(dev):
    """ActuatorNetLSTM against the reference class's fixture (8 substeps, a reset before step 5)."""
    from isaaclab_amd.actuators import ActuatorNetLSTM, ActuatorNetLSTMCfg
    from isaaclab_amd.utils.types import ArticulationActions

    g = load_golden("actuators_lstm")
    for tag, actname in (("anydrive", "softsign"), ("wide", "tanh")):
        n, j = g[f"{tag}_q"].shape[1:]
        cfg = ActuatorNetLSTMCfg(network=_lstm_net(g, tag, actname), effort_limit=80.0, velocity_limit=7.5, saturation_effort=120.0)
        act = ActuatorNetLSTM(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        for s in range(g[f"{tag}_q"].shape[0]):
            if s == 5:
                act.reset([1, 3])
            out = act.compute(ArticulationActions(joint_positions=g[f"{tag}_q_des"][s].to(dev)), g[f"{tag}_q"][s].to(dev),
                              g[f"{tag}_qd"][s].to(dev))
            # gates go through exp / tanh: states are in (-1, 1) (cell: a few units), compared at 1e-5 with a 5e-6 floor
            assert_close(act.sea_hidden_state, g[f"{tag}_hidden"][s], atol=5e-6, what=f"{tag} hidden state step {s}")
            assert_close(act.sea_cell_state, g[f"{tag}_cell"][s], atol=5e-6, what=f"{tag} cell state step {s}")
            assert_close(act.computed_effort, g[f"{tag}_computed"][s], atol=2e-5, what=f"{tag} computed step {s}")
            assert_close(act.applied_effort, g[f"{tag}_applied"][s], atol=2e-5, what=f"{tag} applied step {s}")
            assert out.joint_positions is None and out.joint_efforts is act.applied_effort


def test_actuator_net_lstm_anymal_shape_vs_oracle(dev):
    """16k-env shape (196,608 rows) and a ragged row count against the oracle; hidden size not a multiple of 4 as well."""
    from isaaclab_amd.actuators import ActuatorNetLSTM, ActuatorNetLSTMCfg
    from isaaclab_amd.utils.types import ArticulationActions

    gen = torch.Generator().manual_seed(11)
    for n, hidden, layers, head in ((16384, 8, 2, [32]), (37, 8, 2, [32]), (53, 6, 1, [])):
        j = 12
        lstm = {}
        for l in range(layers):
            in_l = 2 if l == 0 else hidden
            lstm[f"weight_ih_l{l}"] = torch.randn(4 * hidden, in_l, generator=gen) * 0.4
            lstm[f"weight_hh_l{l}"] = torch.randn(4 * hidden, hidden, generator=gen) * 0.4
            lstm[f"bias_ih_l{l}"] = torch.randn(4 * hidden, generator=gen) * 0.1
            lstm[f"bias_hh_l{l}"] = torch.randn(4 * hidden, generator=gen) * 0.1
        dims = [hidden] + head + [1]
        ws = [torch.randn(dims[i + 1], dims[i], generator=gen) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
        bs = [torch.randn(dims[i + 1], generator=gen) * 0.1 for i in range(len(dims) - 1)]
        cfg = ActuatorNetLSTMCfg(network=dict(lstm=lstm, weights=ws, biases=bs, activation="softsign"), effort_limit=80.0,
                                 velocity_limit=7.5, saturation_effort=120.0)
        act = ActuatorNetLSTM(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device=dev)
        params = tuple([lstm[f"{nm}_l{l}"] for l in range(layers)] for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
        h, c = torch.zeros(layers, n * j, hidden), torch.zeros(layers, n * j, hidden)
        lim, vl = torch.full((n, j), 80.0), torch.full((n, j), 7.5)
        for s in range(3):
            q_des = torch.randn(n, j, generator=gen)
            q = q_des + torch.randn(n, j, generator=gen) * 0.3
            qd = torch.randn(n, j, generator=gen) * 3.0
            comp, app, h, c = OA.actuator_net_lstm(q_des, q, qd, h, c, params, ws, bs, "softsign", lim, vl, 120.0)
            act.compute(ArticulationActions(joint_positions=q_des.to(dev)), q.to(dev), qd.to(dev))
            assert_close(act.sea_hidden_state, h, atol=5e-6, what=f"n={n} hidden step {s}")
            assert_close(act.sea_cell_state, c, atol=5e-6, what=f"n={n} cell step {s}")
            assert_close(act.computed_effort, comp, atol=2e-5, what=f"n={n} computed step {s}")
            assert_close(act.applied_effort, app, atol=2e-5, what=f"n={n} applied step {s}")


def test_articulation_data(dev):
    from isaaclab_amd.assets.articulation_data import ArticulationData

    g = load_golden("articulation_data")
    n, j = g["joint_vels"].shape[1:]
    data = ArticulationData(n, j, 1, dev)
    data._previous_joint_vel = g["joint_vels"][0].to(dev).clone()
    data.set_sim_state(root_transforms_xyzw=g["root_transforms_xyzw"].to(dev), root_velocities=g["root_velocities"].to(dev))
    assert_exact(data.root_state_w, g["root_state_w"], "root_state_w")
    assert_exact(data.GRAVITY_VEC_W[0], g["gravity_vec"], "gravity direction")
    for i in range(g["joint_acc"].shape[0]):
        data.set_sim_state(joint_vel=g["joint_vels"][i + 1].to(dev))
        data.update(float(g["dts"][i]))
        assert_exact(data.joint_acc, g["joint_acc"][i], f"joint_acc step {i}")
    assert_exact(data.root_lin_vel_b, g["root_lin_vel_b"], "root_lin_vel_b")
    assert_exact(data.root_ang_vel_b, g["root_ang_vel_b"], "root_ang_vel_b")
    assert_exact(data.projected_gravity_b, g["projected_gravity_b"], "projected_gravity_b")
    assert_close(data.heading_w, g["heading_w"], what="heading_w")


def _drive_contact(dev, g, period, reset_step=None, reset_ids=None):
    from isaaclab_amd.sensors.contact_sensor import ContactSensor, ContactSensorCfg

    f = g["forces"]
    steps, n, b, _ = f.shape
    h = g["hist"].shape[2]
    s = ContactSensor(ContactSensorCfg(update_period=period, history_length=h, track_air_time=True, force_threshold=1.0),
                      n, [f"body_{i}" for i in range(b)], dev)
    for k in range(steps):
        if reset_step is not None and k == reset_step:
            s.reset(reset_ids)
        s.set_net_contact_forces(f[k].to(dev))
        s.update(0.005)
        yield k, s


def test_contact_sensor(dev):
    g = load_golden("contact_sensor")
    for k, s in _drive_contact(dev, g, 0.005, int(g["reset_step"]), g["reset_ids"].tolist()):
        d = s.data
        for key, val in (("net", d.net_forces_w), ("hist", d.net_forces_w_history), ("cur_air", d.current_air_time),
                         ("last_air", d.last_air_time), ("cur_contact", d.current_contact_time),
                         ("last_contact", d.last_contact_time), ("ts", s._timestamp), ("ts_last", s._timestamp_last_update)):
            assert_exact(val, g[key][k], f"{key} step {k}")
        assert_exact(s.compute_first_contact(0.02), g["first_contact"][k], f"first_contact step {k}")
        assert_exact(s.compute_first_air(0.02), g["first_air"][k], f"first_air step {k}")


def test_contact_sensor_net_forces_from_history(dev):
    """A sensor that leaves the copy into net_forces_w out (net_forces_w = NULL at the C ABI) hands out the same
    net_forces_w -- the newest history entry -- and the same everything else as the reference class, reset included."""
    g = load_golden("contact_sensor")
    for period, net_key, hist_key in ((0.005, "net", "hist"),):
        for k, s in _drive_contact(dev, g, period, int(g["reset_step"]), g["reset_ids"].tolist()):
            s.net_forces_from_history = True
            if k == 0:
                s._data.net_forces_w.fill_(123.0)        # never written from here on
            d = s.data
            for key, val in ((net_key, d.net_forces_w), (hist_key, d.net_forces_w_history), ("cur_air", d.current_air_time),
                             ("last_air", d.last_air_time), ("cur_contact", d.current_contact_time),
                             ("last_contact", d.last_contact_time), ("ts", s._timestamp), ("ts_last", s._timestamp_last_update)):
                assert_exact(val, g[key][k], f"{key} step {k}")
        reset_ids = g["reset_ids"].tolist()
        keep = [i for i in range(s._data.net_forces_w.shape[0]) if i not in reset_ids]
        assert (s._data.net_forces_w[keep] == 123.0).all()


def test_contact_sensor_slow_period(dev):
    g = load_golden("contact_sensor")
    for k, s in _drive_contact(dev, g, 0.02):
        assert_exact(s.ordered_history(), g["slow_hist"][k], f"slow hist step {k}")
        assert_exact(s._data.current_air_time, g["slow_cur_air"][k], f"slow cur_air step {k}")
        assert_exact(s._data.current_contact_time, g["slow_cur_contact"][k], f"slow cur_contact step {k}")
        assert_exact(s._is_outdated, g["slow_outdated"][k], f"slow outdated step {k}")


def test_contact_sensor_many_bodies(dev):
    """More bodies than threads in a block (the strided path), vs the oracle."""
    from isaaclab_amd.sensors.contact_sensor import ContactSensor, ContactSensorCfg
    from oracle import contact as OC

    gen = torch.Generator().manual_seed(5)
    n, b, h = 5, 300, 2
    s = ContactSensor(ContactSensorCfg(update_period=0.0, history_length=h, track_air_time=True), n,
                      [f"b{i}" for i in range(b)], dev)
    st = dict(net=torch.zeros(n, b, 3), hist=torch.zeros(n, h, b, 3), ca=torch.zeros(n, b), la=torch.zeros(n, b),
              cc=torch.zeros(n, b), lc=torch.zeros(n, b), ts=torch.zeros(n), tl=torch.zeros(n),
              od=torch.ones(n, dtype=torch.bool))
    for k in range(4):
        f = torch.randn(n, b, 3, generator=gen) * (torch.rand(n, b, 1, generator=gen) < 0.5)
        OC.contact_update(f, st["net"], st["hist"], st["ca"], st["la"], st["cc"], st["lc"], st["ts"], st["tl"], st["od"],
                          0.01, 0.0, 1.0)
        s.set_net_contact_forces(f.to(dev))
        s.update(0.01)
        assert_exact(s.data.net_forces_w_history, st["hist"], f"hist step {k}")
        assert_exact(s.data.current_air_time, st["ca"], f"cur_air step {k}")
        assert_exact(s.data.last_contact_time, st["lc"], f"last_contact step {k}")


def test_command_list_of_one_call_and_the_refusal_of_two_streams(dev):
    """A list records any entry point: one math call kept, then issued by the replay (its output buffer is written only
    then); calls on two streams inside one recording are refused when the recording ends."""
    from isaaclab_amd.utils import math as m
    from isaaclab_amd.utils.command_list import CommandList

    q = torch.rand(64, 4, device=dev) - 0.5
    direct = m.quat_mul(q, q).clone()
    commands = CommandList()
    with commands.record():
        out = m.quat_mul(q, q)
    assert len(commands) == 1
    out.zero_()
    torch.cuda.synchronize()
    commands.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, direct)
    side = torch.cuda.Stream()
    with pytest.raises(Exception, match="more than one stream"):
        with commands.record():
            m.quat_mul(q, q)
            with torch.cuda.stream(side):
                m.quat_mul(q, q)
    with pytest.raises(Exception, match="more than one stream"):
        commands.replay()
