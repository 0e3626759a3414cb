"""Golden fixtures for the actuator models and the derived ArticulationData quantities, produced by the
reference's own classes (actuator_pd.py, actuator_net.py, articulation_data.py)."""

from __future__ import annotations

import io
import sys
import types

import torch

import _refload as R
from gen_golden import gen, save

SECTIONS = {}


def section(fn):
    SECTIONS[fn.__name__] = fn
    return fn


class Cfg(types.SimpleNamespace):
    pass


def _names(j):
    return [f"joint_{i}" for i in range(j)]


def _run_pd_like(cls, cfg, n, j, steps, g, reset_at=None):
    act = cls(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device="cpu")
    Types = R.bootstrap()["types"]
    rec = dict(q_des=[], qd_des=[], tau_ff=[], q=[], qd=[], computed=[], applied=[])
    for s in range(steps):
        q_des = torch.randn(n, j, generator=g)
        qd_des = torch.randn(n, j, generator=g) * 0.5
        tau_ff = torch.randn(n, j, generator=g) * 2.0
        q = q_des + torch.randn(n, j, generator=g) * 0.6
        qd = torch.randn(n, j, generator=g) * 6.0
        if reset_at is not None and s == reset_at:
            act.reset([0, 2, 5])
        ca = Types.ArticulationActions(joint_positions=q_des.clone(), joint_velocities=qd_des.clone(),
                                       joint_efforts=tau_ff.clone(), joint_indices=slice(None))
        act.compute(ca, q.clone(), qd.clone())
        for k, v in zip(("q_des", "qd_des", "tau_ff", "q", "qd", "computed", "applied"),
                        (q_des, qd_des, tau_ff, q, qd, act.computed_effort.clone(), act.applied_effort.clone())):
            rec[k].append(v)
    out = {k: torch.stack(v) for k, v in rec.items()}
    out.update(stiffness=act.stiffness, damping=act.damping, effort_limit=act.effort_limit,
               velocity_limit=act.velocity_limit)
    return out


@section
def actuators_pd():
    m = R.bootstrap()
    pd = m["actuator_pd"]
    g = gen(2001)
    n, j = 96, 12
    base = dict(joint_names_expr=[".*"], armature=None, friction=None)
    # ANYmal-C simple DC motor constants (isaaclab_assets/robots/anymal.py:34-41)
    cfg_dc = Cfg(stiffness={".*": 40.0}, damping={".*": 5.0}, effort_limit=80.0, velocity_limit=7.5,
                 saturation_effort=120.0, **base)
    cfg_pd = Cfg(stiffness=25.0, damping=0.5, effort_limit=23.5, velocity_limit=30.0, **base)
    out = {}
    for tag, cls, cfg in (("ideal", pd.IdealPDActuator, cfg_pd), ("implicit", pd.ImplicitActuator, cfg_pd),
                          ("dcmotor", pd.DCMotor, cfg_dc)):
        r = _run_pd_like(cls, cfg, n, j, 6, g)
        out.update({f"{tag}_{k}": v for k, v in r.items()})
    # per-joint dict parameters (Unitree H1 legs style, unitree.py:216-260), DCMotor
    names = ["left_hip_yaw", "left_knee", "torso", "left_ankle", "right_knee"]
    cfg_h1 = Cfg(joint_names_expr=[".*"], armature=None, friction=None,
                 stiffness={".*_hip_yaw": 150.0, ".*_knee": 200.0, "torso": 200.0, ".*_ankle": 20.0},
                 damping={".*_hip_yaw": 5.0, ".*_knee": 5.0, "torso": 5.0, ".*_ankle": 4.0},
                 effort_limit={".*_hip_yaw": 300.0, ".*_knee": 300.0, "torso": 300.0, ".*_ankle": 100.0},
                 velocity_limit=100.0, saturation_effort=300.0)
    act = pd.DCMotor(cfg_h1, joint_names=names, joint_ids=slice(None), num_envs=4, device="cpu")
    out["h1_names_stiffness"] = act.stiffness
    out["h1_names_damping"] = act.damping
    out["h1_names_effort_limit"] = act.effort_limit
    save("actuators_pd", **out)


def make_mlp(g, in_dim=6, hidden=32, layers=3, act="softsign"):
    acts = {"softsign": torch.nn.Softsign, "elu": torch.nn.ELU, "tanh": torch.nn.Tanh, "relu": torch.nn.ReLU}
    mods, d = [], in_dim
    for _ in range(layers):
        mods += [torch.nn.Linear(d, hidden), acts[act]()]
        d = hidden
    mods.append(torch.nn.Linear(d, 1))
    net = torch.nn.Sequential(*mods)
    with torch.no_grad():
        for mod in net:
            if isinstance(mod, torch.nn.Linear):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) / (mod.in_features ** 0.5))
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    return net


@section
def actuators_mlp():
    m = R.bootstrap()
    Types = m["types"]
    netmod = m["actuator_net"]
    g = gen(7)
    n, j, steps = 40, 12, 8
    out = {}
    for tag, act_name, order, idx, hidden, layers in (
        ("go1", "softsign", "pos_vel", [0, 1, 2], 32, 3),  # Unitree Go1 parameters (unitree.py:30-41)
        ("alt", "elu", "vel_pos", [0, 2, 4], 32, 2),
    ):
        net = make_mlp(g, in_dim=2 * len(idx), hidden=hidden, layers=layers, act=act_name)
        buf = io.BytesIO()
        torch.jit.save(torch.jit.script(net), buf)
        buf.seek(0)
        sys.modules["isaaclab.utils.assets"].read_file = lambda p, b=buf: b
        netmod.read_file = lambda p, b=buf: b
        cfg = Cfg(joint_names_expr=[".*"], armature=None, friction=None, stiffness=None, damping=None,
                  network_file="synthetic", pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0, input_order=order,
                  input_idx=idx, effort_limit=23.7 if tag == "go1" else 80.0,
                  velocity_limit=30.0 if tag == "go1" else 7.5, saturation_effort=23.7 if tag == "go1" else 120.0)
        act = netmod.ActuatorNetMLP(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device="cpu",
                                    stiffness=0.0, damping=0.0)
        rec = dict(q_des=[], q=[], qd=[], computed=[], applied=[], pe_hist=[], v_hist=[])
        for s in range(steps):
            q_des = torch.randn(n, j, generator=g)
            q = q_des + torch.randn(n, j, generator=g) * 0.3
            qd = torch.randn(n, j, generator=g) * (12.0 if tag == "go1" else 4.0)
            if s == 5:
                act.reset([1, 3])
            ca = Types.ArticulationActions(joint_positions=q_des.clone(), joint_velocities=torch.zeros(n, j),
                                           joint_efforts=torch.zeros(n, j), joint_indices=slice(None))
            with torch.no_grad():
                act.compute(ca, q.clone(), qd.clone())
            for k, v in zip(rec, (q_des, q, qd, act.computed_effort.detach().clone(), act.applied_effort.detach().clone(),
                                  act._joint_pos_error_history.clone(), act._joint_vel_history.clone())):
                rec[k].append(v)
        out.update({f"{tag}_{k}": torch.stack(v) for k, v in rec.items()})
        lin = [mod for mod in net if isinstance(mod, torch.nn.Linear)]
        for li, mod in enumerate(lin):
            out[f"{tag}_w{li}"] = mod.weight.detach()
            out[f"{tag}_b{li}"] = mod.bias.detach()
        out[f"{tag}_nlayers"] = torch.tensor(len(lin))
    save("actuators_mlp", **out)


class LstmNet(torch.nn.Module):
    """Shape of the reference's LSTM actuator nets (actuator_net.py:48-54 reads ``network.lstm``): an ``nn.LSTM`` over
    (position error, velocity) followed by a small head on the last layer's output."""

    def __init__(self, hidden: int, layers: int, head: list[int], act: str):
        super().__init__()
        acts = {"softsign": torch.nn.Softsign, "elu": torch.nn.ELU, "tanh": torch.nn.Tanh, "relu": torch.nn.ReLU}
        self.lstm = torch.nn.LSTM(2, hidden, layers, batch_first=True)
        mods, d = [], hidden
        for w in head:
            mods += [torch.nn.Linear(d, w), acts[act]()]
            d = w
        mods.append(torch.nn.Linear(d, 1))
        self.mlp = torch.nn.Sequential(*mods)

    def forward(self, x: torch.Tensor, hc: tuple[torch.Tensor, torch.Tensor]):
        out, hc2 = self.lstm(x, hc)
        return self.mlp(out[:, -1]), hc2


@section
def actuators_lstm():
    m = R.bootstrap()
    Types = m["types"]
    netmod = m["actuator_net"]
    g = gen(11)
    n, j, steps = 40, 12, 8
    out = {}
    for tag, hidden, layers, head, act_name in (("anydrive", 8, 2, [32], "softsign"), ("wide", 16, 3, [], "tanh")):
        net = LstmNet(hidden, layers, head, act_name)
        with torch.no_grad():
            for prm in net.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.4 if prm.dim() > 1 else 0.1))
        buf = io.BytesIO()
        torch.jit.save(torch.jit.script(net), buf)
        buf.seek(0)
        sys.modules["isaaclab.utils.assets"].read_file = lambda p, b=buf: b
        netmod.read_file = lambda p, b=buf: b
        # ANYmal-C ANYdrive 3 constants (isaaclab_assets/robots/anymal.py:43-51)
        cfg = Cfg(joint_names_expr=[".*"], armature=None, friction=None, stiffness=None, damping=None,
                  network_file="synthetic", effort_limit=80.0, velocity_limit=7.5, saturation_effort=120.0)
        act = netmod.ActuatorNetLSTM(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device="cpu",
                                     stiffness=0.0, damping=0.0)
        rec = dict(q_des=[], q=[], qd=[], computed=[], applied=[], hidden=[], cell=[])
        for s in range(steps):
            q_des = torch.randn(n, j, generator=g)
            q = q_des + torch.randn(n, j, generator=g) * 0.3
            qd = torch.randn(n, j, generator=g) * 4.0
            if s == 5:
                act.reset([1, 3])
            ca = Types.ArticulationActions(joint_positions=q_des.clone(), joint_velocities=torch.zeros(n, j),
                                           joint_efforts=torch.zeros(n, j), joint_indices=slice(None))
            act.compute(ca, q.clone(), qd.clone())
            for k, v in zip(rec, (q_des, q, qd, act.computed_effort.detach().clone(), act.applied_effort.detach().clone(),
                                  act.sea_hidden_state.clone(), act.sea_cell_state.clone())):
                rec[k].append(v)
        out.update({f"{tag}_{k}": torch.stack(v) for k, v in rec.items()})
        sd = net.lstm.state_dict()
        for li in range(layers):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                out[f"{tag}_{nm}_l{li}"] = sd[f"{nm}_l{li}"].detach()
        lin = [mod for mod in net.mlp if isinstance(mod, torch.nn.Linear)]
        for li, mod in enumerate(lin):
            out[f"{tag}_head_w{li}"] = mod.weight.detach()
            out[f"{tag}_head_b{li}"] = mod.bias.detach()
        out[f"{tag}_shape"] = torch.tensor([hidden, layers, len(lin)])
    save("actuators_lstm", **out)


@section
def actuators_delayed():
    """DelayedPDActuator and RemotizedPDActuator (actuator_pd.py:236-363) on the reference's own DelayBuffer,
    CircularBuffer and LinearInterpolation."""
    m = R.bootstrap()
    Types = m["types"]
    pd = m["actuator_pd"]
    cb = R.load("isaaclab.utils.buffers.circular_buffer", f"{R.L}/utils/buffers/circular_buffer.py")
    sys.modules["isaaclab.utils.buffers"] = types.ModuleType("isaaclab.utils.buffers")
    sys.modules["isaaclab.utils.buffers"].circular_buffer = cb
    db = R.load("isaaclab.utils.buffers.delay_buffer", f"{R.L}/utils/buffers/delay_buffer.py")
    li = R.load("isaaclab.utils.interpolation.linear_interpolation", f"{R.L}/utils/interpolation/linear_interpolation.py")
    pd.DelayBuffer, pd.LinearInterpolation = db.DelayBuffer, li.LinearInterpolation
    g = gen(4101)
    torch.manual_seed(4101)  # reset() draws the time lags from the global generator (actuator_pd.py:270-276)
    n, j, steps = 48, 12, 9
    base = dict(joint_names_expr=[".*"], armature=None, friction=None)
    # Spot's knee lookup shape (isaaclab_assets/robots/spot.py): angle, transmission ratio, max torque; ascending angles
    ang = torch.linspace(-2.79, -0.25, 12)
    lookup = torch.stack([ang, torch.linspace(-24.8, -8.9, 12), 37.2 + 80.0 * torch.sin((ang + 2.79) * 1.1)], dim=1).tolist()
    out = {"lookup": torch.tensor(lookup)}
    for tag, cls, cfg in (
        ("delayed", pd.DelayedPDActuator, Cfg(stiffness=60.0, damping=1.5, effort_limit=45.0, velocity_limit=20.0, min_delay=0,
                                              max_delay=4, **base)),
        ("remotized", pd.RemotizedPDActuator, Cfg(stiffness=60.0, damping=1.5, effort_limit=None, velocity_limit=None,
                                                  min_delay=1, max_delay=3, joint_parameter_lookup=lookup, **base)),
    ):
        act = cls(cfg, joint_names=_names(j), joint_ids=slice(None), num_envs=n, device="cpu")
        rec = dict(q_des=[], qd_des=[], tau_ff=[], q=[], qd=[], computed=[], applied=[], lags=[])
        for s in range(steps):
            q_des = torch.randn(n, j, generator=g) - (1.5 if tag == "remotized" else 0.0)
            qd_des = torch.randn(n, j, generator=g) * 0.5
            tau_ff = torch.randn(n, j, generator=g) * 2.0
            q = q_des + torch.randn(n, j, generator=g) * 0.8
            qd = torch.randn(n, j, generator=g) * 6.0
            if s == 0:
                act.reset(slice(None))   # as Articulation.reset does before the first step
            if s == 5:
                act.reset([1, 3, 8])
            ca = Types.ArticulationActions(joint_positions=q_des.clone(), joint_velocities=qd_des.clone(),
                                           joint_efforts=tau_ff.clone(), joint_indices=slice(None))
            act.compute(ca, q.clone(), qd.clone())
            for k, v in zip(rec, (q_des, qd_des, tau_ff, q, qd, act.computed_effort.clone(), act.applied_effort.clone(),
                                  act.positions_delay_buffer.time_lags.clone())):
                rec[k].append(v)
        out.update({f"{tag}_{k}": torch.stack(v) for k, v in rec.items()})
        out[f"{tag}_effort_limit"] = act.effort_limit
        assert len(set(rec["lags"][0].tolist())) > 1
    save("actuators_delayed", **out)


@section
def articulation_data():
    """Real ArticulationData on a fake physics view: derived root quantities and joint_acc."""
    m = R.bootstrap()
    physx = sys.modules["omni.physics.tensors.impl.api"]

    class SimView:
        def set_subspace_roots(self, p):
            pass

        def get_gravity(self):
            return (0.0, 0.0, -9.81)

    physx.create_simulation_view = lambda backend: SimView()
    ad = R.load("isaaclab.assets.articulation.articulation_data", f"{R.L}/assets/articulation/articulation_data.py")
    g = gen(3001)
    n, j = 128, 12
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    root_tf = torch.cat([torch.randn(n, 3, generator=g), q[:, [1, 2, 3, 0]]], dim=-1)  # xyzw like PhysX
    root_vel = torch.randn(n, 6, generator=g)
    vels = [torch.randn(n, j, generator=g) * 2.0 for _ in range(4)]

    class View:
        count = n

        def __init__(self):
            self.k = 0

        def get_root_transforms(self):
            return root_tf.clone()

        def get_root_velocities(self):
            return root_vel.clone()

        def get_dof_positions(self):
            return torch.zeros(n, j)

        def get_dof_velocities(self):
            return vels[self.k].clone()

    view = View()
    data = ad.ArticulationData(view, "cpu")
    dts = [0.005, 0.005, 0.005]
    accs = []
    for i, dt in enumerate(dts):
        view.k = i + 1
        data.update(dt)
        accs.append(data.joint_acc.clone())
    out = dict(root_transforms_xyzw=root_tf, root_velocities=root_vel, joint_vels=torch.stack(vels),
               dts=torch.tensor(dts, dtype=torch.float64), joint_acc=torch.stack(accs),
               root_state_w=data.root_state_w, root_lin_vel_b=data.root_lin_vel_b, root_ang_vel_b=data.root_ang_vel_b,
               projected_gravity_b=data.projected_gravity_b, heading_w=data.heading_w,
               gravity_vec=data.GRAVITY_VEC_W[0])
    # the elapsed times the reference divided by (sim timestamp differences in double)
    # (TimestampedBuffer.timestamp starts at -1.0, so the first difference is 1 + dt: timestamped_buffer.py)
    ts, el = 0.0, []
    last = -1.0
    for dt in dts:
        ts += dt
        el.append(ts - last)
        last = ts
    out["time_elapsed"] = torch.tensor(el, dtype=torch.float64)
    save("articulation_data", **out)
