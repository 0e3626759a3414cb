"""CPU: oracle actuators / articulation data / contact bookkeeping against reference-generated fixtures."""

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import actuators as A
from oracle import articulation as AD
from oracle import contact as C


def test_pd_actuators_bit_exact():
    g = load_golden("actuators_pd")
    for tag in ("ideal", "implicit", "dcmotor"):
        for s in range(g[f"{tag}_q"].shape[0]):
            args = [g[f"{tag}_{k}"][s] for k in ("q_des", "qd_des", "tau_ff", "q", "qd")]
            kp, kd, lim, vl = (g[f"{tag}_{k}"] for k in ("stiffness", "damping", "effort_limit", "velocity_limit"))
            if tag == "dcmotor":
                comp, app = A.dc_motor(*args, kp, kd, lim, vl, 120.0)
            else:
                comp, app = A.ideal_pd(*args, kp, kd, lim)
            assert_exact(comp, g[f"{tag}_computed"][s], f"{tag} computed step {s}")
            assert_exact(app, g[f"{tag}_applied"][s], f"{tag} applied step {s}")
    # the DC-motor envelope actually clips some rows, both ways
    assert (g["dcmotor_applied"] != g["dcmotor_computed"]).float().mean() > 0.02


def _mlp_params(g, tag):
    nl = int(g[f"{tag}_nlayers"])
    return [g[f"{tag}_w{i}"] for i in range(nl)], [g[f"{tag}_b{i}"] for i in range(nl)]


def test_actuator_net_mlp():
    g = load_golden("actuators_mlp")
    for tag, act, order, idx, lim, vl, sat in (("go1", "softsign", "pos_vel", [0, 1, 2], 23.7, 30.0, 23.7),
                                               ("alt", "elu", "vel_pos", [0, 2, 4], 80.0, 7.5, 120.0)):
        w, b = _mlp_params(g, tag)
        n, j = g[f"{tag}_q"].shape[1:]
        k = max(idx) + 1
        pe, vh = torch.zeros(n, k, j), torch.zeros(n, k, j)
        limt, vlt = torch.full((n, j), lim), torch.full((n, j), vl)
        for s in range(g[f"{tag}_q"].shape[0]):
            if s == 5:  # the fixture resets envs 1 and 3 before step 5 (actuator_net.py:142-145)
                pe[[1, 3]] = 0.0
                vh[[1, 3]] = 0.0
            comp, app, pe, vh = A.actuator_net_mlp(g[f"{tag}_q_des"][s], g[f"{tag}_q"][s], g[f"{tag}_qd"][s], pe, vh, w, b,
                                                   act, idx, order, -1.0, 1.0, 1.0, limt, vlt, sat)
            assert_exact(pe, g[f"{tag}_pe_hist"][s], f"{tag} pos-error history step {s}")
            assert_exact(vh, g[f"{tag}_v_hist"][s], f"{tag} velocity history step {s}")
            assert_close(comp, g[f"{tag}_computed"][s], atol=1e-5, what=f"{tag} computed step {s}")
            assert_close(app, g[f"{tag}_applied"][s], atol=1e-5, what=f"{tag} applied step {s}")


def lstm_params(g, tag):
    hidden, layers, nhead = (int(v) for v in g[f"{tag}_shape"])
    lstm = tuple([g[f"{tag}_{nm}_l{l}"] for l in range(layers)] for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    return hidden, layers, lstm, [g[f"{tag}_head_w{i}"] for i in range(nhead)], [g[f"{tag}_head_b{i}"] for i in range(nhead)]


def test_actuator_net_lstm():
    g = load_golden("actuators_lstm")
    for tag, act in (("anydrive", "softsign"), ("wide", "tanh")):
        hidden, layers, lstm, hw, hb = lstm_params(g, tag)
        n, j = g[f"{tag}_q"].shape[1:]
        h, c = torch.zeros(layers, n * j, hidden), torch.zeros(layers, n * j, hidden)
        limt, vlt = torch.full((n, j), 80.0), torch.full((n, j), 7.5)
        for s in range(g[f"{tag}_q"].shape[0]):
            if s == 5:  # the fixture resets envs 1 and 3 before step 5 (actuator_net.py:70-74)
                h.view(layers, n, j, hidden)[:, [1, 3]] = 0.0
                c.view(layers, n, j, hidden)[:, [1, 3]] = 0.0
            comp, app, h, c = A.actuator_net_lstm(g[f"{tag}_q_des"][s], g[f"{tag}_q"][s], g[f"{tag}_qd"][s], h, c, lstm, hw, hb,
                                                  act, limt, vlt, 120.0)
            assert_close(h, g[f"{tag}_hidden"][s], atol=2e-6, what=f"{tag} hidden state step {s}")
            assert_close(c, g[f"{tag}_cell"][s], atol=2e-6, what=f"{tag} cell state step {s}")
            assert_close(comp, g[f"{tag}_computed"][s], atol=1e-5, what=f"{tag} computed step {s}")
            assert_close(app, g[f"{tag}_applied"][s], atol=1e-5, what=f"{tag} applied step {s}")
        assert (g[f"{tag}_applied"] != g[f"{tag}_computed"]).any()


def test_delayed_and_remotized_pd():
    g = load_golden("actuators_delayed")
    for tag, max_delay in (("delayed", 4), ("remotized", 3)):
        n, j = g[f"{tag}_q"].shape[1:]
        lines = [A.DelayLine(max_delay, n) for _ in range(3)]
        kp, kd = torch.full((n, j), 60.0), torch.full((n, j), 1.5)
        lookup = (g["lookup"][:, 0], g["lookup"][:, 2]) if tag == "remotized" else None
        for s in range(g[f"{tag}_q"].shape[0]):
            if s in (0, 5):  # resets draw new lags (recorded in the fixture) and clear the rings (actuator_pd.py:262-281)
                ids = slice(None) if s == 0 else [1, 3, 8]
                for ln in lines:
                    ln.time_lags[ids] = g[f"{tag}_lags"][s][ids]
                    ln.reset(ids)
            comp, app = A.delayed_pd(lines, g[f"{tag}_q_des"][s], g[f"{tag}_qd_des"][s], g[f"{tag}_tau_ff"][s], g[f"{tag}_q"][s],
                                     g[f"{tag}_qd"][s], kp, kd, g[f"{tag}_effort_limit"], lookup)
            assert_exact(comp, g[f"{tag}_computed"][s], f"{tag} computed step {s}")
            assert_exact(app, g[f"{tag}_applied"][s], f"{tag} applied step {s}")
        assert (g[f"{tag}_applied"] != g[f"{tag}_computed"]).float().mean() > 0.02


def test_articulation_data():
    g = load_golden("articulation_data")
    rs = AD.root_state_from_view(g["root_transforms_xyzw"], g["root_velocities"])
    assert_exact(rs, g["root_state_w"], "root_state_w")
    assert_exact(AD.gravity_dir(), g["gravity_vec"], "gravity direction")
    assert_exact(AD.root_lin_vel_b(rs), g["root_lin_vel_b"], "root_lin_vel_b")
    assert_exact(AD.root_ang_vel_b(rs), g["root_ang_vel_b"], "root_ang_vel_b")
    assert_exact(AD.projected_gravity_b(rs, g["gravity_vec"]), g["projected_gravity_b"], "projected_gravity_b")
    assert_close(AD.heading_w(rs), g["heading_w"], what="heading_w")
    prev = g["joint_vels"][0]
    for i in range(g["joint_acc"].shape[0]):
        acc, prev = AD.joint_acc(g["joint_vels"][i + 1], prev, float(g["time_elapsed"][i]))
        assert_exact(acc, g["joint_acc"][i], f"joint_acc step {i}")


def _run_contact(g, period, steps, reset_step=None, reset_ids=None):
    f = g["forces"]
    _, n, b, _ = f.shape
    h = g["hist"].shape[2]
    st = dict(net=torch.zeros(n, b, 3), hist=torch.zeros(n, h, b, 3), cur_air=torch.zeros(n, b), last_air=torch.zeros(n, b),
              cur_contact=torch.zeros(n, b), last_contact=torch.zeros(n, b), ts=torch.zeros(n), ts_last=torch.zeros(n),
              outdated=torch.ones(n, dtype=torch.bool))
    for k in range(steps):
        if reset_step is not None and k == reset_step:
            C.contact_reset(reset_ids, st["net"], st["hist"], st["cur_air"], st["last_air"], st["cur_contact"],
                            st["last_contact"], st["ts"], st["ts_last"], st["outdated"])
        C.contact_update(f[k], st["net"], st["hist"], st["cur_air"], st["last_air"], st["cur_contact"], st["last_contact"],
                         st["ts"], st["ts_last"], st["outdated"], 0.005, period, 1.0)
        yield k, st


def test_contact_sensor_every_substep():
    g = load_golden("contact_sensor")
    steps = g["forces"].shape[0]
    for k, st in _run_contact(g, 0.005, steps, int(g["reset_step"]), g["reset_ids"].tolist()):
        for key in ("net", "hist", "cur_air", "last_air", "cur_contact", "last_contact", "ts", "ts_last"):
            assert_exact(st[key], g[key][k], f"{key} step {k}")
        assert_exact(C.compute_first_contact(st["cur_contact"], 0.02), g["first_contact"][k], f"first_contact step {k}")
        assert_exact(C.compute_first_air(st["cur_air"], 0.02), g["first_air"][k], f"first_air step {k}")
    assert g["first_contact"].any() and g["first_air"].any()


def test_contact_sensor_slow_period():
    g = load_golden("contact_sensor")
    for k, st in _run_contact(g, 0.02, g["forces"].shape[0]):
        assert_exact(st["hist"], g["slow_hist"][k], f"slow hist step {k}")
        assert_exact(st["cur_air"], g["slow_cur_air"][k], f"slow cur_air step {k}")
        assert_exact(st["cur_contact"], g["slow_cur_contact"][k], f"slow cur_contact step {k}")
        assert_exact(st["outdated"], g["slow_outdated"][k], f"slow outdated step {k}")
