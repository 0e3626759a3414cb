"""EventManager mode dispatch against the script recorded from the reference's EventManager: the same term calls with
the same env ids in the same order, the same interval timers after every call (CPU: the manager is host-side logic on
torch tensors), plus one GPU case driving a HIP event term."""

from types import SimpleNamespace as NS

import pytest
import torch

from conftest import assert_exact, load_golden


def _replay(device):
    from isaaclab_amd.managers import EventManager, EventTermCfg, ManagerTermBase

    g = load_golden("event_manager")
    n, dt = int(g["num_envs"]), float(g["dt"])
    calls = []

    def make_term(tag):
        def term(env, env_ids, bias=0.0):
            if env_ids is None:
                ids = torch.tensor([-1])
            elif isinstance(env_ids, slice):
                ids = torch.arange(n)
            else:
                ids = torch.as_tensor(env_ids).cpu().clone()
            calls.append((tag, ids))
        return term

    class ClassTerm(ManagerTermBase):
        def reset(self, env_ids=None):
            calls.append((4, torch.tensor([-2]) if env_ids is None else torch.as_tensor(env_ids).cpu() + 1000))

        def __call__(self, env, env_ids, bias=0.0):
            calls.append((3, torch.as_tensor(env_ids).cpu().clone()))

    draws = list(torch.split(g["draws"], g["draw_sizes"].tolist()))

    class Replayed(EventManager):
        def _rand(self, count, device):
            d = draws.pop(0)
            assert d.numel() == count, "the manager draws what the reference draws, in the same order"
            return d.to(device)

    cfg = NS(
        startup_mass=EventTermCfg(func=make_term(0), mode="startup", params={"bias": 1.0}),
        push_each=EventTermCfg(func=make_term(1), mode="interval", interval_range_s=(0.05, 0.12)),
        gust_global=EventTermCfg(func=make_term(2), mode="interval", interval_range_s=(0.06, 0.09), is_global_time=True),
        class_push=EventTermCfg(func=ClassTerm, mode="interval", interval_range_s=(0.03, 0.2)),
        reset_always=EventTermCfg(func=make_term(5), mode="reset"),
        reset_throttled=EventTermCfg(func=make_term(6), mode="reset", min_step_count_between_reset=3),
        disabled=None,
    )
    mgr = Replayed(cfg, NS(num_envs=n, device=device, scene=NS()))
    assert mgr.active_terms == {"startup": ["startup_mass"], "interval": ["push_each", "gust_global", "class_push"],
                                "reset": ["reset_always", "reset_throttled"]}
    assert mgr.available_modes == ["startup", "interval", "reset"]
    timers = []

    def snap():
        t = mgr._interval_term_time_left
        timers.append(torch.stack([t[0].cpu(), t[1].expand(n).cpu(), t[2].cpu()]))

    snap()
    for kind, arg in zip(g["script_kinds"].tolist(), torch.split(g["script_args"], g["script_sizes"].tolist())):
        if kind == 0:
            mgr.apply("startup")
        elif kind == 1:
            mgr.apply("interval", dt=dt)
            snap()
        elif kind == 3:
            assert mgr.reset(arg.to(device)) == {}
        else:
            step, ids = int(arg[0]), arg[1:]
            mgr.apply("reset", env_ids=ids.to(device) if len(ids) else None, global_env_step_count=step)
            snap()
    assert not draws
    assert [c[0] for c in calls] == g["call_tags"].tolist()
    for k, (c, want) in enumerate(zip(calls, torch.split(g["call_ids"], g["call_sizes"].tolist()))):
        assert_exact(c[1], want, f"env ids of call {k} (term {c[0]})")
    assert_exact(torch.stack(timers), g["timers"], "interval timers")
    assert_exact(mgr._reset_term_last_triggered_step_id[0], g["last_step_always"], "last triggered (every reset)")
    assert_exact(mgr._reset_term_last_triggered_step_id[1], g["last_step_throttled"], "last triggered (throttled)")
    return mgr


def test_event_manager_matches_reference_script_cpu():
    _replay("cpu")


def test_event_manager_argument_errors():
    from isaaclab_amd.managers import EventManager, EventTermCfg

    env = NS(num_envs=4, device="cpu", scene=NS())
    f = lambda env, env_ids, scale: None  # noqa: E731
    with pytest.raises(TypeError, match="is not of type EventTermCfg"):
        EventManager(dict(a=3), env)
    with pytest.raises(ValueError, match="'interval_range_s' is not specified"):
        EventManager(dict(a=EventTermCfg(func=f, mode="interval", params={"scale": 1.0})), env)
    with pytest.raises(ValueError, match="is negative"):
        EventManager(dict(a=EventTermCfg(func=f, mode="reset", params={"scale": 1.0}, min_step_count_between_reset=-1)), env)
    with pytest.raises(ValueError, match="expects mandatory parameters"):
        EventManager(dict(a=EventTermCfg(func=f, mode="reset")), env)
    with pytest.raises(AttributeError, match="not callable"):
        EventManager(dict(a=EventTermCfg(func=3, mode="reset")), env)
    mgr = EventManager(dict(a=EventTermCfg(func=f, mode="interval", interval_range_s=(1.0, 2.0), params={"scale": 1.0}),
                            b=EventTermCfg(func=f, mode="reset", params={"scale": 1.0})), env)
    with pytest.raises(ValueError, match="requires the time-step"):
        mgr.apply("interval")
    with pytest.raises(ValueError, match="does not require environment indices"):
        mgr.apply("interval", env_ids=[0], dt=0.1)
    with pytest.raises(ValueError, match="total number of environment steps"):
        mgr.apply("reset", env_ids=[0])
    with pytest.warns(UserWarning, match="not defined"):
        mgr.apply("startup")
    assert mgr.get_term_cfg("b").mode == "reset"
    with pytest.raises(ValueError, match="not found"):
        mgr.get_term_cfg("c")


@pytest.mark.gpu
def test_event_manager_on_device_with_hip_term(dev):
    """The same script with the manager's tensors on the GPU, then an interval push through the HIP event kernel."""
    from isaaclab_amd.envs import mdp
    from isaaclab_amd.managers import EventManager, EventTermCfg, SceneEntityCfg

    _replay(dev)
    n = 64
    data = NS(root_state_w=torch.zeros(n, 13, device=dev))
    scene = {"robot": NS(data=data)}
    env = NS(num_envs=n, device=dev, scene=scene)
    cfg = dict(push_robot=EventTermCfg(func=mdp.push_by_setting_velocity, mode="interval", interval_range_s=(0.1, 0.3),
                                       params={"velocity_range": {"x": (-0.5, 0.5), "y": (-0.5, 0.5)},
                                               "asset_cfg": SceneEntityCfg("robot")}))
    mgr = EventManager(cfg, env)
    pushes = torch.zeros(n)
    for _ in range(10):
        before = mgr._interval_term_time_left[0].clone()
        mgr.apply("interval", dt=0.05)
        pushes += (before - 0.05 < 1e-6).cpu()
        v = data.root_state_w.cpu()
        assert (v[pushes == 0, 7:9] == 0).all(), "only the due envs were pushed"
        assert (v[:, 7:9].abs() <= 0.5 * pushes[:, None] + 1e-6).all(), "each push adds a draw from the range to the velocity"
    assert (pushes > 0).all() and (data.root_state_w[:, 7:9].abs().sum(-1) > 0).all()
