"""Golden vectors for the EventManager's mode dispatch, produced by the REFERENCE class
(``isaaclab/managers/event_manager.py``) driving recording terms: which env ids each term call receives over a scripted
sequence of ``apply`` / ``reset`` calls, the interval timers after every call, and the ``torch.rand`` draws consumed
(recorded in call order so that the replay is independent of the generator).  Writes tests/golden/event_manager.npz."""

import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
em = R.load("isaaclab.managers.event_manager", f"{R.L}/managers/event_manager.py")
EventTermCfg = m["manager_term_cfg"].EventTermCfg
ManagerTermBase = m["manager_base"].ManagerTermBase

draws = []
_rand = torch.rand


def recording_rand(*size, **kw):
    r = _rand(*size, **{k: v for k, v in kw.items() if k != "device"})
    draws.append(r.clone().flatten())
    return r


em.torch = NS(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
em.torch.rand = recording_rand

n = 24
calls = []   # (term tag, env ids or -1 for None)


def make_term(tag):
    def term(env, env_ids, bias=0.0):
        if env_ids is None:
            ids = torch.tensor([-1])
        elif isinstance(env_ids, slice):
            ids = torch.arange(n)
        else:
            ids = torch.as_tensor(env_ids).clone()
        calls.append((tag, ids))
    return term


class ClassTerm(ManagerTermBase):
    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        self.resets = []

    def reset(self, env_ids=None):
        calls.append((4, torch.tensor([-2]) if env_ids is None else torch.as_tensor(env_ids).clone() + 1000))

    def __call__(self, env, env_ids, bias=0.0):
        calls.append((3, torch.as_tensor(env_ids).clone()))


torch.manual_seed(4242)
env = NS(num_envs=n, device="cpu", scene=NS(), sim=NS(is_playing=lambda: False))
cfg = NS(
    startup_mass=EventTermCfg(func=make_term(0), mode="startup", params={"bias": 1.0}),
    push_each=EventTermCfg(func=make_term(1), mode="interval", interval_range_s=(0.05, 0.12)),
    gust_global=EventTermCfg(func=make_term(2), mode="interval", interval_range_s=(0.06, 0.09), is_global_time=True),
    class_push=EventTermCfg(func=ClassTerm, mode="interval", interval_range_s=(0.03, 0.2)),
    reset_always=EventTermCfg(func=make_term(5), mode="reset"),
    reset_throttled=EventTermCfg(func=make_term(6), mode="reset", min_step_count_between_reset=3),
    disabled=None,
)
mgr = em.EventManager(cfg, env)
print(mgr.active_terms)
g = torch.Generator().manual_seed(77)
script, timers = [], []


def snap():
    timers.append(torch.stack([mgr._interval_term_time_left[0].clone(), mgr._interval_term_time_left[1].expand(n).clone(),
                               mgr._interval_term_time_left[2].clone()]))


snap()
mgr.apply("startup")
script.append((0, torch.tensor([-1])))
dt = 0.02
for step in range(1, 13):
    mgr.apply("interval", dt=dt)
    script.append((1, torch.tensor([-1])))
    snap()
    if step % 2 == 0:  # a batch of envs finishes its episode
        ids = torch.randperm(n, generator=g)[: 5 + step % 3].sort().values
        mgr.reset(ids)
        script.append((3, ids))
        mgr.apply("reset", env_ids=ids, global_env_step_count=step)
        script.append((2, torch.cat([torch.tensor([step]), ids])))
        snap()
mgr.apply("reset", global_env_step_count=20)   # env_ids None -> every env
script.append((2, torch.tensor([20])))
snap()

out = {"num_envs": torch.tensor(n), "dt": torch.tensor(dt, dtype=torch.float64)}
out["draws"] = torch.cat(draws)
out["draw_sizes"] = torch.tensor([len(d) for d in draws])
out["call_tags"] = torch.tensor([c[0] for c in calls])
out["call_sizes"] = torch.tensor([len(c[1]) for c in calls])
out["call_ids"] = torch.cat([c[1] for c in calls])
out["script_kinds"] = torch.tensor([s[0] for s in script])
out["script_sizes"] = torch.tensor([len(s[1]) for s in script])
out["script_args"] = torch.cat([s[1] for s in script])
out["timers"] = torch.stack(timers)
out["last_step_throttled"] = mgr._reset_term_last_triggered_step_id[1].clone()
out["last_step_always"] = mgr._reset_term_last_triggered_step_id[0].clone()
np.savez_compressed(os.path.join(HERE, "event_manager.npz"), **{k: v.numpy() for k, v in out.items()})
print("calls", len(calls), "draws", len(draws), "timers", out["timers"].shape)
