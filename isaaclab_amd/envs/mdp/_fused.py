"""Glue between the reference's term-function seam and the fused MDP kernel.

A term function of the reference is ``func(env, **params) -> Tensor`` (``manager_term_cfg.py:28-52``).  Here every
library term is declared by a *builder* that turns ``params`` into a :class:`Spec` (kernel term kind + parameters +
which scene entities it reads).  The decorators wrap the builder into a function with the reference's call
signature -- calling it evaluates a cached one-term plan through ``ilx_mdp_step`` -- and keep the builder on the
function (``func._ilx_build``) so that the managers can compile whole term lists into one plan.
"""

from __future__ import annotations

import functools
from dataclasses import dataclass, field

import torch

from ... import _abi
from .. import fused_mdp as FM


@dataclass
class Spec:
    kind: str
    p: tuple = (0.0, 0.0, 0.0, 0.0)
    ids: object = None
    ids2: object = None
    flags: int = 0
    aux: float = 0.0
    asset: str | None = "robot"      # articulation entity the term reads
    contact: str | None = None       # contact sensor entity
    ray: str | None = None           # ray caster entity
    command: str | None = None       # command term name
    term_names: list | None = None   # is_terminated_term: termination terms to count
    external: object = None          # kind "external": env -> the (num_envs, ...) tensor the term hands out as it is

    def bindings(self):
        return (self.asset, self.contact, self.ray, self.command)


def _key(v):
    if hasattr(v, "__dict__") and not isinstance(v, type):
        return tuple(sorted((k, _key(x)) for k, x in vars(v).items()))
    if isinstance(v, dict):
        return tuple(sorted((k, _key(x)) for k, x in v.items()))
    if isinstance(v, slice):
        return ("slice", v.start, v.stop, v.step)
    if isinstance(v, (list, tuple)):
        return tuple(_key(x) for x in v)
    if isinstance(v, torch.Tensor):
        return tuple(v.tolist())
    return v


def _entity(env, name):
    scene = env.scene
    sensors = getattr(scene, "sensors", None)
    if sensors is not None and name in sensors:
        return sensors[name]
    return scene[name]


def sizes(env, asset, contact, ray, command):
    """Row widths of the scene entities a plan reads."""
    out = dict(num_joints=1, num_bodies=1, num_sensor_bodies=0, history_len=0, num_rays=0, action_dim=0, command_dim=0)
    if asset is not None:
        d = _entity(env, asset).data
        out["num_joints"], out["num_bodies"] = d.joint_pos.shape[1], d.body_state_w.shape[1]
    if contact is not None:
        h = _entity(env, contact).data.net_forces_w_history
        out["history_len"], out["num_sensor_bodies"] = h.shape[1], h.shape[2]
    if ray is not None:
        out["num_rays"] = _entity(env, ray).data.ray_hits_w.shape[1]
    am = getattr(env, "action_manager", None)
    if am is not None:
        out["action_dim"] = am.action.shape[1]
    if command is not None:
        out["command_dim"] = env.command_manager.get_command(command).shape[1]
    return out


def bind_env(mdp: FM.FusedMdp, env, asset, contact, ray, command):
    """Point the plan's descriptor at the env's current tensors (cheap: pointer stores)."""
    b = {}
    if asset is None:  # no articulation involved: the kernel still wants a root state row per env
        zero = mdp.__dict__.get("_zero_root")
        if zero is None:
            zero = mdp.__dict__["_zero_root"] = torch.zeros(env.num_envs, 13, device=mdp.device)
            zero[:, 3] = 1.0
        b["root_state_w"] = zero
    if asset is not None:
        d = _entity(env, asset).data
        b.update(root_state_w=d.root_state_w, joint_pos=d.joint_pos, joint_vel=d.joint_vel, joint_acc=d.joint_acc,
                 applied_torque=d.applied_torque, computed_torque=d.computed_torque, default_joint_pos=d.default_joint_pos,
                 default_joint_vel=d.default_joint_vel, soft_joint_pos_limits=d.soft_joint_pos_limits,
                 soft_joint_vel_limits=d.soft_joint_vel_limits, body_state_w=d.body_state_w)
        acc = getattr(d, "body_lin_acc_w", None)
        if acc is None and getattr(d, "body_acc_w", None) is not None:
            acc = d.body_acc_w[..., 0:3].contiguous()
        if acc is not None:
            b["body_lin_acc_w"] = acc
    if contact is not None:
        ent = _entity(env, contact)
        # the kernels take a maximum over the history and need no order: bind the sensor's own (ring) storage
        c = ent.raw_data if hasattr(ent, "raw_data") else ent.data
        b["net_forces_w_history"] = c.net_forces_w_history
        if c.current_air_time is not None:
            b.update(current_air_time=c.current_air_time, last_air_time=c.last_air_time,
                     current_contact_time=c.current_contact_time, last_contact_time=c.last_contact_time)
    if ray is not None:
        r = _entity(env, ray).data
        b.update(ray_hits_w=r.ray_hits_w, sensor_pos_w=r.pos_w)
    am = getattr(env, "action_manager", None)
    if am is not None:
        b.update(action=am.action, prev_action=am.prev_action)
    if command is not None:
        b["command"] = env.command_manager.get_command(command)
    origins = getattr(env.scene, "env_origins", None)
    if origins is not None:
        b["env_origins"] = origins
    mdp.bind(**{k: _abi.f32c(v) for k, v in b.items()})
    if getattr(env, "episode_length_buf", None) is not None:
        mdp.bind(episode_length_buf=env.episode_length_buf)


def merge_bindings(a, b):
    """Union of two binding tuples, or None when they name different entities for the same slot."""
    out = []
    for x, y in zip(a, b):
        if x is not None and y is not None and x != y:
            return None
        out.append(x if x is not None else y)
    return tuple(out)


def group_specs(specs):
    """Partition term indices into groups whose bindings can share one plan: [(bindings, [indices])]."""
    groups = []
    for i, s in enumerate(specs):
        for g in groups:
            m = merge_bindings(g[0], s.bindings())
            if m is not None:
                g[0], _ = m, g[1].append(i)
                break
        else:
            groups.append([s.bindings(), [i]])
    return [(tuple(g[0]), g[1]) for g in groups]


def to_termspec(group: str, spec: Spec, weight: float = 1.0, post=None) -> FM.TermSpec:
    if group == "observation":
        post = post or {}
        t = FM.obs_term(spec.kind, noise=post.get("noise"), clip=post.get("clip"), scale=post.get("scale"), ids=spec.ids,
                        offset=spec.aux, make_quat_unique=bool(spec.flags & FM.OBS_QUAT_UNIQUE))
        return t
    return FM.TermSpec(spec.kind, weight=weight, p=spec.p, ids=spec.ids, ids2=spec.ids2, flags=spec.flags, aux=spec.aux)


def make_plan(env, group: str, specs, weights=None, posts=None, step_dt: float = 1.0, time_out_flags=None,
              bindings=None) -> FM.FusedMdp:
    """One plan for terms that share their bindings."""
    asset, contact, ray, command = bindings if bindings is not None else specs[0].bindings()
    sz = sizes(env, asset, contact, ray, command)
    terms = []
    external_dims = []
    for i, s in enumerate(specs):
        if s.kind == "external":      # the term's slot among the plan's external buffers travels in `aux`
            n = env.num_envs
            s.aux = float(len(external_dims))
            external_dims.append(s.external(env).numel() // max(n, 1))
        t = to_termspec(group, s, 1.0 if weights is None else weights[i], None if posts is None else posts[i])
        if group == "termination" and time_out_flags is not None and time_out_flags[i]:
            t.flags |= 1
        terms.append(t)
    extra = {}
    if group == "reward" and any(s.kind in ("is_alive", "is_terminated", "is_terminated_term") for s in specs):
        # rewards that read the termination manager's buffers: mirror its term list (never evaluated here)
        names = list(env.termination_manager.active_terms)
        extra["terminations"] = [FM.TermSpec("time_out") for _ in names]
        for s, t in zip(specs, terms):
            if s.kind == "is_terminated_term":
                wanted = names if s.term_names is None else s.term_names
                t.flags = sum(1 << names.index(n) for n in wanted)
    kw = {"termination": "terminations", "reward": "rewards", "observation": "observations"}[group]
    extra[kw] = terms
    max_len = int(getattr(env, "max_episode_length", 1000))
    return FM.FusedMdp(env.num_envs, env.device, step_dt=step_dt, max_episode_length=max_len,
                       command={"kind": "none"}, external_obs_dims=external_dims, seed=None, **sz, **extra)


_PHASE = {"termination": FM.PHASE_TERMINATION, "reward": FM.PHASE_REWARD, "observation": FM.PHASE_OBSERVATION}


def run_plan(mdp: FM.FusedMdp, env, group: str, specs, bindings=None):
    bind_env(mdp, env, *(bindings if bindings is not None else specs[0].bindings()))
    for s in specs:
        if s.kind == "external":
            mdp.bind_external(int(s.aux), _abi.f32c(s.external(env)).reshape(env.num_envs, -1))
    if group == "reward" and mdp.desc.num_terminations > 0:
        tm = env.termination_manager
        mdp.bind(terminated=tm.terminated, time_outs=tm.time_outs, term_dones=tm.term_dones_tensor())
    if group == "termination" and any(s.kind == "command_resample" for s in specs):
        term = env.command_manager.get_term(next(s.command for s in specs if s.kind == "command_resample"))
        mdp.bind(time_left=term.time_left, command_counter=term.command_counter)
    if group == "termination" and getattr(env, "episode_length_buf", None) is not None:
        # the kernel's termination phase is the env's "episode_length_buf += 1; compute()" (manager_based_rl_env.py:199);
        # a term function / manager is called after the env has counted the step, so hand the kernel the count before it
        env.episode_length_buf.sub_(1)
    mdp.step(_PHASE[group])


def evaluate(env, group: str, func, spec: Spec, key):
    """Single-term evaluation (a term function called directly)."""
    cache = env.__dict__.setdefault("_ilx_term_plans", {})
    k = (group, func.__name__, key)
    mdp = cache.get(k)
    if mdp is None:
        mdp = cache[k] = make_plan(env, group, [spec])
    run_plan(mdp, env, group, [spec])
    if group == "reward":
        return mdp.reward_buf.clone()
    if group == "termination":
        return mdp.term_dones[0].clone()
    return mdp.obs.clone()


def _term(group):
    def deco(builder):
        @functools.wraps(builder)
        def func(env, **params):
            spec = builder(**params)
            return evaluate(env, group, func, spec, _key(params))
        func._ilx_build = builder
        func._ilx_group = group
        return func
    return deco


reward_term, termination_term, observation_term = _term("reward"), _term("termination"), _term("observation")
