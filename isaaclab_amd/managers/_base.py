"""Shared pieces of the managers: term discovery and parameter resolution (``manager_base.py:243-298``)."""

from __future__ import annotations

import inspect

from ..envs.mdp import _fused as F
from .manager_term_cfg import ManagerTermBaseCfg
from .scene_entity_cfg import SceneEntityCfg


def iter_terms(cfg, term_type):
    """(name, term cfg) pairs of a cfg object or dict, in declaration order; ``None`` entries are skipped
    (how the task cfgs of the reference switch terms off)."""
    items = cfg.items() if isinstance(cfg, dict) else vars(cfg).items()
    for name, term in items:
        if term is None or name.startswith("_") or not isinstance(term, ManagerTermBaseCfg):
            continue
        if not isinstance(term, term_type):
            raise TypeError(f"Configuration for the term '{name}' is not of type {term_type.__name__}. Received: '{type(term)}'.")
        yield name, term


def resolve_term(env, name: str, term: ManagerTermBaseCfg) -> F.Spec:
    """Resolve SceneEntityCfg parameters against the scene and build the kernel term (manager_base.py:250-298)."""
    if not callable(term.func):
        raise AttributeError(f"The term '{name}' is not callable. Received: {term.func}")
    build = getattr(term.func, "_ilx_build", None)
    if build is None:
        raise TypeError(f"The term '{name}' uses '{getattr(term.func, '__name__', term.func)}', which is not a term of "
                        "isaaclab_amd.envs.mdp: only library terms can be compiled into the fused kernel")
    for value in term.params.values():
        if isinstance(value, SceneEntityCfg):
            value.resolve(env.scene)
    params = dict(term.params)
    if getattr(term.func, "_ilx_needs_step_dt", False):
        params["step_dt"] = env.step_dt
    try:
        return build(**params)
    except TypeError as e:
        sig = inspect.signature(build)
        raise ValueError(f"The term '{name}' expects parameters {list(sig.parameters)} but received {list(term.params)}.") from e
