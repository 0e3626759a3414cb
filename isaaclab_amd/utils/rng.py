"""The state every device-side generator shares: one seed and one draw counter.

A kernel's random numbers are ``philox(seed, draw, row, column)``.  Every launch that draws takes a fresh ``draw`` from
the single counter kept here, tagged with the drawing module's stream id in its top 16 bits, so no two launches -- of
one module or of two -- ever share a key.  :func:`seed_all` is what the environments call where the reference calls
``torch.manual_seed`` (``ManagerBasedEnv.seed``, ``envs/manager_based_env.py``; ``DirectRLEnv.seed``,
``envs/direct_rl_env.py``): it sets the seed and restarts the counter, so a run re-seeded with the same value replays
its draws, and two seeds give different ones.  Ranks of a multi-GPU run get different draws by seeding each rank
differently, as the reference's training scripts do (``seed + local_rank``).
"""

from __future__ import annotations

STREAM = {"sampling": 1, "noise": 2, "events": 3, "velocity_command": 4, "pose_2d_command": 5, "terrain": 6, "fused_mdp": 7}
_state = {"seed": 0, "draw": 0}


def seed_all(seed: int) -> int:
    """Seed of every device generator; the draw counter restarts."""
    _state["seed"], _state["draw"] = int(seed), 0
    return _state["seed"]


def seed() -> int:
    return _state["seed"]


def next_draw(stream: str) -> int:
    """The ``draw`` word of the next launch of module ``stream``: unique over all modules until the next seed_all."""
    _state["draw"] += 1
    return (STREAM[stream] << 48) | _state["draw"]
