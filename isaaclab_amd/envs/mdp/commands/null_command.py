"""``NullCommand`` (interface of ``isaaclab/envs/mdp/commands/null_command.py``): a command term that does nothing."""

from __future__ import annotations

from collections.abc import Sequence


class NullCommand:
    """null_command.py:19-69: no buffers, no launches; reading ``command`` is an error."""

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self.num_envs, self.device = env.num_envs, env.device
        self.metrics: dict = {}

    def __str__(self) -> str:
        return f"NullCommand:\n\tCommand dimension: N/A\n\tResampling time range: {self.cfg.resampling_time_range}"

    @property
    def command(self):
        raise RuntimeError("NullCommandTerm does not generate any commands.")

    def reset(self, env_ids: Sequence[int] | None = None) -> dict[str, float]:
        return {}

    def compute(self, dt: float):
        pass
