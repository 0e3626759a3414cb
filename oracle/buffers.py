"""CPU restatement of the batched ring buffers (test infrastructure only; never imported by the product).

Follows ``isaaclab/utils/buffers/circular_buffer.py`` (append :107-135, buffer :78-86, __getitem__ :137-166, reset :92-105)
and ``delay_buffer.py:158-177`` with an explicit per-env list of entries instead of a shared ring pointer.  Pinned by
tests/golden/buffers.npz, recorded from the reference classes.
"""

from __future__ import annotations

import torch


class Ring:
    def __init__(self, max_len: int, n: int, d: int):
        self.max_len, self.hist = max_len, torch.zeros(n, max_len, d)   # hist[:, -1] is the newest entry
        self.pushes = torch.zeros(n, dtype=torch.long)

    def reset(self, ids=None):
        ids = slice(None) if ids is None else ids
        self.hist[ids] = 0.0
        self.pushes[ids] = 0

    def append(self, x):
        fresh = self.pushes == 0
        self.hist = torch.cat([self.hist[:, 1:], x.unsqueeze(1)], dim=1)
        self.hist[fresh] = x[fresh].unsqueeze(1).expand(-1, self.max_len, -1)
        self.pushes += 1

    def get(self, key):
        lag = torch.minimum(key.long(), self.pushes - 1)
        return self.hist[torch.arange(len(key)), self.max_len - 1 - lag]
