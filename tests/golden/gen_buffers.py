"""Golden vectors for CircularBuffer / DelayBuffer, produced by the REFERENCE classes
(``isaaclab/utils/buffers/circular_buffer.py``, ``delay_buffer.py``).  Writes tests/golden/buffers.npz."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

m = R.bootstrap()
CB = m["circular_buffer"].CircularBuffer
db = R.load("isaaclab.utils.buffers.delay_buffer", f"{R.L}/utils/buffers/delay_buffer.py")

out = {}
g = torch.Generator().manual_seed(1234)
n, d, L, steps = 40, 5, 4, 9
data = torch.randn(steps, n, d, generator=g)
keys = torch.randint(0, L + 2, (steps, n), generator=g)
out["data"], out["keys"] = data, keys
reset_at = {3: torch.tensor([0, 7, 39]), 6: torch.tensor([7, 8])}
cb = CB(L, n, "cpu")
hist, got, lens = [], [], []
for k in range(steps):
    if k in reset_at:
        cb.reset(reset_at[k])
        out[f"reset_ids_{k}"] = reset_at[k]
        out[f"after_reset_{k}"] = cb.buffer.clone()
    cb.append(data[k])
    hist.append(cb.buffer.clone()), got.append(cb[keys[k]].clone()), lens.append(cb.current_length.clone())
out["history"], out["getitem"], out["current_length"] = torch.stack(hist), torch.stack(got), torch.stack(lens)

delay = db.DelayBuffer(3, n, "cpu")
lags = torch.randint(0, 4, (n,), generator=g, dtype=torch.int)
delay.set_time_lag(lags)
out["time_lags"] = lags
delayed = []
for k in range(steps):
    if k == 4:
        delay.reset(torch.tensor([1, 2, 30]))
        delay.set_time_lag(2, torch.tensor([1, 2, 30]))
    delayed.append(delay.compute(data[k]).clone())
out["delayed"] = torch.stack(delayed)
out["time_lags_after"] = delay.time_lags.clone()
np.savez_compressed(os.path.join(HERE, "buffers.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote buffers.npz with", len(out), "arrays")
