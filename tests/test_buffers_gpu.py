"""GPU: CircularBuffer / DelayBuffer (HIP ``ilx_circular_buffer_append``) against the fixture recorded from the reference."""

import pytest
import torch

from conftest import assert_exact, load_golden

pytestmark = pytest.mark.gpu


def test_circular_buffer_matches_reference():
    from isaaclab_amd.utils.buffers import CircularBuffer

    dev = torch.device("cuda:0")
    g = load_golden("buffers")
    data, keys = g["data"].to(dev), g["keys"].to(dev)
    steps, n, d = data.shape
    L = g["history"].shape[2]
    cb = CircularBuffer(L, n, dev)
    with pytest.raises(RuntimeError, match="empty circular buffer"):
        cb[keys[0]]
    wide = torch.zeros(n, L * d + 6, device=dev)
    for k in range(steps):
        if f"reset_ids_{k}" in g:
            cb.reset(g[f"reset_ids_{k}"].tolist())
            assert_exact(cb.buffer, g[f"after_reset_{k}"], f"history after reset before push {k}")
        # the flattened history lands in a column range of a wider row from the same launch
        cb.append(data[k], ordered_out=wide[:, 3:3 + L * d])
        assert_exact(wide[:, 3:3 + L * d].reshape(n, L, d), g["history"][k], f"ordered history from the append {k}")
        assert_exact(cb.buffer, g["history"][k], f"ordered history after push {k}")
        assert_exact(cb[keys[k]], g["getitem"][k], f"lagged read after push {k}")
        assert_exact(cb.current_length, g["current_length"][k], f"current length after push {k}")
    assert float(wide[:, :3].abs().max()) == 0.0 and float(wide[:, 3 + L * d:].abs().max()) == 0.0
    with pytest.raises(ValueError, match="greater than zero"):
        CircularBuffer(0, n, dev)
    with pytest.raises(ValueError, match="environments while expecting"):
        cb.append(data[0][:5])
    fresh = CircularBuffer(3, n, dev)
    fresh.reset()          # nothing appended yet: only the counts
    assert int(fresh.current_length.sum()) == 0


def test_delay_buffer_matches_reference():
    from isaaclab_amd.utils.buffers import DelayBuffer

    dev = torch.device("cuda:0")
    g = load_golden("buffers")
    data = g["data"].to(dev)
    delay = DelayBuffer(3, data.shape[1], dev)
    delay.set_time_lag(g["time_lags"].to(dev))
    for k in range(data.shape[0]):
        if k == 4:
            delay.reset([1, 2, 30])
            delay.set_time_lag(2, [1, 2, 30])
        assert_exact(delay.compute(data[k]), g["delayed"][k], f"delayed output {k}")
    assert_exact(delay.time_lags, g["time_lags_after"], "time lags")
    with pytest.raises(ValueError, match="cannot be larger than the history length"):
        delay.set_time_lag(4)
    with pytest.raises(TypeError, match="Invalid dtype"):
        delay.set_time_lag(torch.zeros(data.shape[1]))
