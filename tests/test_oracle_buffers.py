"""CPU: the ring buffer oracle against the fixture recorded from the reference classes."""

from conftest import assert_exact, load_golden
from oracle import buffers as O


def test_circular_and_delay_buffer():
    g = load_golden("buffers")
    data, keys = g["data"], g["keys"]
    steps, n, d = data.shape
    ring = O.Ring(g["history"].shape[2], n, d)
    for k in range(steps):
        if f"reset_ids_{k}" in g:
            ring.reset(g[f"reset_ids_{k}"])
        ring.append(data[k])
        assert_exact(ring.hist, g["history"][k], f"ordered history after push {k}")
        assert_exact(ring.get(keys[k]), g["getitem"][k], f"lagged read after push {k}")
        assert_exact(ring.pushes.clamp(max=ring.max_len), g["current_length"][k], f"current length after push {k}")
    delay, lags = O.Ring(4, n, d), g["time_lags"].clone()
    for k in range(steps):
        if k == 4:
            delay.reset([1, 2, 30])
            lags[[1, 2, 30]] = 2
        delay.append(data[k])
        assert_exact(delay.get(lags), g["delayed"][k], f"delayed output {k}")
