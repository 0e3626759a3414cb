"""CPU: the modifier oracle against the fixture recorded from the reference's functions and classes."""

from conftest import assert_close, assert_exact, load_golden
from oracle import modifiers as O


def test_function_modifiers():
    g = load_golden("modifiers")
    x = g["x"]
    assert_exact(O.scale(x, 1.7), g["scale"], "scale")
    assert_exact(O.bias(x, -0.35), g["bias"], "bias")
    assert_exact(O.clip(x, (-1.5, 2.0)), g["clip_both"], "clip")
    assert_exact(O.clip(x, (-0.5, None)), g["clip_lo"], "clip, lower bound only")
    assert_exact(O.clip(x, (None, 0.25)), g["clip_hi"], "clip, upper bound only")


def test_digital_filter_and_integrator():
    g = load_golden("modifiers")
    steps, shape = g["filter_in"], tuple(g["x"].shape)
    f = O.DigitalFilter(g["filter_A"].tolist(), g["filter_B"].tolist(), shape)
    for k in range(steps.shape[0]):
        if k == 5:
            f.reset(g["filter_reset_ids"])
        if k == 7:
            f.reset()
        assert_close(f(steps[k]), g["filter_out"][k], rtol=1e-6, atol=1e-6, what=f"filter step {k}")
    assert_close(f.x_n, g["filter_x_n"], atol=1e-6, what="input history")
    assert_close(f.y_n, g["filter_y_n"], atol=1e-6, what="output history")
    for tag, a, b in (("delay", [0.0], [0.0, 1.0]), ("average", [0.0], [0.25] * 4), ("lowpass", [-0.8], [0.2])):
        f = O.DigitalFilter(a, b, shape)
        for k in range(6):
            assert_close(f(steps[k]), g[f"{tag}_out"][k], rtol=1e-6, atol=1e-6, what=f"{tag} step {k}")
    it = O.Integrator(0.02, shape)
    for k in range(steps.shape[0]):
        if k == 4:
            it.reset(g["filter_reset_ids"])
        assert_exact(it(steps[k]), g["integrator_out"][k], f"integrator step {k}")
    assert_exact(it.y_prev, g["integrator_y_prev"], "integrator previous sample")
