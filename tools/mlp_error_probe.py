"""GPU probe: error distribution of the actuator network's three evaluations (default split-f16 hidden layers,
ILX_MLP_EXACT_F32, generic fmaf chains) against float64 on the headline network shape.
    python tools/mlp_error_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_components_gpu as T
from isaaclab_amd import synth

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(41)
cases = [("bench net (synth.actuator_mlp_weights)", *synth.actuator_mlp_weights())]
for gain in (1.0, 6.0):
    cases.append((f"random net, gain {gain}", *T._random_net(gen, gain)))
for name, ws, bs in cases:
    ws, bs = [torch.as_tensor(w) for w in ws], [torch.as_tensor(b) for b in bs]
    for flags, what in ((0, "default (split f16)"), (2, "ILX_MLP_EXACT_F32"), (1, "generic fmaf")):
        act = T._run_mlp(dev, ws, bs, "softsign", flags, n=16384)
        ref = T._mlp_float64(act, ws, bs, "softsign")
        err = (act.computed_effort.double().cpu() - ref).abs().flatten()
        rel = err / ref.abs().flatten().clamp_min(1e-30)
        q = lambda t, p: float(torch.quantile(t[:: max(1, t.numel() // 1000000)], p))
        print(f"{name:42s} {what:20s} abs err median {q(err, 0.5):.2e} 99% {q(err, 0.99):.2e} max {float(err.max()):.2e} | "
              f"rel err median {q(rel, 0.5):.2e} 99% {q(rel, 0.99):.2e} max {float(rel.max()):.2e} | |y| up to {float(ref.abs().max()):.3g}")
