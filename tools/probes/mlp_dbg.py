import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import load_golden
from isaaclab_amd.actuators import ActuatorNetMLP, ActuatorNetMLPCfg
from isaaclab_amd.utils.types import ArticulationActions
dev = torch.device("cuda:0")
g = load_golden("actuators_mlp")
tag = "go1"
nl = int(g[f"{tag}_nlayers"])
net = dict(weights=[g[f"{tag}_w{i}"] for i in range(nl)], biases=[g[f"{tag}_b{i}"] for i in range(nl)], activation="softsign")
print([tuple(w.shape) for w in net["weights"]])
n, j = g[f"{tag}_q"].shape[1:]
cfg = ActuatorNetMLPCfg(network=net, pos_scale=-1.0, vel_scale=1.0, torque_scale=1.0, input_order="pos_vel", input_idx=[0, 1, 2], effort_limit=23.7, velocity_limit=30.0, saturation_effort=23.7)
for flags in (0, 2, 1):
    act = ActuatorNetMLP(cfg, joint_names=[f"j{i}" for i in range(j)], joint_ids=slice(None), num_envs=n, device=dev)
    act._flags = flags
    for s in range(2):
        ca = ArticulationActions(joint_positions=g[f"{tag}_q_des"][s].to(dev))
        act.compute(ca, g[f"{tag}_q"][s].to(dev), g[f"{tag}_qd"][s].to(dev))
        ref = g[f"{tag}_computed"][s]
        got = act.computed_effort.cpu()
        err = (got - ref).abs()
        print(flags, s, "max err", float(err.max()), "ref max", float(ref.abs().max()), "nan", int(torch.isnan(got).sum()), got.flatten()[:4].tolist(), ref.flatten()[:4].tolist())
