"""GPU probe: only the actuator-net launch of the bench workload (diagnostics for counter passes / timing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd import _abi
import bench

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
# a small flat mesh instead of the terrain: the actuator does not read it
from isaaclab_amd import synth
real = synth.rough_terrain_mesh
synth.rough_terrain_mesh = lambda seed=0: real(seed=seed, rows=1, cols=1)
env, state, _ = bench.build_env(dev, envs, 3000)
from isaaclab_amd.utils.types import ArticulationActions
d = env.data
def actuator():
    ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target, joint_efforts=d.joint_effort_target)
    env.actuator.compute(ca, d.joint_pos, d.joint_vel)
for _ in range(3): actuator()
torch.cuda.synchronize()
blk = torch.randn(4096, 4096, device=dev); out = torch.empty_like(blk)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.mm(blk, blk, out=out); torch.mm(blk, blk, out=out)
a.record()
for _ in range(reps): actuator()
b.record(); torch.cuda.synchronize()
print("actuator_net_mlp us per launch:", a.elapsed_time(b) * 1e3 / reps)
