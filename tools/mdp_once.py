"""GPU probe: a few launches of the fused MDP kernel and nothing else of size (for counter passes).
    python tools/mdp_once.py [config] [envs] [phases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC
from isaaclab_amd.envs import fused_mdp as FM

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else BC.ENVS[name]
cfg = BC.build(name, torch.device("cuda:0"), envs, 3000)
for _ in range(3):
    cfg.step()
phases = int(sys.argv[3]) if len(sys.argv) > 3 else FM.PHASE_ALL
for _ in range(5):
    cfg.env.mdp.step(phases)
torch.cuda.synchronize()
