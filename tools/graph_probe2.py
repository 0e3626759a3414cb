"""GPU probe (diagnostics, breaks the step's semantics on purpose): how the graph executor schedules three independent
branches -- actuators | scene updates | scanner refresh (forced due) -- before the managers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC
from isaaclab_amd.tasks import locomotion_velocity as LV

dev = torch.device("cuda:0")
cfg = BC.build("anymal", dev, 16384, 3000)
env = cfg.env
raw = cfg.state["raw_action"].to(dev)
env.count_steps_on_device()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def scenes():
    d = env.data
    for k in range(LV.DECIMATION):
        d.joint_pos, d.joint_vel = env.physics["joint_pos"][k], env.physics["joint_vel"][k]
        env.contact.set_net_contact_forces(env.physics["net_forces"][k])
        env.scene.update(LV.PHYSICS_DT)

def rays():
    d = env.data
    d.set_sim_state(root_state_w=env.physics["root_state_w"])
    env.scanner.set_pose(d.root_state_w[:, :3], d.root_state_w[:, 3:7])
    env.scanner._is_outdated.fill_(True)
    env.scanner._update_outdated_buffers(height_out=env.height_scan, height_offset=env.height_scan_offset)

def step(order):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    parts = {"A": (main, lambda: env.step_actuators(raw)), "S": (s1, scenes), "R": (s2, rays)}
    for key in order:
        st, fn = parts[key]
        with torch.cuda.stream(st):
            fn()
    main.wait_stream(s1); main.wait_stream(s2)
    env.step_managers()

for order in ("ASR", "SRA", "RSA", "RAS", "ARS", "SAR"):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(order)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step(order)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    print(f"issue order {order}: {1e6 * (time.perf_counter() - t0) / K:.1f} us per step")
