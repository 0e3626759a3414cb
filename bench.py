#!/usr/bin/env python3
"""Benchmark of the non-physics env step (BASELINE.json metric): Isaac-Velocity-Rough-Anymal-C-v0,
16384 envs per GPU, ActuatorNetMLP + contact bookkeeping + 187-ray height scan + fused managers.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full env step of the hot path over one batch of synthetic physics state that is already resident
in HBM: process_action, 4 x (actuator net + scene update: joint_acc, contact sensor, clocks), height-scan ray cast, and the fused
termination / reward / reset / command / observation launch.  Prints ONE JSON line (rank 0).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_ENVS = 16384
TERM_EVALS_PER_ENV_STEP = 29  # 8 obs + 9 active rewards + 2 terminations + 1 command + 4 x (actuator + contact) + 1 ray cast
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E peak (spec); measured float4-copy ceiling is ~6300 GB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (256 CUs x 256 FLOP/clk/CU x 2.4 GHz)
# 6-32-32-32-1 MLP: multiply-adds per row x 2, rows = envs x 12 joints
MLP_FLOP_PER_ENV = 12 * 2 * (6 * 32 + 32 * 32 + 32 * 32 + 32 * 1)

# Algorithmic bytes per env per launch (each distinct input read once, each output written once; fp32 = 4 B,
# bool = 1 B, int64 = 8 B; per-joint constants, network weights and the terrain mesh excluded).  J=12, B=17, R=187.
BYTES = {
    # read q_des, q, qd (3*48); read+write pos-error / velocity histories (2 * 2 * 144); write computed, applied (2*48)
    "actuator_net_mlp": 3 * 48 + 4 * 144 + 2 * 48,
    # InteractiveScene.update in one launch -- contact: read net forces 204, write net_forces_w 204, write one history
    # row 204 (ring) + head 8, read+write 4 timers 8*68, clock 18; joint_acc: read vel 48, read+write previous vel 96,
    # write acc 48; scanner clock 18
    "scene_update": 204 + 204 + 204 + 8 + 8 * 68 + 18 + 48 + 96 + 48 + 18,
    # read pose 28 + clock 9, write pose 28, ray hits 187*12, clock 5
    "raycaster_refresh": 28 + 9 + 28 + 187 * 12 + 5,
    # read root 52, joints (pos, vel, acc, torque, default) 5*48, actions 2*48, contact history 612, timers 2*68,
    # ray hit z 748, sensor pos 12, command state r+w ~2*41, ep_len r+w 16, noise 880 (parity mode only: not counted);
    # write obs 940, reward 4, step_reward 44, episode sums r+w 72, flags 5
    "mdp_step": 52 + 240 + 96 + 612 + 136 + 748 + 12 + 82 + 16 + 940 + 4 + 44 + 72 + 5,
    "joint_action_process": 48 + 2 * 48 + 48 + 48,
    # --actuator lstm: read q_des, q, qd (3*48); read+write hidden and cell state (2 * 2 * 2 layers * 8 * 4 B * 12 joints);
    # write computed, applied (2*48)
    "actuator_net_lstm": 3 * 48 + 2 * 2 * 768 + 2 * 48,
}


def bytes_per_env_step(actuator_kernel):
    return (4 * (BYTES[actuator_kernel] + BYTES["scene_update"]) + BYTES["raycaster_refresh"] + BYTES["mdp_step"]
            + BYTES["joint_action_process"])


def build_env(device, num_envs, seed, actuator="mlp"):
    from isaaclab_amd import synth
    from isaaclab_amd.envs.locomotion_step import LocomotionVelocityStep
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.mesh import convert_to_mesh
    from isaaclab_amd.utils.string import resolve_matching_names_values

    verts, tris, height_at = synth.rough_terrain_mesh(seed=3)
    mesh = convert_to_mesh(verts, tris, str(device))
    ws, bs = synth.actuator_mlp_weights()
    idx, _, vals = resolve_matching_names_values(LV.ANYMAL_DEFAULT_JOINT_POS, LV.ANYMAL_JOINTS)
    default = torch.zeros(12)
    default[idx] = torch.tensor(vals)
    state = synth.locomotion_state(num_envs, seed, num_joints=12, num_bodies=17, default_joint_pos=default,
                                   feet_ids=[13, 14, 15, 16], height_at=height_at)
    net = synth.actuator_lstm_weights() if actuator == "lstm" else dict(weights=ws, biases=bs, activation="softsign")
    env = LocomotionVelocityStep(num_envs, device, mesh, "anymal_c", actuator_net=net, seed=seed)
    env.load_state(state)
    return env, state, (verts, tris, ws, bs, default)


def time_kernels(env, state, device, actuator_kernel, reps=20):
    """Average duration of each kernel of the step, measured with events on the stream the kernels run on."""
    from isaaclab_amd import _abi
    from isaaclab_amd.envs import fused_mdp as FM
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.types import ArticulationActions

    d = env.data
    raw = state["raw_action"].to(device)

    def actuator():
        ca = ArticulationActions(joint_positions=d.joint_pos_target, joint_velocities=d.joint_vel_target,
                                 joint_efforts=d.joint_effort_target)
        env.actuator.compute(ca, d.joint_pos, d.joint_vel)

    def scene():
        env.scene.update(LV.PHYSICS_DT)

    def rays():
        env.scanner._update_outdated_buffers()

    def mdp():
        env.mdp.step(FM.PHASE_ALL)

    def action():
        _abi.check(_abi.joint_action_process(_abi.ptr(raw), _abi.ptr(env.action), _abi.ptr(env.prev_action), None, 0.5,
                                             _abi.ptr(d.default_joint_pos), 0.0, None, _abi.ptr(d.joint_pos_target), env.n, 12,
                                             _abi.stream()))

    # The launches of one measurement are queued behind a long-running blocker, so that they run back to back on the
    # device: the host (Python + ctypes, ~10 us per call) would otherwise pace the short kernels and the events would
    # time its gaps.  Measured per launch = (event span over `reps` queued launches) / reps.
    blocker_a = torch.randn(4096, 4096, device=device)
    blocker_out = torch.empty_like(blocker_a)

    def span(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            torch.mm(blocker_a, blocker_a, out=blocker_out)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps  # microseconds per call

    def due():
        env.scanner._is_outdated.fill_(True)  # every env is due, as once per env step

    def rays_due():
        due()
        rays()

    out = {}
    for name, fn in ((actuator_kernel, actuator), ("scene_update", scene), ("mdp_step", mdp), ("joint_action_process", action)):
        out[name] = span(fn)
    out["raycaster_refresh"] = span(rays_due) - span(due)
    return out


def cpu_baseline(seed, envs=NUM_ENVS, warmup=2, steps=10):
    """The oracle step (unfused PyTorch CPU ops + C ray cast) at the workload's full env count on the host's cores:
    `warmup` untimed steps, then `steps` timed ones between perf_counter reads -- the step-loop timing of
    scripts/benchmarks/benchmark_non_rl.py:154-179 without the simulator inside the step."""
    from isaaclab_amd import synth
    from isaaclab_amd.tasks import locomotion_velocity as LV
    from isaaclab_amd.utils.string import resolve_matching_names_values
    from oracle import raycast as RC
    from oracle.step_locomotion import AnymalRoughOracle

    cores = os.cpu_count() or 1
    # every core this process may run on (a GPU box gives one GPU's share of the machine, not all of os.cpu_count())
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    threads = max(1, min(threads, int(os.environ.get("OMP_NUM_THREADS", threads))))
    torch.set_num_threads(threads)
    sample_envs = envs
    verts, tris, height_at = synth.rough_terrain_mesh(seed=3)
    ws, bs = synth.actuator_mlp_weights()
    idx, _, vals = resolve_matching_names_values(LV.ANYMAL_DEFAULT_JOINT_POS, LV.ANYMAL_JOINTS)
    default = torch.zeros(12)
    default[idx] = torch.tensor(vals)
    state = synth.locomotion_state(sample_envs, seed, num_joints=12, num_bodies=17, default_joint_pos=default,
                                   feet_ids=[13, 14, 15, 16], height_at=height_at)
    cm = RC.CMesh(verts, tris)
    cm.bvh()
    o = AnymalRoughOracle(sample_envs, state, ws, bs, default, cm)
    g = torch.Generator().manual_seed(1)
    noise = torch.rand(sample_envs, 235, generator=g)
    samples = torch.rand(sample_envs, 7, generator=g)
    for _ in range(warmup):
        o.step(state["raw_action"], noise, samples, samples)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step(state["raw_action"], noise, samples, samples)
    dt = (time.perf_counter() - t0) / steps
    return {"value": sample_envs / dt, "unit": "env-steps/s", "cores": cores, "threads": threads, "kind": "port",
            "sample": f"{warmup} warm-up + {steps} timed steps of all {sample_envs} envs (same step, same terrain mesh of "
                      f"{tris.shape[0]} triangles): {dt * 1e3:.1f} ms/step; torch {torch.__version__} CPU ops on {threads} "
                      f"threads + C BVH ray cast"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=NUM_ENVS, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--actuator", choices=("mlp", "lstm"), default="mlp",
                    help="actuator network: the headline 6-32-32-32-1 MLP, or the ANYdrive-shaped LSTM(2, 8, 2) + head "
                         "(no CPU baseline, reported as a side measurement)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)

    from isaaclab_amd.distributed import OverlappedLogReducer

    env, state, _ = build_env(device, args.envs, seed=3000 + rank, actuator=args.actuator)
    raw = state["raw_action"].to(device)
    n_log = env.mdp.log_sums.numel()

    # episode statistics of the done envs: the only cross-rank exchange of the path, one all-reduce per step, started
    # after the step and completed beside the next two (nothing in the step consumes it)
    reducer = OverlappedLogReducer(n_log, device)

    def one_step():
        buf, _ = reducer.begin_step()
        env.mdp.bind(log_sums=buf)
        env.step(raw)
        reducer.end_step()

    for _ in range(args.warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    reducer.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # what the collective group saw: the number of ranks that took part (a sum of ones over RCCL) and the spread of the
    # per-rank step time; `value` is priced on the slowest rank
    rccl_ranks, rank_ms = None, [elapsed / args.steps * 1e3] * 2
    if dist is not None:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        t_min = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN)
        elapsed = float(t.item())
        rank_ms = [float(t_min.item()) / args.steps * 1e3, elapsed / args.steps * 1e3]

    act_kernel = "actuator_net_" + args.actuator
    kern = time_kernels(env, state, device, act_kernel)
    BYTES_PER_ENV_STEP = bytes_per_env_step(act_kernel)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.envs * world * args.steps / elapsed
        launches = {act_kernel: 4, "scene_update": 4, "raycaster_refresh": 1, "mdp_step": 1, "joint_action_process": 1}
        dominant = max(kern, key=lambda k: kern[k] * launches[k])
        # HBM-side bytes per launch of that kernel from the committed counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # of this same command, corrected as stated in the file); only meaningful for the env count it was taken at
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")) as f:
                prof = json.load(f)
            if prof.get("envs") == args.envs:
                traffic = prof["kernels"][dominant]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        if dominant == "actuator_net_mlp":  # the one kernel of the step that is matrix work
            achieved = MLP_FLOP_PER_ENV * args.envs / (kern[dominant] * 1e-6) / 1e12
            roofline = {"bound": "mfma", "kernel": dominant, "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "launch_us": kern[dominant],
                        "algorithmic_flop_per_launch": MLP_FLOP_PER_ENV * args.envs}
        else:
            achieved = BYTES[dominant] * args.envs / (kern[dominant] * 1e-6) / 1e9
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launch_us": kern[dominant],
                        "algorithmic_bytes_per_launch": BYTES[dominant] * args.envs}
        line = {
            "metric": "env-steps/sec (non-physics step) ANYmal-C rough", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Isaac-Velocity-Rough-Anymal-C-v0 non-physics step", "envs_per_gpu": args.envs,
                       "decimation": 4, "joints": 12, "bodies": 17, "rays": 187, "obs_dim": 235,
                       "terrain_triangles": int(env.scanner.meshes["/World/ground"].num_triangles),
                       "actuator": ("ActuatorNetMLP 6-32-32-32-1 softsign (fp32 MFMA)" if args.actuator == "mlp" else
                                    "ActuatorNetLSTM LSTM(2, 8, 2 layers) + 8-32-1 softsign head"),
                       "parallelism": f"env-shard x{world}"},
            "env_term_steps_per_s": value * TERM_EVALS_PER_ENV_STEP,
            "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
            "step_hbm_frac": BYTES_PER_ENV_STEP * args.envs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline": roofline,
            "kernels_us": {k: round(v, 2) for k, v in kern.items()},
            "mfma_frac": (MLP_FLOP_PER_ENV * args.envs / (kern[act_kernel] * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS
                          if args.actuator == "mlp" else None),
            "rccl_ranks": rccl_ranks, "ms_per_step_min_rank": rank_ms[0], "ms_per_step_max_rank": rank_ms[1],
        }
        if not args.no_cpu_baseline and world == 1 and args.actuator == "mlp":
            line["cpu_baseline"] = cpu_baseline(3000)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
