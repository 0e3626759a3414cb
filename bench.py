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
MFMA_F16_PEAK_TFLOPS = 2500.0 # dense f16 MFMA peak
# 6-32-32-32-1 MLP: multiply-adds per row x 2, rows = envs x 12 joints
MLP_FLOP_PER_ENV = 12 * 2 * (6 * 32 + 32 * 32 + 32 * 32 + 32 * 1)
# ... and as the default kernel executes it: the first layer on the f32 MFMA, the two hidden 32 x 32 layers on the f16 matrix
# cores as three products each (head x head, head x remainder, remainder x head)
MLP_F32_MFMA_FLOP_PER_ENV = 12 * 2 * (6 * 32)
MLP_F16_MFMA_FLOP_PER_ENV = 12 * 3 * 2 * (32 * 32 + 32 * 32)
# the byte table of the workload's measurement recipe for the headline configuration (4 substeps x 1 960 B + 3 645 B per
# env step: without the second actuator history and the dense height scan that this build's launches also move)
RECIPE_BYTES_PER_ENV_STEP = 4 * 1960 + 3645
GRAPH_WARMUP_REPLAYS = 20      # per recorded graph, whatever --warmup says
CUS = 256                      # compute units of an MI355X (4 SIMDs each)

def build_env(device, num_envs, seed, actuator="mlp"):
    """The headline configuration's step object and its synthetic state (what the probe scripts under tools/ use)."""
    import bench_configs as BC

    cfg = BC.build("anymal", device, num_envs, seed, actuator)
    return cfg.env, cfg.state, None


def time_kernels(cfg, device, reps=20, only=None):
    """Average duration of each launch of the step, measured with events on the stream the kernels run on."""
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

    out = {}
    for name, (fn, _) in cfg.launches.items():
        if only is not None and name not in only:
            continue
        # (fn, setup): the launch needs a setup launch before each call, whose own span is taken off
        out[name] = span(fn[0]) - span(fn[1]) if isinstance(fn, tuple) else span(fn)
    return out


def capture_step(cfg, reducer):
    """The env step recorded once into a HIP graph per statistics vector (the only argument that alternates between
    steps); a replay issues the same launches with the same arguments without the host's launch calls.  The fused MDP
    launch keeps its generator's step count in device memory, advanced inside the graph, so replays draw fresh numbers;
    sensor clocks and histories live in device memory anyway."""
    mdp = getattr(cfg.env, "mdp", None)
    if hasattr(cfg.env, "count_steps_on_device"):
        cfg.env.count_steps_on_device()
    elif mdp is not None:
        mdp.count_steps_on_device()
    overlap = getattr(cfg, "overlap_in_graph", False)
    if overlap:
        cfg.env.concurrent_chains = True   # the sensor chain as a branch beside the actuator chain
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):       # per-stream workspaces are created here, outside the capture
            cfg.step()
    side.synchronize()
    graphs = []
    # the statistics vector starts every replay at zero: the step clears it itself where it can (behind its fork, so that
    # neither chain waits for the clearing launch), else the clearing is recorded ahead of the step
    own_clear = reducer is not None and hasattr(cfg.env, "clear_log_sums")
    if own_clear:
        cfg.env.clear_log_sums = True
    for i in range(2 if reducer is not None else 1):
        if reducer is not None:
            cfg.bind_log_sums(reducer.buffer(i))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            if reducer is not None and not own_clear:
                reducer.buffer(i).zero_()
            cfg.step()
        graphs.append(graph)
    torch.cuda.synchronize()
    if own_clear:
        cfg.env.clear_log_sums = False
    if overlap:
        cfg.env.concurrent_chains = False
    return graphs


def capture_commands(cfg, reducer):
    """The env step recorded once into a command list per statistics vector (isaaclab_amd.utils.command_list): a replay
    issues the same launches with the same arguments from a C loop, one hipLaunchKernel each -- 2.9 us of host time per
    launch where a graph replay costs 14 us however few nodes it has.  Call after capture_step (the generator's step
    count is in device memory by then).  None: the step has work that a list does not hold (a statistics vector that
    only a torch launch could clear)."""
    from isaaclab_amd.utils.command_list import CommandList

    own_clear = reducer is not None and hasattr(cfg.env, "clear_log_sums")
    if reducer is not None and not own_clear:
        return None
    if own_clear:
        cfg.env.clear_log_sums = True
    lists = []
    for i in range(2 if reducer is not None else 1):
        if reducer is not None:
            cfg.bind_log_sums(reducer.buffer(i))
        commands = CommandList()
        with commands.record():
            cfg.step()
        lists.append(commands)
    if own_clear:
        cfg.env.clear_log_sums = False
    return lists


def cpu_baseline(seed, envs=NUM_ENVS, warmup=2, steps=10, root_scatter="origins"):
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
    origins = synth.curriculum_env_origins(sample_envs, seed) if root_scatter == "origins" else None   # as the GPU leg
    state = synth.locomotion_state(sample_envs, seed, num_joints=12, num_bodies=17, default_joint_pos=default,
                                   feet_ids=[13, 14, 15, 16], height_at=height_at, env_origins=origins)
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
    import bench_configs as BC

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=tuple(BC.ENVS), default="anymal",
                    help="which configuration's step to run; the default is the one the headline metric is quoted on")
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: the configuration's own count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="measure the step issued launch by launch from the host (default: replayed from a HIP graph)")
    ap.add_argument("--actuator", choices=("mlp", "lstm"), default="mlp",
                    help="anymal only -- actuator network: the headline 6-32-32-32-1 MLP, or the ANYdrive-shaped LSTM(2, 8, 2) + "
                         "head (no CPU baseline, reported as a side measurement)")
    ap.add_argument("--robot", choices=("franka", "xarm6"), default="franka",
                    help="reach only -- the arm: Franka Panda (9 joints), or the fork's UFactory xArm6 with gripper (8 joints)")
    ap.add_argument("--root-scatter", choices=("origins", "uniform"), default="origins",
                    help="anymal / h1 -- where the robots stand: within 0.5 m of the env origins that the reference's terrain "
                         "importer deals (default), or uniformly over the whole terrain (no two envs near each other)")
    args = ap.parse_args()
    args.envs = args.envs or BC.ENVS[args.config]

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

    cfg_id = list(BC.ENVS).index(args.config) + 1
    cfg = BC.build(args.config, device, args.envs, seed=cfg_id * 1000 + rank, actuator=args.actuator,
                   root_scatter=args.root_scatter, robot=args.robot)

    # episode statistics of the done envs: the only cross-rank exchange of the path, one all-reduce per step, started
    # after the step and completed beside the next two (nothing in the step consumes it).  The in-hand env reduces its two
    # success sums inside the step instead (the statistic of the same step needs them).
    reducer = OverlappedLogReducer(cfg.log_sums.numel(), device) if cfg.log_sums is not None else None

    def eager_step():
        if reducer is not None:
            buf, _ = reducer.begin_step()
            cfg.bind_log_sums(buf)
        cfg.step()
        if reducer is not None:
            reducer.end_step()

    graphs = None

    def graph_step():
        i = 0
        if reducer is not None:
            i = reducer.index
            reducer.begin_step(zero=False)   # the graph zeroes the vector itself
        graphs[i].replay()
        if reducer is not None:
            reducer.end_step()

    lists = None

    def commands_step():
        i = 0
        if reducer is not None:
            i = reducer.index
            reducer.begin_step(zero=False)   # the step's first launch zeroes the vector itself
        lists[i].replay()
        if reducer is not None:
            reducer.end_step()

    restart_clocks = getattr(cfg, "restart_sensor_clocks", None)
    warm_steps = {}   # untimed steps actually run ahead of each timed region

    def timed(one_step, replayed=False):
        if restart_clocks is not None:   # the sensors' clocks start every timed region where an env reset leaves them
            restart_clocks()
        # a recorded graph settles over its first replays: each gets its share of warm-up whatever --warmup says
        warm = max(args.warmup, GRAPH_WARMUP_REPLAYS * (2 if reducer is not None else 1)) if replayed else args.warmup
        for _ in range(warm):
            one_step()
        warm_steps[one_step.__name__] = warm
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    # The step is three to seven short launches (2-17 us each): issued one by one from Python they are bound by the host, so
    # the step is recorded -- in a HIP graph, and in a command list (below) -- and replayed.  The in-hand env (four launches,
    # a collective between them on several GPUs) is timed eagerly first (a graph replay of it is slower than that; on one
    # GPU both and the command list are reported).  Every mode is timed and reported in the line.
    use_graph = not args.eager and args.config != "shadow"   # four launches: a replay's fixed cost is above their host time
    other_s = None
    # The headline is the step recorded as ONE chain in the reference's own order (manager_based_rl_env.py:172-195; scene
    # update k and the actuator model of substep k + 1 paired in one launch): nothing runs ahead of a physics step it would
    # depend on, so the number holds with a simulator between the substeps.  Beside it, for the network actuators: the
    # recording with the sensor chain as a branch parallel to the actuator chain -- possible here only because all four
    # substeps' physics results are already in memory.
    two_branch_s = None
    if use_graph:
        other_s = timed(eager_step)
        if getattr(cfg, "overlap_in_graph", False):
            graphs = capture_step(cfg, reducer)
            two_branch_s = timed(graph_step, replayed=True)
            cfg.overlap_in_graph = False
        graphs = capture_step(cfg, reducer)
        elapsed = timed(graph_step, replayed=True)
    else:
        elapsed = timed(eager_step)
        if world == 1:
            cfg.overlap_in_graph = False
            graphs = capture_step(cfg, reducer)
            other_s = timed(graph_step, replayed=True)
    # The step re-issued from a command list: the same launches, the same single chain in the reference's order, issued by a
    # C loop on the stream instead of replayed as graph nodes.  Where that is faster -- the host pays 2.9 us per launch
    # instead of 14 us per replay, which decides the configurations of two to four short launches, and the launches follow
    # each other more closely on a plain stream than as nodes of a graph -- it is the line's value and the graph replay is
    # reported beside it.  (The in-hand step on several GPUs has a collective inside and is not recorded.)
    commands_s, headline_mode, replaced_s = None, ("hip_graph" if use_graph else "eager"), None
    if graphs is not None and (world == 1 or args.config != "shadow"):
        lists = capture_commands(cfg, reducer)
        if lists is not None:
            commands_s = timed(commands_step, replayed=True)
            mine = torch.tensor([commands_s, elapsed], device=device, dtype=torch.float64)
            if dist is not None:   # every rank takes the same decision: on the slowest rank's times
                dist.all_reduce(mine, op=dist.ReduceOp.MAX)
            if not args.eager and float(mine[0]) < float(mine[1]):   # (--eager asks for the launch-by-launch number)
                replaced_s, elapsed, headline_mode = elapsed, commands_s, "command_list"
    # The same run of steps once more, untimed, with a look at the sensor clocks after every step: on how many steps did
    # every env's contact sensor and height scanner refresh (a step on which the clocks skip the refresh does less work).
    refresh_check = None
    if restart_clocks is not None:
        restart_clocks()
        one_step = commands_step if headline_mode == "command_list" else (graph_step if use_graph else eager_step)
        total = args.warmup + args.steps
        full = 0
        for _ in range(total):
            one_step()
            torch.cuda.synchronize()
            full += bool(cfg.sensors_refreshed())
        if reducer is not None:
            reducer.finish()
        refresh_check = {"steps": total, "steps_with_every_sensor_refreshed": full}
        if full != total:
            print(f"bench.py: the sensors skipped refreshes on {total - full} of {total} steps: the timed region did less "
                  "than the full step's work (run fewer steps per timed region)", file=sys.stderr)
    if reducer is not None:
        cfg.bind_log_sums(cfg.log_sums)
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

    kern = time_kernels(cfg, device)
    exact_us = None
    if args.config == "anymal" and args.actuator == "mlp":   # the same network launch with every product in f32 (ILX_MLP_EXACT_F32)
        from isaaclab_amd import _abi
        cfg.env.actuator._flags |= _abi.MLP_EXACT_F32
        exact_us = time_kernels(cfg, device, only=("actuator_net_mlp",))["actuator_net_mlp"]
        cfg.env.actuator._flags &= ~_abi.MLP_EXACT_F32
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.envs * world * args.steps / elapsed
        launches = {k: n for k, (_, n) in cfg.launches.items()}
        nbytes = cfg.bytes_per_env
        # Per kernel: the fraction of the roof it is bound by.  Every launch of this path streams its inputs and outputs once,
        # so the roof is HBM bandwidth (algorithmic bytes / duration / 8 TB/s); the actuator network also has matrix work,
        # priced per data type as it executes (f32 MFMA flops / 157.3 TF + f16 MFMA flops / 2.5 PF = the time the matrix
        # pipes need), and is bound by whichever of the two fractions is larger.  `roofline` is the kernel furthest below
        # its roof among those that take at least 5 % of the step.
        mlp = "actuator_net_mlp"
        profiles = {}
        for key, fname in (("traffic", "r3_hbm_traffic.json"), ("sq", "r3_sq_counters.json")):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    prof = json.load(f)
                if prof.get("envs") == args.envs and prof.get("config", "anymal") == args.config:
                    profiles[key] = prof["kernels"]
            except (OSError, KeyError, ValueError):
                pass
        step_us = ms * 1e3
        per_kernel = {}
        for k, us in kern.items():
            entry = {"launch_us": round(us, 2), "algorithmic_bytes_per_launch": nbytes[k] * args.envs,
                     "hbm_frac": nbytes[k] * args.envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "hbm"}
            entry["frac"] = entry["hbm_frac"]
            if k.startswith(mlp) and args.actuator == "mlp":
                f32, f16 = MLP_F32_MFMA_FLOP_PER_ENV * args.envs, MLP_F16_MFMA_FLOP_PER_ENV * args.envs
                matrix_us = (f32 / (MFMA_F32_PEAK_TFLOPS * 1e12) + f16 / (MFMA_F16_PEAK_TFLOPS * 1e12)) * 1e6
                entry.update(mfma_f32_flop_per_launch=f32, mfma_f16_flop_per_launch=f16, mfma_frac=matrix_us / us,
                             f32_grade_tflops=MLP_FLOP_PER_ENV * args.envs / (us * 1e-6) / 1e12)
                if entry["mfma_frac"] > entry["hbm_frac"]:
                    entry["bound"], entry["frac"] = "mfma", entry["mfma_frac"]
            per_kernel[k] = entry
        share = {k: kern[k] * launches[k] / step_us for k in kern}
        candidates = [k for k in kern if share[k] >= 0.05] or list(kern)
        worst = min(candidates, key=lambda k: per_kernel[k]["frac"])
        w = per_kernel[worst]
        if w["bound"] == "mfma":
            achieved = (w["mfma_f32_flop_per_launch"] + w["mfma_f16_flop_per_launch"]) / (kern[worst] * 1e-6) / 1e12
            roofline = {"bound": "mfma", "kernel": worst, "achieved": achieved, "peak": achieved / w["frac"], "unit": "TFLOP/s",
                        "frac": w["frac"]}
        else:
            roofline = {"bound": "hbm", "kernel": worst, "achieved": w["hbm_frac"] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": w["hbm_frac"]}
        roofline.update(launch_us=kern[worst], launches_per_step=launches[worst], share_of_step=share[worst],
                        algorithmic_bytes_per_launch=nbytes[worst] * args.envs, traffic=None)
        # the same kernel's average in the committed rocprofv3 kernel trace of this command: there every launch runs inside
        # the step, behind another kernel and on data that kernel or the step before left, not back to back with itself
        trace_symbol = {"raycaster_cast": "raycaster_update_lattice_kernel", "mdp_step": "mdp_fast_kernel",
                        "actuator_net_mlp+scene_update": "mlp_split_scene_kernel", "actuator_net_mlp": "mlp_split_persistent_kernel",
                        "actuator_net_mlp_joint_action": "mlp_split_joint_action_kernel",
                        "scene_update_prepare": "scene_update_prepare_kernel", "scene_update": "scene_update_kernel"}.get(worst)
        if trace_symbol and args.config == "anymal" and args.envs == NUM_ENVS:
            try:
                import csv
                with open(os.path.join(ROOT, "profiles", "r3_bench_kernel_stats.csv")) as f:
                    for row in csv.DictReader(f):
                        if trace_symbol in row["Name"]:
                            in_step_us = float(row["AverageNs"]) / 1e3
                            roofline["launch_us_in_step_rocprof"] = round(in_step_us, 2)
                            roofline["frac_in_step_rocprof"] = round(roofline["frac"] * kern[worst] / in_step_us, 4)
                            break
            except (OSError, KeyError, ValueError):
                pass
        if "traffic" in profiles and worst in profiles["traffic"]:   # HBM-side bytes per launch from the committed counter passes
            roofline["traffic"] = profiles["traffic"][worst]["hbm_bytes_per_launch"]
        if "sq" in profiles and worst in profiles["sq"] and "SQ_INSTS_VALU" in profiles["sq"][worst]:
            # what a kernel far below both roofs is busy with: vector instructions issued per SIMD and launch (committed counter pass)
            roofline["valu_insts_per_simd"] = round(profiles["sq"][worst]["SQ_INSTS_VALU"] / (4 * CUS))
        if "mfma_frac" in w:
            roofline.update(mfma_frac=w["mfma_frac"], hbm_frac=w["hbm_frac"])
            if "sq" in profiles and worst in profiles["sq"] and "mfma_busy_frac" in profiles["sq"][worst]:
                roofline["mfma_busy_frac_counters"] = profiles["sq"][worst]["mfma_busy_frac"]
        mfma_frac = per_kernel[mlp]["mfma_frac"] if mlp in per_kernel and "mfma_frac" in per_kernel[mlp] else None
        per_step = cfg.bytes_per_env_step()
        line = {
            "metric": "env-steps/sec (non-physics step) ANYmal-C rough" if args.config == "anymal" else
                      f"env-steps/sec (non-physics step) {args.config}",
            "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": BC.WORKLOAD[args.config] + (", the task on the fork's UFactory xArm6 with gripper"
                                                                   if args.config == "reach" and args.robot == "xarm6" else ""),
                            "envs_per_gpu": args.envs}, **cfg.info,
                           parallelism=f"env-shard x{world}"),
            "algorithmic_bytes_per_env_step": per_step,
            "step_hbm_frac": per_step * args.envs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline": roofline,
            "kernel_rooflines": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in per_kernel.items()},
            "kernels_us": {k: round(v, 2) for k, v in kern.items()},
            "launches_per_step": launches,
            "bytes_per_env_per_launch": nbytes,
            "mfma_frac": mfma_frac,
            "rccl_ranks": rccl_ranks, "ms_per_step_min_rank": rank_ms[0], "ms_per_step_max_rank": rank_ms[1],
        }
        if refresh_check is not None:
            line["sensor_refresh_check"] = refresh_check
        cast = "raycaster_cast" if "raycaster_cast" in kern else "raycaster_refresh"
        if cast in kern and "rays" in cfg.info:   # the height scanner's rate: rays of one refresh / its (cast's) duration
            line["raycast_mrays_per_s"] = round(args.envs * cfg.info["rays"] / kern[cast], 1)
        if args.config == "anymal":
            line["env_term_steps_per_s"] = value * TERM_EVALS_PER_ENV_STEP
            # the same step against the byte table of the measurement recipe (which leaves out what this build adds)
            line["step_hbm_frac_recipe_bytes"] = RECIPE_BYTES_PER_ENV_STEP * args.envs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if exact_us is not None:
                line["mlp_exact_f32_us"] = round(exact_us, 2)
        if use_graph:
            line["graph_warmup_replays"] = warm_steps.get("graph_step", 0) // len(graphs)
        notes = {"hip_graph": "hip-graph replay of the step recorded as one chain in the reference's order (no launch ahead of a physics "
                              "step it would depend on), once per statistics vector",
                 "eager": "eager: one host call per launch",
                 "command_list": "command list: the step's launches, recorded once as one chain in the reference's order, issued by a C "
                                 "loop (ilx_commands_replay), once per statistics vector"}
        line["launch_mode"] = notes[headline_mode]

        def side(seconds, note):
            return {"ms_per_step": seconds / args.steps * 1e3, "value": args.envs * world * args.steps / seconds, "note": note}

        if other_s is not None:
            line["eager" if use_graph else "hip_graph"] = side(
                other_s, "the same step " + ("issued launch by launch from the host" if use_graph else "replayed from a captured graph"))
        if replaced_s is not None:   # the mode that the command list took the line's value from
            line["hip_graph" if use_graph else "eager"] = side(
                replaced_s, "the same step " + ("replayed from a captured graph" if use_graph else "issued launch by launch from the host"))
        elif commands_s is not None:
            line["command_list"] = side(commands_s, "the same step, one chain, issued from a command list (ilx_commands_replay)")
        if two_branch_s is not None:
            line["hip_graph_two_branches"] = {
                "ms_per_step": two_branch_s / args.steps * 1e3, "value": args.envs * world * args.steps / two_branch_s,
                "note": "the same step recorded with the sensor chain (scene updates, scanner refresh) as a branch beside the actuator "
                        "chain: an overlap that needs all substeps' physics results in memory beforehand, so not the headline"}
        if not args.no_cpu_baseline and world == 1 and args.config == "anymal" and args.actuator == "mlp":
            line["cpu_baseline"] = cpu_baseline(3000, envs=args.envs, root_scatter=args.root_scatter)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
