#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/profile_bench.sh into the summaries kept under profiles/:

    python tools/reduce_profiles.py <tag> [envs]

profiles/<tag>_bench_kernel_stats.csv  the --kernel-trace --stats table, as rocprofv3 wrote it
profiles/<tag>_bench_line.json         the bench line of the same tree
profiles/<tag>_sq_counters.json        per-kernel SQ counters per launch (busy cycles, waves, VALU / SALU / MFMA instructions)
profiles/<tag>_hbm_traffic.json        per-kernel FETCH_SIZE / WRITE_SIZE per launch (KB; FETCH_SIZE doubled: gfx950 tallies its
                                       128-byte requests as 64 bytes) -> bytes that crossed the L2 <-> fabric boundary
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {  # substring of the kernel symbol -> bench kernel name (launches of one bench kernel are summed per call)
    "mlp_split_joint_action": "actuator_net_mlp_joint_action", "mlp_split_scene": "actuator_net_mlp+scene_update",
    "joint_action_kernel": "joint_action_process", "mlp_split_persistent": "actuator_net_mlp", "mlp_mfma": "actuator_net_mlp", "actuator_lstm_kernel": "actuator_net_lstm", "scene_update_kernel": "scene_update",
    "raycaster_update": "raycaster_cast", "raycaster_prepare_kernel": "raycaster_prepare", "sensor_refreshed_kernel": "raycaster_cast",
    "scene_update_prepare_kernel": "scene_update_prepare",
    "mdp_step_kernel": "mdp_step", "mdp_fast_kernel": "mdp_step",
}
PRIMARY = {"actuator_net_mlp_joint_action": "mlp_split_joint_action", "actuator_net_mlp+scene_update": "mlp_split_scene", "actuator_net_lstm": "actuator_lstm_kernel", "joint_action_process": "joint_action_kernel", "actuator_net_mlp": "mlp_", "scene_update": "scene_update_kernel",
           "raycaster_cast": "raycaster_update_lattice_kernel", "raycaster_prepare": "raycaster_prepare_kernel",
           "scene_update_prepare": "scene_update_prepare_kernel", "mdp_step": "mdp_"}


def counter_per_launch(path, counter):
    """Sum of `counter` over the launches that belong to one call of each bench kernel, divided by the number of calls
    (= launches of the kernel's primary symbol), at full grid size only (the bench's 16384-env launches)."""
    rows = list(csv.DictReader(open(path)))
    total, calls, grid = {}, {}, {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        for sub, name in KERNELS.items():   # the first (most specific) entry that matches names the kernel
            if sub in r["Kernel_Name"]:
                if PRIMARY[name] in r["Kernel_Name"]:
                    grid[name] = max(grid.get(name, 0), int(r["Grid_Size"]))
                break
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        for sub, name in KERNELS.items():
            if sub in r["Kernel_Name"]:
                primary = PRIMARY[name] in r["Kernel_Name"]
                if primary and int(r["Grid_Size"]) != grid[name]:
                    break  # smaller launches belong to the set-up, not to the step
                total[name] = total.get(name, 0.0) + float(r["Counter_Value"])
                if primary:
                    calls[name] = calls.get(name, 0) + 1
                break
    return {k: (total[k] / calls[k], calls[k]) for k in total if calls.get(k)}


def main():
    tag = sys.argv[1]
    envs = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "bench_line.json")):
        shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{tag}_bench_line.json"))
    fetch = glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
    write = glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
    if fetch and write:
        f = counter_per_launch(fetch[0], "FETCH_SIZE")
        w = counter_per_launch(write[0], "WRITE_SIZE")
        out = {"command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline  (and a separate pass "
                          "with --pmc WRITE_SIZE)",
               "correction": "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes); counts L2 fabric-side requests, "
                             "Infinity Cache hits included",
               "envs": envs, "kernels": {}}
        for k in f:
            fk, n = f[k]
            wk = w.get(k, (0.0, 0))[0]
            out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(2.0 * fk, 1), "launches_sampled": n,
                                 "WRITE_SIZE_KB_per_launch": round(wk, 1), "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
        json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    sq = glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
    if sq:
        rows = list(csv.DictReader(open(sq[0])))
        names = sorted({r["Counter_Name"] for r in rows})
        out = {"command": "rocprofv3 --pmc " + " ".join(names) + " -- python3 bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline",
               "note": "per launch, summed over the shader engines as rocprofv3 reports them; full-size launches only", "envs": envs,
               "kernels": {}}
        for c in names:
            for k, (v, n) in counter_per_launch(sq[0], c).items():
                out["kernels"].setdefault(k, {"launches_sampled": n})[c] = round(v, 1)
        json.dump(out, open(os.path.join(dst, f"{tag}_sq_counters.json"), "w"), indent=1)
    print("profiles written for", tag)


if __name__ == "__main__":
    main()
