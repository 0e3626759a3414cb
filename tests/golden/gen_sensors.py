"""Golden fixtures for the contact-sensor bookkeeping and the ray-caster ray set-up, produced by the
reference's SensorBase / ContactSensor / RayCaster code driven on fake physics views."""

from __future__ import annotations

import sys
import types

import torch

import _refload as R
from gen_golden import gen, save

SECTIONS = {}


def section(fn):
    SECTIONS[fn.__name__] = fn
    return fn


def _load_sensor_modules():
    R.bootstrap()
    physx = sys.modules["omni.physics.tensors.impl.api"]
    physx.ArticulationView = type("ArticulationView", (), {})
    physx.RigidBodyView = type("RigidBodyView", (), {})
    R.placeholder("omni.timeline")
    sys.modules["isaacsim.core.prims"].XFormPrim = type("XFormPrim", (), {})
    calls = []

    def recorder(ray_starts, ray_directions, max_dist=1e6, mesh=None, **kw):
        calls.append((ray_starts.clone(), ray_directions.clone(), max_dist))
        return (torch.zeros_like(ray_starts), None, None, None)

    R.placeholder("isaaclab.utils.warp", raycast_mesh=recorder, convert_to_warp_mesh=lambda *a, **k: None)
    R.placeholder("isaaclab.sensors")
    sb = R.load("isaaclab.sensors.sensor_base", f"{R.L}/sensors/sensor_base.py")
    R.placeholder("isaaclab.sensors.contact_sensor")
    csd = R.load("isaaclab.sensors.contact_sensor.contact_sensor_data", f"{R.L}/sensors/contact_sensor/contact_sensor_data.py")
    cs = R.load("isaaclab.sensors.contact_sensor.contact_sensor", f"{R.L}/sensors/contact_sensor/contact_sensor.py")
    R.placeholder("isaaclab.sensors.ray_caster")
    R.placeholder("isaaclab.sensors.ray_caster.patterns")
    pat = R.load("isaaclab.sensors.ray_caster.patterns.patterns", f"{R.L}/sensors/ray_caster/patterns/patterns.py")
    rcd = R.load("isaaclab.sensors.ray_caster.ray_caster_data", f"{R.L}/sensors/ray_caster/ray_caster_data.py")
    rc = R.load("isaaclab.sensors.ray_caster.ray_caster", f"{R.L}/sensors/ray_caster/ray_caster.py")
    return dict(sb=sb, csd=csd, cs=cs, pat=pat, rcd=rcd, rc=rc, calls=calls, physx=physx)


def make_contact_sensor(mods, n, b, h, forces_seq, update_period, track_air_time=True, thr=1.0):
    cs = mods["cs"]
    s = object.__new__(cs.ContactSensor)
    s.cfg = types.SimpleNamespace(update_period=update_period, history_length=h, track_air_time=track_air_time,
                                  force_threshold=thr, filter_prim_paths_expr=[], track_pose=False, debug_vis=False)
    s._is_visualizing = False
    s._is_initialized = True
    s._initialize_handle = s._invalidate_initialize_handle = s._debug_vis_handle = None
    s._num_envs, s._num_bodies, s._device = n, b, "cpu"
    s._sim_physics_dt = 0.005
    s._is_outdated = torch.ones(n, dtype=torch.bool)
    s._timestamp = torch.zeros(n)
    s._timestamp_last_update = torch.zeros(n)
    d = mods["csd"].ContactSensorData()
    d.net_forces_w = torch.zeros(n, b, 3)
    d.net_forces_w_history = torch.zeros(n, h, b, 3) if h > 0 else d.net_forces_w.unsqueeze(1)
    if track_air_time:
        d.last_air_time = torch.zeros(n, b)
        d.current_air_time = torch.zeros(n, b)
        d.last_contact_time = torch.zeros(n, b)
        d.current_contact_time = torch.zeros(n, b)
    s._data = d
    state = {"k": 0}

    class View:
        def get_net_contact_forces(self, dt):
            return forces_seq[state["k"]].reshape(-1, 3).clone()

    s._contact_physx_view = View()
    return s, state


def synth_forces(g, steps, n, b):
    """Forces that switch contact on/off per body with values straddling the 1.0 N threshold."""
    on = torch.rand(steps, n, b, generator=g) < 0.55
    # hold contact states for a few steps so that timers accumulate
    for s in range(1, steps):
        keep = torch.rand(n, b, generator=g) < 0.6
        on[s] = torch.where(keep, on[s - 1], on[s])
    mag = torch.rand(steps, n, b, generator=g) * 40.0
    f = torch.randn(steps, n, b, 3, generator=g)
    f = f / f.norm(dim=-1, keepdim=True) * mag.unsqueeze(-1) * on.unsqueeze(-1)
    # a few forces right at the threshold (norm within 1 ulp-ish of 1.0)
    f[:, 0, 0] = torch.tensor([0.6, 0.8, 0.0])
    f[:, 1, 0] = torch.tensor([0.6, 0.8, 1e-4])
    f[:, 2, 0] = torch.tensor([0.6, 0.79999, 0.0])
    return f


@section
def contact_sensor():
    mods = _load_sensor_modules()
    g = gen(4001)
    n, b, h, steps = 48, 17, 3, 12
    forces = synth_forces(g, steps, n, b)
    s, st = make_contact_sensor(mods, n, b, h, forces, update_period=0.005)
    rec = {k: [] for k in ("net", "hist", "cur_air", "last_air", "cur_contact", "last_contact", "ts", "ts_last",
                           "first_contact", "first_air")}
    reset_ids = [3, 7, 20]
    for k in range(steps):
        st["k"] = k
        if k == 7:
            s.reset(reset_ids)
        s.update(0.005)
        d = s._data
        for key, v in zip(rec, (d.net_forces_w, d.net_forces_w_history, d.current_air_time, d.last_air_time,
                                d.current_contact_time, d.last_contact_time, s._timestamp, s._timestamp_last_update,
                                s.compute_first_contact(0.02), s.compute_first_air(0.02))):
            rec[key].append(v.clone())
    out = {k: torch.stack(v) for k, v in rec.items()}
    out.update(forces=forces, reset_ids=torch.tensor(reset_ids), reset_step=torch.tensor(7), dt=torch.tensor(0.005),
               period=torch.tensor(0.005))
    # slower sensor: period 0.02 with dt 0.005 and history: only every 4th tick refreshes
    s2, st2 = make_contact_sensor(mods, n, b, h, forces, update_period=0.02)
    rec2 = {k: [] for k in ("hist", "cur_air", "cur_contact", "outdated")}
    for k in range(steps):
        st2["k"] = k
        s2.update(0.005)
        d = s2._data
        for key, v in zip(rec2, (d.net_forces_w_history, d.current_air_time, d.current_contact_time, s2._is_outdated)):
            rec2[key].append(v.clone())
    out.update({f"slow_{k}": torch.stack(v) for k, v in rec2.items()})
    save("contact_sensor", **out)


@section
def ray_caster_rays():
    """Grid pattern + RayCaster._initialize_rays_impl/_update_buffers_impl up to the ray-cast call."""
    mods = _load_sensor_modules()
    rc, pat, physx = mods["rc"], mods["pat"], mods["physx"]
    g = gen(5001)
    n = 64
    # velocity_env_cfg.py:66-72: GridPatternCfg(resolution=0.1, size=[1.6, 1.0]), offset z=20, yaw only
    pcfg = types.SimpleNamespace(resolution=0.1, size=(1.6, 1.0), direction=(0.0, 0.0, -1.0), ordering="xy",
                                 func=pat.grid_pattern)
    starts, dirs = pat.grid_pattern(pcfg, "cpu")
    pcfg_yx = types.SimpleNamespace(resolution=0.25, size=(1.0, 0.5), direction=(0.0, 0.0, -1.0), ordering="yx",
                                    func=pat.grid_pattern)
    starts_yx, dirs_yx = pat.grid_pattern(pcfg_yx, "cpu")
    out = dict(grid_starts=starts, grid_dirs=dirs, grid_yx_starts=starts_yx, grid_yx_dirs=dirs_yx)

    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    q[0] = torch.tensor([1.0, 0, 0, 0])
    pos = torch.randn(n, 3, generator=g) * 5.0
    root_tf = torch.cat([pos, q[:, [1, 2, 3, 0]]], dim=-1)

    class AView(physx.ArticulationView):
        count = n

        def get_root_transforms(self):
            return root_tf.clone()

    for tag, yaw_only in (("yaw", True), ("full", False)):
        s = object.__new__(rc.RayCaster)
        s._initialize_handle = s._invalidate_initialize_handle = s._debug_vis_handle = None
        s.cfg = types.SimpleNamespace(pattern_cfg=pcfg, attach_yaw_only=yaw_only, max_distance=1e6,
                                      mesh_prim_paths=["/World/ground"], drift_range=(0.0, 0.0),
                                      offset=types.SimpleNamespace(pos=(0.0, 0.0, 20.0), rot=(1.0, 0.0, 0.0, 0.0)))
        s._device = "cpu"
        s._view = AView()
        s._data = mods["rcd"].RayCasterData()
        s.meshes = {"/World/ground": None}
        s._initialize_rays_impl()
        s.drift[:] = torch.randn(n, 3, generator=g) * 0.01
        mods["calls"].clear()
        s._update_buffers_impl(torch.arange(n))
        ray_starts_w, ray_dirs_w, max_dist = mods["calls"][-1]
        out.update({f"{tag}_ray_starts_local": s.ray_starts[0], f"{tag}_ray_dirs_local": s.ray_directions[0],
                    f"{tag}_drift": s.drift.clone(), f"{tag}_ray_starts_w": ray_starts_w,
                    f"{tag}_ray_dirs_w": ray_dirs_w, f"{tag}_pos_w": s._data.pos_w, f"{tag}_quat_w": s._data.quat_w})
    out["root_transforms_xyzw"] = root_tf
    save("ray_caster_rays", **out)


@section
def ray_patterns():
    """lidar / bpearl / pinhole patterns of the reference (patterns.py:61-179) for a few configurations."""
    mods = _load_sensor_modules()
    pat = mods["pat"]
    NS = types.SimpleNamespace
    out = {}
    for tag, cfg in (("lidar_full", NS(channels=16, vertical_fov_range=(-15.0, 15.0), horizontal_fov_range=(0.0, 360.0), horizontal_res=2.0)),
                     ("lidar_part", NS(channels=5, vertical_fov_range=(-30.0, 10.0), horizontal_fov_range=(-60.0, 75.0), horizontal_res=7.5))):
        s, d = pat.lidar_pattern(cfg, "cpu")
        out[f"{tag}_starts"], out[f"{tag}_dirs"] = s, d
    angles = [89.5, 86.6875, 83.875, 81.0625, 78.25, 75.4375, 72.625, 69.8125, 67.0, 64.1875, 61.375, 58.5625, 55.75, 52.9375,
              50.125, 47.3125, 44.5, 41.6875, 38.875, 36.0625, 33.25, 30.4375, 27.625, 24.8125, 22, 19.1875, 16.375, 13.5625,
              10.75, 7.9375, 5.125, 2.3125]   # the RS-Bpearl ring table (patterns_cfg.py:186-190, data)
    s, d = pat.bpearl_pattern(NS(horizontal_fov=360.0, horizontal_res=10.0, vertical_ray_angles=angles), "cpu")
    out["bpearl_starts"], out["bpearl_dirs"] = s, d
    s, d = pat.bpearl_pattern(NS(horizontal_fov=90.0, horizontal_res=4.0, vertical_ray_angles=[60.0, 30.0, 5.0]), "cpu")
    out["bpearl_small_starts"], out["bpearl_small_dirs"] = s, d
    k = torch.tensor([[[380.0, 0.0, 32.5], [0.0, 375.0, 23.0], [0.0, 0.0, 1.0]], [[120.0, 0.0, 30.0], [0.0, 130.0, 26.0], [0.0, 0.0, 1.0]]])
    s, d = pat.pinhole_camera_pattern(NS(width=64, height=48), k, "cpu")
    out["pinhole_k"], out["pinhole_starts"], out["pinhole_dirs"] = k, s, d
    save("ray_patterns", **out)
