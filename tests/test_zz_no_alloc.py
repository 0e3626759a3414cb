"""GPU: the sensor refresh that allocates nothing (ilx_raycaster_refresh_static, a buffer built into the library in
place of the workspace ilx_raycaster_refresh grows with hipMalloc) against ilx_raycaster_refresh, bit for bit: two
calls in a row, the second with fewer envs, then the size limit and the hint it refuses."""

import ctypes

import pytest
import torch

from conftest import assert_exact

pytestmark = pytest.mark.gpu


def make_sensor(n, mesh, dev, seed):
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg

    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True,
                       pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)), update_period=0.02, drift_range=(-0.05, 0.05))
    rc = RayCaster(cfg, n, mesh, dev)
    gen = torch.Generator().manual_seed(seed)
    pos = torch.cat([torch.rand(n, 1, generator=gen) * 9.0 + 9.5, torch.rand(n, 1, generator=gen) * 11.0 + 19.5,
                     torch.rand(n, 1, generator=gen) + 0.5], -1)
    yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
    quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
    rc.set_pose(pos.to(dev), quat.to(dev))
    rc.drift[:] = (torch.rand(n, 3, generator=gen) * 0.1 - 0.05).to(dev)
    return rc


def refreshed(rc, static, due):
    rc.static_workspace = static
    rc._timestamp[:] = 0.25
    rc._timestamp_last_update[:] = 0.0
    rc._is_outdated[:] = due.to(rc._is_outdated.device)
    rc._data.ray_hits_w.fill_(-7.0)
    d = rc.data
    torch.cuda.synchronize()
    return d.ray_hits_w.clone(), d.pos_w.clone(), d.quat_w.clone(), rc._timestamp_last_update.clone(), rc._is_outdated.clone()


def test_static_refresh_matches_refresh(dev):
    from isaaclab_amd import _abi
    from isaaclab_amd.utils.mesh import convert_to_mesh
    from test_oracle_heightfield import hf_case

    v, t, rows, cols = hf_case(rows=101, cols=121, seed=31, origin=(9.0, 19.0))
    mesh = convert_to_mesh(v, t, str(dev))
    assert mesh.lattice_stats()["patches"] == 1   # the refresh takes the path that needs the buffer between launches
    for n, seed in ((700, 1), (93, 2)):           # the second call is smaller: stale records of the first stay behind it
        due = torch.rand(n, generator=torch.Generator().manual_seed(seed)) < 0.8
        want = refreshed(make_sensor(n, mesh, dev, seed), False, due)
        mesh.lattice_counter(True)
        got = refreshed(make_sensor(n, mesh, dev, seed), True, due)
        assert mesh.lattice_counter(False) > 0.7 * int(due.sum()) * 187
        for a, b, what in zip(got, want, ("ray_hits_w", "pos_w", "quat_w", "timestamp_last_update", "is_outdated")):
            assert_exact(a, b, f"{what}, {n} envs")
        assert (got[0][~due] == -7.0).all() and not bool(got[4].any())
    # beyond the built-in buffer: a clear error, nothing launched
    rc = make_sensor(8, mesh, dev, 3)
    big = _abi.RayCasterDesc(
        mesh=mesh.handle, num_envs=40000, num_rays=rc.num_rays, attach_yaw_only=1, hints=rc._hints, max_distance=1e6,
        pose_pos=_abi.ptr(rc._pose_pos), pose_quat=_abi.ptr(rc._pose_quat), pose_pos_stride=3, pose_quat_stride=4, drift=None,
        ray_starts_local=_abi.ptr(rc.ray_starts), ray_directions_local=_abi.ptr(rc.ray_directions), pos_w=_abi.ptr(rc._data.pos_w),
        quat_w=_abi.ptr(rc._data.quat_w), ray_hits_w=_abi.ptr(rc._data.ray_hits_w), height_scan=None, height_offset=0.0,
        timestamp=_abi.ptr(rc._timestamp), timestamp_last_update=_abi.ptr(rc._timestamp_last_update),
        is_outdated=rc._is_outdated.data_ptr())
    status = _abi.raycaster_refresh_static(ctypes.byref(big), _abi.stream())
    assert status == -2 and b"32768 envs" in _abi.lib.ilx_last_error()   # ILX_ESHAPE
    # a preparation kept in the per-stream workspace is not what this entry point reads: refused, nothing launched
    big.num_envs, big.hints = 8, rc._hints | _abi.RAYS_PREPARED
    status = _abi.raycaster_refresh_static(ctypes.byref(big), _abi.stream())
    assert status == -1 and b"ILX_RAYS_PREPARED" in _abi.lib.ilx_last_error()   # ILX_EINVAL
    torch.cuda.synchronize()
