"""GPU: BVH ray cast and the RayCaster sensor (HIP through the C ABI) against the oracle."""

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import raycast as RC
from test_oracle_raycast import make_rays, small_mesh

pytestmark = pytest.mark.gpu


def test_small_mesh_all_outputs(dev):
    from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

    verts, tris, starts, dirs = make_rays()
    mesh = convert_to_mesh(verts.numpy(), tris.numpy().reshape(-1), str(dev))
    assert mesh.num_triangles == tris.shape[0]
    hits, dist, normal, face = raycast_mesh(starts.to(dev), dirs.to(dev), mesh, return_distance=True, return_normal=True,
                                            return_face_id=True)
    h, t, f = RC.raycast_brute(starts, dirs, verts, tris)
    assert_exact(face, f, "face id")
    assert_exact(dist, t, "distance")
    assert_exact(hits, h, "hit position")
    ok = f >= 0
    v0, v1, v2 = (verts[tris[f[ok].long(), k].long()] for k in range(3))
    n = torch.cross(v1 - v0, v2 - v0, dim=-1)
    n = n / n.norm(dim=-1, keepdim=True)
    assert_close(normal.cpu()[ok], n, atol=1e-5, what="normal")
    assert torch.isinf(normal.cpu()[~ok]).all()
    # default call returns Nones like the reference
    out = raycast_mesh(starts.to(dev), dirs.to(dev), mesh)
    assert out[1] is None and out[2] is None and out[3] is None
    assert_exact(out[0], h, "hit position (default call)")
    # (N, R, 3) shaped input keeps its shape
    out = raycast_mesh(starts[:600].view(3, 200, 3).to(dev), dirs[:600].view(3, 200, 3).to(dev), mesh, return_distance=True)
    assert out[0].shape == (3, 200, 3) and out[1].shape == (3, 200)


def test_z_parallel_hint_never_changes_results(dev):
    """The column-grid kernel (z-parallel hint) must return what the tree walk returns -- for z-parallel rays (grid
    look-up), for rays that break the hint (in-kernel tree walk), on seams, vertices and outside the footprint."""
    from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

    verts, tris, starts, dirs = make_rays()
    mesh = convert_to_mesh(verts.numpy(), tris.numpy().reshape(-1), str(dev))
    gen = torch.Generator().manual_seed(5)
    lo, hi = verts.min(0).values, verts.max(0).values
    m = 4096
    xy = lo[:2] - 0.5 + torch.rand(m, 2, generator=gen) * (hi[:2] - lo[:2] + 1.0)
    xy[:verts.shape[0]] = verts[:, :2][:m]                       # straight through mesh vertices
    mid = 0.5 * (verts[tris[:, 0].long()] + verts[tris[:, 1].long()])[:, :2]
    xy[1024:1024 + min(1024, mid.shape[0])] = mid[:1024]         # on triangle edges
    down = torch.cat([xy, torch.full((m, 1), float(hi[2]) + 3.0)], -1)
    up = torch.cat([xy, torch.full((m, 1), float(lo[2]) - 3.0)], -1)
    s_all = torch.cat([down, up, starts])
    d_all = torch.cat([torch.tensor([[0.0, 0.0, -1.0]]).repeat(m, 1), torch.tensor([[0.0, 0.0, 2.5]]).repeat(m, 1), dirs])
    ref = raycast_mesh(s_all.to(dev), d_all.to(dev), mesh, return_distance=True, return_normal=True, return_face_id=True)
    got = raycast_mesh(s_all.to(dev), d_all.to(dev), mesh, return_distance=True, return_normal=True, return_face_id=True,
                       z_parallel_hint=True)
    for a, b, what in zip(got, ref, ("hits", "distance", "normal", "face")):
        assert_exact(a, b, f"{what} with the z-parallel hint")
    h, t, f = RC.raycast_brute(s_all[:2 * m], d_all[:2 * m], verts, tris)
    assert_exact(got[3][:2 * m], f, "column-grid face ids vs brute force")
    assert_exact(got[1][:2 * m], t, "column-grid distances vs brute force")
    assert (f >= 0).float().mean() > 0.5


def test_tiny_and_degenerate_meshes(dev):
    from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

    verts = torch.tensor([[-5.0, -5.0, 0.25], [5.0, -5.0, 0.25], [-5.0, 5.0, 0.25], [5.0, 5.0, 0.25]])
    tris = torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32)
    mesh = convert_to_mesh(verts.numpy(), tris.numpy(), str(dev))
    starts = torch.tensor([[0.3, -0.7, 20.0], [4.0, 4.0, 3.0], [7.0, 0.0, 20.0], [1.0, 1.0, 20.0], [0.0, 0.0, 1.0]])
    dirs = torch.tensor([[0.0, 0.0, -1.0]] * 3 + [[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]])
    hits, dist, _, face = raycast_mesh(starts.to(dev), dirs.to(dev), mesh, return_distance=True, return_face_id=True)
    h, t, f = RC.raycast_brute(starts, dirs, verts, tris)
    assert_exact(face, f, "face")
    assert_exact(hits, h, "hits")
    assert face.cpu().tolist() == [0, 1, -1, -1, 0]
    _, _, _, f2 = raycast_mesh(starts[:1].to(dev), dirs[:1].to(dev), mesh, max_dist=19.75, return_face_id=True)
    assert f2.item() == -1, "max_dist is exclusive"
    hz, dz, _, fz = raycast_mesh(starts.to(dev), dirs.to(dev), mesh, return_distance=True, return_face_id=True,
                                 z_parallel_hint=True)
    assert_exact(fz, f, "face (column grid)")
    assert_exact(hz, h, "hits (column grid)")
    # empty ray set
    out = raycast_mesh(torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev), mesh)
    assert out[0].shape == (0, 3)
    with pytest.raises(Exception):
        convert_to_mesh(verts.numpy(), np.array([[0, 1, 7]], dtype=np.int32), str(dev))


def test_raycaster_against_reference_ray_setup(dev):
    """A flat plane turns hit x/y into the world ray start x/y: pins the sensor's ray set-up to the reference."""
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    g = load_golden("ray_caster_rays")
    verts = np.array([[-100, -100, 0], [100, -100, 0], [-100, 100, 0], [100, 100, 0]], dtype=np.float32)
    mesh = convert_to_mesh(verts, np.array([[0, 1, 2], [1, 3, 2]], dtype=np.int32), str(dev))
    n = g["root_transforms_xyzw"].shape[0]
    for tag, yaw in (("yaw", True), ("full", False)):
        cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=yaw,
                           pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)), update_period=0.02)
        rc = RayCaster(cfg, n, mesh, dev)
        assert rc.num_rays == 187
        assert_exact(rc.ray_starts, g[f"{tag}_ray_starts_local"], "local ray starts")
        rc.drift[:] = g[f"{tag}_drift"].to(dev)
        rc.set_root_transforms(g["root_transforms_xyzw"].to(dev))
        rc.update(0.02)
        d = rc.data
        assert_exact(d.pos_w, g[f"{tag}_pos_w"], "pos_w")
        assert_exact(d.quat_w, g[f"{tag}_quat_w"], "quat_w")
        sw, dw = g[f"{tag}_ray_starts_w"].view(n, 187, 3), g[f"{tag}_ray_dirs_w"].view(n, 187, 3)
        if yaw:
            assert_close(d.ray_hits_w[..., :2], sw[..., :2], what="yaw-only world ray starts (x, y)")
            assert (d.ray_hits_w[..., 2].abs() < 1e-5).all()
        else:
            h, _, _ = RC.raycast_brute(sw, dw, torch.from_numpy(verts), torch.tensor([[0, 1, 2], [1, 3, 2]], dtype=torch.int32))
            hit = ~torch.isinf(h[:, 0])
            got = d.ray_hits_w.cpu().view(-1, 3)
            assert_exact(torch.isinf(got[:, 0]), ~hit, "hit/miss pattern")
            # hit positions are start + t*dir with t up to ~1e3 for grazing rays: relative to the ray length
            assert_close(got[hit], h[hit], rtol=1e-5, atol=1e-4, what="full-orientation hits")


def test_raycaster_lazy_update_and_mask(dev):
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    verts, tris, _ = small_mesh(n=41, size=8.0)
    mesh = convert_to_mesh(verts.numpy(), tris.numpy(), str(dev))
    n = 6
    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True,
                       pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)), update_period=0.02)
    rc = RayCaster(cfg, n, mesh, dev)
    pos = torch.tensor([[4.0, 4.0, 0.6]]).repeat(n, 1) + torch.arange(n).view(n, 1) * torch.tensor([[0.1, 0.05, 0.0]])
    quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1)
    rc.set_pose(pos.to(dev), quat.to(dev))
    rc.update(0.005)           # period not elapsed, but every sensor starts outdated
    first = rc.data.ray_hits_w.clone()
    s, d = RC.sensor_rays(*RC.grid_pattern((1.6, 1.0), 0.1), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
    sw, dw = RC.world_rays(pos, quat, s, d, True)
    h, _, _ = RC.raycast_brute(sw, dw, verts, tris)
    assert_exact(first, h.view(n, -1, 3), "hits after first read")
    # move the robots: nothing changes until the period elapses ...
    rc.set_pose((pos + 0.3).to(dev), quat.to(dev))
    rc.update(0.005)
    assert_exact(rc.data.ray_hits_w, first, "not refreshed before the period elapses")
    # ... except for sensors that were reset (sensor_base.py:182-195)
    rc.reset([1, 4])
    rc.update(0.005)
    now = rc.data.ray_hits_w
    assert_exact(now[[0, 2, 3, 5]], first[[0, 2, 3, 5]], "untouched envs keep their hits")
    sw2, dw2 = RC.world_rays(pos + 0.3, quat, s, d, True)
    h2, _, _ = RC.raycast_brute(sw2, dw2, verts, tris)
    assert_exact(now[[1, 4]], h2.view(n, -1, 3)[[1, 4]], "reset envs are recast")


def test_lidar_pattern_sensor(dev):
    """A lidar pattern on a sensor attached with its full orientation (tree kernel; with the z-parallel hint forced on as
    well: the hint must not change anything for rays that are not vertical)."""
    from isaaclab_amd import _abi
    from isaaclab_amd.sensors.patterns import LidarPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    verts, tris, _ = small_mesh(n=41, size=8.0)
    mesh = convert_to_mesh(verts.numpy(), tris.numpy(), str(dev))
    gen = torch.Generator().manual_seed(5)
    n = 9
    pat = LidarPatternCfg(channels=6, vertical_fov_range=(-60.0, -5.0), horizontal_fov_range=(0.0, 360.0), horizontal_res=12.0)
    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.1, 0.0, 0.4)), attach_yaw_only=False, pattern_cfg=pat, update_period=0.0, max_distance=30.0)
    pos = torch.cat([torch.rand(n, 2, generator=gen) * 4.0 + 2.0, torch.full((n, 1), 1.5)], -1)
    quat = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.2 * torch.randn(n, 4, generator=gen), dim=-1)
    s_local, d_local = pat.func(pat, "cpu")
    s_local = s_local + torch.tensor([0.1, 0.0, 0.4])
    sw, dw = RC.world_rays(pos, quat, s_local, d_local, False)
    h, _, _ = RC.raycast_brute(sw, dw, verts, tris, max_dist=30.0)
    hit = ~torch.isinf(h[:, 0])
    assert hit.float().mean() > 0.5 and (~hit).any()
    for force_hint in (False, True):
        rc = RayCaster(cfg, n, mesh, dev)
        assert rc.num_rays == 6 * 29
        if force_hint:
            rc._hints = _abi.RAYS_Z_PARALLEL
        rc.set_pose(pos.to(dev), quat.to(dev))
        rc.update(0.005)
        got = rc.data.ray_hits_w.cpu().view(-1, 3)
        assert_exact(torch.isinf(got[:, 0]), ~hit, f"hit/miss pattern (hint forced: {force_hint})")
        assert_close(got[hit], h[hit], rtol=1e-5, atol=1e-5, what=f"lidar hits (hint forced: {force_hint})")
        assert not bool(rc._is_outdated.any())


def test_full_size_terrain_sample(dev):
    """2.56 M-triangle rough terrain: 64 envs x 187 rays against the C oracle (BVH, and brute force on a subset)."""
    from isaaclab_amd import synth
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

    verts, tris, height_at = synth.rough_terrain_mesh(seed=3)
    mesh = convert_to_mesh(verts, tris, str(dev))
    assert mesh.depth <= 12
    cm = RC.CMesh(verts, tris)
    gen = torch.Generator().manual_seed(9)
    n = 64
    xy = torch.rand(n, 2, generator=gen) * torch.tensor([78.0, 158.0]) + 1.0
    xy[0] = torch.tensor([-5.0, 40.0])        # over the border ring
    xy[1] = torch.tensor([40.0, 190.0])       # outside everything -> misses
    xy[2] = torch.tensor([8.0, 16.0])         # tile corner: rays on tile seams / grid lines
    z = torch.from_numpy(height_at(xy[:, 0].numpy(), xy[:, 1].numpy())) + 0.6
    pos = torch.cat([xy, z[:, None]], -1)
    yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
    yaw[2] = 0.0
    quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True,
                       pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)), update_period=0.02)
    rc = RayCaster(cfg, n, mesh, dev)
    rc.set_pose(pos.to(dev), quat.to(dev))
    rc.update(0.02)
    hits = rc.data.ray_hits_w.cpu()
    # the world ray starts the kernel used are not observable; rebuild them from the hits (x, y exact for
    # vertical rays) so that the oracle casts the very same rays
    s, d = RC.sensor_rays(*RC.grid_pattern((1.6, 1.0), 0.1), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
    sw, dw = RC.world_rays(pos, quat, s, d, True)
    got_hit = ~torch.isinf(hits[..., 0])
    assert_close(hits[..., :2][got_hit], sw[..., :2][got_hit], what="ray x/y")
    sw_k = sw.clone()
    sw_k[..., :2] = torch.where(got_hit.unsqueeze(-1), hits[..., :2], sw[..., :2])
    h, t, f = RC.raycast_c(sw_k, dw, cm)
    h = h.view(n, -1, 3)
    assert_exact(torch.isinf(h[..., 0]), ~got_hit, "hit / miss pattern")
    assert_exact(hits[got_hit], h[got_hit], "hit positions on the full terrain")
    assert (~got_hit[1]).all() and got_hit[0].all()
    # generic entry point incl. face ids on the same rays, brute-force subset
    sub = torch.randperm(n * 187, generator=gen)[:192]
    hh, dist, _, face = raycast_mesh(sw_k.view(-1, 3)[sub].to(dev), dw.reshape(-1, 3)[sub].to(dev), mesh,
                                     return_distance=True, return_face_id=True)
    hb, tb, fb = RC.raycast_c(sw_k.view(-1, 3)[sub], dw.reshape(-1, 3)[sub], cm, brute=True)
    assert_exact(face, fb, "face ids vs brute force")
    assert_exact(dist, tb, "distance vs brute force")
    # all sensor rays again through both kernels of the generic entry point
    for hint in (False, True):
        ha, ta, _, fa = raycast_mesh(sw_k.view(-1, 3).to(dev), dw.reshape(-1, 3).to(dev), mesh, return_distance=True,
                                     return_face_id=True, z_parallel_hint=hint)
        assert_exact(fa, f, f"face ids, hint={hint}")
        assert_exact(ta, t, f"distances, hint={hint}")
    # rays exactly on the 0.1 m lattice (shared edges and vertices: ties between up to six faces)
    k = torch.randint(0, 800, (2048, 2), generator=gen)
    lat = torch.cat([k[:, :1] * 0.1, k[:, 1:] * 0.2, torch.full((2048, 1), 25.0)], -1)
    dl = torch.tensor([[0.0, 0.0, -1.0]]).repeat(2048, 1)
    _, tl, _, fl = raycast_mesh(lat.to(dev), dl.to(dev), mesh, return_distance=True, return_face_id=True, z_parallel_hint=True)
    _, tlc, flc = RC.raycast_c(lat, dl, cm)
    assert_exact(fl, flc, "lattice face ids (column grid)")
    assert_exact(tl, tlc, "lattice distances (column grid)")
    # oblique rays through the terrain
    o = torch.cat([torch.rand(512, 2, generator=gen) * torch.tensor([80.0, 160.0]), torch.full((512, 1), 3.0)], -1)
    dd = torch.randn(512, 3, generator=gen)
    dd[:, 2] = -dd[:, 2].abs() * 0.3 - 0.05
    ho, to, _, fo = raycast_mesh(o.to(dev), dd.to(dev), mesh, return_distance=True, return_face_id=True)
    hc, tc, fc = RC.raycast_c(o, dd, cm)
    assert_exact(fo, fc, "oblique face ids")
    assert_exact(to, tc, "oblique distances")
    assert (fc >= 0).float().mean() > 0.9
