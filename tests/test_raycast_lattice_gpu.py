"""GPU: the height-field path of the sensor refresh (one workgroup per env, heights staged in LDS) against the
brute-force oracle, bit for bit, on meshes that exercise every way a ray can leave that path; and hit heights / face
indices against the float64 closed form that shares no formula with the intersection test."""

import numpy as np
import pytest
import torch

from conftest import assert_exact
from oracle import heightfield as HF
from oracle import raycast as RC
from test_oracle_heightfield import hf_case, step_case

pytestmark = pytest.mark.gpu

IDENTITY = torch.tensor([[1.0, 0.0, 0.0, 0.0]])


def cast_exact_rays(mesh, dev, xy, z=20.0, pos=None, max_dist=1e6):
    """Sensor refresh whose world ray starts are exactly the given (x, y, z): one env with the identity orientation at
    the origin, the rays handed over as its 'pattern' (yaw-only rotation by the identity quaternion and adding a zero
    position are exact).  Returns hits (r,3) and the height-scan column."""
    from isaaclab_amd import _abi

    r = xy.shape[0]
    starts = torch.cat([xy, torch.full((r, 1), z)], -1).float().to(dev).contiguous()
    dirs = torch.tensor([[0.0, 0.0, -1.0]]).repeat(r, 1).to(dev)
    p = (torch.zeros(1, 3) if pos is None else pos).to(dev)
    q = IDENTITY.to(dev)
    hits = torch.full((1, r, 3), 7.0, device=dev)
    height = torch.zeros(1, r, device=dev)
    _abi.check(_abi.raycaster_update(mesh.handle, _abi.ptr(p), _abi.ptr(q), _abi.ptr(starts), _abi.ptr(dirs), None, 1,
                                     float(max_dist), _abi.ptr(hits), _abi.ptr(height), 0.5, 1, r, _abi.RAYS_Z_PARALLEL,
                                     _abi.stream()), "ilx_raycaster_update")
    torch.cuda.synchronize()
    return hits[0].cpu(), height[0].cpu()


def brute(xy, verts, tris, z=20.0, max_dist=1e6):
    n = xy.shape[0]
    s = torch.cat([xy, torch.full((n, 1), z)], -1).float()
    d = torch.tensor([[0.0, 0.0, -1.0]]).repeat(n, 1)
    return RC.raycast_c(s, d, RC.CMesh(np.asarray(verts), np.asarray(tris)), max_dist=max_dist, brute=True)


def clusters(gen, lo, hi, n_clusters, per, spread=2.5):
    """Ray clusters the size of a scan pattern (the staged window holds 32 lattice lines)."""
    out = []
    for _ in range(n_clusters):
        c = lo + torch.rand(2, generator=gen) * (hi - lo)
        out.append(c + (torch.rand(per, 2, generator=gen) - 0.5) * spread)
    return out


def test_single_tile_every_kind_of_ray(dev):
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v, t, rows, cols = hf_case()
    mesh = convert_to_mesh(v, t, str(dev))
    assert mesh.lattice_stats()["patches"] == 1
    mesh.lattice_counter(True)
    gen = torch.Generator().manual_seed(3)
    lo, hi = torch.from_numpy(v[:, :2].min(0)), torch.from_numpy(v[:, :2].max(0))
    xs, ys = torch.from_numpy(v.reshape(rows, cols, 3)[:, 0, 0].copy()), torch.from_numpy(v.reshape(rows, cols, 3)[0, :, 1].copy())
    total = 0
    sets = clusters(gen, lo + 1.0, hi - 1.0, 6, 900, spread=1.8)
    # on lattice lines, on vertices, on cell diagonals, one float step to either side of a line -- all inside one
    # sub-rectangle of the tile, so that the env's window can hold them
    k = torch.randint(5, 26, (600,), generator=gen)
    l = torch.randint(5, 31, (600,), generator=gen)
    rx = xs[5] + torch.rand(600, generator=gen) * (xs[25] - xs[5])
    ry = ys[5] + torch.rand(600, generator=gen) * (ys[30] - ys[5])
    on_lines = torch.stack([xs[k], ry], -1)
    on_lines[200:400] = torch.stack([rx[200:400], ys[l[200:400]]], -1)
    on_lines[400:] = torch.stack([xs[k[400:]], ys[l[400:]]], -1)
    near = torch.stack([torch.nextafter(xs[k], torch.tensor(1e9)), torch.nextafter(ys[l], torch.tensor(-1e9))], -1)
    f = torch.rand(600, generator=gen)
    diag = torch.stack([xs[k] + f * (xs[k + 1] - xs[k]), ys[l] + f * (ys[l + 1] - ys[l])], -1)
    sets += [on_lines, near, diag, torch.cat([on_lines[:300], rx[:300, None].expand(300, 2) * 0 + torch.stack([rx[:300], ry[:300]], -1)])]
    # a cluster over the mesh edge (the rectangle leaves the patch: general path for the whole env) and one outside
    sets.append(torch.stack([lo[0] - 0.3 + torch.rand(500, generator=gen) * 1.0, lo[1] + torch.rand(500, generator=gen) * 2.0], -1))
    sets.append(hi + 5.0 + torch.rand(64, 2, generator=gen))
    for xy in sets:
        xy = xy.float()
        hits, height = cast_exact_rays(mesh, dev, xy)
        hb, tb, fb = brute(xy, v, t)
        assert_exact(hits, hb, "hits vs brute force")
        assert_exact(height, (0.0 - hb[:, 2]) - 0.5, "height-scan column")
        total += xy.shape[0]
    served = mesh.lattice_counter(False)
    assert 0.5 * total < served < total - 1500, (served, total)   # the height grid served the ordinary rays, the special ones left it


def test_closed_form_pins_height_and_face(dev):
    """Independent of Moller-Trumbore: float64 plane interpolation from the height grid (face from cell + diagonal side)."""
    from isaaclab_amd.utils.mesh import convert_to_mesh, raycast_mesh

    v, t, rows, cols = hf_case(rows=65, cols=81, seed=9, origin=(-20.0, 33.0))
    mesh = convert_to_mesh(v, t, str(dev))
    gen = torch.Generator().manual_seed(8)
    lo, hi = torch.from_numpy(v[:, :2].min(0)), torch.from_numpy(v[:, :2].max(0))
    for xy in clusters(gen, lo + 1.5, hi - 1.5, 8, 1000):
        xy = xy.float()
        hits, _ = cast_exact_rays(mesh, dev, xy)
        z, f, clear = HF.closed_form_hit(v, rows, cols, xy[:, 0].numpy(), xy[:, 1].numpy())
        assert clear.sum() > 900
        assert np.abs(hits[:, 2].numpy()[clear] - z[clear]).max() <= 1e-5, "sensor path: hit height vs closed form"
        # face ids come from the generic entry point (tree and column grid)
        s = torch.cat([xy, torch.full((xy.shape[0], 1), 20.0)], -1)
        d = torch.tensor([[0.0, 0.0, -1.0]]).repeat(xy.shape[0], 1)
        for hint in (False, True):
            h2, _, _, face = raycast_mesh(s.to(dev), d.to(dev), mesh, return_face_id=True, z_parallel_hint=hint)
            assert np.array_equal(face.cpu().numpy()[clear], f[clear]), f"face index vs closed form (hint={hint})"
            assert np.abs(h2[:, 2].cpu().numpy()[clear] - z[clear]).max() <= 1e-5


def test_tiles_border_overhang_and_slope_correction(dev):
    """Several patches side by side with a flat border and a slab hovering over one tile; a slope-corrected tile
    (moved vertices, stretched and degenerate faces).  Everything equals brute force whatever path served it."""
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v1, t1, rows, cols = hf_case(seed=4, origin=(0.0, 0.0))
    w, h = (rows - 1) * 0.1, (cols - 1) * 0.1
    parts_v, parts_t, base = [], [], 0
    for a in range(2):
        for b in range(2):
            vv, tt, _, _ = hf_case(seed=10 + 2 * a + b, origin=(a * w, b * h))
            parts_v.append(vv), parts_t.append(tt + base)
            base += vv.shape[0]
    vs, ts, _ = step_case()
    parts_v.append(vs + np.array([2 * w, 0.0, 0.0], dtype=np.float32)), parts_t.append(ts + base)
    base += vs.shape[0]
    extra_v = np.array([[-3, -3, 0], [3 * w + 3, -3, 0], [-3, 0, 0], [3 * w + 3, 0, 0],                       # border strip
                        [1.0, 1.0, 1.5], [2.2, 1.0, 1.5], [1.0, 2.4, 1.5], [2.2, 2.4, 1.7]], dtype=np.float32)  # hovering slab
    extra_t = np.array([[0, 1, 2], [1, 3, 2], [4, 5, 6], [5, 7, 6]], dtype=np.int32) + base
    v = np.concatenate(parts_v + [extra_v])
    t = np.concatenate(parts_t + [extra_t])
    mesh = convert_to_mesh(v, t, str(dev))
    st = mesh.lattice_stats()
    assert st["patches"] == 5 and st["simple_cells"] < st["cells"]
    mesh.lattice_counter(True)
    gen = torch.Generator().manual_seed(12)
    total = 0
    lo, hi = torch.tensor([-0.5, -0.5]), torch.tensor([3 * w + 0.5, 2 * h + 0.5])
    sets = clusters(gen, lo, hi, 24, 800, spread=2.0)
    sets += clusters(gen, torch.tensor([0.8, 0.8]), torch.tensor([2.4, 2.6]), 4, 800, spread=1.5)         # under the slab
    sets += clusters(gen, torch.tensor([2 * w + 1.6, 0.5]), torch.tensor([2 * w + 2.4, 3.5]), 4, 800, spread=1.2)  # the step
    sets += clusters(gen, torch.tensor([w - 0.2, h - 0.2]), torch.tensor([w + 0.2, h + 0.2]), 3, 800, spread=1.0)  # tile corner
    for xy in sets:
        xy = xy.float()
        hits, _ = cast_exact_rays(mesh, dev, xy, z=25.0)
        hb, _, _ = brute(xy, v, t, z=25.0)
        assert_exact(hits, hb, "hits vs brute force")
        total += xy.shape[0]
    served = mesh.lattice_counter(False)
    assert 0.2 * total < served < total, (served, total)
    # a range limit that cuts some hits off, and the sensor position entering the height column
    xy = sets[0].float()
    hits, height = cast_exact_rays(mesh, dev, xy, z=2.0, pos=torch.tensor([[0.0, 0.0, 0.375]]), max_dist=2.35)
    s = torch.cat([xy, torch.full((xy.shape[0], 1), 2.0)], -1) + torch.tensor([0.0, 0.0, 0.375])
    hb, _, fb = RC.raycast_c(s, torch.tensor([[0.0, 0.0, -1.0]]).repeat(xy.shape[0], 1), RC.CMesh(v, t), max_dist=2.35, brute=True)
    assert_exact(hits, hb, "hits with a range limit")
    assert (fb < 0).any() and (fb >= 0).any()
    assert_exact(height, (0.375 - hb[:, 2]) - 0.5, "height column with the sensor position")


def test_sensor_envs_masks_clocks_and_long_patterns(dev):
    """The RayCaster class on a lattice mesh: random yaw, envs masked out, clocks advanced by the same launch, a pattern
    longer than one workgroup, and a pattern wider than the staged window (falls back as a whole)."""
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v, t, rows, cols = hf_case(rows=81, cols=97, seed=21, origin=(10.0, 20.0))
    mesh = convert_to_mesh(v, t, str(dev))
    cm = RC.CMesh(v, t)
    gen = torch.Generator().manual_seed(2)
    n = 48
    for size, res, expect_served in (((1.6, 1.0), 0.1, True), ((2.0, 1.6), 0.0625, True), ((5.0, 4.5), 0.25, False)):
        cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True,
                           pattern_cfg=GridPatternCfg(resolution=res, size=size), update_period=0.02)
        rc = RayCaster(cfg, n, mesh, dev)
        pos = torch.cat([torch.rand(n, 1, generator=gen) * 7.0 + 10.5, torch.rand(n, 1, generator=gen) * 8.6 + 20.5,
                         torch.rand(n, 1, generator=gen) + 0.5], -1)
        pos[0, :2] = torch.tensor([10.1, 20.1])     # pattern hangs over the mesh edge
        pos[1, :2] = torch.tensor([40.0, 40.0])     # off the mesh
        yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
        quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
        rc.set_pose(pos.to(dev), quat.to(dev))
        rc.reset()
        rc._is_outdated[:] = True
        rc._is_outdated[5:9] = False               # not due: must keep their (zero) hits and clocks
        rc._timestamp[:] = 0.125
        mesh.lattice_counter(True)
        hits = rc.data.ray_hits_w.cpu()
        served = mesh.lattice_counter(False)
        assert (served > 0) == expect_served, (size, served)
        assert not bool(rc._is_outdated[:5].any()) and not bool(rc._is_outdated[9:].any())
        assert torch.equal(rc._timestamp_last_update.cpu()[:5], torch.full((5,), 0.125))
        assert torch.equal(rc._timestamp_last_update.cpu()[5:9], torch.zeros(4))
        assert (hits[5:9] == 0).all()
        s, d = RC.sensor_rays(*RC.grid_pattern(size, res), (0.0, 0.0, 20.0), (1.0, 0.0, 0.0, 0.0))
        sw, dw = RC.world_rays(pos, quat, s, d, True)
        got_hit = ~torch.isinf(hits[..., 0])
        sw_k = sw.clone()   # the kernel's own ray x/y (exact for vertical rays) so that the oracle casts the same rays
        sw_k[..., :2] = torch.where(got_hit.unsqueeze(-1), hits[..., :2], sw[..., :2])
        keep = torch.ones(n, dtype=torch.bool)
        keep[5:9] = False
        assert (sw_k[keep][..., :2] - sw[keep][..., :2]).abs().max() < 2e-5
        h, _, _ = RC.raycast_c(sw_k, dw, cm)
        h = h.view(n, -1, 3)
        assert_exact(torch.isinf(h[keep][..., 0]), ~got_hit[keep], "hit / miss pattern")
        assert_exact(hits[keep][got_hit[keep]], h[keep][got_hit[keep]], f"hits, pattern {size} @ {res}")
        assert (~got_hit[1]).all() and (size[0] > 1.7 or got_hit[2].all())


def test_kept_pattern_is_reused_only_for_the_same_pattern(dev):
    """ILX_RAYS_PATTERN_STATIC: the packed pattern and its box stay in the stream's workspace between refreshes.  Two
    sensors with different patterns that take turns on one mesh and stream, and a sensor that refreshes several times in a
    row, give what a sensor gives on its first refresh."""
    from isaaclab_amd import _abi
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v, t, rows, cols = hf_case(rows=81, cols=97, seed=22, origin=(10.0, 20.0))
    mesh = convert_to_mesh(v, t, str(dev))
    gen = torch.Generator().manual_seed(5)
    n = 64
    pos = torch.cat([torch.rand(n, 1, generator=gen) * 6.0 + 11.0, torch.rand(n, 1, generator=gen) * 7.0 + 21.0,
                     torch.rand(n, 1, generator=gen) + 0.5], -1).to(dev)
    yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
    quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1).to(dev)

    def sensor(size, res):
        cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True,
                           pattern_cfg=GridPatternCfg(resolution=res, size=size), update_period=0.02)
        rc = RayCaster(cfg, n, mesh, dev)
        assert rc._hints & _abi.RAYS_PATTERN_STATIC
        rc.set_pose(pos, quat)
        return rc

    def refresh(rc):
        rc._is_outdated[:] = True
        return rc.data.ray_hits_w.clone()

    first = {}
    for key, size, res in (("a", (1.6, 1.0), 0.1), ("b", (1.0, 1.4), 0.2)):   # fresh mesh workspace state for neither: just first use
        first[key] = refresh(sensor(size, res))
    a, b = sensor((1.6, 1.0), 0.1), sensor((1.0, 1.4), 0.2)
    for _ in range(2):
        assert torch.equal(refresh(a), first["a"]) and torch.equal(refresh(a), first["a"])   # second one reuses
        assert torch.equal(refresh(b), first["b"])                                           # other pattern: packed anew
    assert torch.equal(refresh(a), first["a"])


def test_lean_refresh_leaves_the_hit_records_to_their_reader(dev):
    """``lean_hits``: refreshes that write only the bound height scan (ILX_RAYS_HEIGHT_ONLY) and hit records formed when
    ``data.ray_hits_w`` is read, against a sensor that stores the records eagerly -- bit for bit, over three refreshes with
    moving poses, drift, envs that are not due, and a read in between."""
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v, t, rows, cols = hf_case(rows=81, cols=97, seed=23, origin=(10.0, 20.0))
    mesh = convert_to_mesh(v, t, str(dev))
    gen = torch.Generator().manual_seed(9)
    n = 200
    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True, pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)),
                       update_period=0.02, drift_range=(-0.05, 0.05))
    eager, lean = RayCaster(cfg, n, mesh, dev), RayCaster(cfg, n, mesh, dev)
    lean.lean_hits = True
    scans = [torch.zeros(n, eager.num_rays, device=dev) for _ in range(2)]
    for rc, scan in zip((eager, lean), scans):
        rc.bind_height_scan(scan, 0.5)
        rc.reset()
    lean.drift.copy_(eager.drift)
    for step in range(3):
        pos = torch.cat([torch.rand(n, 1, generator=gen) * 7.0 + 10.5, torch.rand(n, 1, generator=gen) * 8.6 + 20.5,
                         torch.rand(n, 1, generator=gen) + 0.5], -1).to(dev)
        pos[3, :2] = torch.tensor([40.0, 40.0])     # off the mesh: misses
        yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
        quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1).to(dev)
        due = torch.rand(n, generator=gen) < (1.0 if step == 0 else 0.6)
        for rc in (eager, lean):
            rc.set_pose(pos, quat)
            rc._timestamp += 0.02
            rc._is_outdated.copy_(due)
            rc._update_outdated_buffers()
        assert lean._hits_behind and not eager._hits_behind
        assert_exact(scans[1], scans[0].cpu(), f"height scan, refresh {step}")
        if step == 1:
            continue   # the records of two refreshes are formed at once by the read after the third
        untouched = lean._data.hits_storage.clone()
        got = lean.data.ray_hits_w
        assert not lean._hits_behind and got.data_ptr() == lean._data.hits_storage.data_ptr()
        assert_exact(got, eager.data.ray_hits_w.cpu(), f"hit records, refresh {step}")
        assert step > 0 or not torch.equal(untouched, got)   # ... and they were not there before the read
        assert_exact(lean._data.pos_w, eager._data.pos_w.cpu(), "pos_w")
    assert torch.isinf(eager.data.ray_hits_w[3]).all()


def test_first_refresh_of_a_stream_can_be_recorded(dev):
    """``RayCaster`` sizes the refresh's per-stream workspaces when it is set up (ilx_raycaster_reserve), so the very first
    refresh on a stream needs no allocation: it is recorded into a graph on a stream the sensor has never run on -- an
    allocation inside a capture would fail it -- and the replay equals a sensor refreshed launch by launch."""
    from isaaclab_amd.sensors.patterns import GridPatternCfg
    from isaaclab_amd.sensors.ray_caster import OffsetCfg, RayCaster, RayCasterCfg
    from isaaclab_amd.utils.mesh import convert_to_mesh

    v, t, rows, cols = hf_case(rows=81, cols=97, seed=41, origin=(10.0, 20.0))
    mesh = convert_to_mesh(v, t, str(dev))
    gen = torch.Generator().manual_seed(3)
    n = 300
    cfg = RayCasterCfg(offset=OffsetCfg(pos=(0.0, 0.0, 20.0)), attach_yaw_only=True, pattern_cfg=GridPatternCfg(resolution=0.1, size=(1.6, 1.0)),
                       update_period=0.02)
    recorded, eager = RayCaster(cfg, n, mesh, dev), RayCaster(cfg, n, mesh, dev)
    pos = torch.cat([torch.rand(n, 1, generator=gen) * 7.0 + 10.5, torch.rand(n, 1, generator=gen) * 8.6 + 20.5,
                     torch.rand(n, 1, generator=gen) + 0.5], -1).to(dev)
    yaw = torch.rand(n, generator=gen) * 6.28 - 3.14
    quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1).to(dev)
    for rc in (recorded, eager):
        rc.reset()
        rc.set_pose(pos, quat)
        rc._timestamp += 0.02
        rc._is_outdated.fill_(True)
    eager._update_outdated_buffers()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        recorded._update_outdated_buffers()
    assert bool(recorded._is_outdated.all()), "recording must not run the refresh"
    graph.replay()
    torch.cuda.synchronize()
    assert_exact(recorded._data.hits_storage, eager._data.hits_storage.cpu(), "hit records of the replayed first refresh")
    assert_exact(recorded._data.pos_w, eager._data.pos_w.cpu(), "pos_w")
    assert not bool(recorded._is_outdated.any())
