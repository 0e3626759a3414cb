"""Oracle for the mesh ray cast and the RayCaster ray set-up (test infrastructure, CPU only).

PARITY UNPINNED for the intersection itself: the reference hands it to warp's ``mesh_query_ray``
(``isaaclab/utils/warp/kernels.py:66``), a third-party package that is not available here and whose version the
reference does not pin.  What *is* pinned against the reference (fixtures ``ray_caster_rays.npz``): the grid
pattern (``sensors/ray_caster/patterns/patterns.py:16-58``) and the world-frame ray starts / directions that
``RayCaster._update_buffers_impl`` (``sensors/ray_caster/ray_caster.py:220-260``) hands to the cast.

The intersection follows the contract documented in ``kernels.py:27-75`` / ``ops.py:47-58``: closest hit with
``0 <= t < max_dist``, ``hit = start + t * dir``, misses are ``inf`` / ``-1``; ties go to the lowest face index.
``raycast_brute`` (torch, every ray against every triangle) is the definition; ``raycast_c`` calls the C
restatement in ``raycast_ref.c`` (brute force or its own binary BVH) for sizes torch cannot brute-force.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import mathx as M

BARY_EPS = 1.0e-6


# ---- ray set-up (pinned) ---------------------------------------------------------------------------
def grid_pattern(size, resolution: float, direction=(0.0, 0.0, -1.0), ordering: str = "xy"):
    """patterns.py:16-58.  Uses the same ``torch.arange`` call so the fp32 grid coordinates are identical."""
    if ordering not in ("xy", "yx"):
        raise ValueError(f"Ordering must be 'xy' or 'yx'. Received: '{ordering}'.")
    if resolution <= 0:
        raise ValueError(f"Resolution must be greater than 0. Received: '{resolution}'.")
    x = torch.arange(start=-size[0] / 2, end=size[0] / 2 + 1.0e-9, step=resolution)
    y = torch.arange(start=-size[1] / 2, end=size[1] / 2 + 1.0e-9, step=resolution)
    gx, gy = torch.meshgrid(x, y, indexing="xy" if ordering == "xy" else "ij")
    starts = torch.zeros(gx.numel(), 3)
    starts[:, 0] = gx.flatten()
    starts[:, 1] = gy.flatten()
    dirs = torch.zeros_like(starts)
    dirs[..., :] = torch.tensor(list(direction))
    return starts, dirs


def sensor_rays(ray_starts, ray_dirs, offset_pos, offset_rot):
    """ray_caster.py:201-209: apply the sensor offset to the pattern (directions rotated, starts shifted)."""
    q = torch.tensor(list(offset_rot), dtype=torch.float32).repeat(len(ray_dirs), 1)
    return ray_starts + torch.tensor(list(offset_pos), dtype=torch.float32), M.quat_apply(q, ray_dirs)


def world_rays(pos_w, quat_w, ray_starts, ray_dirs, attach_yaw_only: bool):
    """ray_caster.py:242-252: per-env world-frame ray starts (N,R,3) and directions (N,R,3)."""
    n, r = pos_w.shape[0], ray_starts.shape[0]
    q = quat_w.unsqueeze(1).expand(n, r, 4)
    s = ray_starts.unsqueeze(0).expand(n, r, 3)
    d = ray_dirs.unsqueeze(0).expand(n, r, 3)
    if attach_yaw_only:
        starts_w = M.quat_apply_yaw(q, s) + pos_w.unsqueeze(1)
        dirs_w = d.clone()
    else:
        starts_w = M.quat_apply(q, s) + pos_w.unsqueeze(1)
        dirs_w = M.quat_apply(q, d)
    return starts_w, dirs_w


# ---- intersection (unpinned; definition of record) -------------------------------------------------
def raycast_brute(starts, dirs, verts, tris, max_dist: float = 1e6, chunk: int = 256):
    """Every ray against every triangle with the pinned Moller-Trumbore order.  Returns (hits, t, face)."""
    starts, dirs = starts.reshape(-1, 3).float(), dirs.reshape(-1, 3).float()
    v0 = verts[tris[:, 0].long()]
    e1 = verts[tris[:, 1].long()] - v0
    e2 = verts[tris[:, 2].long()] - v0
    n = starts.shape[0]
    t_out = torch.full((n,), float("inf"))
    f_out = torch.full((n,), -1, dtype=torch.int32)
    md = float(M.f32(max_dist))
    eps = float(M.f32(BARY_EPS))
    one_eps = float(M.f32(1.0) + M.f32(BARY_EPS))
    for lo in range(0, n, chunk):
        o = starts[lo:lo + chunk, None, :]
        d = dirs[lo:lo + chunk, None, :]
        c = lambda a: (a[..., 0], a[..., 1], a[..., 2])  # noqa: E731
        p = M.cross3(c(d), c(e2[None]))
        det = M.dot3(c(e1[None]), p)
        ok = det != 0
        inv = torch.ones(()) / det
        s = (o[..., 0] - v0[None, :, 0], o[..., 1] - v0[None, :, 1], o[..., 2] - v0[None, :, 2])
        u = M.dot3(s, p) * inv
        ok &= (u >= -eps) & (u <= one_eps)
        q = M.cross3(s, c(e1[None]))
        v = M.dot3(c(d), q) * inv
        ok &= (v >= -eps) & ((u + v) <= one_eps)
        t = M.dot3(c(e2[None]), q) * inv
        ok &= (t >= 0) & (t < md)
        t = torch.where(ok, t, torch.full_like(t, float("inf")))
        tmin, _ = t.min(dim=1)
        # lowest face index among the ties
        is_min = (t == tmin[:, None]) & ok
        face = torch.where(is_min.any(dim=1), is_min.float().argmax(dim=1), torch.full((t.shape[0],), -1)).to(torch.int32)
        t_out[lo:lo + chunk] = tmin
        f_out[lo:lo + chunk] = face
    hit = f_out >= 0
    tt = torch.where(hit, t_out, torch.zeros(()))
    hits = starts + tt[:, None] * dirs
    hits = torch.where(hit[:, None], hits, torch.full_like(hits, float("inf")))
    return hits, t_out, f_out


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libraycast_ref.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        P, I64, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
        L.rc_brute.argtypes = [P, P, I64, P, P, I64, F, P, P, P]
        L.rc_brute.restype = None
        L.rc_bvh_build.argtypes = [P, P, I64]
        L.rc_bvh_build.restype = P
        L.rc_bvh_free.argtypes = [P]
        L.rc_bvh_free.restype = None
        L.rc_bvh_cast.argtypes = [P, P, P, I64, F, P, P, P]
        L.rc_bvh_cast.restype = None
        _LIB = L
    return _LIB


class CMesh:
    """Mesh + binary BVH owned by the C oracle."""

    def __init__(self, verts, tris):
        self.verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float32))
        self.tris = np.ascontiguousarray(np.asarray(tris, dtype=np.int32))
        self._bvh = None

    def bvh(self):
        if self._bvh is None:
            self._bvh = _lib().rc_bvh_build(self.verts.ctypes.data, self.tris.ctypes.data, self.tris.shape[0])
        return self._bvh

    def __del__(self):
        if getattr(self, "_bvh", None) and _LIB is not None:
            _LIB.rc_bvh_free(self._bvh)
            self._bvh = None


def raycast_c(starts, dirs, mesh: CMesh, max_dist: float = 1e6, brute: bool = False):
    """C oracle.  Returns torch tensors (hits (n,3), t (n), face (n) int32)."""
    s = np.ascontiguousarray(starts.reshape(-1, 3).numpy().astype(np.float32))
    d = np.ascontiguousarray(dirs.reshape(-1, 3).numpy().astype(np.float32))
    n = s.shape[0]
    hits = np.empty((n, 3), np.float32)
    dist = np.empty(n, np.float32)
    face = np.empty(n, np.int32)
    if brute:
        _lib().rc_brute(mesh.verts.ctypes.data, mesh.tris.ctypes.data, mesh.tris.shape[0], s.ctypes.data, d.ctypes.data,
                        n, max_dist, hits.ctypes.data, dist.ctypes.data, face.ctypes.data)
    else:
        _lib().rc_bvh_cast(mesh.bvh(), s.ctypes.data, d.ctypes.data, n, max_dist, hits.ctypes.data, dist.ctypes.data,
                           face.ctypes.data)
    return torch.from_numpy(hits), torch.from_numpy(dist), torch.from_numpy(face)


def height_scan(pos_w, ray_hits_w, offset: float = 0.5):
    """observations.py:165-173."""
    return pos_w[:, 2].unsqueeze(1) - ray_hits_w[..., 2] - M.f32(offset)
