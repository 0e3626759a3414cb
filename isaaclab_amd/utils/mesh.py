"""Triangle meshes and ray casting on the HIP BVH kernels (interface of ``isaaclab/utils/warp/ops.py``).

``raycast_mesh`` keeps the reference's signature and return convention (``ops.py:24-32``): a tuple
``(hits, distance | None, normal | None, face_id | None)`` with ``inf`` / ``-1`` for missed rays.  Unlike the
reference there is no device-wide synchronisation after the launch; the work is ordered on the current stream.
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _abi


class Mesh:
    """Owns an ``ilx_mesh_t``: the 4-wide BVH and the packed triangles in GPU memory."""

    def __init__(self, points, indices, device="cuda:0"):
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int32).reshape(-1, 3))
        self.device = torch.device(device)
        self.num_vertices, self.num_triangles = pts.shape[0], idx.shape[0]
        self.bounds_min, self.bounds_max = (pts.min(axis=0), pts.max(axis=0)) if pts.shape[0] else (np.zeros(3), np.zeros(3))
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _abi.check(_abi.mesh_create(pts.ctypes.data, pts.shape[0], idx.ctypes.data, idx.shape[0],
                                        ctypes.byref(self.handle)), "ilx_mesh_create")
        nt, nn, depth = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        _abi.check(_abi.mesh_info(self.handle, ctypes.byref(nt), ctypes.byref(nn), ctypes.byref(depth)), "ilx_mesh_info")
        self.num_nodes, self.depth = nn.value, depth.value

    def grid_stats(self) -> dict:
        """How the column grid behind ``RAYS_Z_PARALLEL`` casts encodes its cells (diagnostics)."""
        out = (ctypes.c_int64 * 12)()
        _abi.check(_abi.mesh_grid_stats(self.handle, out), "ilx_mesh_grid_stats")
        names = ("cells", "empty", "run1", "run2", "run3", "list_cells", "lists", "list_words", "candidates", "cells_per_row",
                 "pairs", "triples")
        return dict(zip(names, (int(v) for v in out)))

    def lattice_stats(self) -> dict:
        """Height-field patches recognised in the mesh and how many of their cells serve z-parallel sensor rays from
        the height grid alone (diagnostics)."""
        out = (ctypes.c_int64 * 4)()
        _abi.check(_abi.mesh_lattice_stats(self.handle, out), "ilx_mesh_lattice_stats")
        return dict(zip(("patches", "faces", "cells", "simple_cells"), (int(v) for v in out)))

    def lattice_counter(self, enable: bool = True) -> int:
        """Switch the count of height-grid-served sensor rays on or off and return the count so far (diagnostics;
        synchronises the device)."""
        served = ctypes.c_int64()
        _abi.check(_abi.mesh_lattice_counter(self.handle, int(enable), ctypes.byref(served)), "ilx_mesh_lattice_counter")
        return served.value

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _abi is not None and getattr(_abi, "lib", None) is not None:
            _abi.lib.ilx_mesh_destroy(h)
            self.handle = None


def lattice_probe(points, indices) -> dict:
    """``Mesh.lattice_stats`` for host arrays, computed on the host alone (no device needed)."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
    idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int32).reshape(-1, 3))
    out = (ctypes.c_int64 * 4)()
    _abi.check(_abi.mesh_lattice_probe(pts.ctypes.data, pts.shape[0], idx.ctypes.data, idx.shape[0], out), "ilx_mesh_lattice_probe")
    return dict(zip(("patches", "faces", "cells", "simple_cells"), (int(v) for v in out)))


def convert_to_mesh(points: np.ndarray, indices: np.ndarray, device: str) -> Mesh:
    """ops.py:130-145 (``convert_to_warp_mesh``): vertices (V,3) + flat or (F,3) triangle indices -> mesh handle."""
    return Mesh(points, indices, device)


def raycast_mesh(ray_starts: torch.Tensor, ray_directions: torch.Tensor, mesh: Mesh, max_dist: float = 1e6,
                 return_distance: bool = False, return_normal: bool = False, return_face_id: bool = False,
                 z_parallel_hint: bool = False):
    """ops.py:24-127.  ``z_parallel_hint`` (no counterpart in the reference) tells the library that the directions
    are expected to be (0, 0, +-z); it selects the column-grid kernel and never changes the result."""
    shape = ray_starts.shape
    device = ray_starts.device
    starts = _abi.f32c(ray_starts.to(mesh.device)).view(-1, 3)
    dirs = _abi.f32c(ray_directions.to(mesh.device)).view(-1, 3)
    n = starts.shape[0]
    hits = torch.empty((n, 3), dtype=torch.float32, device=mesh.device)
    dist = torch.empty((n,), dtype=torch.float32, device=mesh.device) if return_distance else None
    normal = torch.empty((n, 3), dtype=torch.float32, device=mesh.device) if return_normal else None
    face = torch.empty((n,), dtype=torch.int32, device=mesh.device) if return_face_id else None
    _abi.check(_abi.raycast_mesh(mesh.handle, _abi.ptr(starts), _abi.ptr(dirs), n, float(max_dist), _abi.ptr(hits),
                                 _abi.ptr(dist), _abi.ptr(normal), _abi.ptr(face),
                                 _abi.RAYS_Z_PARALLEL if z_parallel_hint else 0, _abi.stream()), "ilx_raycast_mesh")
    if return_distance:
        dist = dist.to(device).view(shape[0], shape[1]) if len(shape) == 3 else dist.to(device)
    if return_normal:
        normal = normal.to(device).view(shape)
    if return_face_id:
        face = face.to(device).view(shape[0], shape[1]) if len(shape) == 3 else face.to(device)
    return hits.to(device).view(shape), dist, normal, face
