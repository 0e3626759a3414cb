#!/usr/bin/env python3
"""Generate the golden fixtures under ``tests/golden/`` by running the REFERENCE's own Python files.

Run only in the build container (needs ``/root/reference``):

    python tests/golden/gen_golden.py            # all sections
    python tests/golden/gen_golden.py math terms # selected sections

Every fixture is a small ``.npz`` holding seeded inputs and the outputs the reference code produced
for them on this machine's torch CPU build.  The fixtures are data; no reference source is stored.
"""

from __future__ import annotations

import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402

torch.set_num_threads(1)
SECTIONS = {}


def section(fn):
    SECTIONS[fn.__name__] = fn
    return fn


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def rand_quat(n, g, normalized=True):
    q = torch.randn(n, 4, generator=g)
    if normalized:
        q = q / q.norm(dim=-1, keepdim=True)
    return q


# --------------------------------------------------------------------------------------------------
@section
def math_fns():
    """Every rotation / transformation function of utils/math.py used on the path."""
    m = R.bootstrap()["math"]
    g = gen(1001)
    n = 384
    q1 = rand_quat(n, g)
    q2 = rand_quat(n, g)
    qu = rand_quat(n, g, normalized=False) * 1.7  # un-normalised
    v = torch.randn(n, 3, generator=g) * 2.0
    v2 = torch.randn(n, 3, generator=g)
    # adversarial rows: identity, pure-yaw, w<0, zero vector part, tiny rotations
    q1[0] = torch.tensor([1.0, 0, 0, 0])
    q1[1] = torch.tensor([0.0, 0, 0, 1.0])
    q1[2] = torch.tensor([-1.0, 0, 0, 0])
    q1[3] = torch.tensor([math.cos(1e-7), math.sin(1e-7), 0, 0])
    q1[4] = torch.tensor([math.cos(2e-7), 0, math.sin(2e-7), 0])
    q1[5] = -q1[6]
    q2[0] = torch.tensor([1.0, 0, 0, 0])
    q2[1] = q1[1]
    q2[7] = q1[7]  # identical pair -> zero error
    v[0] = 0.0
    out = dict(q1=q1, q2=q2, qu=qu, v=v, v2=v2)
    out["quat_rotate"] = m.quat_rotate(q1, v)
    out["quat_rotate_inverse"] = m.quat_rotate_inverse(q1, v)
    out["quat_rotate_inverse_unnorm"] = m.quat_rotate_inverse(qu, v)
    out["quat_apply"] = m.quat_apply(q1, v)
    out["quat_apply_yaw"] = m.quat_apply_yaw(q1, v)
    out["yaw_quat"] = m.yaw_quat(q1)
    out["quat_mul"] = m.quat_mul(q1, q2)
    out["quat_mul_unnorm"] = m.quat_mul(qu, q2)
    out["quat_conjugate"] = m.quat_conjugate(q1)
    out["quat_inv"] = m.quat_inv(qu)
    out["quat_unique"] = m.quat_unique(q1)
    out["normalize3"] = m.normalize(v)
    out["normalize4"] = m.normalize(qu)
    out["matrix_from_quat"] = m.matrix_from_quat(qu)
    out["quat_box_minus"] = m.quat_box_minus(q1, q2)
    out["axis_angle_from_quat"] = m.axis_angle_from_quat(q1)
    out["quat_error_magnitude"] = m.quat_error_magnitude(q1, q2)
    out["skew"] = m.skew_symmetric_matrix(v)
    angle = (torch.rand(n, generator=g) * 2 - 1) * 3.0
    angle[0] = 0.0
    out["angle"] = angle
    out["quat_from_angle_axis"] = m.quat_from_angle_axis(angle, v2)
    rpy = (torch.rand(n, 3, generator=g) * 2 - 1) * math.pi
    out["rpy"] = rpy
    out["quat_from_euler_xyz"] = m.quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    e = m.euler_xyz_from_quat(q1)
    out["euler_xyz_from_quat"] = torch.stack(e, dim=-1)
    t01, t02, t12 = v, v2, torch.randn(n, 3, generator=g)
    out["t12"] = t12
    p, q = m.combine_frame_transforms(t01, q1, t12, q2)
    out["combine_pos"], out["combine_quat"] = p, q
    p, q = m.subtract_frame_transforms(t01, qu, t02, q2)
    out["subtract_pos"], out["subtract_quat"] = p, q
    p, q = m.subtract_frame_transforms(t01, q1)
    out["subtract_pos_none"], out["subtract_quat_none"] = p, q
    pe, re = m.compute_pose_error(t01, q1, t02, q2, rot_error_type="axis_angle")
    out["pose_err_pos"], out["pose_err_aa"] = pe, re
    pe, re = m.compute_pose_error(t01, qu, t02, q2, rot_error_type="quat")
    out["pose_err_quat"] = re
    delta = torch.randn(n, 6, generator=g) * 0.3
    delta[0, 3:] = 0.0  # zero rotation -> NaN axis masked out
    delta[1, 3:] = torch.tensor([3e-7, 0.0, 0.0])
    out["delta"] = delta
    p, q = m.apply_delta_pose(t01, q1, delta)
    out["delta_pos"], out["delta_quat"] = p, q
    # wrap_to_pi incl. the reference's own table (test_math.py:246-278 values)
    ang = torch.cat([
        (torch.rand(n, generator=g) * 2 - 1) * 20.0,
        torch.tensor([0.0, math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 4 * math.pi, -4 * math.pi,
                      math.pi / 2, -math.pi / 2, 1.5 * math.pi, 7 * math.pi / 4, -1.5 * math.pi, 2 * math.pi]),
    ])
    out["wrap_in"] = ang
    out["wrap_to_pi"] = m.wrap_to_pi(ang)
    lo = -torch.rand(n, 5, generator=g) - 0.1
    hi = torch.rand(n, 5, generator=g) + 0.1
    x = torch.randn(n, 5, generator=g)
    out["st_x"], out["st_lo"], out["st_hi"] = x, lo, hi
    out["scale_transform"] = m.scale_transform(x, lo, hi)
    out["unscale_transform"] = m.unscale_transform(x, lo, hi)
    out["saturate"] = m.saturate(x, lo, hi)
    pts = torch.randn(16, 7, 3, generator=g)
    out["tp_points"] = pts
    out["transform_points"] = m.transform_points(pts, v[:16], qu[:16])
    save("math_fns", **out)


@section
def math_kats():
    """Known-answer tables of the reference's own unit tests (source/isaaclab/test/utils/test_math.py),
    re-evaluated through the reference functions so the stored answers are what the reference returns."""
    m = R.bootstrap()["math"]
    # axis_angle_from_quat table (test_math.py:66-85): quaternions of rotations about unit axes
    quats = torch.tensor([
        [1.0, 0.0, 0.0, 0.0],
        [0.8418536, 0.142006, 0.0, 0.5206887],
        [0.9850375, 0.0995007, 0.0995007, 0.0995007],
        [0.7071068, 0.7071068, 0.0, 0.0],
        [0.0, 1.0, 0.0, 0.0],
        [-0.7071068, 0.0, 0.7071068, 0.0],
        [0.5, 0.5, 0.5, 0.5],
    ])
    out = dict(aa_q=quats, aa_out=m.axis_angle_from_quat(quats))
    # small-angle branch (test_math.py:87-109)
    th = torch.tensor([0.0, 1e-8, 1e-7, 5e-7, 9.9e-7, 1.1e-6, 1e-5])
    qs = torch.stack([torch.cos(th / 2), torch.sin(th / 2), torch.zeros_like(th), torch.zeros_like(th)], dim=-1)
    out["aa_small_q"] = qs
    out["aa_small_out"] = m.axis_angle_from_quat(qs)
    # quat_error_magnitude cases (test_math.py:111-139)
    qa = torch.tensor([[1.0, 0, 0, 0], [1.0, 0, 0, 0], [1.0, 0, 0, 0], [0.7071068, 0.7071068, 0, 0]])
    qb = torch.tensor([[1.0, 0, 0, 0], [0.7071068, 0.7071068, 0, 0], [0.0, 1.0, 0, 0], [0.7071068, -0.7071068, 0, 0]])
    out["qem_a"], out["qem_b"], out["qem_out"] = qa, qb, m.quat_error_magnitude(qa, qb)
    save("math_kats", **out)


# --------------------------------------------------------------------------------------------------
def main(argv):
    if not R.available():
        print("reference checkout not found; nothing generated")
        return 1
    names = argv or list(SECTIONS)
    for n in names:
        print(f"== {n}")
        SECTIONS[n]()
    return 0


if __name__ == "__main__":
    # sections defined in sibling modules register themselves on import
    for extra in ("gen_terms", "gen_actuators", "gen_sensors", "gen_ik", "gen_shadow", "gen_managers", "gen_events"):
        if os.path.exists(os.path.join(HERE, extra + ".py")):
            mod = __import__(extra)
            for k, f in getattr(mod, "SECTIONS", {}).items():
                SECTIONS[k] = f
    sys.exit(main(sys.argv[1:]))
