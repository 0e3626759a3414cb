// Ray / triangle intersection with a pinned evaluation order (Moller-Trumbore, two-sided).
//
// The reference delegates the intersection to warp's mesh_query_ray (isaaclab/utils/warp/kernels.py:66),
// whose source is not part of the reference checkout; the documented contract (kernels.py:27-75,
// ops.py:47-58) is: closest hit with 0 <= t < max_dist, hit = start + t * dir, misses stay +inf / -1.
// This header defines the formula the HIP kernels use; oracle/raycast.py and oracle/raycast_ref.c restate
// it operation for operation, so hit distance and face index agree bit for bit.  Ties in t (shared edges)
// go to the lowest face index.
#pragma once

#include "math_dev.h"

namespace ilx {

// Barycentric slack: a ray through a shared edge must not fall into a rounding crack between two faces.
constexpr float kBaryEps = 1.0e-6f;

// Triangle given as (v0, e1 = v1 - v0, e2 = v2 - v0).  Returns true and t when the ray hits.
ILX_DEV bool ray_triangle(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float max_dist, float& t_out) {
    const f3 p = cross3(d, e2);
    const float det = dot3(e1, p);
    if (!(det != 0.0f)) return false;  // parallel (or NaN)
    const float inv = div_(1.0f, det);
    const f3 s = {o.x - v0.x, o.y - v0.y, o.z - v0.z};
    const float u = dot3(s, p) * inv;
    if (!(u >= -kBaryEps && u <= 1.0f + kBaryEps)) return false;
    const f3 q = cross3(s, e1);
    const float v = dot3(d, q) * inv;
    if (!(v >= -kBaryEps && (u + v) <= 1.0f + kBaryEps)) return false;
    const float t = dot3(e2, q) * inv;
    if (!(t >= 0.0f && t < max_dist)) return false;
    t_out = t;
    return true;
}

// The same test for a ray parallel to the z axis (d.x == d.y == 0), split so that the barycentric decision needs
// only the xy parts of the triangle: with those two components zero, p.z, the z terms of det / u and the x, y terms
// of v are products with zero, so every value computed below is the value ray_triangle computes (a zero may carry
// the other sign, which no comparison and no output can tell).  Stage 1 rejects or yields (qz, inv); stage 2 needs
// the z parts and produces t.
struct ColumnHit {
    float sx, sy, qz, inv;
};
ILX_DEV bool ray_triangle_z_inside(float ox, float oy, float dz, float v0x, float v0y, float e1x, float e1y, float e2x,
                                   float e2y, ColumnHit& h) {
    const float px = -(dz * e2y), py = dz * e2x;
    const float det = e1x * px + e1y * py;
    if (!(det != 0.0f)) return false;
    const float inv = div_(1.0f, det);
    const float sx = ox - v0x, sy = oy - v0y;
    const float u = (sx * px + sy * py) * inv;
    if (!(u >= -kBaryEps && u <= 1.0f + kBaryEps)) return false;
    const float qz = fmaf_(sx, e1y, -(sy * e1x));
    const float v = (dz * qz) * inv;
    if (!(v >= -kBaryEps && (u + v) <= 1.0f + kBaryEps)) return false;
    h = ColumnHit{sx, sy, qz, inv};
    return true;
}
ILX_DEV bool ray_triangle_z_distance(const ColumnHit& h, float oz, float v0z, float e1x, float e1y, float e1z, float e2x,
                                     float e2y, float e2z, float max_dist, float& t_out) {
    const float sz = oz - v0z;
    const float qx = fmaf_(h.sy, e1z, -(sz * e1y)), qy = fmaf_(sz, e1x, -(h.sx * e1z));
    const float t = ((e2x * qx + e2y * qy) + e2z * h.qz) * h.inv;
    if (!(t >= 0.0f && t < max_dist)) return false;
    t_out = t;
    return true;
}

}  // namespace ilx
