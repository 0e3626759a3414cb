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

}  // namespace ilx
