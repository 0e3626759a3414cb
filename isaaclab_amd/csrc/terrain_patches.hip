// Flat-patch search on a terrain mesh (reference: isaaclab/terrains/utils.py:135-273): rejection sampling of patch
// centres whose ring of downward rays hits heights inside a z range and within a maximum spread.  Two small kernels
// around the ray cast of each round: ray set-up and the validity test.
#include "common.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

// utils.py:236-244: centre (x, y) into the patch row, ray starts = centre + ring offset at z = 100, directions straight down
__global__ void __launch_bounds__(256) flat_patch_rays_kernel(const float* cx, const float* cy, const float* ring, const int64_t* ids,
                                                              int64_t k, int q, float* flat_patches, float* starts, float* dirs) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= k * q) return;
    const int64_t p = i / q;
    const int r = static_cast<int>(i - p * q);
    const float x = cx[p], y = cy[p];
    if (r == 0) flat_patches[3 * ids[p]] = x, flat_patches[3 * ids[p] + 1] = y;
    starts[3 * i] = x + ring[3 * r], starts[3 * i + 1] = y + ring[3 * r + 1], starts[3 * i + 2] = 100.0f;
    dirs[3 * i] = 0.0f, dirs[3 * i + 1] = 0.0f, dirs[3 * i + 2] = -1.0f;
}

// utils.py:248-258: height of the last ring point into the patch row; invalid when any height leaves [z_lo, z_hi] or
// max - min exceeds the allowed spread (a missed ray has height +inf and fails the range test)
__global__ void __launch_bounds__(256) flat_patch_check_kernel(const float* hits, const int64_t* ids, int64_t k, int q, float z_lo,
                                                               float z_hi, float max_height_diff, float* flat_patches,
                                                               uint8_t* not_valid) {
    const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= k) return;
    const float* h = hits + 3 * p * q + 2;
    float lo = h[0], hi = h[0];
    bool bad = false;
    for (int r = 0; r < q; ++r) {
        const float z = h[3 * r];
        bad = bad || z < z_lo || z > z_hi;
        lo = fminf(lo, z), hi = fmaxf(hi, z);
    }
    flat_patches[3 * ids[p] + 2] = h[3 * (q - 1)];
    not_valid[p] = bad || (hi - lo) > max_height_diff;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_flat_patch_rays(const float* centre_x, const float* centre_y, const float* ring, const int64_t* patch_ids, int64_t count,
                                   int32_t ring_points, float* flat_patches, float* ray_starts, float* ray_directions, void* stream) {
    ILX_REQUIRE(count >= 0 && ring_points > 0, ILX_EINVAL, "ilx_flat_patch_rays: bad shape (%lld, %d)", static_cast<long long>(count), ring_points);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(centre_x && centre_y && ring && patch_ids && flat_patches && ray_starts && ray_directions, ILX_EINVAL,
                "ilx_flat_patch_rays: NULL operand");
    ILX_LAUNCH(flat_patch_rays_kernel, dim3(blocks_for(count * ring_points)), dim3(256), 0, as_stream(stream), centre_x, centre_y,
                       ring, patch_ids, count, ring_points, flat_patches, ray_starts, ray_directions);
    ILX_LAUNCH_CHECK("ilx_flat_patch_rays");
    return ILX_OK;
}

extern "C" int ilx_flat_patch_check(const float* ray_hits, const int64_t* patch_ids, int64_t count, int32_t ring_points, float z_lo,
                                    float z_hi, float max_height_diff, float* flat_patches, uint8_t* not_valid, void* stream) {
    ILX_REQUIRE(count >= 0 && ring_points > 0, ILX_EINVAL, "ilx_flat_patch_check: bad shape (%lld, %d)", static_cast<long long>(count), ring_points);
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(ray_hits && patch_ids && flat_patches && not_valid, ILX_EINVAL, "ilx_flat_patch_check: NULL operand");
    ILX_LAUNCH(flat_patch_check_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), ray_hits, patch_ids, count,
                       ring_points, z_lo, z_hi, max_height_diff, flat_patches, not_valid);
    ILX_LAUNCH_CHECK("ilx_flat_patch_check");
    return ILX_OK;
}
