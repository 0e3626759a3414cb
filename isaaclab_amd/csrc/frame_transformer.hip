// FrameTransformer sensor (reference: isaaclab/sensors/frame_transformer/frame_transformer.py:322-385): world poses of a
// source frame and of F target frames per env, and the targets expressed in the source frame.  One thread per
// (env, target frame); the gathers of the reference (view order -> env order -> body -> frame) arrive as one row index.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

ILX_DEV void view_pose(const float* transforms, int64_t row, f3& p, f4& q) {
    const float* t = transforms + 7 * row;
    p = {t[0], t[1], t[2]};
    q = {t[6], t[3], t[4], t[5]};  // the physics view hands out x, y, z, w
}

__global__ void __launch_bounds__(256) frame_transformer_kernel(ilx_frame_transformer s) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= s.num_envs * s.num_frames) return;
    const int64_t env = i / s.num_frames;
    f3 sp, tp;
    f4 sq, tq;
    view_pose(s.transforms, s.source_rows[env], sp, sq);
    if (s.source_offset_pos) {  // frame_transformer.py:341-347
        f3 p;
        f4 q;
        combine_frame_transforms(sp, sq, load3(s.source_offset_pos + 3 * env), load4u(s.source_offset_quat + 4 * env), p, q);
        sp = p, sq = q;
    }
    view_pose(s.transforms, s.target_rows[i], tp, tq);
    if (s.target_offset_pos) {  // :358-364
        f3 p;
        f4 q;
        combine_frame_transforms(tp, tq, load3(s.target_offset_pos + 3 * i), load4u(s.target_offset_quat + 4 * i), p, q);
        tp = p, tq = q;
    }
    f3 rp;
    f4 rq;
    subtract_frame_transforms(sp, sq, tp, tq, rp, rq);  // :371-376
    if (i == env * s.num_frames) {
        store3(s.source_pos_w + 3 * env, sp);
        store4u(s.source_quat_w + 4 * env, sq);
    }
    store3(s.target_pos_w + 3 * i, tp);
    store4u(s.target_quat_w + 4 * i, tq);
    store3(s.target_pos_source + 3 * i, rp);
    store4u(s.target_quat_source + 4 * i, rq);
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_frame_transformer_update(const ilx_frame_transformer* sensor, void* stream) {
    ILX_REQUIRE(sensor, ILX_EINVAL, "ilx_frame_transformer_update: descriptor is NULL");
    const ilx_frame_transformer& s = *sensor;
    ILX_REQUIRE(s.num_envs >= 0 && s.num_frames >= 0, ILX_EINVAL, "ilx_frame_transformer_update: negative shape");
    if (s.num_envs == 0 || s.num_frames == 0) return ILX_OK;
    ILX_REQUIRE(s.transforms && s.source_rows && s.target_rows, ILX_EINVAL, "ilx_frame_transformer_update: the view's transforms or the row indices are NULL");
    ILX_REQUIRE((s.source_offset_pos != nullptr) == (s.source_offset_quat != nullptr) &&
                    (s.target_offset_pos != nullptr) == (s.target_offset_quat != nullptr),
                ILX_EINVAL, "ilx_frame_transformer_update: an offset needs both its position and its orientation");
    ILX_REQUIRE(s.source_pos_w && s.source_quat_w && s.target_pos_w && s.target_quat_w && s.target_pos_source && s.target_quat_source,
                ILX_EINVAL, "ilx_frame_transformer_update: NULL data buffer");
    ILX_LAUNCH(frame_transformer_kernel, dim3(blocks_for(s.num_envs * s.num_frames)), dim3(256), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_frame_transformer_update");
    return ILX_OK;
}
