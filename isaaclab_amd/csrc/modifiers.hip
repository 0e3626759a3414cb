// Observation modifiers (reference: isaaclab/utils/modifiers/modifier.py): scale, clip, bias, DigitalFilter, Integrator,
// plus a strided copy.  One thread per element of an (n, d) block whose rows may be strided (a column range of a wider
// observation row); the filter and integrator states are dense, (n, d, taps) and (n, d).
#include "common.h"

#include "../../include/isaaclab_amd.h"

namespace ilx {

struct ModifierArgs {
    ilx_modifier m;
    int kind;
};

__global__ void __launch_bounds__(256) modifier_kernel(ModifierArgs a) {
    const ilx_modifier& m = a.m;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= m.n * m.d) return;
    const int64_t row = i / m.d, col = i - row * m.d;
    const float x = m.in[row * m.in_stride + col];
    float y = x;
    switch (a.kind) {
        case ILX_MOD_COPY: break;
        case ILX_MOD_SCALE: y = x * m.p0; break;   // modifier.py:22-32
        case ILX_MOD_CLIP:                           // modifier.py:35-46; a NaN input stays NaN as in torch.clip
            if (m.has_lo) y = (y < m.p0) ? m.p0 : y;
            if (m.has_hi) y = (y > m.p1) ? m.p1 : y;
            break;
        case ILX_MOD_BIAS: y = x + m.p0; break;     // modifier.py:49-59
        case ILX_MOD_DIGITAL_FILTER: {               // modifier.py:170-191
            float* xs = m.x_n + i * m.nb;
            float* ys = m.y_n + i * m.na;
            float hx[ILX_MOD_MAX_TAPS], hy[ILX_MOD_MAX_TAPS];
#pragma unroll
            for (int j = 0; j < ILX_MOD_MAX_TAPS; ++j) {
                hx[j] = (j >= 1 && j < m.nb) ? xs[j - 1] : 0.0f;   // the history moved one slot towards the past
                hy[j] = (j < m.na) ? ys[j] : 0.0f;
            }
            hx[0] = x;
            float fx = 0.0f, fy = 0.0f;
#pragma unroll
            for (int j = 0; j < ILX_MOD_MAX_TAPS; ++j) {
                if (j < m.nb) fx += hx[j] * m.B[j];
                if (j < m.na) fy += hy[j] * m.A[j];
            }
            y = fx - fy;
#pragma unroll
            for (int j = 0; j < ILX_MOD_MAX_TAPS; ++j) {
                if (j < m.nb) xs[j] = hx[j];
                if (j < m.na) ys[j] = (j == 0) ? y : hy[j - 1];
            }
            break;
        }
        case ILX_MOD_INTEGRATOR: {                   // modifier.py:245-259
            const float prev = m.y_n[i];
            y = m.x_n[i] + (x + prev) * 0.5f * m.p0;  // halving is exact
            m.x_n[i] = y;
            m.y_n[i] = x;
            break;
        }
    }
    m.out[row * m.out_stride + col] = y;
}

// zero the state rows of the listed envs (DigitalFilter.reset / Integrator.reset, modifier.py:157-168, 232-243)
__global__ void __launch_bounds__(256) modifier_reset_kernel(float* a, int64_t la, float* b, int64_t lb, const int64_t* env_ids, int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t per = la + lb;
    if (i >= count * per) return;
    const int64_t k = i / per, j = i - k * per;
    const int64_t env = env_ids ? env_ids[k] : k;
    if (j < la) a[env * la + j] = 0.0f;
    else b[env * lb + (j - la)] = 0.0f;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_modifier_apply(int32_t kind, const ilx_modifier* mod, void* stream) {
    ILX_REQUIRE(mod, ILX_EINVAL, "ilx_modifier_apply: descriptor is NULL");
    const ilx_modifier& m = *mod;
    ILX_REQUIRE(kind >= ILX_MOD_COPY && kind <= ILX_MOD_INTEGRATOR, ILX_EINVAL, "ilx_modifier_apply: unknown kind %d", kind);
    ILX_REQUIRE(m.n >= 0 && m.d >= 0, ILX_EINVAL, "ilx_modifier_apply: negative shape");
    if (m.n == 0 || m.d == 0) return ILX_OK;
    ILX_REQUIRE(m.in && m.out, ILX_EINVAL, "ilx_modifier_apply: in or out is NULL");
    ILX_REQUIRE(m.in_stride >= m.d && m.out_stride >= m.d, ILX_ESHAPE, "ilx_modifier_apply: row strides (%lld, %lld) below the width %d",
                static_cast<long long>(m.in_stride), static_cast<long long>(m.out_stride), m.d);
    if (kind == ILX_MOD_DIGITAL_FILTER) {
        ILX_REQUIRE(m.nb >= 1 && m.nb <= ILX_MOD_MAX_TAPS && m.na >= 1 && m.na <= ILX_MOD_MAX_TAPS, ILX_ESHAPE,
                    "ilx_modifier_apply: filter lengths A=%d, B=%d outside 1..%d", m.na, m.nb, ILX_MOD_MAX_TAPS);
        ILX_REQUIRE(m.x_n && m.y_n, ILX_EINVAL, "ilx_modifier_apply: the filter needs its input and output histories");
    }
    if (kind == ILX_MOD_INTEGRATOR)
        ILX_REQUIRE(m.x_n && m.y_n, ILX_EINVAL, "ilx_modifier_apply: the integrator needs its integral and previous-value buffers");
    ModifierArgs a{m, kind};
    ILX_LAUNCH(modifier_kernel, dim3(blocks_for(m.n * m.d)), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_modifier_apply");
    return ILX_OK;
}

extern "C" int ilx_modifier_reset(float* state_a, int64_t len_a, float* state_b, int64_t len_b, const int64_t* env_ids, int64_t count,
                                  void* stream) {
    ILX_REQUIRE(count >= 0 && len_a >= 0 && len_b >= 0, ILX_EINVAL, "ilx_modifier_reset: negative size");
    if (count == 0 || len_a + len_b == 0) return ILX_OK;
    ILX_REQUIRE((state_a || len_a == 0) && (state_b || len_b == 0), ILX_EINVAL, "ilx_modifier_reset: NULL state buffer");
    ILX_LAUNCH(modifier_reset_kernel, dim3(blocks_for(count * (len_a + len_b))), dim3(256), 0, as_stream(stream), state_a, len_a,
                       state_b, len_b, env_ids, count);
    ILX_LAUNCH_CHECK("ilx_modifier_reset");
    return ILX_OK;
}
