// Definitions shared by the actuator-network kernels: the Linear-stack handle behind ilx_mlp_t, its activations and
// the DC-motor clip (reference: isaaclab/actuators/actuator_net.py, actuator_pd.py:223-233).
#pragma once

#include "common.h"
#include "math_dev.h"

namespace ilx {

constexpr int kMaxLayers = 8;
constexpr int kMaxWidth = 64;

template <int ACT>
ILX_DEV float activate(float x) {
    if (ACT == ILX_ACT_SOFTSIGN) return div_(x, 1.0f + fabsf(x));
    if (ACT == ILX_ACT_ELU) return x > 0.0f ? x : (expf(x) - 1.0f);
    if (ACT == ILX_ACT_TANH) return tanhf(x);
    if (ACT == ILX_ACT_RELU) return fmaxf(x, 0.0f);
    return x;
}

// actuator_pd.py:223-233
ILX_DEV float dc_motor_clip(float effort, float qd, float sat, float vel_limit, float effort_limit) {
    const float ratio = div_(qd, vel_limit);
    float mx = sat * (1.0f - ratio);
    mx = fminf(fmaxf(mx, 0.0f), effort_limit);
    float mn = sat * (-1.0f - ratio);
    mn = fminf(fmaxf(mn, -effort_limit), 0.0f);
    return fminf(fmaxf(effort, mn), mx);
}

struct MlpDev {
    int num_layers;               // linear layers
    int dims[kMaxLayers + 1];     // in, h1, ..., out
    int act;
    const float* raw;             // per layer: W (out x in, row-major) then b (out)
    const float* packed;          // MFMA operand order, or nullptr
    int ks0;                      // k-steps of the first layer (ceil(in/2))
    const float* image;           // the weights as mlp_mfma_tile_kernel keeps them in LDS (with `packed`), or nullptr
    int image_floats;
    const float* image16;         // the image of mlp_mfma_split_kernel (hidden weights as f16 head / remainder), or nullptr
    int image16_floats;
};

struct MlpHost {
    MlpDev dev;
    float* d_raw = nullptr;
    float* d_packed = nullptr;
    float* d_image = nullptr;
    float* d_image16 = nullptr;
    int device = 0;
    int raw_floats = 0;           // length of `raw`
};

}  // namespace ilx

struct ilx_mlp_s {
    ilx::MlpHost h;
};
