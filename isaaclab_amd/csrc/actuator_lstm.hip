// ActuatorNetLSTM (reference: isaaclab/actuators/actuator_net.py:30-99): per (env, joint) row one time step of a small
// multi-layer LSTM over (position error, velocity), a Linear-stack head on the last layer's output, DC-motor clip.
//
// One lane per row.  The recurrent state (L, N*J, H) is the traffic: 2 * L * H floats read and written per row, as
// whole 16-byte words per lane (H contiguous floats per row and layer, so a wave reads one contiguous span).  The
// weights are staged once per block into LDS, one padded record per gate row (b_ih, b_hh, W_ih row, W_hh row), and read
// back as wave-uniform 16-byte broadcasts; a lane keeps one layer's input, state and the four gates of the unit it is
// working on in registers.  Hidden sizes up to 16, four layers, head widths up to 32.
#include "common.h"
#include "math_dev.h"
#include "mlp_net.h"

#include <vector>

namespace ilx {

constexpr int kLstmMaxLayers = 4;
constexpr int kLstmMaxHidden = 16;
constexpr int kLstmMaxInput = 4;
constexpr int kHeadMaxWidth = 32;
constexpr int kLstmBlock = 256;

struct LstmDev {
    int num_layers, input_size, hidden;
    int row_floats[kLstmMaxLayers];   // padded floats per gate row of layer l: 2 + in_l + H rounded up to 4
    int layer_offset[kLstmMaxLayers]; // float offset of layer l in `packed`
    int packed_floats;
    const float* packed;
};

struct LstmArgs {
    LstmDev net;
    MlpDev head;
    int head_floats;
    const float* q_des;
    const float* q;
    const float* qd;
    float* hidden;  // (L, rows, H)
    float* cell;
    const float* effort_limit;
    const float* vel_limit;
    int per_env;
    float sat;
    float* computed;
    float* applied;
    int64_t rows;
    int j;
};

ILX_DEV float sigmoid_(float x) { return div_(1.0f, 1.0f + expf(-x)); }

// H contiguous floats of one row, as 16-byte words when H is a multiple of 4.
template <int HP>
ILX_DEV void load_state(const float* p, int h, float (&v)[HP]) {
    if ((h & 3) == 0) {
#pragma unroll
        for (int k = 0; k < HP; k += 4) {
            if (k < h) {
                const float4 w = *reinterpret_cast<const float4*>(p + k);
                v[k] = w.x, v[k + 1] = w.y, v[k + 2] = w.z, v[k + 3] = w.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < HP; ++k)
            if (k < h) v[k] = p[k];
    }
}
template <int HP>
ILX_DEV void store_state(float* p, int h, const float (&v)[HP]) {
    if ((h & 3) == 0) {
#pragma unroll
        for (int k = 0; k < HP; k += 4)
            if (k < h) *reinterpret_cast<float4*>(p + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < HP; ++k)
            if (k < h) p[k] = v[k];
    }
}

// Gate row `g` of a layer against the layer input and the previous hidden state:
// (b_ih + W_ih x) + (b_hh + W_hh h), each sum accumulated left to right.
template <int HP>
ILX_DEV float gate_row(const float* rec, int in_dim, int h, const float (&x)[HP], const float (&hp)[HP]) {
    float a = rec[0], b = rec[1];
    const float* wi = rec + 2;
    const float* wh = wi + in_dim;
#pragma unroll
    for (int k = 0; k < HP; ++k)
        if (k < in_dim) a = fmaf_(wi[k], x[k], a);
#pragma unroll
    for (int k = 0; k < HP; ++k)
        if (k < h) b = fmaf_(wh[k], hp[k], b);
    return a + b;
}

template <int HP, int ACT>
__global__ void __launch_bounds__(kLstmBlock) actuator_lstm_kernel(LstmArgs a) {
    extern __shared__ float lds[];
    float* w_lstm = lds;
    float* w_head = lds + a.net.packed_floats;
    for (int i = threadIdx.x; i < a.net.packed_floats; i += blockDim.x) w_lstm[i] = a.net.packed[i];
    for (int i = threadIdx.x; i < a.head_floats; i += blockDim.x) w_head[i] = a.head.raw[i];
    __syncthreads();
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= a.rows) return;
    const int h = a.net.hidden;
    const float vel = a.qd[row];

    float x[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) x[k] = 0.0f;
    x[0] = a.q_des[row] - a.q[row];  // actuator_net.py:80-81
    x[1] = vel;
    int in_dim = a.net.input_size;
    for (int l = 0; l < a.net.num_layers; ++l) {
        const int64_t at = (static_cast<int64_t>(l) * a.rows + row) * h;
        float hp[HP], cp[HP], hn[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) hp[k] = cp[k] = hn[k] = 0.0f;
        load_state<HP>(a.hidden + at, h, hp);
        load_state<HP>(a.cell + at, h, cp);
        const float* layer = w_lstm + a.net.layer_offset[l];
        const int rf = a.net.row_floats[l];
#pragma unroll
        for (int u = 0; u < HP; ++u) {
            if (u < h) {  // torch.nn.LSTM gate order: input, forget, cell, output
                const float gi = gate_row<HP>(layer + (0 * h + u) * rf, in_dim, h, x, hp);
                const float gf = gate_row<HP>(layer + (1 * h + u) * rf, in_dim, h, x, hp);
                const float gg = gate_row<HP>(layer + (2 * h + u) * rf, in_dim, h, x, hp);
                const float go = gate_row<HP>(layer + (3 * h + u) * rf, in_dim, h, x, hp);
                const float c = sigmoid_(gf) * cp[u] + sigmoid_(gi) * tanhf(gg);
                cp[u] = c;
                hn[u] = sigmoid_(go) * tanhf(c);
            }
        }
        store_state<HP>(a.hidden + at, h, hn);
        store_state<HP>(a.cell + at, h, cp);
#pragma unroll
        for (int k = 0; k < HP; ++k) x[k] = hn[k];
        in_dim = h;
    }

    // head: Linear -> act -> ... -> Linear(.., 1) on the last layer's output
    float cur[kHeadMaxWidth], nxt[kHeadMaxWidth];
#pragma unroll
    for (int k = 0; k < kHeadMaxWidth; ++k) cur[k] = k < HP ? x[k < HP ? k : 0] : 0.0f;
    const float* p = w_head;
    for (int l = 0; l < a.head.num_layers; ++l) {
        const int din = a.head.dims[l], dout = a.head.dims[l + 1];
        const float* w = p;
        const float* b = p + din * dout;
        const bool last = l + 1 == a.head.num_layers;
#pragma unroll
        for (int o = 0; o < kHeadMaxWidth; ++o) {
            float acc = 0.0f;
            if (o < dout) {
                acc = b[o];
#pragma unroll
                for (int i = 0; i < kHeadMaxWidth; ++i)
                    if (i < din) acc = fmaf_(w[o * din + i], cur[i], acc);
                if (!last) acc = activate<ACT>(acc);
            }
            nxt[o] = acc;
        }
#pragma unroll
        for (int k = 0; k < kHeadMaxWidth; ++k) cur[k] = nxt[k];
        p = b + dout;
    }
    const int64_t pi = a.per_env ? row : (row % a.j);
    a.computed[row] = cur[0];  // actuator_net.py:90
    a.applied[row] = dc_motor_clip(cur[0], vel, a.sat, a.vel_limit[pi], a.effort_limit[pi]);
}

struct LstmHost {
    LstmDev dev{};
    float* d_packed = nullptr;
};

}  // namespace ilx

using namespace ilx;

struct ilx_lstm_s {
    LstmHost h;
};

extern "C" int ilx_lstm_create(int num_layers, int input_size, int hidden_size, const float* const* weight_ih_host,
                               const float* const* weight_hh_host, const float* const* bias_ih_host,
                               const float* const* bias_hh_host, ilx_lstm_t* out) {
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_lstm_create: out is NULL");
    *out = nullptr;
    ILX_REQUIRE(num_layers >= 1 && num_layers <= kLstmMaxLayers, ILX_ESHAPE, "ilx_lstm_create: %d layers (1..%d supported)",
                num_layers, kLstmMaxLayers);
    ILX_REQUIRE(hidden_size >= 1 && hidden_size <= kLstmMaxHidden, ILX_ESHAPE,
                "ilx_lstm_create: hidden size %d (1..%d supported)", hidden_size, kLstmMaxHidden);
    ILX_REQUIRE(input_size >= 2 && input_size <= kLstmMaxInput, ILX_ESHAPE,
                "ilx_lstm_create: input size %d (2..%d supported; the actuator feeds position error and velocity)", input_size,
                kLstmMaxInput);
    ILX_REQUIRE(weight_ih_host && weight_hh_host && bias_ih_host && bias_hh_host, ILX_EINVAL, "ilx_lstm_create: NULL argument");
    auto* obj = new ilx_lstm_s();
    LstmDev& d = obj->h.dev;
    d.num_layers = num_layers, d.input_size = input_size, d.hidden = hidden_size;
    std::vector<float> pk;
    for (int l = 0; l < num_layers; ++l) {
        if (!weight_ih_host[l] || !weight_hh_host[l] || !bias_ih_host[l] || !bias_hh_host[l]) {
            delete obj;
            return fail(ILX_EINVAL, "ilx_lstm_create: NULL parameter of layer %d", l);
        }
        const int in_l = l == 0 ? input_size : hidden_size;
        const int rf = (2 + in_l + hidden_size + 3) / 4 * 4;
        d.row_floats[l] = rf;
        d.layer_offset[l] = static_cast<int>(pk.size());
        for (int g = 0; g < 4 * hidden_size; ++g) {  // weight_ih_l: (4H, in), weight_hh_l: (4H, H), rows ordered i, f, g, o
            std::vector<float> rec(rf, 0.0f);
            rec[0] = bias_ih_host[l][g], rec[1] = bias_hh_host[l][g];
            for (int k = 0; k < in_l; ++k) rec[2 + k] = weight_ih_host[l][g * in_l + k];
            for (int k = 0; k < hidden_size; ++k) rec[2 + in_l + k] = weight_hh_host[l][g * hidden_size + k];
            pk.insert(pk.end(), rec.begin(), rec.end());
        }
    }
    d.packed_floats = static_cast<int>(pk.size());
    int rc = check_hip(hipMalloc(&obj->h.d_packed, pk.size() * sizeof(float)), "ilx_lstm_create: hipMalloc");
    if (rc == ILX_OK)
        rc = check_hip(hipMemcpy(obj->h.d_packed, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice),
                       "ilx_lstm_create: hipMemcpy");
    if (rc != ILX_OK) {
        if (obj->h.d_packed) (void)hipFree(obj->h.d_packed);
        delete obj;
        return rc;
    }
    d.packed = obj->h.d_packed;
    *out = obj;
    return ILX_OK;
}

extern "C" void ilx_lstm_destroy(ilx_lstm_t net) {
    if (!net) return;
    if (net->h.d_packed) (void)hipFree(net->h.d_packed);
    delete net;
}

template <int HP>
static void launch_lstm(const LstmArgs& a, size_t lds, hipStream_t s) {
    const dim3 g(blocks_for(a.rows, kLstmBlock)), b(kLstmBlock);
    switch (a.head.act) {
        case ILX_ACT_SOFTSIGN: hipLaunchKernelGGL((actuator_lstm_kernel<HP, ILX_ACT_SOFTSIGN>), g, b, lds, s, a); break;
        case ILX_ACT_ELU: hipLaunchKernelGGL((actuator_lstm_kernel<HP, ILX_ACT_ELU>), g, b, lds, s, a); break;
        case ILX_ACT_TANH: hipLaunchKernelGGL((actuator_lstm_kernel<HP, ILX_ACT_TANH>), g, b, lds, s, a); break;
        case ILX_ACT_RELU: hipLaunchKernelGGL((actuator_lstm_kernel<HP, ILX_ACT_RELU>), g, b, lds, s, a); break;
        default: hipLaunchKernelGGL((actuator_lstm_kernel<HP, ILX_ACT_IDENTITY>), g, b, lds, s, a); break;
    }
}

extern "C" int ilx_actuator_net_lstm(ilx_lstm_t lstm, ilx_mlp_t head, const float* q_des, const float* q, const float* qd,
                                     float* hidden_state, float* cell_state, const float* effort_limit,
                                     const float* velocity_limit, int params_per_env, float saturation_effort, float* computed,
                                     float* applied, int64_t n, int j, void* stream) {
    ILX_REQUIRE(lstm && head, ILX_EINVAL, "ilx_actuator_net_lstm: network handle is NULL");
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_actuator_net_lstm: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(q_des && q && qd && hidden_state && cell_state && effort_limit && velocity_limit && computed && applied,
                ILX_EINVAL, "ilx_actuator_net_lstm: NULL operand");
    const MlpDev& hd = head->h.dev;
    ILX_REQUIRE(hd.dims[0] == lstm->h.dev.hidden, ILX_ESHAPE, "ilx_actuator_net_lstm: head input width %d != hidden size %d",
                hd.dims[0], lstm->h.dev.hidden);
    ILX_REQUIRE(hd.dims[hd.num_layers] == 1, ILX_ESHAPE, "ilx_actuator_net_lstm: head output width must be 1");
    for (int l = 0; l <= hd.num_layers; ++l)
        ILX_REQUIRE(hd.dims[l] <= kHeadMaxWidth, ILX_ESHAPE, "ilx_actuator_net_lstm: head layer width %d (max %d)", hd.dims[l],
                    kHeadMaxWidth);
    ILX_REQUIRE((reinterpret_cast<uintptr_t>(hidden_state) & 15) == 0 && (reinterpret_cast<uintptr_t>(cell_state) & 15) == 0,
                ILX_EINVAL, "ilx_actuator_net_lstm: state buffers must be 16-byte aligned");
    LstmArgs a;
    a.net = lstm->h.dev, a.head = hd, a.head_floats = head->h.raw_floats;
    a.q_des = q_des, a.q = q, a.qd = qd, a.hidden = hidden_state, a.cell = cell_state;
    a.effort_limit = effort_limit, a.vel_limit = velocity_limit, a.per_env = params_per_env, a.sat = saturation_effort;
    a.computed = computed, a.applied = applied, a.rows = n * j, a.j = j;
    if (a.rows == 0) return ILX_OK;
    const size_t lds = (static_cast<size_t>(a.net.packed_floats) + a.head_floats) * sizeof(float);
    ILX_REQUIRE(lds <= 60 * 1024, ILX_ESHAPE, "ilx_actuator_net_lstm: %zu bytes of weights do not fit the LDS stage", lds);
    hipStream_t s = as_stream(stream);
    if (a.net.hidden <= 8)
        launch_lstm<8>(a, lds, s);
    else
        launch_lstm<16>(a, lds, s);
    ILX_LAUNCH_CHECK("ilx_actuator_net_lstm");
    return ILX_OK;
}
