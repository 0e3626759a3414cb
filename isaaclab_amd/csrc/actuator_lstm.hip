// ActuatorNetLSTM (reference: isaaclab/actuators/actuator_net.py:30-99): per (env, joint) row one time step of a small
// multi-layer LSTM over (position error, velocity), a Linear-stack head on the last layer's output, DC-motor clip.
//
// One lane per row.  The recurrent state (L, N*J, H) is the traffic: 2 * L * H floats read and written per row, as
// whole 16-byte words per lane (H contiguous floats per row and layer, so a wave reads one contiguous span).  The
// weights are wave-uniform: they are read through the scalar cache, one padded record per gate row (b_ih, b_hh, W_ih
// row, W_hh row), and enter the FMAs as scalar operands; a lane keeps one layer's input, state and the four gates of
// the unit it is working on in registers.  Hidden sizes up to 16, four layers, head widths up to 32.
#include "common.h"
#include "math_dev.h"
#include "mlp_net.h"

#include <vector>

namespace ilx {

constexpr int kLstmMaxLayers = 4;
constexpr int kLstmMaxHidden = 16;
constexpr int kLstmInput = 4;      // layer-0 input width the records are padded to (the actuator feeds 2 values)
constexpr int kHeadMaxWidth = 32;
constexpr int kLstmBlock = 256;

// Weight records, one per gate row, every part padded with zeros to whole 16-byte words so that all inner loops
// have compile-time bounds and no guards (a zero-padded hidden unit stays exactly 0 for ever):
//   layer l, gate row g:  (b_ih, b_hh, 0, 0 | W_ih row padded to IN_l | W_hh row padded to HP),  IN_0 = 4, IN_l = HP
// Head records: layer 0 rows are (bias, 0, 0, 0 | W row padded to HP), later layers (bias, 0, 0, 0 | W row padded to 32).
struct LstmDev {
    int num_layers, input_size, hidden, hp;  // hp = hidden size padded to 8 or 16
    int packed_floats;
    const float* packed;
};

struct LstmArgs {
    LstmDev net;
    int head_layers, head_act;
    int head_dims[kMaxLayers + 1];
    const float* head_packed;  // device, head_floats long, hidden padding = net.hp
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

// Gate non-linearities on the fast exponential (v_exp_f32) and reciprocal (1 ulp): absolute error ~1e-7 on values
// in (-1, 1), inside the 1e-5 bar of the parity tests.
ILX_DEV float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
ILX_DEV float tanh_(float x) { return fmaf_(2.0f, sigmoid_(2.0f * x), -1.0f); }

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

// Read-only, wave-uniform data (the weight records): the constant address space makes every read of it a scalar load,
// so a weight reaches the FMA as a scalar operand and costs neither a vector register nor an LDS / vector-memory
// instruction.
typedef const float __attribute__((address_space(4))) * UniformPtr;
ILX_DEV UniformPtr as_uniform(const float* p) { return (UniformPtr)(p); }

// One gate row against the layer input and the previous hidden state: (b_ih + W_ih x) + (b_hh + W_hh h), each sum
// accumulated left to right.
template <int IN, int HP>
ILX_DEV float gate_row(UniformPtr rec, const float (&x)[IN], const float (&hp)[HP]) {
    float a = rec[0], b = rec[1];
#pragma unroll
    for (int k = 0; k < IN; ++k) a = fmaf_(rec[4 + k], x[k], a);
#pragma unroll
    for (int k = 0; k < HP; ++k) b = fmaf_(rec[4 + IN + k], hp[k], b);
    return a + b;
}

// One LSTM layer for this lane's row: reads and rewrites the row's state, returns the new hidden vector in `out`.
template <int IN, int HP>
ILX_DEV void lstm_layer(UniformPtr layer, const float (&x)[IN], float* hidden, float* cell, int h, float (&out)[HP]) {
    constexpr int kRec = 4 + IN + HP;
    float hp[HP], cp[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) hp[k] = cp[k] = 0.0f;
    load_state<HP>(hidden, h, hp);
    load_state<HP>(cell, h, cp);
#pragma unroll
    for (int u = 0; u < HP; ++u) {  // torch.nn.LSTM gate order: input, forget, cell, output
        const float gi = gate_row<IN, HP>(layer + (0 * HP + u) * kRec, x, hp);
        const float gf = gate_row<IN, HP>(layer + (1 * HP + u) * kRec, x, hp);
        const float gg = gate_row<IN, HP>(layer + (2 * HP + u) * kRec, x, hp);
        const float go = gate_row<IN, HP>(layer + (3 * HP + u) * kRec, x, hp);
        const float c = sigmoid_(gf) * cp[u] + sigmoid_(gi) * tanh_(gg);
        cp[u] = c;
        out[u] = sigmoid_(go) * tanh_(c);
    }
    store_state<HP>(hidden, h, out);
    store_state<HP>(cell, h, cp);
}

template <int HP, int ACT>
__global__ void __launch_bounds__(kLstmBlock) actuator_lstm_kernel(LstmArgs a) {
    extern __shared__ float stage[];  // [kHeadMaxWidth][kLstmBlock]: a head layer's outputs, one column per lane
    const UniformPtr w_lstm = as_uniform(a.net.packed), w_head = as_uniform(a.head_packed);
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= a.rows) return;
    const int h = a.net.hidden;
    const float vel = a.qd[row];

    float hv[HP];
    {
        float x[kLstmInput] = {a.q_des[row] - a.q[row], vel, 0.0f, 0.0f};  // actuator_net.py:80-81
        lstm_layer<kLstmInput, HP>(w_lstm, x, a.hidden + row * h, a.cell + row * h, h, hv);
    }
    UniformPtr layer = w_lstm + 4 * HP * (4 + kLstmInput + HP);
    for (int l = 1; l < a.net.num_layers; ++l) {
        const int64_t at = (static_cast<int64_t>(l) * a.rows + row) * h;
        float x[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) x[k] = hv[k];
        lstm_layer<HP, HP>(layer, x, a.hidden + at, a.cell + at, h, hv);
        layer += 4 * HP * (4 + 2 * HP);
    }

    // head: Linear -> act -> ... -> Linear(.., 1) on the last layer's output.  The rows of a layer are walked by a
    // run-time loop; their results go through one LDS column per lane back into registers for the next layer.
    float out = 0.0f;
    {
        const int dout = a.head_dims[1];
        const bool last = a.head_layers == 1;
        for (int o = 0; o < dout; ++o) {
            const UniformPtr rec = w_head + o * (4 + HP);
            float acc = rec[0];
#pragma unroll
            for (int i = 0; i < HP; ++i) acc = fmaf_(rec[4 + i], hv[i], acc);
            if (!last) acc = activate<ACT>(acc);
            stage[o * kLstmBlock + threadIdx.x] = acc;
            out = acc;
        }
    }
    UniformPtr p = w_head + a.head_dims[1] * (4 + HP);
    for (int l = 1; l < a.head_layers; ++l) {
        const int din = a.head_dims[l], dout = a.head_dims[l + 1];
        float cur[kHeadMaxWidth];
#pragma unroll
        for (int i = 0; i < kHeadMaxWidth; ++i) cur[i] = i < din ? stage[i * kLstmBlock + threadIdx.x] : 0.0f;
        const bool last = l + 1 == a.head_layers;
        for (int o = 0; o < dout; ++o) {
            const UniformPtr rec = p + o * (4 + kHeadMaxWidth);
            float acc = rec[0];
#pragma unroll
            for (int i = 0; i < kHeadMaxWidth; ++i) acc = fmaf_(rec[4 + i], cur[i], acc);
            if (!last) acc = activate<ACT>(acc);
            stage[o * kLstmBlock + threadIdx.x] = acc;  // only this lane reads its column: no barrier needed
            out = acc;
        }
        p += dout * (4 + kHeadMaxWidth);
    }
    const int64_t pi = a.per_env ? row : (row % a.j);
    a.computed[row] = out;  // actuator_net.py:90
    a.applied[row] = dc_motor_clip(out, vel, a.sat, a.vel_limit[pi], a.effort_limit[pi]);
}

struct LstmHost {
    LstmDev dev{};
    float* d_packed = nullptr;
    float* d_head = nullptr;
    int head_layers = 0, head_act = 0, head_floats = 0;
    int head_dims[kMaxLayers + 1] = {0};
};

}  // namespace ilx

using namespace ilx;

struct ilx_lstm_s {
    LstmHost h;
};

extern "C" int ilx_lstm_create(int num_layers, int input_size, int hidden_size, const float* const* weight_ih_host,
                               const float* const* weight_hh_host, const float* const* bias_ih_host,
                               const float* const* bias_hh_host, int head_layers, const int32_t* head_dims, int head_activation,
                               const float* const* head_weights_host, const float* const* head_biases_host, ilx_lstm_t* out) {
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_lstm_create: out is NULL");
    *out = nullptr;
    ILX_REQUIRE(num_layers >= 1 && num_layers <= kLstmMaxLayers, ILX_ESHAPE, "ilx_lstm_create: %d layers (1..%d supported)",
                num_layers, kLstmMaxLayers);
    ILX_REQUIRE(hidden_size >= 1 && hidden_size <= kLstmMaxHidden, ILX_ESHAPE,
                "ilx_lstm_create: hidden size %d (1..%d supported)", hidden_size, kLstmMaxHidden);
    ILX_REQUIRE(input_size >= 2 && input_size <= kLstmInput, ILX_ESHAPE,
                "ilx_lstm_create: input size %d (2..%d supported; the actuator feeds position error and velocity)", input_size,
                kLstmInput);
    ILX_REQUIRE(weight_ih_host && weight_hh_host && bias_ih_host && bias_hh_host, ILX_EINVAL, "ilx_lstm_create: NULL argument");
    ILX_REQUIRE(head_layers >= 1 && head_layers <= kMaxLayers && head_dims && head_weights_host && head_biases_host, ILX_EINVAL,
                "ilx_lstm_create: the head needs 1..%d Linear layers", kMaxLayers);
    ILX_REQUIRE(head_activation >= 0 && head_activation <= ILX_ACT_IDENTITY, ILX_EINVAL, "ilx_lstm_create: unknown activation %d",
                head_activation);
    ILX_REQUIRE(head_dims[0] == hidden_size, ILX_ESHAPE, "ilx_lstm_create: head input width %d != hidden size %d", head_dims[0],
                hidden_size);
    ILX_REQUIRE(head_dims[head_layers] == 1, ILX_ESHAPE, "ilx_lstm_create: head output width must be 1 (one torque per joint)");
    for (int l = 1; l <= head_layers; ++l)
        ILX_REQUIRE(head_dims[l] >= 1 && head_dims[l] <= kHeadMaxWidth, ILX_ESHAPE, "ilx_lstm_create: head layer width %d (1..%d)",
                    head_dims[l], kHeadMaxWidth);
    for (int l = 0; l < num_layers; ++l)
        ILX_REQUIRE(weight_ih_host[l] && weight_hh_host[l] && bias_ih_host[l] && bias_hh_host[l], ILX_EINVAL,
                    "ilx_lstm_create: NULL parameter of layer %d", l);
    for (int l = 0; l < head_layers; ++l)
        ILX_REQUIRE(head_weights_host[l] && head_biases_host[l], ILX_EINVAL, "ilx_lstm_create: NULL head parameter of layer %d", l);

    const int H = hidden_size, HP = hidden_size <= 8 ? 8 : 16;
    std::vector<float> pk;
    for (int l = 0; l < num_layers; ++l) {
        const int in_l = l == 0 ? input_size : H, inp = l == 0 ? kLstmInput : HP;
        for (int gate = 0; gate < 4; ++gate)
            for (int u = 0; u < HP; ++u) {  // weight_ih_l: (4H, in), weight_hh_l: (4H, H), rows ordered i, f, g, o
                std::vector<float> rec(4 + inp + HP, 0.0f);
                if (u < H) {
                    const int g = gate * H + u;
                    rec[0] = bias_ih_host[l][g], rec[1] = bias_hh_host[l][g];
                    for (int k = 0; k < in_l; ++k) rec[4 + k] = weight_ih_host[l][g * in_l + k];
                    for (int k = 0; k < H; ++k) rec[4 + inp + k] = weight_hh_host[l][g * H + k];
                }
                pk.insert(pk.end(), rec.begin(), rec.end());
            }
    }
    std::vector<float> hd;
    for (int l = 0; l < head_layers; ++l) {
        const int din = head_dims[l], dout = head_dims[l + 1], pad = l == 0 ? HP : kHeadMaxWidth;
        for (int o = 0; o < dout; ++o) {  // torch.nn.Linear: weight (out, in)
            std::vector<float> rec(4 + pad, 0.0f);
            rec[0] = head_biases_host[l][o];
            for (int i = 0; i < din; ++i) rec[4 + i] = head_weights_host[l][o * din + i];
            hd.insert(hd.end(), rec.begin(), rec.end());
        }
    }
    auto* obj = new ilx_lstm_s();
    LstmHost& h = obj->h;
    h.dev.num_layers = num_layers, h.dev.input_size = input_size, h.dev.hidden = H, h.dev.hp = HP;
    h.dev.packed_floats = static_cast<int>(pk.size());
    h.head_layers = head_layers, h.head_act = head_activation, h.head_floats = static_cast<int>(hd.size());
    for (int l = 0; l <= head_layers; ++l) h.head_dims[l] = head_dims[l];
    int rc = check_hip(hipMalloc(&h.d_packed, pk.size() * sizeof(float)), "ilx_lstm_create: hipMalloc");
    if (rc == ILX_OK) rc = check_hip(hipMalloc(&h.d_head, hd.size() * sizeof(float)), "ilx_lstm_create: hipMalloc");
    if (rc == ILX_OK)
        rc = check_hip(hipMemcpy(h.d_packed, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice), "ilx_lstm_create: hipMemcpy");
    if (rc == ILX_OK)
        rc = check_hip(hipMemcpy(h.d_head, hd.data(), hd.size() * sizeof(float), hipMemcpyHostToDevice), "ilx_lstm_create: hipMemcpy");
    if (rc != ILX_OK) {
        if (h.d_packed) (void)hipFree(h.d_packed);
        if (h.d_head) (void)hipFree(h.d_head);
        delete obj;
        return rc;
    }
    h.dev.packed = h.d_packed;
    *out = obj;
    return ILX_OK;
}

extern "C" void ilx_lstm_destroy(ilx_lstm_t net) {
    if (!net) return;
    if (net->h.d_packed) (void)hipFree(net->h.d_packed);
    if (net->h.d_head) (void)hipFree(net->h.d_head);
    delete net;
}

template <int HP>
static void launch_lstm(const LstmArgs& a, size_t lds, hipStream_t s) {
    const dim3 g(blocks_for(a.rows, kLstmBlock)), b(kLstmBlock);
    switch (a.head_act) {
        case ILX_ACT_SOFTSIGN: ILX_LAUNCH((actuator_lstm_kernel<HP, ILX_ACT_SOFTSIGN>), g, b, lds, s, a); break;
        case ILX_ACT_ELU: ILX_LAUNCH((actuator_lstm_kernel<HP, ILX_ACT_ELU>), g, b, lds, s, a); break;
        case ILX_ACT_TANH: ILX_LAUNCH((actuator_lstm_kernel<HP, ILX_ACT_TANH>), g, b, lds, s, a); break;
        case ILX_ACT_RELU: ILX_LAUNCH((actuator_lstm_kernel<HP, ILX_ACT_RELU>), g, b, lds, s, a); break;
        default: ILX_LAUNCH((actuator_lstm_kernel<HP, ILX_ACT_IDENTITY>), g, b, lds, s, a); break;
    }
}

extern "C" int ilx_actuator_net_lstm(ilx_lstm_t net, const float* q_des, const float* q, const float* qd, float* hidden_state,
                                     float* cell_state, const float* effort_limit, const float* velocity_limit,
                                     int params_per_env, float saturation_effort, float* computed, float* applied, int64_t n,
                                     int j, void* stream) {
    ILX_REQUIRE(net, ILX_EINVAL, "ilx_actuator_net_lstm: network handle is NULL");
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_actuator_net_lstm: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(q_des && q && qd && hidden_state && cell_state && effort_limit && velocity_limit && computed && applied,
                ILX_EINVAL, "ilx_actuator_net_lstm: NULL operand");
    ILX_REQUIRE((reinterpret_cast<uintptr_t>(hidden_state) & 15) == 0 && (reinterpret_cast<uintptr_t>(cell_state) & 15) == 0,
                ILX_EINVAL, "ilx_actuator_net_lstm: state buffers must be 16-byte aligned");
    const LstmHost& h = net->h;
    LstmArgs a;
    a.net = h.dev, a.head_layers = h.head_layers, a.head_act = h.head_act, a.head_packed = h.d_head, a.head_floats = h.head_floats;
    for (int l = 0; l <= kMaxLayers; ++l) a.head_dims[l] = h.head_dims[l];
    a.q_des = q_des, a.q = q, a.qd = qd, a.hidden = hidden_state, a.cell = cell_state;
    a.effort_limit = effort_limit, a.vel_limit = velocity_limit, a.per_env = params_per_env, a.sat = saturation_effort;
    a.computed = computed, a.applied = applied, a.rows = n * j, a.j = j;
    if (a.rows == 0) return ILX_OK;
    const size_t lds = static_cast<size_t>(kHeadMaxWidth) * kLstmBlock * sizeof(float);
    hipStream_t s = as_stream(stream);
    if (a.net.hp == 8)
        launch_lstm<8>(a, lds, s);
    else
        launch_lstm<16>(a, lds, s);
    ILX_LAUNCH_CHECK("ilx_actuator_net_lstm");
    return ILX_OK;
}
