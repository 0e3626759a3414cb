// Actuator models (reference: isaaclab/actuators/actuator_pd.py, actuator_net.py).
//
//  * ilx_actuator_pd       IdealPD / Implicit torque estimate / DCMotor: one thread per (env, joint), streaming.
//  * ilx_actuator_net_mlp  ActuatorNetMLP: history roll + per-(env,joint) MLP + DC-motor clip in ONE launch.
//      - fast path: hidden width 32, output width 1 -> chained v_mfma_f32_32x32x2_f32 (exact fp32 MFMA).
//        One wave owns 32 rows (= (env, joint) pairs).  The network is evaluated transposed,
//        H_out^T[unit][row] = W[unit][k] * H_in^T[k][row], so a layer's 32x32 accumulator (row on the lane,
//        units in the 16 registers) is directly the B operand of the next layer's MFMAs: no LDS, no shuffles.
//        Weights are re-packed once at ilx_mlp_create into the per-lane operand order, so every weight load
//        is a coalesced 256-byte read.
//      - generic path: any widths <= 64, one thread per row, activations staged through LDS.
#include "common.h"
#include "math_dev.h"
#include "mlp_net.h"
#include "scene_update_dev.h"
#include "uniform_access.h"

#include <algorithm>
#include <vector>

namespace ilx {

// ------------------------------------------------------------------------------------------------
// PD family
// ------------------------------------------------------------------------------------------------
// `block`: the workgroup's index among the actuator's workgroups (the launch may hold others: pd_scene_kernel).
template <int MODEL>
ILX_DEV void actuator_pd_block(const float* __restrict__ q_des, const float* __restrict__ qd_des, const float* __restrict__ tau_ff,
                               const float* __restrict__ q, const float* __restrict__ qd, const float* __restrict__ kp,
                               const float* __restrict__ kd, const float* __restrict__ effort_limit,
                               const float* __restrict__ vel_limit, int per_env, float sat, float* __restrict__ computed,
                               float* __restrict__ applied, int64_t total, int j, unsigned block) {
    const int64_t i = static_cast<int64_t>(block) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pi = per_env ? i : (i % j);
    const float vel = qd[i];
    // actuator_pd.py:135-138, evaluated left to right
    const float e_pos = q_des[i] - q[i];
    const float e_vel = (qd_des ? qd_des[i] : 0.0f) - vel;
    const float tau = kp[pi] * e_pos + kd[pi] * e_vel + (tau_ff ? tau_ff[i] : 0.0f);
    const float lim = effort_limit[pi];
    float out;
    if (MODEL == 1) {
        out = dc_motor_clip(tau, vel, sat, vel_limit[pi], lim);
    } else {
        out = fminf(fmaxf(tau, -lim), lim);  // actuator_base.py:252-261
    }
    computed[i] = tau;
    applied[i] = out;
}

template <int MODEL>
__global__ void __launch_bounds__(256)
actuator_pd_kernel(const float* __restrict__ q_des, const float* __restrict__ qd_des, const float* __restrict__ tau_ff,
                   const float* __restrict__ q, const float* __restrict__ qd, const float* __restrict__ kp,
                   const float* __restrict__ kd, const float* __restrict__ effort_limit,
                   const float* __restrict__ vel_limit, int per_env, float sat, float* __restrict__ computed,
                   float* __restrict__ applied, int64_t total, int j) {
    actuator_pd_block<MODEL>(q_des, qd_des, tau_ff, q, qd, kp, kd, effort_limit, vel_limit, per_env, sat, computed, applied, total, j,
                             blockIdx.x);
}

// The PD actuator model of one physics substep and the scene update of the substep before it as workgroups of one grid
// (see mlp_split_scene_kernel below for why the two go together): both are short and bound by the launch's fixed cost.
template <int MODEL>
__global__ void __launch_bounds__(256)
pd_scene_kernel(const float* __restrict__ q_des, const float* __restrict__ qd_des, const float* __restrict__ tau_ff,
                const float* __restrict__ q, const float* __restrict__ qd, const float* __restrict__ kp,
                const float* __restrict__ kd, const float* __restrict__ effort_limit, const float* __restrict__ vel_limit,
                int per_env, float sat, float* __restrict__ computed, float* __restrict__ applied, int64_t total, int j,
                SceneArgs s, unsigned su_blocks) {
    if (blockIdx.x < su_blocks) scene_update_thread(s, blockIdx.x);
    else actuator_pd_block<MODEL>(q_des, qd_des, tau_ff, q, qd, kp, kd, effort_limit, vel_limit, per_env, sat, computed, applied, total,
                                  j, blockIdx.x - su_blocks);
}

// The PD actuator model of an env step's FIRST physics substep together with what precedes it in the step: the action
// manager's bookkeeping and a joint action term with consecutive joints (action_manager.py:331-332, joint_actions.py:130-160,
// 214-225), whose processed action becomes the position or effort target of joints first .. first + width - 1 before
// the model reads it; the first workgroup also opens the step for the fused managers (as joint_action_kernel does).
// A thread owns one (env, joint) and, where the joint is the action term's, the action element that goes with it.
struct PdJointAction {
    const float* raw;      // (n, width)
    float* action;         // (n, width) or null
    float* prev_action;    // (n, width) or null
    const float* offset;   // (n, width) or null
    float scale;
    int target;            // 0: q_des, 1: tau_ff
    int first, width;
    float* log_sums;       // the step's opening (or null / 0 / null)
    int log_count;
    uint64_t* step_index_device;
};
template <int MODEL>
__global__ void __launch_bounds__(256)
pd_joint_action_kernel(PdJointAction ja, float* __restrict__ q_des, const float* __restrict__ qd_des, float* __restrict__ tau_ff,
                       const float* __restrict__ q, const float* __restrict__ qd, const float* __restrict__ kp,
                       const float* __restrict__ kd, const float* __restrict__ effort_limit, const float* __restrict__ vel_limit,
                       int per_env, float sat, float* __restrict__ computed, float* __restrict__ applied, int64_t total, int j) {
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < ja.log_count; k += blockDim.x) ja.log_sums[k] = 0.0f;
        if (ja.step_index_device && threadIdx.x == 0) *ja.step_index_device += 1;
    }
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t env = i / j;
    const int joint = static_cast<int>(i - env * j);
    const int64_t pi = per_env ? i : joint;
    float target_q = q_des[i], target_tau = tau_ff ? tau_ff[i] : 0.0f;
    const int a = joint - ja.first;
    if (a >= 0 && a < ja.width) {
        const int64_t ai = env * ja.width + a;
        const float r = ja.raw[ai];
        if (ja.prev_action) ja.prev_action[ai] = ja.action[ai];
        if (ja.action) ja.action[ai] = r;
        const float v = r * ja.scale + (ja.offset ? ja.offset[ai] : 0.0f);   // joint_actions.py:134-135
        if (ja.target == 0) q_des[i] = v, target_q = v;
        else tau_ff[i] = v, target_tau = v;
    }
    const float vel = qd[i];
    // actuator_pd.py:135-138, evaluated left to right
    const float e_pos = target_q - q[i];
    const float e_vel = (qd_des ? qd_des[i] : 0.0f) - vel;
    const float tau = kp[pi] * e_pos + kd[pi] * e_vel + target_tau;
    const float lim = effort_limit[pi];
    float out;
    if (MODEL == 1) {
        out = dc_motor_clip(tau, vel, sat, vel_limit[pi], lim);
    } else {
        out = fminf(fmaxf(tau, -lim), lim);  // actuator_base.py:252-261
    }
    computed[i] = tau;
    applied[i] = out;
}

// ------------------------------------------------------------------------------------------------
// Delayed / remotized PD (actuator_pd.py:236-363)
// ------------------------------------------------------------------------------------------------
struct DelayArgs {
    const float* set[3];   // q_des, qd_des, tau_ff of this call (the last two may be null: zeros, nothing buffered)
    float* ring[3];        // (len, n, j) rings of the three set-points, one write pointer for all envs
    int len, pointer;      // slots per ring; the slot this call writes
    int64_t* num_pushes;   // (n) pushes since the env's last reset
    const int32_t* time_lags;  // (n)
    const float* q;
    const float* qd;
    const float* kp;
    const float* kd;
    const float* effort_limit;
    int per_env;
    const float* lookup_x;  // remotized: ascending joint angles and the torque limit at each, or null
    const float* lookup_y;
    int lookup_n;
    float* computed;
    float* applied;
    int64_t n;
    int j, envs_per_block;
};

// utils/interpolation/linear_interpolation.py:55-88
ILX_DEV float interpolate_held(const float* x, const float* y, int n, float q) {
    int smaller = 0;
    for (int i = 0; i < n; ++i) smaller += x[i] < q ? 1 : 0;
    const int lo = smaller - 1 < 0 ? 0 : smaller - 1;
    const int hi = smaller > n - 1 ? n - 1 : smaller;
    const float w = hi == lo ? 0.0f : div_(q - x[lo], x[hi] - x[lo]);
    return y[lo] + w * (y[hi] - y[lo]);
}

// LinearInterpolation.compute (linear_interpolation.py:55-88) for arbitrary query points, one thread per point.
__global__ void __launch_bounds__(256) linear_interpolation_kernel(const float* x, const float* y, int n, const float* q, float* out,
                                                                   int64_t count) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) out[i] = interpolate_held(x, y, n, q[i]);
}

// A block owns whole envs (its threads read an env's push count before any of them advances it); one thread per
// (env, joint): append the three set-points to their rings (circular_buffer.py:121-143: the first push after a reset
// fills every slot), read them back time_lag steps late (delay_buffer.py:158-177), then the ideal PD law.
__global__ void __launch_bounds__(256) actuator_delayed_pd_kernel(DelayArgs a) {
    const int e_local = threadIdx.x / a.j, joint = threadIdx.x - e_local * a.j;
    const int64_t env = static_cast<int64_t>(blockIdx.x) * a.envs_per_block + e_local;
    const bool on = e_local < a.envs_per_block && env < a.n;
    int64_t pushes = 0;
    if (on) {
        pushes = a.num_pushes[env];
        const int64_t lag64 = a.time_lags[env] < pushes ? a.time_lags[env] : pushes;  // min(lag, pushes_after - 1)
        int slot = (a.pointer - static_cast<int>(lag64)) % a.len;
        slot = slot < 0 ? slot + a.len : slot;
        const int64_t at = env * a.j + joint, plane = a.n * a.j;
        float late[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            late[k] = 0.0f;
            if (a.set[k]) {
                const float v = a.set[k][at];
                if (pushes == 0) {
                    for (int s = 0; s < a.len; ++s) a.ring[k][s * plane + at] = v;
                    late[k] = v;
                } else {
                    a.ring[k][a.pointer * plane + at] = v;
                    late[k] = slot == a.pointer ? v : a.ring[k][slot * plane + at];
                }
            }
        }
        const int64_t pi = a.per_env ? at : joint;
        const float pos = a.q[at];
        const float tau = a.kp[pi] * (late[0] - pos) + a.kd[pi] * (late[1] - a.qd[at]) + late[2];  // actuator_pd.py:135-138
        const float lim = a.effort_limit[pi];
        float out = fminf(fmaxf(tau, -lim), lim);
        if (a.lookup_x) {  // actuator_pd.py:354-361
            const float cap = interpolate_held(a.lookup_x, a.lookup_y, a.lookup_n, pos);
            out = fminf(fmaxf(out, -cap), cap);
        }
        a.computed[at] = tau;
        a.applied[at] = out;
    }
    __syncthreads();
    if (on && joint == 0) a.num_pushes[env] = pushes + 1;
}

// ------------------------------------------------------------------------------------------------
// MLP
// ------------------------------------------------------------------------------------------------
constexpr int kMaxHist = 8;

struct NetArgs {
    MlpDev net;
    const float* q_des;
    const float* q;
    const float* qd;
    float* pe_hist;
    float* v_hist;
    int hist_len;  // K
    // feature f of the network input (actuator_net.py:160-171: input_idx gather, pos_vel / vel_pos concatenation):
    // history slot feat_slot[f] (0 = the sample of this call) of the position errors (bit f of feat_is_pos) or velocities
    int n_feat;
    int feat_slot[2 * kMaxHist];
    unsigned feat_is_pos;
    float pos_scale, vel_scale, torque_scale;
    const float* effort_limit;
    const float* vel_limit;
    int per_env;
    float sat;
    float* computed;
    float* applied;
    int64_t rows;  // N * J
    int j;
    float inv_j;   // 1.0f / j
    // The first actuator launch of an env step can do the joint action's processing on the way (process_action is the
    // launch just before it, joint_actions.py:130-160 / action_manager.py:318-339): with ja_raw set, q_des is an OUTPUT --
    // prev_action <- action <- raw, q_des <- raw * ja_scale + ja_offset -- and the network runs on that q_des.  The
    // first workgroup also opens the step for the fused managers (begin_*: as ilx_joint_action_process_begin).
    const float* ja_raw;
    float* ja_action;
    float* ja_prev_action;
    const float* ja_offset;
    float* ja_target;
    float ja_scale;
    float* begin_log;
    int begin_count;
    uint64_t* begin_counter;
};

// Reads the row's inputs, forms the network input features, and hands back the old history values.
// feat[f] follows actuator_net.py:160-171 (input_idx gather, scaling, pos_vel / vel_pos concatenation).
// H bounds the history length and the number of input indices the instance can hold.
template <int H>
struct RowInT {
    float pe_new0, vel;    // q_des - q, qd
    float vel_limit, effort_limit;
    float pe_old[H - 1];   // previous slots 0..K-2
    float v_old[H - 1];
    float feat[2 * H];
};
using RowIn = RowInT<kMaxHist>;

// Where a row (one (env, joint) pair) lives.  I = int64_t in general; the resident kernel uses uint32_t when the
// host has checked that every index fits, which keeps 64-bit divisions and address arithmetic off the VALU.
template <class I>
struct RowAt {
    I row, hbase, pidx;  // flat (env, joint) index; index of history slot 0; index of the per-joint parameters
};
template <class I>
ILX_DEV RowAt<I> row_at(const NetArgs& a, I row) {
    const I env = row / static_cast<I>(a.j);
    const I joint = row - env * static_cast<I>(a.j);
    return {row, env * static_cast<I>(a.hist_len) * static_cast<I>(a.j) + joint, a.per_env ? row : joint};
}

// Element `idx` of an array whose pointer is a kernel argument.  With 32-bit indices (the tile kernels: the host has
// checked that every byte offset fits) the address is the scalar base plus the lane's 32-bit byte offset.
template <class I>
ILX_DEV float rd(const float* p, I idx) {
    if constexpr (sizeof(I) == 4) return ld_at(as_global(p), 4u * static_cast<uint32_t>(idx));
    else return p[idx];
}
template <class I>
ILX_DEV void wr(float* p, I idx, float v) {
    if constexpr (sizeof(I) == 4) st_at(as_global(p), 4u * static_cast<uint32_t>(idx), v);
    else p[idx] = v;
}

// Row inputs as they come from memory; issue_row only issues the loads (nothing consumes them), so a caller can
// put a tile's worth of compute between issue_row and finish_row.
template <int H>
struct RowRawT {
    float q_des, q, qd, vel_limit, effort_limit;
    float pe_old[H - 1], v_old[H - 1];
    float slot[2 * H];  // history value feeding feature f when it is not the newest sample
};

template <int H, class I>
ILX_DEV void issue_row(const NetArgs& a, const RowAt<I>& at, bool valid, RowRawT<H>& w, bool feature_slots = true) {
    w.q_des = w.q = w.qd = 0.0f;
    w.vel_limit = w.effort_limit = 1.0f;
    if (valid) {
        w.q_des = rd(a.q_des, at.row), w.q = rd(a.q, at.row), w.qd = rd(a.qd, at.row);
        w.vel_limit = rd(a.vel_limit, at.pidx), w.effort_limit = rd(a.effort_limit, at.pidx);
    }
#pragma unroll
    for (int k = 0; k < H - 1; ++k) {
        w.pe_old[k] = w.v_old[k] = 0.0f;
        if (valid && k < a.hist_len - 1) {
            w.pe_old[k] = rd(a.pe_hist, at.hbase + static_cast<I>(k * a.j));
            w.v_old[k] = rd(a.v_hist, at.hbase + static_cast<I>(k * a.j));
        }
    }
    // which history slot feeds feature f is the same for all rows (resolved on the host): one more (cached) read
#pragma unroll
    for (int f = 0; f < 2 * H; ++f) {
        w.slot[f] = 0.0f;
        if (feature_slots && f < a.n_feat && a.feat_slot[f] > 0 && valid)
            w.slot[f] = rd(((a.feat_is_pos >> f) & 1) ? a.pe_hist : a.v_hist, at.hbase + static_cast<I>((a.feat_slot[f] - 1) * a.j));
    }
}

template <int H>
ILX_DEV void finish_row(const NetArgs& a, const RowRawT<H>& w, RowInT<H>& r) {
    r.pe_new0 = w.q_des - w.q;
    r.vel = w.qd;
    r.vel_limit = w.vel_limit, r.effort_limit = w.effort_limit;
#pragma unroll
    for (int k = 0; k < H - 1; ++k) r.pe_old[k] = w.pe_old[k], r.v_old[k] = w.v_old[k];
#pragma unroll
    for (int f = 0; f < 2 * H; ++f) {
        float val = 0.0f;
        if (f < a.n_feat) {
            const bool pos = (a.feat_is_pos >> f) & 1;
            const float v = a.feat_slot[f] > 0 ? w.slot[f] : (pos ? r.pe_new0 : r.vel);
            val = v * (pos ? a.pos_scale : a.vel_scale);
        }
        r.feat[f] = val;
    }
}

// finish_row for the usual feature list -- position errors of samples 0 .. K-1, then velocities of samples 0 .. K-1
// (input_idx = range(K), "pos_vel" order: actuator_net.py:160-171) -- where every feature is a known register: no
// per-feature selects on the host's tables.
template <int H, int K>
ILX_DEV void finish_row_pos_vel(const NetArgs& a, const RowRawT<H>& w, RowInT<H>& r) {
    static_assert(K >= 1 && K <= H, "history length of the specialisation");
    r.pe_new0 = w.q_des - w.q;
    r.vel = w.qd;
    r.vel_limit = w.vel_limit, r.effort_limit = w.effort_limit;
#pragma unroll
    for (int k = 0; k < H - 1; ++k) r.pe_old[k] = w.pe_old[k], r.v_old[k] = w.v_old[k];
#pragma unroll
    for (int f = 0; f < 2 * H; ++f) r.feat[f] = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        r.feat[k] = (k == 0 ? r.pe_new0 : w.pe_old[k > 0 ? k - 1 : 0]) * a.pos_scale;
        r.feat[K + k] = (k == 0 ? r.vel : w.v_old[k > 0 ? k - 1 : 0]) * a.vel_scale;
    }
}

template <int H, class I>
ILX_DEV void load_row(const NetArgs& a, const RowAt<I>& at, bool valid, RowInT<H>& r) {
    RowRawT<H> w;
    issue_row(a, at, valid, w);
    finish_row(a, w, r);
}

// actuator_net.py:150-156: roll both histories by one slot and write the newest sample into slot 0.
template <int H, class I>
ILX_DEV void store_history(const NetArgs& a, const RowAt<I>& at, const RowInT<H>& r, bool do_pos, bool do_vel) {
    if (do_pos) wr(a.pe_hist, at.hbase, r.pe_new0);
    if (do_vel) wr(a.v_hist, at.hbase, r.vel);
#pragma unroll
    for (int k = 0; k < H - 1; ++k) {
        if (k < a.hist_len - 1) {
            const I o = at.hbase + static_cast<I>((k + 1) * a.j);
            if (do_pos) wr(a.pe_hist, o, r.pe_old[k]);
            if (do_vel) wr(a.v_hist, o, r.v_old[k]);
        }
    }
}

template <int H, class I>
ILX_DEV void store_torque(const NetArgs& a, const RowAt<I>& at, float net_out, const RowInT<H>& r) {
    const float tau = net_out * a.torque_scale;  // actuator_net.py:179
    wr(a.computed, at.row, tau);
    wr(a.applied, at.row, dc_motor_clip(tau, r.vel, a.sat, r.vel_limit, r.effort_limit));
}

// ---- generic path ------------------------------------------------------------------------------
template <int ACT>
__global__ void __launch_bounds__(128) mlp_generic_kernel(NetArgs a) {
    // activations of the two most recent layers: [buffer][unit][thread] -> conflict-free LDS columns
    __shared__ float act[2][kMaxWidth][128];
    const int t = threadIdx.x;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + t;
    const bool valid = row < a.rows;
    RowIn r;
    const RowAt<int64_t> at = row_at<int64_t>(a, valid ? row : 0);
    load_row(a, at, valid, r);
#pragma unroll
    for (int f = 0; f < 2 * kMaxHist; ++f) act[0][f][t] = r.feat[f];
    const float* p = a.net.raw;
    int cur = 0;
    float out = 0.0f;
    for (int l = 0; l < a.net.num_layers; ++l) {
        const int din = a.net.dims[l], dout = a.net.dims[l + 1];
        const float* w = p;
        const float* b = p + static_cast<int64_t>(din) * dout;
        const bool last = (l + 1 == a.net.num_layers);
        for (int o = 0; o < dout; ++o) {
            float acc = b[o];
            for (int i = 0; i < din; ++i) acc = fmaf_(w[o * din + i], act[cur][i][t], acc);
            if (!last) acc = activate<ACT>(acc);
            act[cur ^ 1][o][t] = acc;
            out = acc;
        }
        p = b + dout;
        cur ^= 1;
    }
    if (valid) {
        store_torque(a, at, out, r);
        store_history(a, at, r, true, true);
    }
}

// ---- MFMA path -----------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Packed layout (floats), L = number of hidden layers (all 32 wide), lane = 0..63, r = 0..15:
//   [ks0][64]        first-layer A operands: W0[lane&31][2s + (lane>>5)] (0 beyond in_dim)
//   L x [16][64]     hidden biases in accumulator order: b_l[unit(r, lane>>5)]
//   (L-1) x [16][64] hidden A operands: W_l[lane&31][unit(r, lane>>5)]
//   [16][64]         output weights in accumulator order: w_out[unit(r, lane>>5)]
//   [1]              output bias
// where unit(r, h) = (r & 3) + 8 * (r >> 2) + 4 * h is the MFMA 32x32 accumulator row of register r.
__host__ __device__ inline int acc_unit(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int ACT>
ILX_DEV f32x16 activate16(f32x16 v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (ACT == ILX_ACT_SOFTSIGN) {
            v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + fabsf(v[r]));  // 1 ulp reciprocal; value path only
        } else {
            v[r] = activate<ACT>(v[r]);
        }
    }
    return v;
}

// Softsign of sixteen values with one reciprocal per four.  v_rcp_f32 issues at a quarter of the rate of a multiply, and
// the split kernel is bound by vector issue, so the four denominators d = 1 + |x| of a group share one:
//   r = 1 / (da dc db dd),  1 / (da dc) = r (db dd),  1 / da = dc / (da dc), ...
// in packed multiplies (two values per instruction): 16 issue slots per four values instead of 24.  Every d is >= 1, so
// the product cannot underflow; it must stay finite, which |x| <= 2^24 guarantees (2^100 at most).  CLAMP: the inputs
// are held to that range first -- beyond it softsign is +-1 to within 6e-8 either way; without it the caller knows the
// bound.  Each result carries the reciprocal's 1 ulp and four roundings (<= 3e-7 relative).
using f2 = __attribute__((ext_vector_type(2))) float;
template <bool CLAMP>
ILX_DEV f32x16 softsign16_shared_rcp(f32x16 v) {
    constexpr float kBound = 16777216.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f2 ab = {v[4 * g], v[4 * g + 1]}, cd = {v[4 * g + 2], v[4 * g + 3]};
        if (CLAMP) {
            ab = {__builtin_amdgcn_fmed3f(ab.x, -kBound, kBound), __builtin_amdgcn_fmed3f(ab.y, -kBound, kBound)};
            cd = {__builtin_amdgcn_fmed3f(cd.x, -kBound, kBound), __builtin_amdgcn_fmed3f(cd.y, -kBound, kBound)};
        }
        const f2 dab = {1.0f + fabsf(ab.x), 1.0f + fabsf(ab.y)}, dcd = {1.0f + fabsf(cd.x), 1.0f + fabsf(cd.y)};
        const f2 p = dab * dcd;                                    // (da dc, db dd)
        const float r = __builtin_amdgcn_rcpf(p.x * p.y);
        const f2 q = f2{p.y, p.x} * r;                             // (1 / (da dc), 1 / (db dd))
        const f2 ya = ab * (q * dcd), yc = cd * (q * dab);         // x / d
        v[4 * g] = ya.x, v[4 * g + 1] = ya.y, v[4 * g + 2] = yc.x, v[4 * g + 3] = yc.y;
    }
    return v;
}

template <int ACT>
__global__ void __launch_bounds__(256) mlp_mfma_kernel(NetArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = tile * 32 + (lane & 31);
    const int half = lane >> 5;
    if (tile * 32 >= a.rows) return;  // wave-uniform
    const bool valid = row < a.rows;
    const int nhid = a.net.num_layers - 1;
    const float* pk = a.net.packed;

    RowIn r;
    const RowAt<int64_t> at = row_at<int64_t>(a, valid ? row : 0);
    load_row(a, at, valid, r);

    // ---- layer 0: K = in_dim, two features per MFMA (half 0 feeds the even one, half 1 the odd one)
    const float* bias = pk + a.net.ks0 * 64;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = bias[k * 64 + lane];
#pragma unroll
    for (int s = 0; s < kMaxHist; ++s) {
        if (s < a.net.ks0) {
            const float bop = half ? r.feat[2 * s + 1] : r.feat[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pk[s * 64 + lane], bop, acc, 0, 0, 0);
        }
    }
    acc = activate16<ACT>(acc);

    // ---- hidden layers 1..L-1: the accumulator registers are the B operands
    const float* wts = pk + a.net.ks0 * 64 + nhid * 1024;
    for (int l = 1; l < nhid; ++l) {
        const float* b = bias + l * 1024;
        const float* w = wts + (l - 1) * 1024;
        f32x16 nxt;
#pragma unroll
        for (int k = 0; k < 16; ++k) nxt[k] = b[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(w[k * 64 + lane], acc[k], nxt, 0, 0, 0);
        acc = activate16<ACT>(nxt);
    }

    // ---- output layer (width 1): per-lane partial dot over its 16 units, then add the other half
    const float* wo = wts + (nhid - 1) * 1024;
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) part = fmaf_(wo[k * 64 + lane], acc[k], part);
    const float other = __shfl_xor(part, 32, 64);
    const float out = (half ? other + part : part + other) + wo[1024];

    if (valid) {
        if (half == 0) store_torque(a, at, out, r);
        store_history(a, at, r, half == 0, half == 1);
    }
}

constexpr int kResidentHist = 4;  // the tile kernels serve histories and input index lists up to this length

// One 32-row tile per wavefront, weights in LDS.  The launch is 6,144 tiles on 1,024 SIMDs: with every tile its own
// wavefront and few registers per wavefront (the A operands come from LDS one instruction ahead of their MFMA instead
// of living in ~100 registers), all tiles of a SIMD are resident together, so one tile's loads, activations (VALU) and
// stores run under the other tiles' MFMAs, and nobody waits for a weight prologue of a hundred loads per lane.
// LDS image (floats), L hidden layers of 32 units, acc_unit as above:
//   [ks0][64]        first-layer A operands, as in `packed`
//   (L-1) x [16][64] hidden A operands, as in `packed` (a permutation of the 32 x 32 matrix: nothing is replicated)
//   L x [32]         hidden biases by unit
//   [32]             output weights by unit
//   [1]              output bias
// Operation order is that of mlp_mfma_kernel, so both give the same bits.
template <int ACT, int NHID>
__global__ void __launch_bounds__(256, 8) mlp_mfma_tile_kernel(NetArgs a) {
    __shared__ float img[kResidentHist * 64 + (NHID - 1) * 1024 + NHID * 32 + 32 + 4];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t tiles = (a.rows + 31) / 32;
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const bool live = tile < tiles;  // wave-uniform

    // the tile's first row is wave-uniform, so its env / joint split is scalar work and a lane only adds its offset
    // (< 32) with a small exact division
    const uint32_t j = static_cast<uint32_t>(a.j);
    const uint32_t row0 = static_cast<uint32_t>(live ? tile : 0) * 32u;
    const uint32_t env0 = row0 / j, joint0 = row0 - env0 * j;
    const uint32_t x = joint0 + (lane & 31);  // < j + 32
    uint32_t q = static_cast<uint32_t>(static_cast<float>(x) * a.inv_j);  // exact up to +-1 for x < 2^22
    q = (q * j > x) ? q - 1 : q;
    q = ((q + 1) * j <= x) ? q + 1 : q;
    const uint32_t env = env0 + q, joint = x - q * j, row = row0 + (lane & 31);
    const bool valid = live && row < static_cast<uint32_t>(a.rows);
    const RowAt<uint32_t> at = {valid ? row : 0u, valid ? env * static_cast<uint32_t>(a.hist_len) * j + joint : 0u,
                                valid ? (a.per_env ? row : joint) : 0u};
    // row inputs first: they travel while the weights are copied
    RowRawT<kResidentHist> raw;
    issue_row(a, at, valid, raw);
    for (int i = threadIdx.x; i < a.net.image_floats; i += 256) img[i] = a.net.image[i];
    __syncthreads();
    if (!live) return;
    RowInT<kResidentHist> cur;
    finish_row(a, raw, cur);

    const int ks0 = a.net.ks0;
    const float* w0 = img + lane;
    const float* wh = img + ks0 * 64 + lane;
    const float* bias = img + ks0 * 64 + (NHID - 1) * 1024 + 4 * half;
    const float* wo = bias + NHID * 32;
    const float bo = img[ks0 * 64 + (NHID - 1) * 1024 + NHID * 32 + 32];

    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = bias[acc_unit(k, 0)];
#pragma unroll
    for (int s = 0; s < kResidentHist; ++s) {
        if (s < ks0) {
            const float bop = half ? cur.feat[2 * s + 1] : cur.feat[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s * 64], bop, acc, 0, 0, 0);
        }
    }
    acc = activate16<ACT>(acc);
#pragma unroll
    for (int l = 1; l < NHID; ++l) {
        f32x16 y;
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = bias[l * 32 + acc_unit(k, 0)];
#pragma unroll
        for (int k = 0; k < 16; ++k) y = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[(l - 1) * 1024 + k * 64], acc[k], y, 0, 0, 0);
        acc = activate16<ACT>(y);
    }
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) part = fmaf_(wo[acc_unit(k, 0)], acc[k], part);
    const float other = __shfl_xor(part, 32, 64);
    const float out = (half ? other + part : part + other) + bo;
    if (valid) {
        if (half == 0) store_torque(a, at, out, cur);
        store_history(a, at, cur, half == 0, half == 1);
    }
}

// The tile kernel with the hidden layers on the f16 matrix cores.  An f32 MFMA runs on the vector ALUs at the packed-f32
// rate, so in mlp_mfma_tile_kernel the matrix products and the activations share one pipe; a 32 x 32 x 16 f16 MFMA is 32
// cycles on the matrix cores proper and runs beside other wavefronts' vector work.  To keep f32-grade results both
// factors are split into an f16 head and an f16 remainder (x = hi + lo, |lo| <= 2^-10 |x|) and three of the four
// products are accumulated in f32 (hi*hi + hi*lo + lo*hi; the dropped lo*lo is below 2^-20 of the product): six MFMAs
// of 32 cycles per hidden layer instead of sixteen of 64, relative error of a product ~1e-6 (f32: 6e-8).  The
// accumulator of one layer is the B operand of the next without lane movement: registers 8s..8s+7 of a lane are the
// eight k-values of k-step s in the permuted order unit = 16 s + 8 (j >> 2) + 4 half + (j & 3), and the weights are
// stored in that order.  The first layer (K = 2 x history length) stays on the f32 MFMA: its inputs are unscaled
// physical quantities.
// LDS image: [ks0][64] f32 first-layer operands | per hidden layer 4 x [64 lanes][8] f16 (hi s0, hi s1, lo s0, lo s1)
//            | L x [32] biases | [32] output weights | output bias.
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) __fp16;

ILX_DEV void split_f16(const f32x16& x, h8 (&hi)[2], h8 (&lo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int jj = 0; jj < 8; jj += 2) {
            const float a = x[8 * s + jj], b = x[8 * s + jj + 1];
            const h2 h = __builtin_amdgcn_cvt_pkrtz(a, b);  // truncation: the remainder is exact in f32
            // a - float(h), b - float(h'): one mixed-precision fused multiply-add each, which reads the f16 half as it is
            // (no conversion instruction); h * -1 + a is a - h exactly
            const uint32_t hp = __builtin_bit_cast(uint32_t, h);
            float ra, rb;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hp), "v"(a));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hp), "v"(b));
            const h2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
            hi[s][jj] = static_cast<_Float16>(h[0]), hi[s][jj + 1] = static_cast<_Float16>(h[1]);
            lo[s][jj] = static_cast<_Float16>(l[0]), lo[s][jj + 1] = static_cast<_Float16>(l[1]);
        }
    }
}

#ifdef ILX_MLP_STAMPS  // diagnostics build only: wall-clock stamps (10 ns ticks) of the first wavefront of the first blocks
__device__ unsigned long long g_mlp_stamps[8 * 2048];
#define ILX_MLP_STAMP(k) \
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_mlp_stamps[blockIdx.x * 8 + (k)] = wall_clock64();
#else
#define ILX_MLP_STAMP(k)
#endif

// POS_VEL: 0, or the history length K of a launch whose feature list is the usual one (see finish_row_pos_vel).
// `block`: the workgroup's index among the network's workgroups (the launch may hold others: mlp_split_scene_kernel).
// W: wavefronts (= tiles) per workgroup; the workgroup's threads copy the weight image to LDS together.
template <int ACT, int NHID, int POS_VEL, bool JA = false, int W = 4>
ILX_DEV void mlp_split_block(const NetArgs& a, unsigned block) {
    constexpr int kHiddenWords = (NHID - 1) * 1024;  // 4 fragments x 64 lanes x 16 bytes per hidden layer
    __shared__ __attribute__((aligned(16))) float img[kResidentHist * 64 + kHiddenWords + NHID * 32 + 32 + 4];
    ILX_MLP_STAMP(0)
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t tiles = (a.rows + 31) / 32;
    const int64_t tile = static_cast<int64_t>(block) * W + (threadIdx.x >> 6);
    const bool live = tile < tiles;  // wave-uniform

    const uint32_t j = static_cast<uint32_t>(a.j);
    const uint32_t row0 = static_cast<uint32_t>(live ? tile : 0) * 32u;
    const uint32_t env0 = row0 / j, joint0 = row0 - env0 * j;
    const uint32_t x = joint0 + (lane & 31);  // < j + 32
    uint32_t q = static_cast<uint32_t>(static_cast<float>(x) * a.inv_j);  // exact up to +-1 for x < 2^22
    q = (q * j > x) ? q - 1 : q;
    q = ((q + 1) * j <= x) ? q + 1 : q;
    const uint32_t env = env0 + q, joint = x - q * j, row = row0 + (lane & 31);
    const bool valid = live && row < static_cast<uint32_t>(a.rows);
    const RowAt<uint32_t> at = {valid ? row : 0u, valid ? env * static_cast<uint32_t>(a.hist_len) * j + joint : 0u,
                                valid ? (a.per_env ? row : joint) : 0u};
    RowRawT<kResidentHist> raw;
    issue_row(a, at, valid, raw, POS_VEL == 0);
    float ja_raw = 0.0f, ja_old = 0.0f;
    if constexpr (JA) {
        if (valid) {
            ja_raw = rd(a.ja_raw, at.row), ja_old = rd(a.ja_action, at.row);
            raw.q_des = rd(a.ja_offset, at.row);   // the offset for now; the processed action below
        }
        if (block == 0) {
            for (int k = threadIdx.x; k < a.begin_count; k += 64 * W) a.begin_log[k] = 0.0f;
            if (a.begin_counter && threadIdx.x == 0) *a.begin_counter += 1;
        }
    }
    ILX_MLP_STAMP(1)
    // the image is padded to whole 16-byte words
    {
        const float4* src = reinterpret_cast<const float4*>(a.net.image16);
        float4* dst = reinterpret_cast<float4*>(img);
        for (int i = threadIdx.x; i < a.net.image16_floats / 4; i += 64 * W) dst[i] = src[i];
    }
    __syncthreads();
    ILX_MLP_STAMP(2)
    if (!live) return;
    if constexpr (JA) {
        raw.q_des = ja_raw * a.ja_scale + raw.q_des;   // joint_actions.py:133: raw * scale + offset
        if (valid && half == 0) {                      // each row is held by a lane of either half: one of them writes
            wr(a.ja_prev_action, at.row, ja_old);      // action_manager.py:331-332
            wr(a.ja_action, at.row, ja_raw);
            wr(a.ja_target, at.row, raw.q_des);
        }
    }
    RowInT<kResidentHist> cur;
    if constexpr (POS_VEL > 0) finish_row_pos_vel<kResidentHist, POS_VEL>(a, raw, cur);
    else finish_row(a, raw, cur);

    const int ks0 = a.net.ks0;
    const float* w0 = img + lane;
    const h8* wh = reinterpret_cast<const h8*>(img + ks0 * 64) + lane;   // fragment f of layer l: wh[(4 (l - 1) + f) * 64]
    const float* bias = img + ks0 * 64 + kHiddenWords + 4 * half;
    const float* wo = bias + NHID * 32;
    const float bo = img[ks0 * 64 + kHiddenWords + NHID * 32 + 32];

    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = bias[acc_unit(k, 0)];
#pragma unroll
    for (int s = 0; s < kResidentHist; ++s) {
        if (s < ks0) {
            const float bop = half ? cur.feat[2 * s + 1] : cur.feat[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s * 64], bop, acc, 0, 0, 0);
        }
    }
    ILX_MLP_STAMP(3)
    // the first layer's inputs are physical quantities of any size; the hidden layers' pre-activations are bounded by
    // the weights (ilx_mlp_create: the split image exists only when 32 |w| + |b| stays far below 2^24)
    if constexpr (ACT == ILX_ACT_SOFTSIGN) acc = softsign16_shared_rcp<true>(acc);
    else acc = activate16<ACT>(acc);
    ILX_MLP_STAMP(4)
#pragma unroll
    for (int l = 1; l < NHID; ++l) {
        h8 xh[2], xl[2];
        split_f16(acc, xh, xl);
        f32x16 y;
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = bias[l * 32 + acc_unit(k, 0)];
        const h8* w = wh + (l - 1) * 256;
        // small terms first
#pragma unroll
        for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(2 + s) * 64], xh[s], y, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s * 64], xl[s], y, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s * 64], xh[s], y, 0, 0, 0);
        if constexpr (ACT == ILX_ACT_SOFTSIGN) acc = softsign16_shared_rcp<false>(y);
        else acc = activate16<ACT>(y);
    }
    ILX_MLP_STAMP(5)
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) part = fmaf_(wo[acc_unit(k, 0)], acc[k], part);
    const float other = __shfl_xor(part, 32, 64);
    const float out = (half ? other + part : part + other) + bo;
    if (valid) {
        if (half == 0) store_torque(a, at, out, cur);
        store_history(a, at, cur, half == 0, half == 1);
    }
    ILX_MLP_STAMP(6)
}

template <int ACT, int NHID, int POS_VEL>
__global__ void __launch_bounds__(256, 8) mlp_mfma_split_kernel(NetArgs a) {
    mlp_split_block<ACT, NHID, POS_VEL>(a, blockIdx.x);
}

// The ANYdrive shape (three hidden layers, three samples in the usual feature order) with wavefronts that stay: the grid
// is a few workgroups of four wavefronts per CU however many tiles there are, wavefront w of all `nblocks * 4` takes tiles
// w, w + nblocks * 4, ... and has the next tile's rows in flight while it computes the current one.  With a wavefront
// per tile (mlp_split_block) all six tiles of a SIMD run their address preamble at once, sharing its issue slots, so the
// first row load leaves after a microsecond, and every workgroup waits for its own copy of the weight image; here a
// SIMD's few wavefronts (kStayPerCu) pay both once and the other tiles' loads travel under compute.  The free wavefront slots
// are what mlp_split_scene_kernel's scene-update workgroups run in.
struct TileAt {
    RowAt<uint32_t> at;
    bool valid;
};
ILX_DEV TileAt split_tile_at(const NetArgs& a, uint32_t tile, int lane) {
    const uint32_t j = static_cast<uint32_t>(a.j);
    const uint32_t row0 = tile * 32u;
    const uint32_t env0 = row0 / j, joint0 = row0 - env0 * j;   // wave-uniform: scalar
    const uint32_t x = joint0 + (lane & 31);  // < j + 32
    uint32_t q = static_cast<uint32_t>(static_cast<float>(x) * a.inv_j);  // exact up to +-1 for x < 2^22
    q = (q * j > x) ? q - 1 : q;
    q = ((q + 1) * j <= x) ? q + 1 : q;
    const uint32_t env = env0 + q, joint = x - q * j, row = row0 + (lane & 31);
    const bool valid = row < static_cast<uint32_t>(a.rows);
    return {{valid ? row : 0u, valid ? env * static_cast<uint32_t>(a.hist_len) * j + joint : 0u, valid ? (a.per_env ? row : joint) : 0u},
            valid};
}
struct JointActionRaw {
    float raw, old, offset;
};
template <bool JA>
ILX_DEV void split_tile_issue(const NetArgs& a, const TileAt& t, RowRawT<kResidentHist>& raw, JointActionRaw& ja) {
    issue_row(a, t.at, t.valid, raw, false);
    ja = {0.0f, 0.0f, 0.0f};
    if constexpr (JA) {
        if (t.valid) ja = {rd(a.ja_raw, t.at.row), rd(a.ja_action, t.at.row), rd(a.ja_offset, t.at.row)};
    }
}

template <int ACT, bool JA>
ILX_DEV void mlp_split_persistent(const NetArgs& a, unsigned block, unsigned nblocks) {
    constexpr int NHID = 3, K = 3, W = 4;
    constexpr int kHiddenWords = (NHID - 1) * 1024;
    __shared__ __attribute__((aligned(16))) float img[kResidentHist * 64 + kHiddenWords + NHID * 32 + 32 + 4];
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const uint32_t tiles = static_cast<uint32_t>((a.rows + 31) / 32);
    const uint32_t stride = nblocks * W;
    uint32_t tile = block * W + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));

    ILX_MLP_STAMP(0)
    TileAt cur{};
    RowRawT<kResidentHist> raw;
    JointActionRaw ja{};
    const bool any = tile < tiles;  // wave-uniform
    if (any) {
        cur = split_tile_at(a, tile, lane);
        split_tile_issue<JA>(a, cur, raw, ja);
    }
    ILX_MLP_STAMP(1)
    if constexpr (JA) {
        if (block == 0) {
            for (int k = threadIdx.x; k < a.begin_count; k += 64 * W) a.begin_log[k] = 0.0f;
            if (a.begin_counter && threadIdx.x == 0) *a.begin_counter += 1;
        }
    }
    {   // the image is padded to whole 16-byte words
        const float4* src = reinterpret_cast<const float4*>(a.net.image16);
        float4* dst = reinterpret_cast<float4*>(img);
        for (int i = threadIdx.x; i < a.net.image16_floats / 4; i += 64 * W) dst[i] = src[i];
    }
    __syncthreads();
    ILX_MLP_STAMP(2)
    if (!any) return;

    const int ks0 = a.net.ks0;
    const float* w0 = img + lane;
    const h8* wh = reinterpret_cast<const h8*>(img + ks0 * 64) + lane;   // fragment f of layer l: wh[(4 (l - 1) + f) * 64]
    const float* bias = img + ks0 * 64 + kHiddenWords + 4 * half;
    const float* wo = bias + NHID * 32;
    const float bo = img[ks0 * 64 + kHiddenWords + NHID * 32 + 32];

    // one tile, up to its stores: joint action (JA), features, the four layers
    struct TileOut {
        RowInT<kResidentHist> in;
        float out, q_des;
    };
    auto compute = [&](RowRawT<kResidentHist>& raw, const JointActionRaw& ja) {
        TileOut r;
        if constexpr (JA) raw.q_des = ja.raw * a.ja_scale + ja.offset;   // joint_actions.py:133: raw * scale + offset
        r.q_des = raw.q_des;
        finish_row_pos_vel<kResidentHist, K>(a, raw, r.in);

        f32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = bias[acc_unit(k, 0)];
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const float bop = half ? r.in.feat[2 * s + 1] : r.in.feat[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s * 64], bop, acc, 0, 0, 0);
        }
        if constexpr (ACT == ILX_ACT_SOFTSIGN) acc = softsign16_shared_rcp<true>(acc);
        else acc = activate16<ACT>(acc);
#pragma unroll
        for (int l = 1; l < NHID; ++l) {
            h8 xh[2], xl[2];
            split_f16(acc, xh, xl);
            f32x16 y;
#pragma unroll
            for (int k = 0; k < 16; ++k) y[k] = bias[l * 32 + acc_unit(k, 0)];
            const h8* w = wh + (l - 1) * 256;
            // small terms first
#pragma unroll
            for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(2 + s) * 64], xh[s], y, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s * 64], xl[s], y, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s * 64], xh[s], y, 0, 0, 0);
            if constexpr (ACT == ILX_ACT_SOFTSIGN) acc = softsign16_shared_rcp<false>(y);
            else acc = activate16<ACT>(y);
        }
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) part = fmaf_(wo[acc_unit(k, 0)], acc[k], part);
        const float other = __shfl_xor(part, 32, 64);
        r.out = (half ? other + part : part + other) + bo;
        return r;
    };
    auto store = [&](const TileAt& t, const TileOut& r, const JointActionRaw& ja) {
        if (!t.valid) return;
        if constexpr (JA) {
            if (half == 0) {                               // each row is held by a lane of either half: one of them writes
                wr(a.ja_prev_action, t.at.row, ja.old);    // action_manager.py:331-332
                wr(a.ja_action, t.at.row, ja.raw);
                wr(a.ja_target, t.at.row, r.q_des);
            }
        }
        if (half == 0) store_torque(a, t.at, r.out, r.in);
        store_history(a, t.at, r.in, half == 0, half == 1);
    };
    // The registers a load fills are named here: the wavefront waits for them at this point and nowhere earlier.
    auto landed = [](const RowRawT<kResidentHist>& w, const JointActionRaw& ja) {
        asm volatile("" ::"v"(w.q_des), "v"(w.q), "v"(w.qd), "v"(w.vel_limit), "v"(w.effort_limit), "v"(w.pe_old[0]), "v"(w.pe_old[1]),
                     "v"(w.v_old[0]), "v"(w.v_old[1]));
        if constexpr (JA) asm volatile("" ::"v"(ja.raw), "v"(ja.old), "v"(ja.offset));
    };
    // Two sets of row registers, used in turn.  Loads and stores share one in-order counter that the compiler can only
    // wait on as a whole once both kinds are pending, so the order within a tile is: next tile's loads out, arithmetic,
    // wait (for those loads and the previous tile's stores, which have had the arithmetic to finish), stores.
    TileAt other{};
    RowRawT<kResidentHist> raw_o;
    JointActionRaw ja_o{};
    landed(raw, ja);
    for (;;) {
        bool more = tile + stride < tiles;  // wave-uniform
        if (more) {
            other = split_tile_at(a, tile + stride, lane);
            split_tile_issue<JA>(a, other, raw_o, ja_o);
        }
        {
            const TileOut r = compute(raw, ja);
            landed(raw_o, ja_o);   // (unconditional: the compiler places the wait by paths, not by conditions)
            store(cur, r, ja);
        }
        ILX_MLP_STAMP(4)
        if (!more) break;
        tile += stride;
        more = tile + stride < tiles;
        if (more) {
            cur = split_tile_at(a, tile + stride, lane);
            split_tile_issue<JA>(a, cur, raw, ja);
        }
        {
            const TileOut r = compute(raw_o, ja_o);
            landed(raw, ja);
            store(other, r, ja_o);
        }
        ILX_MLP_STAMP(5)
        if (!more) break;
        tile += stride;
    }
    ILX_MLP_STAMP(6)
}

template <int ACT>
__global__ void __launch_bounds__(256, 4) mlp_split_persistent_kernel(NetArgs a) {
    request_kernel_arguments<kernel_argument_bytes<NetArgs>()>();
    mlp_split_persistent<ACT, false>(a, blockIdx.x, gridDim.x);
}

// The three-sample launch that also processes the joint action and opens the step (NetArgs::ja_raw).
template <int ACT>
__global__ void __launch_bounds__(256, 4) mlp_split_joint_action_kernel(NetArgs a) {
    request_kernel_arguments<kernel_argument_bytes<NetArgs>()>();
    mlp_split_persistent<ACT, true>(a, blockIdx.x, gridDim.x);
}

// The actuator network of one physics substep and the scene update of the substep before it, which have no physics step
// between them and share no buffer (articulation.py:1399-1436 writes torques and the network's histories;
// interactive_scene.py:496-513 reads joint velocities and contact forces and writes joint_acc, the contact buffers and
// the sensor clocks), as workgroups of one grid: the network's first -- kStayPerCu per CU, which stay for the whole launch and
// are bound by the vector and matrix pipes --, the update's, bound by memory, run in the wavefront slots beside them.  One
// launch and its fixed cost fewer per substep, and the overlap of the two does not depend on how two queues happen to run.
// Registers: six wavefronts per SIMD (80 vector registers, what the network's body needs): kStayPerCu of the network, the rest of
// the update.
// `mlp_first`, `su_first`: where each kind's workgroups start in the grid.
template <int ACT>
__global__ void __launch_bounds__(256, 6) mlp_split_scene_kernel(NetArgs a, SceneArgs s, unsigned mlp_first, unsigned mlp_blocks,
                                                                 unsigned su_first) {
    request_kernel_arguments<kernel_argument_bytes<NetArgs, SceneArgs, unsigned, unsigned, unsigned>()>();
    const unsigned b = blockIdx.x;
    if (b >= mlp_first && b - mlp_first < mlp_blocks) mlp_split_persistent<ACT, false>(a, b - mlp_first, mlp_blocks);
    else scene_update_thread(s, b - su_first);
}

}  // namespace ilx

using namespace ilx;

#ifdef ILX_MLP_STAMPS
extern "C" int ilx_debug_mlp_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ilx::g_mlp_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int ilx_mlp_create(int num_layers, const int32_t* dims, int activation, const float* const* weights_host,
                              const float* const* biases_host, ilx_mlp_t* out) {
    ILX_REQUIRE(out, ILX_EINVAL, "ilx_mlp_create: out is NULL");
    *out = nullptr;
    ILX_REQUIRE(num_layers >= 1 && num_layers <= kMaxLayers, ILX_ESHAPE, "ilx_mlp_create: %d layers (1..%d supported)",
                num_layers, kMaxLayers);
    ILX_REQUIRE(dims && weights_host && biases_host, ILX_EINVAL, "ilx_mlp_create: NULL argument");
    ILX_REQUIRE(activation >= 0 && activation <= ILX_ACT_IDENTITY, ILX_EINVAL, "ilx_mlp_create: unknown activation %d",
                activation);
    for (int l = 0; l <= num_layers; ++l)
        ILX_REQUIRE(dims[l] >= 1 && dims[l] <= kMaxWidth, ILX_ESHAPE, "ilx_mlp_create: layer width %d (1..%d supported)",
                    dims[l], kMaxWidth);
    ILX_REQUIRE(dims[0] <= 2 * kMaxHist, ILX_ESHAPE, "ilx_mlp_create: input width %d (max %d)", dims[0], 2 * kMaxHist);
    ILX_REQUIRE(dims[num_layers] == 1, ILX_ESHAPE, "ilx_mlp_create: output width must be 1 (one torque per joint)");

    auto* obj = new ilx_mlp_s();
    MlpHost& h = obj->h;
    h.dev.num_layers = num_layers;
    for (int l = 0; l <= num_layers; ++l) h.dev.dims[l] = dims[l];
    h.dev.act = activation;
    h.dev.ks0 = (dims[0] + 1) / 2;
    (void)hipGetDevice(&h.device);

    std::vector<float> raw;
    for (int l = 0; l < num_layers; ++l) {
        raw.insert(raw.end(), weights_host[l], weights_host[l] + static_cast<size_t>(dims[l]) * dims[l + 1]);
        raw.insert(raw.end(), biases_host[l], biases_host[l] + dims[l + 1]);
    }
    int rc = check_hip(hipMalloc(&h.d_raw, raw.size() * sizeof(float)), "ilx_mlp_create: hipMalloc");
    if (rc == ILX_OK) rc = check_hip(hipMemcpy(h.d_raw, raw.data(), raw.size() * sizeof(float), hipMemcpyHostToDevice),
                                     "ilx_mlp_create: hipMemcpy");
    h.dev.raw = h.d_raw;
    h.dev.packed = nullptr;
    h.dev.image = nullptr;
    h.dev.image_floats = 0;
    h.dev.image16 = nullptr;
    h.dev.image16_floats = 0;
    h.raw_floats = static_cast<int>(raw.size());

    bool mfma = num_layers >= 2;
    for (int l = 1; l < num_layers; ++l) mfma = mfma && dims[l] == 32;
    if (rc == ILX_OK && mfma) {
        const int nhid = num_layers - 1, ks0 = h.dev.ks0;
        std::vector<float> pk(static_cast<size_t>(ks0) * 64 + static_cast<size_t>(2 * nhid) * 1024 + 1, 0.0f);
        float* p = pk.data();
        for (int s = 0; s < ks0; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int f = 2 * s + (lane >> 5);
                p[s * 64 + lane] = f < dims[0] ? weights_host[0][(lane & 31) * dims[0] + f] : 0.0f;
            }
        p += ks0 * 64;
        for (int l = 0; l < nhid; ++l)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane) p[l * 1024 + r * 64 + lane] = biases_host[l][acc_unit(r, lane >> 5)];
        p += nhid * 1024;
        for (int l = 1; l < nhid; ++l)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane)
                    p[(l - 1) * 1024 + r * 64 + lane] = weights_host[l][(lane & 31) * 32 + acc_unit(r, lane >> 5)];
        p += (nhid - 1) * 1024;
        for (int r = 0; r < 16; ++r)
            for (int lane = 0; lane < 64; ++lane) p[r * 64 + lane] = weights_host[nhid][acc_unit(r, lane >> 5)];
        p[1024] = biases_host[nhid][0];
        rc = check_hip(hipMalloc(&h.d_packed, pk.size() * sizeof(float)), "ilx_mlp_create: hipMalloc");
        if (rc == ILX_OK) rc = check_hip(hipMemcpy(h.d_packed, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice),
                                         "ilx_mlp_create: hipMemcpy");
        h.dev.packed = h.d_packed;
        // the LDS image of mlp_mfma_tile_kernel: the A operands of `packed`, then biases and output weights by unit
        std::vector<float> img;
        img.insert(img.end(), pk.begin(), pk.begin() + ks0 * 64);
        const float* hidden_a = pk.data() + ks0 * 64 + nhid * 1024;
        img.insert(img.end(), hidden_a, hidden_a + (nhid - 1) * 1024);
        for (int l = 0; l < nhid; ++l) img.insert(img.end(), biases_host[l], biases_host[l] + 32);
        img.insert(img.end(), weights_host[nhid], weights_host[nhid] + 32);
        img.push_back(biases_host[nhid][0]);
        if (rc == ILX_OK) rc = check_hip(hipMalloc(&h.d_image, img.size() * sizeof(float)), "ilx_mlp_create: hipMalloc");
        if (rc == ILX_OK) rc = check_hip(hipMemcpy(h.d_image, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice),
                                         "ilx_mlp_create: hipMemcpy");
        h.dev.image = h.d_image;
        h.dev.image_floats = static_cast<int>(img.size());
        // the image of mlp_mfma_split_kernel: hidden weights as f16 head / remainder fragments in its k order
        std::vector<float> img16(pk.begin(), pk.begin() + ks0 * 64);
        for (int l = 1; l < nhid; ++l) {
            std::vector<_Float16> frag(4 * 64 * 8);
            for (int s2 = 0; s2 < 2; ++s2)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int unit = 16 * s2 + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
                        const float w = weights_host[l][(lane & 31) * 32 + unit];
                        const _Float16 hi = static_cast<_Float16>(w);
                        frag[(s2 * 64 + lane) * 8 + jj] = hi;
                        frag[((2 + s2) * 64 + lane) * 8 + jj] = static_cast<_Float16>(w - static_cast<float>(hi));
                    }
            const float* words = reinterpret_cast<const float*>(frag.data());
            img16.insert(img16.end(), words, words + 1024);
        }
        for (int l = 0; l < nhid; ++l) img16.insert(img16.end(), biases_host[l], biases_host[l] + 32);
        img16.insert(img16.end(), weights_host[nhid], weights_host[nhid] + 32);
        img16.push_back(biases_host[nhid][0]);
        while (img16.size() % 4) img16.push_back(0.0f);
        // The split needs every factor inside the f16 range.  Hidden weights: checked here.  Hidden activations: only the
        // bounded ones (softsign, tanh: |x| <= 1) are known to be; behind ELU / ReLU / identity a pre-activation beyond
        // 65 504 would saturate the f16 head, so those networks are served by the f32 kernels.  Small factors are no
        // hazard: below 2^-14 the f16 head and remainder lose relative precision, but each such factor is then off by at
        // most half an f16 subnormal step (3e-8), which the product with a factor of at most one carries into the sum as
        // an absolute error of that size.
        // The same bound on weights and biases keeps a hidden pre-activation (32 products with factors of at most one,
        // and the bias) below 2^22, which the shared reciprocal of the softsign needs (softsign16_shared_rcp).
        bool in_range = activation == ILX_ACT_SOFTSIGN || activation == ILX_ACT_TANH;
        for (int l = 1; l < nhid; ++l) {
            for (int i = 0; i < 32 * 32; ++i) in_range = in_range && std::fabs(weights_host[l][i]) < 6.0e4f;
            for (int i = 0; i < 32; ++i) in_range = in_range && std::fabs(biases_host[l][i]) < 6.0e4f;
        }
        if (rc == ILX_OK && in_range) {
            rc = check_hip(hipMalloc(&h.d_image16, img16.size() * sizeof(float)), "ilx_mlp_create: hipMalloc");
            if (rc == ILX_OK)
                rc = check_hip(hipMemcpy(h.d_image16, img16.data(), img16.size() * sizeof(float), hipMemcpyHostToDevice),
                               "ilx_mlp_create: hipMemcpy");
            h.dev.image16 = h.d_image16;
            h.dev.image16_floats = static_cast<int>(img16.size());
        }
    }
    if (rc != ILX_OK) {
        if (h.d_raw) (void)hipFree(h.d_raw);
        if (h.d_packed) (void)hipFree(h.d_packed);
        if (h.d_image) (void)hipFree(h.d_image);
        if (h.d_image16) (void)hipFree(h.d_image16);
        delete obj;
        return rc;
    }
    *out = obj;
    return ILX_OK;
}

extern "C" void ilx_mlp_destroy(ilx_mlp_t net) {
    if (!net) return;
    if (net->h.d_raw) (void)hipFree(net->h.d_raw);
    if (net->h.d_packed) (void)hipFree(net->h.d_packed);
    if (net->h.d_image) (void)hipFree(net->h.d_image);
    if (net->h.d_image16) (void)hipFree(net->h.d_image16);
    delete net;
}

static int actuator_pd(int model, const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                       const float* qd, const float* stiffness, const float* damping, const float* effort_limit,
                       const float* velocity_limit, int params_per_env, float saturation_effort, float* computed,
                       float* applied, int64_t n, int j, const ilx_scene* scene, void* stream) {
    ILX_REQUIRE(model == ILX_ACTUATOR_IDEAL_PD || model == ILX_ACTUATOR_DC_MOTOR, ILX_EINVAL,
                "ilx_actuator_pd: unknown model %d", model);
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_actuator_pd: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(q_des && q && qd && stiffness && damping && effort_limit && computed && applied, ILX_EINVAL,
                "ilx_actuator_pd: NULL operand");
    ILX_REQUIRE(model != ILX_ACTUATOR_DC_MOTOR || velocity_limit, ILX_EINVAL,
                "ilx_actuator_pd: the DC motor model needs velocity_limit");
    const int64_t total = n * j;
    if (total == 0) return scene ? ilx_scene_update(scene, stream) : ILX_OK;
    dim3 g(blocks_for(total)), b(256);
    hipStream_t s = as_stream(stream);
    if (scene && scene->num_envs > 0) {   // the scene update's workgroups first, the actuator's behind them
        SceneArgs sa{};
        unsigned su_blocks = 0;
        const int rc = scene_args_from(*scene, sa, su_blocks);
        if (rc != ILX_OK) return rc;
        dim3 both(g.x + su_blocks);
        if (model == ILX_ACTUATOR_DC_MOTOR)
            ILX_LAUNCH(pd_scene_kernel<1>, both, b, 0, s, q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit,
                               velocity_limit, params_per_env, saturation_effort, computed, applied, total, j, sa, su_blocks);
        else
            ILX_LAUNCH(pd_scene_kernel<0>, both, b, 0, s, q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit,
                               velocity_limit, params_per_env, saturation_effort, computed, applied, total, j, sa, su_blocks);
    } else if (model == ILX_ACTUATOR_DC_MOTOR)
        ILX_LAUNCH(actuator_pd_kernel<1>, g, b, 0, s, q_des, qd_des, tau_ff, q, qd, stiffness, damping,
                           effort_limit, velocity_limit, params_per_env, saturation_effort, computed, applied, total, j);
    else
        ILX_LAUNCH(actuator_pd_kernel<0>, g, b, 0, s, q_des, qd_des, tau_ff, q, qd, stiffness, damping,
                           effort_limit, velocity_limit, params_per_env, saturation_effort, computed, applied, total, j);
    ILX_LAUNCH_CHECK("ilx_actuator_pd");
    return ILX_OK;
}

extern "C" int ilx_actuator_pd(int model, const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                               const float* qd, const float* stiffness, const float* damping, const float* effort_limit,
                               const float* velocity_limit, int params_per_env, float saturation_effort, float* computed,
                               float* applied, int64_t n, int j, void* stream) {
    return actuator_pd(model, q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit, velocity_limit, params_per_env,
                       saturation_effort, computed, applied, n, j, nullptr, stream);
}

extern "C" int ilx_actuator_pd_scene_update(int model, const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                                            const float* qd, const float* stiffness, const float* damping,
                                            const float* effort_limit, const float* velocity_limit, int params_per_env,
                                            float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                            const ilx_scene* scene, void* stream) {
    ILX_REQUIRE(scene, ILX_EINVAL, "ilx_actuator_pd_scene_update: scene is NULL");
    return actuator_pd(model, q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit, velocity_limit, params_per_env,
                       saturation_effort, computed, applied, n, j, scene, stream);
}

extern "C" int ilx_actuator_pd_joint_action(int model, const float* raw_action, float* action, float* prev_action,
                                            float action_scale, const float* action_offset, int32_t target, int32_t first_joint,
                                            int32_t width, float* q_des, const float* qd_des, float* tau_ff, const float* q,
                                            const float* qd, const float* stiffness, const float* damping,
                                            const float* effort_limit, const float* velocity_limit, int params_per_env,
                                            float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                            float* log_sums, int32_t log_count, uint64_t* step_index_device, void* stream) {
    ILX_REQUIRE(model == ILX_ACTUATOR_IDEAL_PD || model == ILX_ACTUATOR_DC_MOTOR, ILX_EINVAL,
                "ilx_actuator_pd_joint_action: unknown model %d", model);
    ILX_REQUIRE(n > 0 && j > 0, ILX_EINVAL, "ilx_actuator_pd_joint_action: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(raw_action && q_des && q && qd && stiffness && damping && effort_limit && computed && applied, ILX_EINVAL,
                "ilx_actuator_pd_joint_action: NULL operand");
    ILX_REQUIRE(!prev_action || action, ILX_EINVAL, "ilx_actuator_pd_joint_action: prev_action needs action");
    ILX_REQUIRE(target == 0 || (target == 1 && tau_ff), ILX_EINVAL,
                "ilx_actuator_pd_joint_action: target %d (0: q_des, 1: tau_ff, which must then be given)", target);
    ILX_REQUIRE(width > 0 && first_joint >= 0 && first_joint + width <= j, ILX_ESHAPE,
                "ilx_actuator_pd_joint_action: joints %d .. %d of %d", first_joint, first_joint + width - 1, j);
    ILX_REQUIRE(model != ILX_ACTUATOR_DC_MOTOR || velocity_limit, ILX_EINVAL,
                "ilx_actuator_pd_joint_action: the DC motor model needs velocity_limit");
    ILX_REQUIRE(log_count >= 0 && (log_count == 0 || log_sums), ILX_EINVAL,
                "ilx_actuator_pd_joint_action: %d floats at a NULL pointer", log_count);
    const PdJointAction ja{raw_action, action, prev_action, action_offset, action_scale, target, first_joint, width,
                             log_sums, log_sums ? log_count : 0, step_index_device};
    const int64_t total = n * j;
    dim3 g(blocks_for(total)), b(256);
    if (model == ILX_ACTUATOR_DC_MOTOR)
        ILX_LAUNCH(pd_joint_action_kernel<1>, g, b, 0, as_stream(stream), ja, q_des, qd_des, tau_ff, q, qd, stiffness, damping,
                           effort_limit, velocity_limit, params_per_env, saturation_effort, computed, applied, total, j);
    else
        ILX_LAUNCH(pd_joint_action_kernel<0>, g, b, 0, as_stream(stream), ja, q_des, qd_des, tau_ff, q, qd, stiffness, damping,
                           effort_limit, velocity_limit, params_per_env, saturation_effort, computed, applied, total, j);
    ILX_LAUNCH_CHECK("ilx_actuator_pd_joint_action");
    return ILX_OK;
}

// Workgroups per CU of the network kernels whose wavefronts stay (mlp_split_persistent).  Four wavefronts per SIMD: the
// vector pipe takes an instruction every other cycle and a wavefront alone offers one every four, while the matrix
// products and LDS reads of one wavefront want the arithmetic of another to hide behind; more only delays the first row
// loads (every wavefront runs its address preamble first) and, in the launch shared with the scene update, takes the
// wavefront slots the update's workgroups run in.  Measured at 196 608 rows, the whole step of seven launches (four of
// them the network's) on one box: 2 / 3 / 4 per CU = 84.9 / 84.1 / 82.7 us, on another 4 / 5 / 6 = 80.5 / 82.3 / 82.9.
constexpr unsigned kStayPerCu = 4;

// `scene` (or null): the scene update that goes with this launch -- as workgroups of the same grid where the launch is
// the three-sample split kernel's, else as a launch of its own ahead of it.
template <int ACT>
static int launch_mlp(const NetArgs& a, bool force_generic, bool exact_f32, hipStream_t s, const ilx_scene* scene = nullptr) {
    SceneArgs sa{};
    unsigned su_blocks = 0;
    if (scene) {
        const int rc = scene_args_from(*scene, sa, su_blocks);
        if (rc != ILX_OK) return rc;
    }
    bool scene_inside = false;
    if (a.net.packed && !force_generic) {
        const int64_t tiles = (a.rows + 31) / 32;
        const int nhid = a.net.num_layers - 1;
        // the tile kernels index with 32 bits
        const bool small = a.hist_len <= kResidentHist && a.n_feat <= 2 * kResidentHist && a.j < (1 << 20) &&
                           a.rows * a.hist_len < (int64_t(1) << 30);  // byte offsets of the history below 2^32
        const unsigned tile_blocks = static_cast<unsigned>((tiles + 3) / 4);
        // the kernels whose wavefronts stay: kStayPerCu workgroups of four wavefronts per CU, or fewer when the tiles run out
        const unsigned stay_blocks = std::min(tile_blocks, kStayPerCu * static_cast<unsigned>(device_cu_count()));
        const bool split = a.net.image16 && !exact_f32;
        // the usual feature list: position errors of samples 0 .. K-1, then the velocities of the same samples
        bool pos_vel = a.n_feat == 2 * a.hist_len;
        for (int f = 0; f < a.n_feat && pos_vel; ++f)
            pos_vel = a.feat_slot[f] == f % a.hist_len && (((a.feat_is_pos >> f) & 1u) != 0) == (f < a.hist_len);
        const bool three = nhid == 3 && small && split && pos_vel && a.hist_len == 3;
        if (a.ja_raw && three) {   // the step's first actuator launch: the joint action's processing rides on it
            ILX_LAUNCH((mlp_split_joint_action_kernel<ACT>), dim3(stay_blocks), dim3(256), 0, s, a);
            return ILX_OK;
        }
        scene_inside = scene && scene->num_envs > 0 && three;
        if (scene && !scene_inside) {
            const int rc = ilx_scene_update(scene, s);
            if (rc != ILX_OK) return rc;
        }
        if (scene_inside)   // the network's workgroups first in the grid, the update's behind them
            ILX_LAUNCH((mlp_split_scene_kernel<ACT>), dim3(stay_blocks + su_blocks), dim3(256), 0, s, a, sa, 0u, stay_blocks,
                               stay_blocks);
        else if (nhid == 3 && small && split && pos_vel && a.hist_len == 3) {  // the ANYdrive network's shape: three samples
            ILX_LAUNCH((mlp_split_persistent_kernel<ACT>), dim3(stay_blocks), dim3(256), 0, s, a);
        }
        else if (nhid == 3 && small && split)
            ILX_LAUNCH((mlp_mfma_split_kernel<ACT, 3, 0>), dim3(tile_blocks), dim3(256), 0, s, a);
        else if (nhid == 2 && small && split)
            ILX_LAUNCH((mlp_mfma_split_kernel<ACT, 2, 0>), dim3(tile_blocks), dim3(256), 0, s, a);
        else if (nhid == 3 && small)
            ILX_LAUNCH((mlp_mfma_tile_kernel<ACT, 3>), dim3(tile_blocks), dim3(256), 0, s, a);
        else if (nhid == 2 && small)
            ILX_LAUNCH((mlp_mfma_tile_kernel<ACT, 2>), dim3(tile_blocks), dim3(256), 0, s, a);
        else
            ILX_LAUNCH(mlp_mfma_kernel<ACT>, dim3(static_cast<unsigned>((tiles + 3) / 4)), dim3(256), 0, s, a);
    } else {
        if (scene) {
            const int rc = ilx_scene_update(scene, s);
            if (rc != ILX_OK) return rc;
        }
        ILX_LAUNCH(mlp_generic_kernel<ACT>, dim3(blocks_for(a.rows, 128)), dim3(128), 0, s, a);
    }
    return ILX_OK;
}

// What the step's first actuator launch does on the way (ilx_actuator_net_mlp_joint_action), or null.
struct JointActionRide {
    const float* raw;
    float* action;
    float* prev_action;
    float scale;
    const float* offset;
    float* target;
    float* log_sums;
    int32_t log_count;
    uint64_t* step_index_device;
};

static int actuator_net_mlp(ilx_mlp_t net, const float* q_des, const float* q, const float* qd,
                            float* pos_error_history, float* velocity_history, int history_length,
                            const int32_t* input_idx_host, int num_input_idx, int input_order, float pos_scale,
                            float vel_scale, float torque_scale, const float* effort_limit,
                            const float* velocity_limit, int params_per_env, float saturation_effort,
                            float* computed, float* applied, int64_t n, int j, int flags, const ilx_scene* scene, void* stream,
                            const JointActionRide* ride = nullptr) {
    ILX_REQUIRE(net, ILX_EINVAL, "ilx_actuator_net_mlp: net is NULL");
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_actuator_net_mlp: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(q_des && q && qd && pos_error_history && velocity_history && effort_limit && velocity_limit && computed &&
                    applied && input_idx_host,
                ILX_EINVAL, "ilx_actuator_net_mlp: NULL operand");
    ILX_REQUIRE(history_length >= 1 && history_length <= kMaxHist, ILX_ESHAPE,
                "ilx_actuator_net_mlp: history length %d (1..%d supported)", history_length, kMaxHist);
    ILX_REQUIRE(num_input_idx >= 1 && num_input_idx <= kMaxHist, ILX_ESHAPE, "ilx_actuator_net_mlp: %d input indices",
                num_input_idx);
    ILX_REQUIRE(2 * num_input_idx == net->h.dev.dims[0], ILX_ESHAPE,
                "ilx_actuator_net_mlp: network input width %d != 2 * %d input indices", net->h.dev.dims[0], num_input_idx);
    ILX_REQUIRE(input_order == 0 || input_order == 1, ILX_EINVAL, "ilx_actuator_net_mlp: input_order must be 0 or 1");
    NetArgs a{};
    a.net = net->h.dev;
    a.q_des = q_des, a.q = q, a.qd = qd;
    a.pe_hist = pos_error_history, a.v_hist = velocity_history;
    a.hist_len = history_length;
    a.n_feat = 2 * num_input_idx;
    a.feat_is_pos = 0u;
    for (int f = 0; f < 2 * kMaxHist; ++f) a.feat_slot[f] = 0;
    for (int m = 0; m < num_input_idx; ++m) {
        ILX_REQUIRE(input_idx_host[m] >= 0 && input_idx_host[m] < history_length, ILX_EINVAL,
                    "ilx_actuator_net_mlp: input index %d outside the history (length %d)", input_idx_host[m], history_length);
        a.feat_slot[m] = a.feat_slot[m + num_input_idx] = input_idx_host[m];
        a.feat_is_pos |= 1u << (input_order == 0 ? m : m + num_input_idx);  // pos_vel: positions first
    }
    a.pos_scale = pos_scale, a.vel_scale = vel_scale, a.torque_scale = torque_scale;
    a.effort_limit = effort_limit, a.vel_limit = velocity_limit, a.per_env = params_per_env, a.sat = saturation_effort;
    a.computed = computed, a.applied = applied;
    a.rows = n * j, a.j = j, a.inv_j = 1.0f / static_cast<float>(j);
    if (a.rows == 0) return scene ? ilx_scene_update(scene, stream) : ILX_OK;
    hipStream_t s = as_stream(stream);
    const bool generic = (flags & ILX_MLP_FORCE_GENERIC) != 0, exact = (flags & ILX_MLP_EXACT_F32) != 0;
    if (ride) {
        // the launch that can carry the ride is the three-sample split kernel's (as launch_mlp picks it); every other
        // network gets the joint action's own launch ahead of it
        const int nhid = a.net.num_layers - 1;
        bool pos_vel = a.n_feat == 2 * a.hist_len;
        for (int f = 0; f < a.n_feat && pos_vel; ++f)
            pos_vel = a.feat_slot[f] == f % a.hist_len && (((a.feat_is_pos >> f) & 1u) != 0) == (f < a.hist_len);
        const bool small = a.hist_len <= kResidentHist && a.n_feat <= 2 * kResidentHist && a.j < (1 << 20) &&
                           a.rows * a.hist_len < (int64_t(1) << 30);
        const bool carries = a.net.packed && !generic && a.net.image16 && !exact && nhid == 3 && small && pos_vel && a.hist_len == 3;
        if (carries) {
            a.ja_raw = ride->raw, a.ja_action = ride->action, a.ja_prev_action = ride->prev_action, a.ja_scale = ride->scale;
            a.ja_offset = ride->offset, a.ja_target = ride->target;
            a.begin_log = ride->log_sums, a.begin_count = ride->log_sums ? ride->log_count : 0, a.begin_counter = ride->step_index_device;
        } else {
            const int rc = ilx_joint_action_process_begin(ride->raw, ride->action, ride->prev_action, nullptr, ride->scale, ride->offset,
                                                          0.0f, nullptr, ride->target, n, j, ride->log_sums, ride->log_count,
                                                          ride->step_index_device, stream);
            if (rc != ILX_OK) return rc;
        }
    }
    int rc = ILX_OK;
    switch (a.net.act) {
        case ILX_ACT_SOFTSIGN: rc = launch_mlp<ILX_ACT_SOFTSIGN>(a, generic, exact, s, scene); break;
        case ILX_ACT_ELU: rc = launch_mlp<ILX_ACT_ELU>(a, generic, exact, s, scene); break;
        case ILX_ACT_TANH: rc = launch_mlp<ILX_ACT_TANH>(a, generic, exact, s, scene); break;
        case ILX_ACT_RELU: rc = launch_mlp<ILX_ACT_RELU>(a, generic, exact, s, scene); break;
        default: rc = launch_mlp<ILX_ACT_IDENTITY>(a, generic, exact, s, scene); break;
    }
    if (rc != ILX_OK) return rc;
    ILX_LAUNCH_CHECK("ilx_actuator_net_mlp");
    return ILX_OK;
}

extern "C" int ilx_actuator_net_mlp(ilx_mlp_t net, const float* q_des, const float* q, const float* qd,
                                    float* pos_error_history, float* velocity_history, int history_length,
                                    const int32_t* input_idx_host, int num_input_idx, int input_order, float pos_scale,
                                    float vel_scale, float torque_scale, const float* effort_limit,
                                    const float* velocity_limit, int params_per_env, float saturation_effort,
                                    float* computed, float* applied, int64_t n, int j, int flags, void* stream) {
    return actuator_net_mlp(net, q_des, q, qd, pos_error_history, velocity_history, history_length, input_idx_host, num_input_idx,
                            input_order, pos_scale, vel_scale, torque_scale, effort_limit, velocity_limit, params_per_env,
                            saturation_effort, computed, applied, n, j, flags, nullptr, stream);
}

extern "C" int ilx_actuator_net_mlp_joint_action(ilx_mlp_t net, const float* raw_action, float* action, float* prev_action,
                                                 float action_scale, const float* action_offset, float* q_des, const float* q,
                                                 const float* qd, float* pos_error_history, float* velocity_history,
                                                 int history_length, const int32_t* input_idx_host, int num_input_idx,
                                                 int input_order, float pos_scale, float vel_scale, float torque_scale,
                                                 const float* effort_limit, const float* velocity_limit, int params_per_env,
                                                 float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                                 int flags, float* log_sums, int32_t log_count, uint64_t* step_index_device,
                                                 void* stream) {
    ILX_REQUIRE(raw_action && action && prev_action && action_offset && q_des, ILX_EINVAL,
                "ilx_actuator_net_mlp_joint_action: NULL operand");
    ILX_REQUIRE(n > 0, ILX_EINVAL, "ilx_actuator_net_mlp_joint_action: no envs");
    ILX_REQUIRE(log_count >= 0 && (log_count == 0 || log_sums), ILX_EINVAL,
                "ilx_actuator_net_mlp_joint_action: %d floats at a NULL pointer", log_count);
    const JointActionRide ride{raw_action, action, prev_action, action_scale, action_offset, q_des, log_sums, log_count,
                               step_index_device};
    return actuator_net_mlp(net, q_des, q, qd, pos_error_history, velocity_history, history_length, input_idx_host, num_input_idx,
                            input_order, pos_scale, vel_scale, torque_scale, effort_limit, velocity_limit, params_per_env,
                            saturation_effort, computed, applied, n, j, flags, nullptr, stream, &ride);
}

extern "C" int ilx_actuator_net_mlp_scene_update(ilx_mlp_t net, const float* q_des, const float* q, const float* qd,
                                                 float* pos_error_history, float* velocity_history, int history_length,
                                                 const int32_t* input_idx_host, int num_input_idx, int input_order,
                                                 float pos_scale, float vel_scale, float torque_scale,
                                                 const float* effort_limit, const float* velocity_limit, int params_per_env,
                                                 float saturation_effort, float* computed, float* applied, int64_t n, int j,
                                                 int flags, const ilx_scene* scene, void* stream) {
    ILX_REQUIRE(scene, ILX_EINVAL, "ilx_actuator_net_mlp_scene_update: scene is NULL");
    return actuator_net_mlp(net, q_des, q, qd, pos_error_history, velocity_history, history_length, input_idx_host, num_input_idx,
                            input_order, pos_scale, vel_scale, torque_scale, effort_limit, velocity_limit, params_per_env,
                            saturation_effort, computed, applied, n, j, flags, scene, stream);
}

extern "C" int ilx_actuator_delayed_pd(const float* q_des, const float* qd_des, const float* tau_ff, const float* q,
                                       const float* qd, const float* stiffness, const float* damping,
                                       const float* effort_limit, int params_per_env, float* positions_ring,
                                       float* velocities_ring, float* efforts_ring, int ring_length, int pointer,
                                       int64_t* num_pushes, const int32_t* time_lags, const float* lookup_angles,
                                       const float* lookup_max_torque, int lookup_size, float* computed, float* applied,
                                       int64_t n, int j, void* stream) {
    ILX_REQUIRE(n >= 0 && j > 0, ILX_EINVAL, "ilx_actuator_delayed_pd: bad shape (%lld, %d)", (long long)n, j);
    ILX_REQUIRE(j <= 256, ILX_ESHAPE, "ilx_actuator_delayed_pd: %d joints per actuator group (max 256)", j);
    ILX_REQUIRE(q_des && q && qd && stiffness && damping && effort_limit && computed && applied && positions_ring && num_pushes &&
                    time_lags,
                ILX_EINVAL, "ilx_actuator_delayed_pd: NULL operand");
    ILX_REQUIRE((qd_des == nullptr) == (velocities_ring == nullptr) && (tau_ff == nullptr) == (efforts_ring == nullptr), ILX_EINVAL,
                "ilx_actuator_delayed_pd: a set-point and its ring must be given together");
    ILX_REQUIRE(ring_length >= 1 && pointer >= 0 && pointer < ring_length, ILX_EINVAL,
                "ilx_actuator_delayed_pd: pointer %d outside a ring of %d slots", pointer, ring_length);
    ILX_REQUIRE((lookup_angles == nullptr) == (lookup_max_torque == nullptr) && (lookup_angles == nullptr || lookup_size >= 1),
                ILX_EINVAL, "ilx_actuator_delayed_pd: the lookup table needs angles, torques and a size >= 1");
    if (n == 0) return ILX_OK;
    DelayArgs a;
    a.set[0] = q_des, a.set[1] = qd_des, a.set[2] = tau_ff;
    a.ring[0] = positions_ring, a.ring[1] = velocities_ring, a.ring[2] = efforts_ring;
    a.len = ring_length, a.pointer = pointer, a.num_pushes = num_pushes, a.time_lags = time_lags;
    a.q = q, a.qd = qd, a.kp = stiffness, a.kd = damping, a.effort_limit = effort_limit, a.per_env = params_per_env;
    a.lookup_x = lookup_angles, a.lookup_y = lookup_max_torque, a.lookup_n = lookup_size;
    a.computed = computed, a.applied = applied, a.n = n, a.j = j, a.envs_per_block = 256 / j;
    const unsigned blocks = static_cast<unsigned>((n + a.envs_per_block - 1) / a.envs_per_block);
    ILX_LAUNCH(actuator_delayed_pd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_actuator_delayed_pd");
    return ILX_OK;
}

extern "C" int ilx_linear_interpolation(const float* x, const float* y, int32_t num_samples, const float* q, float* out, int64_t count,
                                        void* stream) {
    ILX_REQUIRE(num_samples >= 1, ILX_EINVAL, "Input tensor x is empty!");
    ILX_REQUIRE(count >= 0, ILX_EINVAL, "ilx_linear_interpolation: negative count");
    if (count == 0) return ILX_OK;
    ILX_REQUIRE(x && y && q && out, ILX_EINVAL, "ilx_linear_interpolation: NULL operand");
    ILX_LAUNCH(linear_interpolation_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), x, y, num_samples, q, out, count);
    ILX_LAUNCH_CHECK("ilx_linear_interpolation");
    return ILX_OK;
}
