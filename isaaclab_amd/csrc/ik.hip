// Differential inverse kinematics: controller and action term
// (reference: isaaclab/controllers/differential_ik.py:97-240, isaaclab/envs/mdp/actions/task_space_actions.py:140-229).
//
// One thread per environment.  The only matrix the update needs is the 6x6 Gram matrix A = J J^T: it is accumulated
// column by column while the Jacobian streams through (each column is rotated into the root frame and corrected
// for the frame offset on the fly, nothing is written back), the small problem A y = e is solved in float64 in
// registers (Cholesky for the damped method, cyclic Jacobi eigen-decomposition for the pseudo-inverse and the
// singular-value-clipped variant: J^+ = J^T (J J^T)^+ and sigma_k^2 are the eigenvalues of A), and a second pass
// over the (cache-resident) columns forms dq = J^T y.  Position-only commands run the same code with the rotational
// rows zeroed, which decouples them exactly.
#include "common.h"
#include "math_dev.h"

namespace ilx {

struct IkArgs {
    int64_t n;
    int j;
    int pose;  // 1: 6-row pose command, 0: 3-row position command
    int method;
    float k_val, lambda_val, min_sv;
    // frame pose: either given in the root frame already (frame == 0) or derived from world poses (frame == 1)
    int frame;
    const float* ee_pos;   int64_t ee_pos_stride;    // frame 0: (n,3) root-frame position; frame 1: world position rows
    const float* ee_quat;  int64_t ee_quat_stride;
    const float* root_pos; int64_t root_pos_stride;  // frame 1 only
    const float* root_quat; int64_t root_quat_stride;
    int has_offset;
    float off_pos[3], off_rot[4];
    // Jacobian element (env, row, col) at env * jac_env_stride + row * jac_row_stride + (cols ? cols[col] : col)
    const float* jac;
    int64_t jac_env_stride, jac_row_stride;
    const int32_t* jac_cols;
    const float* joint_pos; int64_t joint_pos_stride; const int32_t* joint_ids;
    const float* pos_des;   // (n,3)
    const float* quat_des;  // (n,4)
    float* out;             int64_t out_stride;  // joint targets, column joint_ids[c] (or c)
};

ILX_DEV void frame_pose(const IkArgs& a, int64_t i, f3& pos_b, f4& quat_b, f4& root_q) {
    const f3 p = load3(a.ee_pos + a.ee_pos_stride * i);
    const f4 q = load4u(a.ee_quat + a.ee_quat_stride * i);
    if (!a.frame) {
        pos_b = p, quat_b = q, root_q = f4{1.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    root_q = load4u(a.root_quat + a.root_quat_stride * i);
    subtract_frame_transforms(load3(a.root_pos + a.root_pos_stride * i), root_q, p, q, pos_b, quat_b);  // :196-200
    if (a.has_offset) {  // :202-206
        f3 p2;
        f4 q2;
        combine_frame_transforms(pos_b, quat_b, f3{a.off_pos[0], a.off_pos[1], a.off_pos[2]},
                                 f4{a.off_rot[0], a.off_rot[1], a.off_rot[2], a.off_rot[3]}, p2, q2);
        pos_b = p2, quat_b = q2;
    }
}

// Column c of the frame Jacobian: task_space_actions.py:140-149 (rotation into the root frame) and :211-229 (offset).
ILX_DEV void jac_column(const IkArgs& a, int64_t i, int c, f4 root_q_inv, float* g) {
    const float* base = a.jac + a.jac_env_stride * i + (a.jac_cols ? a.jac_cols[c] : c);
    f3 v = {base[0], base[a.jac_row_stride], base[2 * a.jac_row_stride]};
    f3 w = {base[3 * a.jac_row_stride], base[4 * a.jac_row_stride], base[5 * a.jac_row_stride]};
    if (a.frame) {
        v = quat_apply(root_q_inv, v);
        w = quat_apply(root_q_inv, w);
        if (a.has_offset) {
            const f3 p = {a.off_pos[0], a.off_pos[1], a.off_pos[2]};
            const f3 pw = cross3(p, w);  // skew(p) w
            v = {v.x - pw.x, v.y - pw.y, v.z - pw.z};
            w = quat_apply(f4{a.off_rot[0], a.off_rot[1], a.off_rot[2], a.off_rot[3]}, w);
        }
    }
    g[0] = v.x, g[1] = v.y, g[2] = v.z;
    if (a.pose) {
        g[3] = w.x, g[4] = w.y, g[5] = w.z;
    } else {
        g[3] = g[4] = g[5] = 0.0f;
    }
}

// In-place Cholesky solve of the symmetric positive definite 6x6 system (float64, fully unrolled).
ILX_DEV void cholesky_solve6(double (&A)[6][6], double (&y)[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < c) A[c][c] -= A[c][k] * A[c][k];
        const double d = sqrt(A[c][c]);
        A[c][c] = d;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r > c) {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < c) A[r][c] -= A[r][k] * A[c][k];
                A[r][c] /= d;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {  // L z = e
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < r) y[r] -= A[r][k] * y[k];
        y[r] /= A[r][r];
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {  // L^T y = z
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > r) y[r] -= A[k][r] * y[k];
        y[r] /= A[r][r];
    }
}

// Cyclic Jacobi eigen-decomposition of the symmetric 6x6 A (float64): on return the diagonal of A holds the
// eigenvalues and the columns of V the eigenvectors.
ILX_DEV void jacobi_eig6(double (&A)[6][6], double (&V)[6][6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            diag += A[r][r] * A[r][r];
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c > r) off += A[r][c] * A[r][c];
        }
        if (off <= 1e-30 * diag || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
#pragma unroll
            for (int q = 1; q < 6; ++q) {
                if (q <= p) continue;
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < 6; ++k) {  // A <- A G (columns p, q)
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = cs * akp - sn * akq;
                    A[k][q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {  // A <- G^T A (rows p, q)
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = cs * apk - sn * aqk;
                    A[q][k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
        }
    }
}

template <int METHOD>
__global__ void __launch_bounds__(64) diff_ik_kernel(IkArgs a) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    f3 pos_b;
    f4 quat_b, root_q;
    frame_pose(a, i, pos_b, quat_b, root_q);
    const f4 root_q_inv = quat_inv(root_q);
    // pose error: differential_ik.py:162-171 (math.py:820-869)
    double y[6];
    {
        const f3 pd = load3(a.pos_des + 3 * i);
        y[0] = pd.x - pos_b.x, y[1] = pd.y - pos_b.y, y[2] = pd.z - pos_b.z;
        if (a.pose) {
            const f3 aa = axis_angle_from_quat(quat_error(quat_b, load4u(a.quat_des + 4 * i)));
            y[3] = aa.x, y[4] = aa.y, y[5] = aa.z;
        } else {
            y[3] = y[4] = y[5] = 0.0;
        }
    }
    if (METHOD != ILX_IK_TRANS) {
        double A[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) A[r][c] = 0.0;
        for (int c = 0; c < a.j; ++c) {
            float g[6];
            jac_column(a, i, c, root_q_inv, g);
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int s = 0; s < 6; ++s)
                    if (s <= r) A[r][s] += static_cast<double>(g[r]) * static_cast<double>(g[s]);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int s = 0; s < 6; ++s)
                if (s > r) A[r][s] = A[s][r];
        if (METHOD == ILX_IK_DLS) {  // differential_ik.py:227-236
            const double l2 = static_cast<double>(a.lambda_val) * static_cast<double>(a.lambda_val);
#pragma unroll
            for (int r = 0; r < 6; ++r) A[r][r] += l2;
            cholesky_solve6(A, y);
        } else {  // pinv :197-203, svd :204-219
            double V[6][6];
            jacobi_eig6(A, V);
            double smax2 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) smax2 = fmax(smax2, A[k][k]);
            const int rows = a.pose ? 6 : 3;
            // torch.linalg.pinv on float32 input drops sigma <= max(rows, cols) * eps32 * sigma_max
            const double rtol = static_cast<double>(rows > a.j ? rows : a.j) * 1.1920928955078125e-7;
            double out[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double s2 = A[k][k];
                const double sigma = s2 > 0.0 ? sqrt(s2) : 0.0;
                const bool keep = METHOD == ILX_IK_PINV ? (sigma > rtol * sqrt(smax2)) : (sigma > static_cast<double>(a.min_sv));
                if (!keep) continue;
                double proj = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) proj += V[r][k] * y[r];
                proj /= s2;
#pragma unroll
                for (int r = 0; r < 6; ++r) out[r] += proj * V[r][k];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) y[r] = out[r];
        }
    }
    const double gain = METHOD == ILX_IK_DLS ? 1.0 : static_cast<double>(a.k_val);
    for (int c = 0; c < a.j; ++c) {
        float g[6];
        jac_column(a, i, c, root_q_inv, g);
        double dq = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) dq += static_cast<double>(g[r]) * y[r];
        const int col = a.joint_ids ? a.joint_ids[c] : c;
        const float q = a.joint_pos[a.joint_pos_stride * i + col];
        a.out[a.out_stride * i + col] = q + static_cast<float>(gain * dq);  // differential_ik.py:174
    }
}

// process_actions (task_space_actions.py:155-167) + set_command (differential_ik.py:97-147)
struct IkCmdArgs {
    IkArgs f;                 // frame inputs only
    const float* raw;         // (n, dim) or null when `command` is already processed
    float* raw_store;         // (n, dim) or null
    float* processed;         // (n, dim) or null
    const float* command;     // (n, dim): used when raw is null
    float scale[7];
    const float* clip;        // (n, dim, 2) or null
    int dim;
    int relative;
    float* pos_des;
    float* quat_des;
};

__global__ void __launch_bounds__(256) ik_command_kernel(IkCmdArgs c) {
    const IkArgs& a = c.f;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float cmd[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        if (k >= c.dim) break;
        if (c.raw) {
            const float r = c.raw[i * c.dim + k];
            if (c.raw_store) c.raw_store[i * c.dim + k] = r;
            float v = r * c.scale[k];
            if (c.clip) v = fminf(fmaxf(v, c.clip[2 * (i * c.dim + k)]), c.clip[2 * (i * c.dim + k) + 1]);
            if (c.processed) c.processed[i * c.dim + k] = v;
            cmd[k] = v;
        } else {
            cmd[k] = c.command[i * c.dim + k];
        }
    }
    f3 pos_b = {0, 0, 0};
    f4 quat_b = {1, 0, 0, 0}, root_q;
    if (a.ee_quat) frame_pose(a, i, pos_b, quat_b, root_q);
    f3 pd;
    f4 qd;
    if (!a.pose) {  // differential_ik.py:118-131
        pd = c.relative ? f3{pos_b.x + cmd[0], pos_b.y + cmd[1], pos_b.z + cmd[2]} : f3{cmd[0], cmd[1], cmd[2]};
        qd = quat_b;
    } else if (c.relative) {  // :134-141 (math.py:872-910)
        apply_delta_pose(pos_b, quat_b, cmd, 1.0e-6f, pd, qd);
    } else {  // :143-145
        pd = {cmd[0], cmd[1], cmd[2]};
        qd = {cmd[3], cmd[4], cmd[5], cmd[6]};
    }
    store3(c.pos_des + 3 * i, pd);
    store4u(c.quat_des + 4 * i, qd);
}

static int check_frame(const ilx_ik_frame* f, IkArgs& a, const char* who) {
    ILX_REQUIRE(f->ee_pos && f->ee_quat, ILX_EINVAL, "%s: end-effector pose is NULL", who);
    ILX_REQUIRE(f->ee_pos_stride >= 3 && f->ee_quat_stride >= 4, ILX_EINVAL, "%s: bad end-effector pose strides", who);
    a.ee_pos = f->ee_pos, a.ee_pos_stride = f->ee_pos_stride, a.ee_quat = f->ee_quat, a.ee_quat_stride = f->ee_quat_stride;
    a.frame = f->root_pos != nullptr;
    if (a.frame) {
        ILX_REQUIRE(f->root_quat && f->root_pos_stride >= 3 && f->root_quat_stride >= 4, ILX_EINVAL, "%s: bad root pose", who);
        a.root_pos = f->root_pos, a.root_pos_stride = f->root_pos_stride;
        a.root_quat = f->root_quat, a.root_quat_stride = f->root_quat_stride;
        a.has_offset = f->has_offset;
        for (int k = 0; k < 3; ++k) a.off_pos[k] = f->offset_pos[k];
        for (int k = 0; k < 4; ++k) a.off_rot[k] = f->offset_rot[k];
    }
    return ILX_OK;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_diff_ik_set_command(const ilx_ik_frame* frame, int command_type, int use_relative_mode,
                                       const float* raw_actions, float* raw_store, float* processed, const float* scale_host,
                                       const float* clip, const float* command, float* ee_pos_des, float* ee_quat_des,
                                       int64_t n, void* stream) {
    ILX_REQUIRE(command_type == ILX_IK_POSITION || command_type == ILX_IK_POSE, ILX_EINVAL,
                "ilx_diff_ik_set_command: unknown command type %d", command_type);
    ILX_REQUIRE(n >= 0, ILX_EINVAL, "ilx_diff_ik_set_command: negative env count");
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(ee_pos_des && ee_quat_des, ILX_EINVAL, "ilx_diff_ik_set_command: desired pose buffers are NULL");
    ILX_REQUIRE((raw_actions != nullptr) != (command != nullptr), ILX_EINVAL,
                "ilx_diff_ik_set_command: pass raw_actions (+ scale) or an already processed command, not both");
    ILX_REQUIRE(!raw_actions || scale_host, ILX_EINVAL, "ilx_diff_ik_set_command: raw_actions need scale");
    IkCmdArgs c{};
    c.f.n = n, c.f.pose = command_type == ILX_IK_POSE;
    c.dim = command_type == ILX_IK_POSITION ? 3 : (use_relative_mode ? 6 : 7);  // differential_ik.py:79-86
    const bool needs_pose = command_type == ILX_IK_POSITION || use_relative_mode;
    if (needs_pose || (frame && frame->ee_quat)) {
        // differential_ik.py:118-141: the current orientation is always needed in position mode, the full pose in relative mode
        ILX_REQUIRE(frame, ILX_EINVAL, "ilx_diff_ik_set_command: this command type needs the current end-effector pose");
        const int rc = check_frame(frame, c.f, "ilx_diff_ik_set_command");
        if (rc != ILX_OK) return rc;
    }
    c.raw = raw_actions, c.raw_store = raw_store, c.processed = processed, c.command = command, c.clip = clip;
    for (int k = 0; k < 7; ++k) c.scale[k] = (raw_actions && k < c.dim) ? scale_host[k] : 1.0f;
    c.relative = use_relative_mode;
    c.pos_des = ee_pos_des, c.quat_des = ee_quat_des;
    ILX_LAUNCH(ik_command_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), c);
    ILX_LAUNCH_CHECK("ilx_diff_ik_set_command");
    return ILX_OK;
}

extern "C" int ilx_diff_ik_compute(const ilx_ik_frame* frame, int command_type, int ik_method, float k_val, float lambda_val,
                                   float min_singular_value, const float* jacobian, int64_t jacobian_env_stride,
                                   int64_t jacobian_row_stride, const int32_t* jacobian_cols, const float* joint_pos,
                                   int64_t joint_pos_stride, const int32_t* joint_ids, const float* ee_pos_des,
                                   const float* ee_quat_des, float* joint_pos_des, int64_t joint_pos_des_stride, int64_t n,
                                   int num_joints, void* stream) {
    ILX_REQUIRE(command_type == ILX_IK_POSITION || command_type == ILX_IK_POSE, ILX_EINVAL,
                "ilx_diff_ik_compute: unknown command type %d", command_type);
    ILX_REQUIRE(ik_method >= ILX_IK_PINV && ik_method <= ILX_IK_DLS, ILX_EINVAL, "ilx_diff_ik_compute: unknown method %d",
                ik_method);
    ILX_REQUIRE(n >= 0 && num_joints > 0, ILX_EINVAL, "ilx_diff_ik_compute: bad shape (%lld, %d)", (long long)n, num_joints);
    if (n == 0) return ILX_OK;
    ILX_REQUIRE(frame && jacobian && joint_pos && ee_pos_des && joint_pos_des, ILX_EINVAL, "ilx_diff_ik_compute: NULL operand");
    ILX_REQUIRE(command_type == ILX_IK_POSITION || ee_quat_des, ILX_EINVAL, "ilx_diff_ik_compute: ee_quat_des is NULL");
    ILX_REQUIRE(jacobian_row_stride >= num_joints && jacobian_env_stride >= 6 * jacobian_row_stride, ILX_EINVAL,
                "ilx_diff_ik_compute: bad Jacobian strides");
    ILX_REQUIRE(joint_pos_stride >= num_joints && joint_pos_des_stride >= num_joints, ILX_EINVAL,
                "ilx_diff_ik_compute: bad joint strides");
    IkArgs a{};
    a.n = n, a.j = num_joints, a.pose = command_type == ILX_IK_POSE, a.method = ik_method;
    a.k_val = k_val, a.lambda_val = lambda_val, a.min_sv = min_singular_value;
    const int rc = check_frame(frame, a, "ilx_diff_ik_compute");
    if (rc != ILX_OK) return rc;
    a.jac = jacobian, a.jac_env_stride = jacobian_env_stride, a.jac_row_stride = jacobian_row_stride, a.jac_cols = jacobian_cols;
    a.joint_pos = joint_pos, a.joint_pos_stride = joint_pos_stride, a.joint_ids = joint_ids;
    a.pos_des = ee_pos_des, a.quat_des = ee_quat_des;
    a.out = joint_pos_des, a.out_stride = joint_pos_des_stride;
    const dim3 grid(blocks_for(n, 64)), block(64);
    switch (ik_method) {
        case ILX_IK_PINV: ILX_LAUNCH(diff_ik_kernel<ILX_IK_PINV>, grid, block, 0, as_stream(stream), a); break;
        case ILX_IK_SVD: ILX_LAUNCH(diff_ik_kernel<ILX_IK_SVD>, grid, block, 0, as_stream(stream), a); break;
        case ILX_IK_TRANS: ILX_LAUNCH(diff_ik_kernel<ILX_IK_TRANS>, grid, block, 0, as_stream(stream), a); break;
        default: ILX_LAUNCH(diff_ik_kernel<ILX_IK_DLS>, grid, block, 0, as_stream(stream), a); break;
    }
    ILX_LAUNCH_CHECK("ilx_diff_ik_compute");
    return ILX_OK;
}
