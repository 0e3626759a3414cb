// Operational-space controller (reference: isaaclab/controllers/operational_space.py).  One thread per env; every matrix
// of the controller is at most 6x6 or num_dof x num_dof (num_dof <= 12), kept in the thread's registers / scratch.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd_osc.h"

namespace ilx {

constexpr int kDof = ILX_OSC_MAX_DOF;

// R diag(d) R^T for a 3-vector d: the 3x3 block of a gain / selection matrix moved from the task to the root frame
// (operational_space.py:299-327; the task-frame matrices are diagonal by construction, :66-100, 236-262).
ILX_DEV void rotate_diag3(const float* R, const float* d, float* out6, int off) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            // (R @ D) @ R^T, left to right as torch.matmul evaluates the chain
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = fmaf_(R[3 * r + k] * d[k], R[3 * c + k], acc);
            out6[6 * (off + r) + off + c] = acc;
        }
}

ILX_DEV void zero36(float* m) {
    for (int i = 0; i < 36; ++i) m[i] = 0.0f;
}

struct OscCmdArgs {
    ilx_osc o;
    const float* command;
    int action_dim, target_dim;
    const float* ee_pose;
    const float* task_pose;
};

__global__ void __launch_bounds__(128) osc_set_command_kernel(OscCmdArgs a) {
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const ilx_osc& o = a.o;
    if (env >= o.num_envs) return;
    const float* cmd = a.command + env * a.action_dim;
    // -- impedance parameters (:216-262)
    float p[6], d[6];
    float* pt = o.motion_p_gains_task + 36 * env;
    float* dt = o.motion_d_gains_task + 36 * env;
    if (o.impedance_mode == ILX_OSC_IMPEDANCE_FIXED) {
        for (int k = 0; k < 6; ++k) p[k] = pt[7 * k], d[k] = dt[7 * k];
    } else {
        for (int k = 0; k < 6; ++k) {
            const float stiff = fminf(fmaxf(cmd[a.target_dim + k], o.motion_stiffness_limits_task[0]), o.motion_stiffness_limits_task[1]);
            p[k] = o.motion_control_axes_task[k] * stiff;
            float ratio = o.motion_damping_ratio_task[k];
            if (o.impedance_mode == ILX_OSC_IMPEDANCE_VARIABLE)
                ratio = fminf(fmaxf(cmd[a.target_dim + 6 + k], o.motion_damping_ratio_limits_task[0]), o.motion_damping_ratio_limits_task[1]);
            d[k] = 2.0f * sqrt_(p[k]) * ratio;
        }
        zero36(pt), zero36(dt);
        for (int k = 0; k < 6; ++k) pt[7 * k] = p[k], dt[7 * k] = d[k];
    }
    float* tgt = o.task_space_target_task + env * a.target_dim;
    for (int k = 0; k < a.target_dim; ++k) tgt[k] = cmd[k];
    // -- task frame (:268-272)
    f3 tp = {0.0f, 0.0f, 0.0f};
    f4 tq = {1.0f, 0.0f, 0.0f, 0.0f};
    if (a.task_pose) tp = load3(a.task_pose + 7 * env), tq = load4u(a.task_pose + 7 * env + 3);
    // -- targets (:275-297)
    bool has_pose = false, has_wrench = false;
    f3 dp = {0.0f, 0.0f, 0.0f};
    f4 dq = {1.0f, 0.0f, 0.0f, 0.0f};
    float wrench[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int at = 0;
    for (int g = 0; g < o.num_targets; ++g) {
        const int type = o.target_types[g];
        if (type == ILX_OSC_POSE_REL) {
            f3 cp;
            f4 cq;
            subtract_frame_transforms(tp, tq, load3(a.ee_pose + 7 * env), load4u(a.ee_pose + 7 * env + 3), cp, cq);
            apply_delta_pose(cp, cq, cmd + at, 1.0e-6f, dp, dq);
            has_pose = true;
            at += 6;
        } else if (type == ILX_OSC_POSE_ABS) {
            dp = load3(cmd + at), dq = load4u(cmd + at + 3);
            has_pose = true;
            at += 7;
        } else {
            for (int k = 0; k < 6; ++k) wrench[k] = cmd[at + k];
            has_wrench = true;
            at += 6;
        }
    }
    // -- gains and selection matrices in the root frame (:299-327)
    float R[9];
    matrix_from_quat(tq, R);
    float* pb = o.motion_p_gains_b + 36 * env;
    float* db = o.motion_d_gains_b + 36 * env;
    zero36(pb), zero36(db);
    rotate_diag3(R, p, pb, 0), rotate_diag3(R, p + 3, pb, 3);
    rotate_diag3(R, d, db, 0), rotate_diag3(R, d + 3, db, 3);
    if (o.contact_wrench_p_gains_task && o.contact_wrench_p_gains_b) {
        float w[6];
        for (int k = 0; k < 6; ++k) w[k] = o.contact_wrench_p_gains_task[36 * env + 7 * k];
        float* wb = o.contact_wrench_p_gains_b + 36 * env;
        zero36(wb);
        rotate_diag3(R, w, wb, 0), rotate_diag3(R, w + 3, wb, 3);
    }
    float* sm = o.selection_matrix_motion_b + 36 * env;
    float* sf = o.selection_matrix_force_b + 36 * env;
    zero36(sm), zero36(sf);
    rotate_diag3(R, o.motion_control_axes_task, sm, 0), rotate_diag3(R, o.motion_control_axes_task + 3, sm, 3);
    rotate_diag3(R, o.contact_wrench_control_axes_task, sf, 0), rotate_diag3(R, o.contact_wrench_control_axes_task + 3, sf, 3);
    // -- targets in the root frame (:329-343)
    if (has_pose) {
        float* t = o.desired_ee_pose_task + 7 * env;
        store3(t, dp);
        t[3] = dq.w, t[4] = dq.x, t[5] = dq.y, t[6] = dq.z;
        f3 bp;
        f4 bq;
        combine_frame_transforms(tp, tq, dp, dq, bp, bq);
        float* b = o.desired_ee_pose_b + 7 * env;
        store3(b, bp);
        b[3] = bq.w, b[4] = bq.x, b[5] = bq.y, b[6] = bq.z;
    }
    if (has_wrench) {
        float* t = o.desired_ee_wrench_task + 6 * env;
        for (int k = 0; k < 6; ++k) t[k] = wrench[k];
        float f[3], mo[3];
        for (int r = 0; r < 3; ++r) {
            f[r] = (R[3 * r] * wrench[0] + R[3 * r + 1] * wrench[1]) + R[3 * r + 2] * wrench[2];
            mo[r] = (R[3 * r] * wrench[3] + R[3 * r + 1] * wrench[4]) + R[3 * r + 2] * wrench[5];
        }
        const f3 lever = cross3(tp, f3{f[0], f[1], f[2]});
        float* b = o.desired_ee_wrench_b + 6 * env;
        b[0] = f[0], b[1] = f[1], b[2] = f[2];
        b[3] = mo[0] + lever.x, b[4] = mo[1] + lever.y, b[5] = mo[2] + lever.z;
    }
}

// In-place Gauss-Jordan inverse with partial pivoting of the leading n x n block of a (row stride ld), double precision.
template <int LD>
ILX_DEV void invert(double (&a)[LD][LD], int n) {
    double inv[LD][LD];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) inv[r][c] = r == c ? 1.0 : 0.0;
    for (int col = 0; col < n; ++col) {
        int piv = col;
        double best = fabs(a[col][col]);
        for (int r = col + 1; r < n; ++r)
            if (fabs(a[r][col]) > best) best = fabs(a[r][col]), piv = r;
        if (piv != col)
            for (int c = 0; c < n; ++c) {
                const double t0 = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t0;
                const double t1 = inv[col][c]; inv[col][c] = inv[piv][c]; inv[piv][c] = t1;
            }
        const double s = 1.0 / a[col][col];
        for (int c = 0; c < n; ++c) a[col][c] *= s, inv[col][c] *= s;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f == 0.0) continue;
            for (int c = 0; c < n; ++c) a[r][c] -= f * a[col][c], inv[r][c] -= f * inv[col][c];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) a[r][c] = inv[r][c];
}

struct OscArgs {
    ilx_osc o;
    int has_pose, has_wrench, n_dof;
    const float* jac;
    const float* ee_pose;
    const float* ee_vel;
    const float* ee_force;
    const float* mass;
    const float* gravity;
    const float* q;
    const float* qd;
    const float* q_null;
    float* tau;
};

__global__ void __launch_bounds__(64) osc_compute_kernel(OscArgs a) {
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const ilx_osc& o = a.o;
    if (env >= o.num_envs) return;
    const int J = a.n_dof;
    const float* jac = a.jac + env * 6 * J;  // (6, J)
    double tau[kDof];
    for (int j = 0; j < J; ++j) tau[j] = 0.0;
    double Minv[kDof][kDof];
    double Lam[6][6];  // operational-space mass matrix
    bool have_minv = false;
    const bool decouple = o.inertial_dynamics_decoupling != 0, partial = o.partial_inertial_dynamics_decoupling != 0;
    // ---- motion control (:400-447)
    if (a.has_pose) {
        const float* dpose = o.desired_ee_pose_b + 7 * env;
        const f3 cp = load3(a.ee_pose + 7 * env), tp = load3(dpose);
        const f4 cq = load4u(a.ee_pose + 7 * env + 3), tq = load4u(dpose + 3);
        // compute_pose_error(..., "axis_angle") (math.py:820-869)
        const f3 aa = axis_angle_from_quat(quat_error(cq, tq));
        const float err[6] = {tp.x - cp.x, tp.y - cp.y, tp.z - cp.z, aa.x, aa.y, aa.z};
        const float* pb = o.motion_p_gains_b + 36 * env;
        const float* db = o.motion_d_gains_b + 36 * env;
        const float* vel = a.ee_vel + 6 * env;
        double acc[6];
        for (int r = 0; r < 6; ++r) {
            double s = 0.0;
            for (int c = 0; c < 6; ++c) s += static_cast<double>(pb[6 * r + c]) * err[c] - static_cast<double>(db[6 * r + c]) * vel[c];
            acc[r] = s;
        }
        double force[6];
        if (decouple) {
            for (int r = 0; r < J; ++r)
                for (int c = 0; c < J; ++c) Minv[r][c] = a.mass[(env * J + r) * J + c];
            invert<kDof>(Minv, J);
            have_minv = true;
            // J Minv J^T
            double JM[6][kDof];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < J; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < J; ++k) s += static_cast<double>(jac[r * J + k]) * Minv[k][c];
                    JM[r][c] = s;
                }
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < J; ++k) s += JM[r][k] * jac[c * J + k];
                    Lam[r][c] = s;
                }
            if (partial) {  // translation and rotation blocks inverted separately, no coupling (:421-428)
                double A3[3][3], B3[3][3];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) A3[r][c] = Lam[r][c], B3[r][c] = Lam[3 + r][3 + c];
                invert<3>(A3, 3), invert<3>(B3, 3);
                // the off-diagonal blocks keep what the buffer held: the reference never writes them (zeros from :83)
                const float* old = o.os_mass_matrix_b + 36 * env;
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) Lam[r][c] = old[6 * r + c];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) Lam[r][c] = A3[r][c], Lam[3 + r][3 + c] = B3[r][c];
            } else {
                invert<6>(Lam, 6);
            }
            float* out = o.os_mass_matrix_b + 36 * env;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) out[6 * r + c] = static_cast<float>(Lam[r][c]);
            for (int r = 0; r < 6; ++r) {
                double s = 0.0;
                for (int c = 0; c < 6; ++c) s += Lam[r][c] * acc[c];
                force[r] = s;
            }
        } else {
            for (int r = 0; r < 6; ++r) force[r] = acc[r];
        }
        const float* sm = o.selection_matrix_motion_b + 36 * env;
        double sel[6];
        for (int r = 0; r < 6; ++r) {
            double s = 0.0;
            for (int c = 0; c < 6; ++c) s += static_cast<double>(sm[6 * r + c]) * force[c];
            sel[r] = s;
        }
        for (int j = 0; j < J; ++j) {
            double s = 0.0;
            for (int r = 0; r < 6; ++r) s += static_cast<double>(jac[r * J + j]) * sel[r];
            tau[j] += s;
        }
    }
    // ---- wrench control (:449-471)
    if (a.has_wrench) {
        const float* w = o.desired_ee_wrench_b + 6 * env;
        double cmdw[6];
        for (int r = 0; r < 6; ++r) cmdw[r] = w[r];
        if (o.contact_wrench_p_gains_b) {
            float* cw = o.ee_contact_wrench_b + 6 * env;
            for (int r = 0; r < 3; ++r) cw[r] = a.ee_force[3 * env + r], cw[3 + r] = w[3 + r];
            const float* g = o.contact_wrench_p_gains_b + 36 * env;
            for (int r = 0; r < 6; ++r) {
                double s = 0.0;
                for (int c = 0; c < 6; ++c) s += static_cast<double>(g[6 * r + c]) * (w[c] - cw[c]);
                cmdw[r] += s;
            }
        }
        const float* sf = o.selection_matrix_force_b + 36 * env;
        double sel[6];
        for (int r = 0; r < 6; ++r) {
            double s = 0.0;
            for (int c = 0; c < 6; ++c) s += static_cast<double>(sf[6 * r + c]) * cmdw[c];
            sel[r] = s;
        }
        for (int j = 0; j < J; ++j) {
            double s = 0.0;
            for (int r = 0; r < 6; ++r) s += static_cast<double>(jac[r * J + j]) * sel[r];
            tau[j] += s;
        }
    }
    // ---- gravity compensation (:473-478)
    if (o.gravity_compensation)
        for (int j = 0; j < J; ++j) tau[j] += a.gravity[env * J + j];
    // ---- null-space joint position control (:480-546)
    if (o.nullspace_position) {
        double Jpt[6][kDof];  // transpose of the (dynamically consistent or Moore-Penrose) pseudo-inverse: (6, J)
        if (decouple && !partial && have_minv) {
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < J; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < 6; ++k) {
                        double jm = 0.0;
                        for (int l = 0; l < J; ++l) jm += static_cast<double>(jac[k * J + l]) * Minv[l][c];
                        s += Lam[r][k] * jm;
                    }
                    Jpt[r][c] = s;
                }
        } else {  // pinv(J)^T = (J J^T)^-1 J for a Jacobian of full row rank
            double G[6][6];
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < J; ++k) s += static_cast<double>(jac[r * J + k]) * jac[c * J + k];
                    G[r][c] = s;
                }
            invert<6>(G, 6);
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < J; ++c) {
                    double s = 0.0;
                    for (int k = 0; k < 6; ++k) s += G[r][k] * jac[k * J + c];
                    Jpt[r][c] = s;
                }
        }
        double accn[kDof], v[kDof];
        for (int j = 0; j < J; ++j) {
            const float target = a.q_null ? a.q_null[env * J + j] : 0.0f;
            const float e = target - a.q[env * J + j];
            accn[j] = static_cast<double>(o.nullspace_p_gain * e + o.nullspace_d_gain * (-a.qd[env * J + j]));
        }
        if (a.mass) {
            for (int r = 0; r < J; ++r) {
                double s = 0.0;
                for (int c = 0; c < J; ++c) s += static_cast<double>(a.mass[(env * J + r) * J + c]) * accn[c];
                v[r] = s;
            }
        } else {
            for (int r = 0; r < J; ++r) v[r] = accn[r];
        }
        // (I - J^T Jpt) v
        double jv[6];
        for (int r = 0; r < 6; ++r) {
            double s = 0.0;
            for (int c = 0; c < J; ++c) s += Jpt[r][c] * v[c];
            jv[r] = s;
        }
        for (int j = 0; j < J; ++j) {
            double s = v[j];
            for (int r = 0; r < 6; ++r) s -= static_cast<double>(jac[r * J + j]) * jv[r];
            tau[j] += s;
        }
    }
    for (int j = 0; j < J; ++j) a.tau[env * J + j] = static_cast<float>(tau[j]);
}

static int target_dim_of(const ilx_osc& o) {
    int dim = 0;
    for (int g = 0; g < o.num_targets; ++g) dim += o.target_types[g] == ILX_OSC_POSE_ABS ? 7 : 6;
    return dim;
}

static int check_cfg(const ilx_osc& o, const char* who) {
    ILX_REQUIRE(o.num_envs >= 0, ILX_EINVAL, "%s: negative env count", who);
    ILX_REQUIRE(o.num_targets >= 1 && o.num_targets <= 3, ILX_EINVAL, "%s: %d task-space targets (1..3)", who, o.num_targets);
    for (int g = 0; g < o.num_targets; ++g)
        ILX_REQUIRE(o.target_types[g] >= ILX_OSC_POSE_REL && o.target_types[g] <= ILX_OSC_WRENCH_ABS, ILX_EINVAL,
                    "%s: Invalid control command: %d.", who, o.target_types[g]);
    ILX_REQUIRE(o.impedance_mode >= ILX_OSC_IMPEDANCE_FIXED && o.impedance_mode <= ILX_OSC_IMPEDANCE_VARIABLE, ILX_EINVAL,
                "%s: Invalid impedance mode: %d.", who, o.impedance_mode);
    return ILX_OK;
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_osc_set_command(const ilx_osc* osc, const float* command, int action_dim, const float* current_ee_pose_b,
                                   const float* current_task_frame_pose_b, void* stream) {
    ILX_REQUIRE(osc, ILX_EINVAL, "ilx_osc_set_command: descriptor is NULL");
    const ilx_osc& o = *osc;
    if (const int s = check_cfg(o, "ilx_osc_set_command")) return s;
    const int target_dim = target_dim_of(o);
    const int want = target_dim + (o.impedance_mode == ILX_OSC_IMPEDANCE_FIXED ? 0 : o.impedance_mode == ILX_OSC_IMPEDANCE_VARIABLE_KP ? 6 : 12);
    ILX_REQUIRE(action_dim == want, ILX_ESHAPE, "ilx_osc_set_command: Invalid command shape '(%lld, %d)'. Expected: '(%lld, %d)'.",
                (long long)o.num_envs, action_dim, (long long)o.num_envs, want);
    if (o.num_envs == 0) return ILX_OK;
    bool rel = false, pose = false, wrench = false;
    for (int g = 0; g < o.num_targets; ++g) {
        rel |= o.target_types[g] == ILX_OSC_POSE_REL;
        pose |= o.target_types[g] != ILX_OSC_WRENCH_ABS;
        wrench |= o.target_types[g] == ILX_OSC_WRENCH_ABS;
    }
    ILX_REQUIRE(!rel || current_ee_pose_b, ILX_EINVAL, "ilx_osc_set_command: Current pose is required for 'pose_rel' command.");
    ILX_REQUIRE(command && o.task_space_target_task && o.motion_p_gains_task && o.motion_d_gains_task && o.motion_p_gains_b &&
                    o.motion_d_gains_b && o.selection_matrix_motion_b && o.selection_matrix_force_b,
                ILX_EINVAL, "ilx_osc_set_command: NULL operand");
    ILX_REQUIRE(!pose || (o.desired_ee_pose_task && o.desired_ee_pose_b), ILX_EINVAL, "ilx_osc_set_command: pose target buffers are NULL");
    ILX_REQUIRE(!wrench || (o.desired_ee_wrench_task && o.desired_ee_wrench_b), ILX_EINVAL,
                "ilx_osc_set_command: wrench target buffers are NULL");
    OscCmdArgs a{o, command, action_dim, target_dim, current_ee_pose_b, current_task_frame_pose_b};
    ILX_LAUNCH(osc_set_command_kernel, dim3(blocks_for(o.num_envs, 128)), dim3(128), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_osc_set_command");
    return ILX_OK;
}

extern "C" int ilx_osc_compute(const ilx_osc* osc, int has_pose_target, int has_wrench_target, const float* jacobian_b, int num_dof,
                               const float* current_ee_pose_b, const float* current_ee_vel_b, const float* current_ee_force_b,
                               const float* mass_matrix, const float* gravity, const float* current_joint_pos,
                               const float* current_joint_vel, const float* nullspace_joint_pos_target, float* joint_efforts,
                               void* stream) {
    ILX_REQUIRE(osc, ILX_EINVAL, "ilx_osc_compute: descriptor is NULL");
    const ilx_osc& o = *osc;
    if (const int s = check_cfg(o, "ilx_osc_compute")) return s;
    ILX_REQUIRE(num_dof >= 1 && num_dof <= ILX_OSC_MAX_DOF, ILX_ESHAPE, "ilx_osc_compute: %d degrees of freedom (1..%d)", num_dof,
                ILX_OSC_MAX_DOF);
    if (o.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(jacobian_b && joint_efforts, ILX_EINVAL, "ilx_osc_compute: NULL operand");
    if (has_pose_target) {
        ILX_REQUIRE(current_ee_pose_b && current_ee_vel_b, ILX_EINVAL,
                    "ilx_osc_compute: Current end-effector pose and velocity are required for motion control.");
        ILX_REQUIRE(o.desired_ee_pose_b && o.motion_p_gains_b && o.motion_d_gains_b && o.selection_matrix_motion_b, ILX_EINVAL,
                    "ilx_osc_compute: motion control buffers are NULL");
        if (o.inertial_dynamics_decoupling) {
            ILX_REQUIRE(mass_matrix, ILX_EINVAL, "ilx_osc_compute: Mass matrix is required for inertial decoupling.");
            ILX_REQUIRE(o.os_mass_matrix_b, ILX_EINVAL, "ilx_osc_compute: os_mass_matrix_b is NULL");
        }
    }
    if (has_wrench_target) {
        ILX_REQUIRE(o.desired_ee_wrench_b && o.selection_matrix_force_b, ILX_EINVAL, "ilx_osc_compute: wrench control buffers are NULL");
        if (o.contact_wrench_p_gains_b) {
            ILX_REQUIRE(current_ee_force_b, ILX_EINVAL,
                        "ilx_osc_compute: Current end-effector force is required for closed-loop force control.");
            ILX_REQUIRE(o.ee_contact_wrench_b, ILX_EINVAL, "ilx_osc_compute: ee_contact_wrench_b is NULL");
        }
    }
    ILX_REQUIRE(!o.gravity_compensation || gravity, ILX_EINVAL, "ilx_osc_compute: Gravity vector is required for gravity compensation.");
    if (o.nullspace_position) {
        ILX_REQUIRE(num_dof > 6, ILX_EINVAL, "ilx_osc_compute: Null-space control is only applicable for redundant manipulators.");
        ILX_REQUIRE(!(o.inertial_dynamics_decoupling && !o.partial_inertial_dynamics_decoupling) || (mass_matrix && has_pose_target),
                    ILX_EINVAL, "ilx_osc_compute: Mass matrix inverse is required for dynamically consistent pseudo-inverse");
        ILX_REQUIRE(current_joint_pos && current_joint_vel, ILX_EINVAL,
                    "ilx_osc_compute: Current joint positions and velocities are required for null-space control.");
    }
    OscArgs a{o, has_pose_target, has_wrench_target, num_dof, jacobian_b, current_ee_pose_b, current_ee_vel_b, current_ee_force_b,
              mass_matrix, gravity, current_joint_pos, current_joint_vel, nullspace_joint_pos_target, joint_efforts};
    ILX_LAUNCH(osc_compute_kernel, dim3(blocks_for(o.num_envs, 64)), dim3(64), 0, as_stream(stream), a);
    ILX_LAUNCH_CHECK("ilx_osc_compute");
    return ILX_OK;
}
