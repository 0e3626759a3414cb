// State assembly of OperationalSpaceControllerAction (reference: isaaclab/envs/mdp/actions/task_space_actions.py:568-662):
// end-effector pose, velocity and contact force in the root frame (with the body offset), the end-effector Jacobian in
// the root frame (offset-corrected), and the gathers of mass matrix, gravity compensation and joint state for the
// controlled joints.  One thread per env; the controller itself is ilx_osc_compute.
#include "common.h"
#include "math_dev.h"

#include "../../include/isaaclab_amd_osc.h"

namespace ilx {

__global__ void __launch_bounds__(128) osc_action_state_kernel(ilx_osc_action_state s) {
    const int64_t env = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (env >= s.num_envs) return;
    const int dof = s.num_dof;
    const float* root = s.root_state_w + 13 * env;
    const f3 root_p = {root[0], root[1], root[2]};
    const f4 root_q = {root[3], root[4], root[5], root[6]};
    const float* ee = s.body_state_w + (env * s.num_bodies + s.ee_body_idx) * 13;
    const f3 ee_p = {ee[0], ee[1], ee[2]};
    const f4 ee_q = {ee[3], ee[4], ee[5], ee[6]};
    const f3 off_p = {s.offset_pos[0], s.offset_pos[1], s.offset_pos[2]};
    const f4 off_q = {s.offset_rot[0], s.offset_rot[1], s.offset_rot[2], s.offset_rot[3]};
    // ---- pose (:597-615)
    f3 pb;
    f4 qb;
    subtract_frame_transforms(root_p, root_q, ee_p, ee_q, pb, qb);
    f3 pbo = pb;
    f4 qbo = qb;
    if (s.has_offset) combine_frame_transforms(pb, qb, off_p, off_q, pbo, qbo);
    if (s.ee_pose_w) {
        store3(s.ee_pose_w + 7 * env, ee_p);
        store4u(s.ee_pose_w + 7 * env + 3, ee_q);
    }
    if (s.ee_pose_b_no_offset) {
        store3(s.ee_pose_b_no_offset + 7 * env, pb);
        store4u(s.ee_pose_b_no_offset + 7 * env + 3, qb);
    }
    store3(s.ee_pose_b + 7 * env, pbo);
    store4u(s.ee_pose_b + 7 * env + 3, qbo);
    if (s.pose_only) return;
    // ---- velocity (:617-634)
    const f3 rel_lin = {ee[7] - root[7], ee[8] - root[8], ee[9] - root[9]};
    const f3 rel_ang = {ee[10] - root[10], ee[11] - root[11], ee[12] - root[12]};
    f3 lin_b = quat_rotate_inverse(root_q, rel_lin);
    const f3 ang_b = quat_rotate_inverse(root_q, rel_ang);
    if (s.has_offset) {
        const f3 r = quat_rotate(qb, off_p);
        const f3 c = cross3(ang_b, r);
        lin_b = {lin_b.x + c.x, lin_b.y + c.y, lin_b.z + c.z};
    }
    if (s.ee_vel_w)
        for (int k = 0; k < 6; ++k) s.ee_vel_w[6 * env + k] = ee[7 + k];
    store3(s.ee_vel_b + 6 * env, lin_b);
    store3(s.ee_vel_b + 6 * env + 3, ang_b);
    // ---- contact force (:636-643)
    if (s.ee_force_w && s.ee_force_b) store3(s.ee_force_b + 3 * env, quat_rotate_inverse(root_q, load3(s.ee_force_w + 3 * env)));
    // ---- Jacobian (:400-410, 576-595): rows rotated into the root frame, then the offset correction
    float R[9], Ro[9];
    matrix_from_quat(quat_inv(root_q), R);
    matrix_from_quat(off_q, Ro);
    const float* J = s.jacobians + (env * s.jac_bodies + s.jac_body_idx) * 6 * s.jac_cols;
    float* Jb = s.jacobian_b + env * 6 * dof;
    for (int c = 0; c < dof; ++c) {
        const int col = s.jac_joint_ids[c];
        const float v[3] = {J[col], J[s.jac_cols + col], J[2 * s.jac_cols + col]};
        const float w[3] = {J[3 * s.jac_cols + col], J[4 * s.jac_cols + col], J[5 * s.jac_cols + col]};
        float vb[3], wb[3];
        for (int r = 0; r < 3; ++r) {
            vb[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1]) + R[3 * r + 2] * v[2];
            wb[r] = (R[3 * r] * w[0] + R[3 * r + 1] * w[1]) + R[3 * r + 2] * w[2];
        }
        if (s.has_offset) {
            // v += -skew(offset) w;  rows of -skew(r): (0, r.z, -r.y), (-r.z, 0, r.x), (r.y, -r.x, 0)
            const float add[3] = {(0.0f * wb[0] + off_p.z * wb[1]) + -off_p.y * wb[2], (-off_p.z * wb[0] + 0.0f * wb[1]) + off_p.x * wb[2],
                                  (off_p.y * wb[0] + -off_p.x * wb[1]) + 0.0f * wb[2]};
            for (int r = 0; r < 3; ++r) vb[r] += add[r];
            float wo[3];
            for (int r = 0; r < 3; ++r) wo[r] = (Ro[3 * r] * wb[0] + Ro[3 * r + 1] * wb[1]) + Ro[3 * r + 2] * wb[2];
            for (int r = 0; r < 3; ++r) wb[r] = wo[r];
        }
        for (int r = 0; r < 3; ++r) Jb[r * dof + c] = vb[r], Jb[(3 + r) * dof + c] = wb[r];
    }
    // ---- dynamics and joint state gathers (:568-574, 645-649)
    const int nj = s.num_joints;
    for (int r = 0; r < dof; ++r) {
        const int jr = s.joint_ids[r];
        if (s.mass_matrices)
            for (int c = 0; c < dof; ++c) s.mass_matrix[(env * dof + r) * dof + c] = s.mass_matrices[(env * nj + jr) * nj + s.joint_ids[c]];
        if (s.gravity_forces) s.gravity[env * dof + r] = s.gravity_forces[env * nj + jr];
        s.joint_pos_out[env * dof + r] = s.joint_pos[env * nj + jr];
        s.joint_vel_out[env * dof + r] = s.joint_vel[env * nj + jr];
    }
}

}  // namespace ilx

using namespace ilx;

extern "C" int ilx_osc_action_state_update(const ilx_osc_action_state* state, void* stream) {
    ILX_REQUIRE(state, ILX_EINVAL, "ilx_osc_action_state_update: descriptor is NULL");
    const ilx_osc_action_state& s = *state;
    ILX_REQUIRE(s.num_envs >= 0, ILX_EINVAL, "ilx_osc_action_state_update: negative env count");
    if (s.num_envs == 0) return ILX_OK;
    ILX_REQUIRE(s.root_state_w && s.body_state_w && s.ee_pose_b, ILX_EINVAL, "ilx_osc_action_state_update: root / body state or ee_pose_b is NULL");
    ILX_REQUIRE(s.num_bodies > 0 && s.ee_body_idx >= 0 && s.ee_body_idx < s.num_bodies, ILX_EINVAL,
                "ilx_osc_action_state_update: end-effector body %d outside the %d bodies", s.ee_body_idx, s.num_bodies);
    if (!s.pose_only) {
        ILX_REQUIRE(s.num_dof >= 1 && s.num_dof <= ILX_OSC_MAX_DOF, ILX_ESHAPE, "ilx_osc_action_state_update: %d controlled joints outside 1..%d",
                    s.num_dof, ILX_OSC_MAX_DOF);
        ILX_REQUIRE(s.jacobians && s.joint_ids && s.jac_joint_ids && s.joint_pos && s.joint_vel, ILX_EINVAL,
                    "ilx_osc_action_state_update: Jacobian, joint state or index lists are NULL");
        ILX_REQUIRE(s.ee_vel_b && s.jacobian_b && s.joint_pos_out && s.joint_vel_out, ILX_EINVAL, "ilx_osc_action_state_update: NULL output");
        ILX_REQUIRE((!s.mass_matrices || s.mass_matrix) && (!s.gravity_forces || s.gravity), ILX_EINVAL,
                    "ilx_osc_action_state_update: a dynamics input without its output");
        ILX_REQUIRE(s.jac_bodies > 0 && s.jac_body_idx >= 0 && s.jac_body_idx < s.jac_bodies && s.jac_cols > 0 && s.num_joints > 0, ILX_EINVAL,
                    "ilx_osc_action_state_update: bad Jacobian shape or body index");
    }
    ILX_LAUNCH(osc_action_state_kernel, dim3((s.num_envs + 127) / 128), dim3(128), 0, as_stream(stream), s);
    ILX_LAUNCH_CHECK("ilx_osc_action_state_update");
    return ILX_OK;
}
