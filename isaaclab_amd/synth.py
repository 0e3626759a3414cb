"""Synthetic stand-ins for what the closed simulator provides: terrain meshes and physics state tensors.

PhysX and the USD terrain importer are outside this package; benchmarks and tests need tensors of the same
shape and statistics instead.  Everything is generated on the CPU from an explicit seed and is deterministic.
"""

from __future__ import annotations

import numpy as np
import torch


def heightfield_mesh(heights: np.ndarray, horizontal_scale: float, origin=(0.0, 0.0)):
    """Triangulate a height grid (nx, ny): vertices on the grid, two triangles per cell, in the face order of the
    reference's ``convert_height_field_to_mesh`` (terrains/height_field/utils.py:157-171: row by row, the two halves
    of a cell one after the other, split along the ind0-ind3 diagonal; 81 x 81 -> 12,800 faces)."""
    nx, ny = heights.shape
    xs = origin[0] + np.arange(nx, dtype=np.float32) * np.float32(horizontal_scale)
    ys = origin[1] + np.arange(ny, dtype=np.float32) * np.float32(horizontal_scale)
    gx, gy = np.meshgrid(xs, ys, indexing="ij")
    verts = np.stack([gx, gy, heights.astype(np.float32)], axis=-1).reshape(-1, 3)
    idx = np.arange(nx * ny, dtype=np.int32).reshape(nx, ny)
    i0, i1, i2, i3 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.stack([np.stack([i0, i3, i1], -1), np.stack([i0, i2, i3], -1)], axis=1).reshape(-1, 3).astype(np.int32)
    return verts, tris


def rough_terrain_mesh(seed: int = 0, rows: int = 10, cols: int = 20, tile: float = 8.0, hscale: float = 0.1,
                       border: float = 20.0):
    """A rows x cols field of height-field tiles (random uniform noise, pyramid slopes, stairs) with a flat border,
    mirroring the layout statistics of the reference's rough terrain (10 x 20 tiles of 8 m, 0.1 m cells).

    Returns (vertices (V,3) float32, triangles (F,3) int32, height_at) where ``height_at(x, y)`` gives the
    analytic grid height used to place robots (nearest grid vertex).
    """
    rng = np.random.default_rng(seed)
    n = int(round(tile / hscale))  # cells per tile edge
    nx, ny = rows * n + 1, cols * n + 1
    h = np.zeros((nx, ny), dtype=np.float32)
    for r in range(rows):
        for c in range(cols):
            kind = (r + c) % 3
            level = (r + 1) / rows
            sl = (slice(r * n, (r + 1) * n + 1), slice(c * n, (c + 1) * n + 1))
            ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
            if kind == 0:  # uniform noise, step 0.02, range up to 0.10
                amp = 0.02 + 0.08 * level
                t = np.round(rng.uniform(-amp, amp, size=(n + 1, n + 1)) / 0.02) * 0.02
            elif kind == 1:  # pyramid slope up to 0.4
                slope = 0.4 * level
                dist = np.minimum(np.minimum(ii, n - ii), np.minimum(jj, n - jj)) * hscale
                t = slope * dist
            else:  # stairs, step height 0.05..0.23, 0.3 m wide
                step = 0.05 + 0.18 * level
                ring = np.minimum(np.minimum(ii, n - ii), np.minimum(jj, n - jj)) // 3
                t = step * np.minimum(ring, 8)
            # tile edges at zero so that neighbouring tiles join
            t = t.astype(np.float32)
            t[0, :] = t[-1, :] = 0.0
            t[:, 0] = t[:, -1] = 0.0
            h[sl] = np.maximum(h[sl], t)
    verts, tris = heightfield_mesh(h, hscale)
    # flat border ring made of large triangles
    x0, y0, x1, y1 = 0.0, 0.0, rows * tile, cols * tile
    bx0, by0, bx1, by1 = x0 - border, y0 - border, x1 + border, y1 + border
    quads = [((bx0, by0), (bx1, y0)), ((bx0, y1), (bx1, by1)), ((bx0, y0), (x0, y1)), ((x1, y0), (bx1, y1))]
    bv, bt = [], []
    base = verts.shape[0]
    for (qx0, qy0), (qx1, qy1) in quads:
        k = base + len(bv)
        bv += [(qx0, qy0, 0.0), (qx1, qy0, 0.0), (qx0, qy1, 0.0), (qx1, qy1, 0.0)]
        bt += [(k, k + 1, k + 2), (k + 1, k + 3, k + 2)]
    verts = np.concatenate([verts, np.asarray(bv, dtype=np.float32)])
    tris = np.concatenate([tris, np.asarray(bt, dtype=np.int32)])

    def height_at(x, y):
        i = np.clip(np.round(np.asarray(x) / hscale).astype(np.int64), 0, nx - 1)
        j = np.clip(np.round(np.asarray(y) / hscale).astype(np.int64), 0, ny - 1)
        return h[i, j]

    return verts, tris, height_at


def curriculum_env_origins(n: int, seed: int, rows: int = 10, cols: int = 20, tile: float = 8.0, max_init_level: int = 5):
    """Env origins as the reference's terrain importer deals them for a curriculum terrain
    (terrains/terrain_importer.py:_compute_env_origins_curriculum): env i belongs to terrain type (column)
    floor(i / (n / cols)) and starts at a random level (row) up to ``max_init_level`` (rough_env_cfg: 5); its origin is
    the centre of that tile.  (n, 2) for the tile layout of :func:`rough_terrain_mesh`."""
    g = torch.Generator().manual_seed(seed)
    levels = torch.randint(0, min(max_init_level, rows - 1) + 1, (n,), generator=g)
    types = torch.div(torch.arange(n), n / cols, rounding_mode="floor").to(torch.long).clamp_(max=cols - 1)
    return torch.stack([(levels.float() + 0.5) * tile, (types.float() + 0.5) * tile], dim=-1)


def _quat_from_rpy(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    q = torch.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
                     sy * cr * cp - cy * sr * sp], dim=-1)
    return q / q.norm(dim=-1, keepdim=True)


def locomotion_state(n: int, seed: int, *, num_joints: int, num_bodies: int, default_joint_pos: torch.Tensor,
                     feet_ids, decimation: int = 4, height_at=None, extent=(80.0, 160.0), history_len: int = 3,
                     max_episode_length: int = 1000, joint_vel_std: float = 2.0, with_bodies: bool = False,
                     env_origins: torch.Tensor | None = None):
    """Seeded stand-in for what PhysX hands the rollout step of a legged robot on rough terrain.

    Root positions: uniform over ``extent``, or -- with ``env_origins`` (n, >=2), e.g. :func:`curriculum_env_origins` --
    each env's origin + U(-0.5, 0.5) in x and y, where the task's reset event puts a robot
    (velocity_env_cfg.py: ``reset_base`` pose_range).

    Per physics substep k < decimation: joint positions / velocities and net contact forces; once per step: root
    state.  Plus the persistent buffers a running env would hold (previous actions, contact history / timers,
    command state, episode length), pre-rolled so that every decision branch occurs: feet contact ~50 %, thigh /
    shank 5 %, base 1 %; ~0.1 % of the envs time out; ~2 % stand still.
    """
    g = torch.Generator().manual_seed(seed)
    j, b, d = num_joints, num_bodies, decimation
    s = {}
    xy = torch.rand(n, 2, generator=g)
    if env_origins is None:
        xy = xy * torch.tensor([extent[0] - 2.0, extent[1] - 2.0]) + 1.0
    else:
        xy = env_origins[:, :2].float().cpu() + (xy - 0.5)
    z0 = torch.zeros(n) if height_at is None else torch.from_numpy(np.asarray(height_at(xy[:, 0].numpy(), xy[:, 1].numpy())))
    pos = torch.cat([xy, (z0 + 0.6 + (torch.rand(n, generator=g) - 0.5) * 0.1).unsqueeze(-1)], dim=-1).float()
    yaw = (torch.rand(n, generator=g) * 2 - 1) * np.pi
    rp = torch.randn(n, 2, generator=g) * 0.1
    quat = _quat_from_rpy(rp[:, 0], rp[:, 1], yaw)
    s["root_state_w"] = torch.cat([pos, quat, torch.randn(n, 3, generator=g) * 0.5, torch.randn(n, 3, generator=g) * 0.5], dim=-1)
    dq = default_joint_pos.reshape(1, j)
    s["joint_pos"] = torch.stack([(dq + torch.randn(n, j, generator=g) * 0.2).clamp(-2.0, 2.0) for _ in range(d)])
    vel0 = torch.randn(n, j, generator=g) * joint_vel_std
    s["joint_vel"] = torch.stack([vel0 + torch.randn(n, j, generator=g) * 0.05 * (k + 1) for k in range(d)])
    s["previous_joint_vel"] = vel0 - torch.randn(n, j, generator=g) * 0.05
    p_contact = torch.full((b,), 0.05)
    p_contact[0] = 0.01
    p_contact[list(feet_ids)] = 0.5

    def forces():
        on = torch.rand(n, b, generator=g) < p_contact
        f = torch.randn(n, b, 3, generator=g) * torch.tensor([3.0, 3.0, 60.0])
        f[..., 2] = f[..., 2].abs()
        return f * on.unsqueeze(-1)

    s["net_forces"] = torch.stack([forces() for _ in range(d)])
    s["net_forces_w_history"] = torch.stack([forces() for _ in range(history_len)], dim=1)
    in_contact = s["net_forces_w_history"][:, 0].norm(dim=-1) > 1.0
    t = torch.round(torch.rand(n, b, generator=g) * 120) * 0.005
    s["current_contact_time"] = torch.where(in_contact, t, torch.zeros(()))
    s["current_air_time"] = torch.where(in_contact, torch.zeros(()), t)
    s["last_air_time"] = torch.rand(n, b, generator=g) * 0.6
    s["last_contact_time"] = torch.rand(n, b, generator=g) * 0.6
    s["action"] = torch.rand(n, j, generator=g) * 2 - 1       # the action of the previous step
    s["prev_action"] = torch.rand(n, j, generator=g) * 2 - 1
    s["raw_action"] = torch.rand(n, j, generator=g) * 2 - 1   # this step's policy output
    s["command"] = torch.rand(n, 3, generator=g) * 2 - 1
    s["heading_target"] = (torch.rand(n, generator=g) * 2 - 1) * np.pi
    s["is_heading_env"] = torch.ones(n, dtype=torch.bool)
    s["is_standing_env"] = torch.rand(n, generator=g) < 0.02
    s["time_left"] = torch.rand(n, generator=g) * 10.0
    s["command_counter"] = torch.ones(n, dtype=torch.long)
    s["episode_length_buf"] = torch.randint(0, max_episode_length, (n,), generator=g)
    s["pos_error_history"] = torch.randn(n, 3, j, generator=g) * 0.2
    s["vel_history"] = torch.randn(n, 3, j, generator=g) * joint_vel_std
    if with_bodies:  # link states (n, b, 13): near the root, orientation the root's plus a small tilt, own velocities
        bpos = pos.unsqueeze(1) + torch.randn(n, b, 3, generator=g) * torch.tensor([0.15, 0.15, 0.3])
        tilt = torch.randn(n, b, 2, generator=g) * 0.05
        bquat = _quat_from_rpy(rp[:, None, 0] + tilt[..., 0], rp[:, None, 1] + tilt[..., 1], yaw[:, None] + torch.randn(n, b, generator=g) * 0.1)
        s["body_state_w"] = torch.cat([bpos, bquat, torch.randn(n, b, 6, generator=g) * 0.5], dim=-1).float()
    return s


def actuator_mlp_weights(seed: int = 7, dims=(6, 32, 32, 32, 1)):
    """Random-init actuator network (the reference's weight files live on a remote asset server):
    N(0, 1/sqrt(fan_in)) weights, small biases."""
    g = torch.Generator().manual_seed(seed)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    return ws, bs


def actuator_lstm_weights(seed: int = 7, hidden: int = 8, layers: int = 2, head=(32,)):
    """Random-init LSTM actuator network of the ANYdrive shape (``torch.nn.LSTM(2, 8, 2)`` + a small head; the
    reference's file is a remote asset): a dict for ``ActuatorNetLSTMCfg.network``."""
    g = torch.Generator().manual_seed(seed)
    lstm = {}
    for l in range(layers):
        in_l = 2 if l == 0 else hidden
        lstm[f"weight_ih_l{l}"] = torch.randn(4 * hidden, in_l, generator=g) / hidden ** 0.5
        lstm[f"weight_hh_l{l}"] = torch.randn(4 * hidden, hidden, generator=g) / hidden ** 0.5
        lstm[f"bias_ih_l{l}"] = torch.randn(4 * hidden, generator=g) * 0.1
        lstm[f"bias_hh_l{l}"] = torch.randn(4 * hidden, generator=g) * 0.1
    dims = [hidden, *head, 1]
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    return dict(lstm=lstm, weights=ws, biases=bs, activation="softsign")


# ---- manipulator kinematics for the DifferentialIK fixtures ---------------------------------------------------------
# xArm6 revolute chain (joint origins xyz / rpy, every axis = local z), from the robot description the reference
# ships (ROBOT/xarm/xarm6_with_gripper.urdf:79-225).  Data only; the forward kinematics below is the textbook product.
XARM6_JOINT_ORIGINS = (
    ((0.0, 0.0, 0.267), (0.0, 0.0, 0.0)),
    ((0.0, 0.0, 0.0), (-1.5708, 0.0, 0.0)),
    ((0.0535, -0.2845, 0.0), (0.0, 0.0, 0.0)),
    ((0.0775, 0.3425, 0.0), (-1.5708, 0.0, 0.0)),
    ((0.0, 0.0, 0.0), (1.5708, 0.0, 0.0)),
    ((0.076, 0.097, 0.0), (-1.5708, 0.0, 0.0)),
)
XARM6_DEFAULT_JOINT_POS = (0.0, -1.22, -0.78, 0.0, 1.30, 0.0)


def _rot_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _quat_from_matrix(m):
    """(…,3,3) rotation matrices -> wxyz quaternions with w >= 0 (float64)."""
    w = np.sqrt(np.maximum(0.0, 1.0 + m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2])) / 2
    x = np.sqrt(np.maximum(0.0, 1.0 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2])) / 2
    y = np.sqrt(np.maximum(0.0, 1.0 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2])) / 2
    z = np.sqrt(np.maximum(0.0, 1.0 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2])) / 2
    x = np.copysign(x, m[..., 2, 1] - m[..., 1, 2])
    y = np.copysign(y, m[..., 0, 2] - m[..., 2, 0])
    z = np.copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return np.stack([w, x, y, z], -1)


def serial_arm_kinematics(joint_pos: np.ndarray, root_pos: np.ndarray, root_quat: np.ndarray, origins=XARM6_JOINT_ORIGINS):
    """World pose of the last link and its geometric Jacobian (N,6,J) for a z-axis revolute chain (float64 inside).

    Returns ``ee_pos_w (N,3), ee_quat_w (N,4 wxyz), jacobian_w (N,6,J)`` as float32 arrays: what a fixed-base
    articulation's ``body_pos_w/body_quat_w`` and ``root_physx_view.get_jacobians()`` would hold for that link."""
    q = np.asarray(joint_pos, dtype=np.float64)
    n, nj = q.shape
    w, x, y, z = (np.asarray(root_quat, dtype=np.float64)[:, k] for k in range(4))
    rot = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                    np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                    np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
    pos = np.asarray(root_pos, dtype=np.float64).copy()
    axes, anchors = [], []
    for k in range(nj):
        xyz, rpy = origins[k]
        pos = pos + np.einsum("nij,j->ni", rot, np.array(xyz))
        rot = rot @ _rot_rpy(*rpy)
        axes.append(rot[:, :, 2].copy())
        anchors.append(pos.copy())
        c, s = np.cos(q[:, k]), np.sin(q[:, k])
        rz = np.zeros((n, 3, 3))
        rz[:, 0, 0], rz[:, 0, 1], rz[:, 1, 0], rz[:, 1, 1], rz[:, 2, 2] = c, -s, s, c, 1.0
        rot = rot @ rz
    jac = np.zeros((n, 6, nj))
    for k in range(nj):
        jac[:, 0:3, k] = np.cross(axes[k], pos - anchors[k])
        jac[:, 3:6, k] = axes[k]
    return pos.astype(np.float32), _quat_from_matrix(rot).astype(np.float32), jac.astype(np.float32)


def reach_ik_state(n: int, seed: int, spread: float = 0.3):
    """Synthetic state of a fixed-base 6-DOF arm for the DifferentialIK action term: joint positions around the
    default pose, root pose, end-effector pose and world Jacobian from the kinematics above, raw actions U(-1, 1)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.tensor(XARM6_DEFAULT_JOINT_POS).repeat(n, 1) + spread * torch.randn(n, 6, generator=g)
    root_pos = torch.cat([torch.rand(n, 2, generator=g) * 20 - 10, torch.zeros(n, 1)], -1)
    yaw = torch.rand(n, generator=g) * 6.28 - 3.14
    root_quat = torch.stack([torch.cos(yaw / 2), torch.zeros(n), torch.zeros(n), torch.sin(yaw / 2)], -1)
    ee_pos, ee_quat, jac = serial_arm_kinematics(q.numpy(), root_pos.numpy(), root_quat.numpy())
    return dict(joint_pos=q, root_pos_w=root_pos, root_quat_w=root_quat, ee_pos_w=torch.from_numpy(ee_pos),
                ee_quat_w=torch.from_numpy(ee_quat), jacobian_w=torch.from_numpy(jac),
                raw_action_pose_rel=torch.rand(n, 6, generator=g) * 2 - 1)


def inhand_state(n: int, seed: int, num_joints: int = 24, num_bodies: int = 26, num_actuated: int = 20, num_fingertips: int = 5,
                 max_episode_length: int = 600):
    """Synthetic state of the in-hand manipulation direct env (Shadow hand shapes: 24 DOF, 20 actuated, 5 fingertips):
    object position = in-hand position + N(0, 0.1) so that a few percent of the envs exceed the fall distance,
    ~3 % of the goal rotations close to the object's (successes), episode lengths up to the limit."""
    g = torch.Generator().manual_seed(seed)

    def quat(k):
        q = torch.randn(k, 4, generator=g)
        return q / q.norm(dim=-1, keepdim=True)

    lower = -0.3 - torch.rand(num_joints, generator=g)
    upper = 0.3 + torch.rand(num_joints, generator=g)
    lower, upper = lower.repeat(n, 1), upper.repeat(n, 1)
    env_origins = torch.cat([torch.rand(n, 2, generator=g) * 60 - 30, torch.zeros(n, 1)], -1)
    in_hand = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1)
    obj = torch.zeros(n, 13)
    obj[:, 0:3] = in_hand + env_origins + 0.1 * torch.randn(n, 3, generator=g)
    obj[:, 3:7] = quat(n)
    obj[:, 7:13] = torch.randn(n, 6, generator=g)
    goal = quat(n)
    near = torch.rand(n, generator=g) < 0.03
    small = torch.cat([torch.ones(n, 1), 0.03 * torch.randn(n, 3, generator=g)], -1)
    small = small / small.norm(dim=-1, keepdim=True)
    # goal = small rotation * object orientation on the "near" envs (quaternion product written out)
    w1, x1, y1, z1 = small.unbind(-1)
    w2, x2, y2, z2 = obj[:, 3:7].unbind(-1)
    prod = torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)
    goal = torch.where(near.unsqueeze(-1), prod, goal)
    body = torch.zeros(n, num_bodies, 13)
    body[..., 0:3] = env_origins.unsqueeze(1) + torch.randn(n, num_bodies, 3, generator=g) * 0.1
    body[..., 3:7] = quat(n * num_bodies).view(n, num_bodies, 4)
    body[..., 7:13] = torch.randn(n, num_bodies, 6, generator=g)
    q = lower + (upper - lower) * torch.rand(n, num_joints, generator=g)
    actuated = sorted(torch.randperm(num_joints, generator=g)[:num_actuated].tolist())
    fingers = sorted(torch.randperm(num_bodies, generator=g)[:num_fingertips].tolist())
    return dict(joint_pos=q, joint_vel=torch.randn(n, num_joints, generator=g), lower=lower, upper=upper, env_origins=env_origins,
                in_hand_pos=in_hand, object_state=obj, goal_rot=goal, body_state=body,
                cur_targets=q.clone(), prev_targets=q + 0.05 * torch.randn(n, num_joints, generator=g),
                successes=torch.randint(0, 5, (n,), generator=g).float(), reset_goal_buf=torch.zeros(n, dtype=torch.bool),
                consecutive_successes=torch.tensor([1.5]), episode_length_buf=torch.randint(0, max_episode_length, (n,), generator=g),
                actions=torch.rand(n, num_actuated, generator=g) * 2 - 1, actuated=actuated, fingers=fingers)
