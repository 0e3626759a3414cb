"""CPU restatement of the joint action terms and the action manager's bookkeeping.  Test infrastructure only: the
product never imports this module.

Follows ``isaaclab/envs/mdp/actions/joint_actions.py:130-139`` (affine map, clamp), ``:158-160, 189-193, 209-211,
223-225`` (what each term writes into the articulation) and ``isaaclab/managers/action_manager.py:296-339``.
"""

import torch


def process_actions(raw: torch.Tensor, scale, offset, clip=None) -> torch.Tensor:
    """joint_actions.py:130-139: ``raw * scale + offset``, then ``clamp(min=clip[..., 0], max=clip[..., 1])``."""
    out = raw * scale + offset
    if clip is not None:
        out = torch.clamp(out, min=clip[:, :, 0], max=clip[:, :, 1])
    return out


def manager_process(action: torch.Tensor, prev_action: torch.Tensor, raw: torch.Tensor, term_dims: list[int]):
    """action_manager.py:318-339: shifts the action history and returns each term's slice of the raw action."""
    if sum(term_dims) != raw.shape[1]:
        raise ValueError(f"Invalid action shape, expected: {sum(term_dims)}, received: {raw.shape[1]}.")
    prev_action[:] = action
    action[:] = raw
    parts, idx = [], 0
    for d in term_dims:
        parts.append(raw[:, idx: idx + d])
        idx += d
    return parts


def manager_reset(action: torch.Tensor, prev_action: torch.Tensor, term_raw: list, env_ids):
    """action_manager.py:296-316 with joint_actions.py:141-142."""
    prev_action[env_ids] = 0.0
    action[env_ids] = 0.0
    for r in term_raw:
        r[env_ids] = 0.0


def to_limits(raw, scale, clip, rescale: bool, lo, hi):
    """joint_actions_to_limits.py:107-129: scale, clamp to ``clip``, then (optionally) clamp to [-1, 1] and map onto the soft
    joint limits with ``unscale_transform`` (math.py:61-78): ``x * (upper - lower) * 0.5 + (lower + upper) * 0.5``."""
    out = raw * scale
    if clip is not None:
        out = torch.clamp(out, min=clip[:, :, 0], max=clip[:, :, 1])
    if rescale:
        x = out.clamp(-1.0, 1.0)
        offset = (lo + hi) * 0.5
        out = x * (hi - lo) * 0.5 + offset
    return out


def ema_to_limits(processed, alpha, prev_applied, lo, hi):
    """joint_actions_to_limits.py:213-229; returns the applied action, which is also the next ``prev_applied``."""
    ema = alpha * processed
    ema = ema + (1.0 - alpha) * prev_applied
    return torch.clamp(ema, lo, hi)


def binary(raw, open_command, close_command, clip=None):
    """binary_joint_actions.py:118-131 (``raw`` (n,1) float: negative closes; bool: False closes)."""
    mask = (raw == 0) if raw.dtype == torch.bool else (raw < 0)
    out = torch.where(mask, close_command, open_command)
    if clip is not None:
        out = torch.clamp(out, min=clip[:, :, 0], max=clip[:, :, 1])
    return out


def non_holonomic(body_quat_w, processed):
    """non_holonomic_actions.py:147-155."""
    from . import mathx as M

    yaw = M.euler_xyz_from_quat(body_quat_w)[2]
    return torch.stack([torch.cos(yaw) * processed[:, 0], torch.sin(yaw) * processed[:, 0], processed[:, 1]], dim=-1)
