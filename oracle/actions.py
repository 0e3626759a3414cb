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
