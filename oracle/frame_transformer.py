"""CPU restatement of the FrameTransformer refresh (test infrastructure only; never imported by the product).

Follows ``isaaclab/sensors/frame_transformer/frame_transformer.py:322-385`` with the gathers given as row indices.  Pinned by
tests/golden/frame_transformer.npz, recorded from the reference class.
"""

from __future__ import annotations

import torch

from . import mathx as M


def refresh(transforms, source_rows, target_rows, num_frames, source_offset=None, target_offsets=None):
    """transforms (rows, 7) position + quaternion x,y,z,w; source_rows (n), target_rows (n*F); offsets as (pos, quat) with
    shapes (3,), (4,) for the source and (F,3), (F,4) for the targets, or None.
    Returns source_pos_w, source_quat_w, target_pos_w, target_quat_w, target_pos_source, target_quat_source."""
    n = source_rows.numel()
    wxyz = torch.cat([transforms[:, :3], transforms[:, 6:7], transforms[:, 3:6]], dim=1)
    src, tgt = wxyz[source_rows], wxyz[target_rows]
    sp, sq = src[:, :3], src[:, 3:]
    if source_offset is not None:
        sp, sq = M.combine_frame_transforms(sp, sq, source_offset[0].expand(n, 3), source_offset[1].expand(n, 4))
    tp, tq = tgt[:, :3], tgt[:, 3:]
    if target_offsets is not None:
        tp, tq = M.combine_frame_transforms(tp, tq, target_offsets[0].repeat(n, 1), target_offsets[1].repeat(n, 1))
    rp, rq = M.subtract_frame_transforms(sp.repeat_interleave(num_frames, 0), sq.repeat_interleave(num_frames, 0), tp, tq)
    f = num_frames
    return sp, sq, tp.view(n, f, 3), tq.view(n, f, 4), rp.view(n, f, 3), rq.view(n, f, 4)
