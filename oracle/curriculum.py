"""CPU restatement of the terrain curriculum.  Test infrastructure only: the product never imports this module.

Follows ``isaaclab_tasks/manager_based/locomotion/velocity/mdp/curriculums.py:44-55`` (move flags) and
``isaaclab/terrains/terrain_importer.py:307-322`` (levels, origins).
"""

import torch


def move_flags(root_pos_w, env_origins, command, env_ids, half_size: float, max_episode_length_s: float):
    distance = torch.norm(root_pos_w[env_ids, :2] - env_origins[env_ids, :2], dim=1)
    move_up = distance > half_size
    move_down = distance < torch.norm(command[env_ids, :2], dim=1) * max_episode_length_s * 0.5
    move_down = move_down & ~move_up
    return move_up, move_down


def update_env_origins(terrain_levels, terrain_types, terrain_origins, env_origins, env_ids, move_up, move_down, random_levels):
    """``random_levels``: what ``torch.randint_like(levels[env_ids], max_level)`` returned (one per listed env)."""
    max_level = terrain_origins.shape[0]
    lv = terrain_levels[env_ids] + 1 * move_up - 1 * move_down
    lv = torch.where(lv >= max_level, random_levels, torch.clip(lv, 0))
    terrain_levels[env_ids] = lv
    env_origins[env_ids] = terrain_origins[lv, terrain_types[env_ids]]
