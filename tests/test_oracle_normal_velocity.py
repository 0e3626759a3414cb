"""CPU: the NormalVelocityCommand oracle against the fixture recorded from the reference class."""

import torch

from conftest import assert_close, assert_exact, load_golden
from oracle import command as C


def test_normal_velocity_command_sequence():
    g = load_golden("normal_velocity_command")
    st = dict(vel_command_b=g["pre_vel_command_b"].clone(), is_standing=g["pre_is_standing_env"].clone(),
              is_zero=torch.stack([g["pre_is_zero_vel_x_env"], g["pre_is_zero_vel_y_env"], g["pre_is_zero_vel_yaw_env"]], dim=1),
              time_left=g["pre_time_left"].clone(), command_counter=g["pre_command_counter"].clone(),
              error_vel_xy=g["pre_error_vel_xy"].clone(), error_vel_yaw=g["pre_error_vel_yaw"].clone())
    for k in range(g["samples"].shape[0]):
        C.normal_velocity_compute(st, g["root_state_w"][k], 0.02, g["samples"][k], zero_prob=(0.1, 0.3, 0.2), rel_standing_envs=0.15,
                                  max_command_step=0.09 / 0.02)
        assert_exact(st["time_left"], g["time_left"][k], f"time_left {k}")
        assert_exact(st["command_counter"], g["command_counter"][k], f"counter {k}")
        assert_exact(st["vel_command_b"], g["vel_command_b"][k], f"command {k}")
        assert_exact(st["is_zero"], g["is_zero"][k], f"zero flags {k}")
        assert_exact(st["is_standing"], g["is_standing_env"][k], f"standing flags {k}")
        assert_close(st["error_vel_xy"], g["error_vel_xy"][k], atol=1e-6, what=f"error_vel_xy {k}")
        assert_close(st["error_vel_yaw"], g["error_vel_yaw"][k], atol=1e-6, what=f"error_vel_yaw {k}")
