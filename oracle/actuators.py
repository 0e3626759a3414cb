"""Oracle for the actuator models (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/actuators/actuator_pd.py`` and ``actuator_net.py`` of the reference.
State that the reference keeps on the actuator object (histories) is passed in and returned.
"""

from __future__ import annotations

import torch

ACTIVATIONS = {
    "softsign": torch.nn.functional.softsign,
    "elu": torch.nn.functional.elu,
    "tanh": torch.tanh,
    "relu": torch.relu,
    "identity": lambda x: x,
}


def pd_torque(q_des, qd_des, tau_ff, q, qd, stiffness, damping):
    """actuator_pd.py:135-138: kp*(q_des-q) + kd*(qd_des-qd) + tau_ff, evaluated left to right."""
    error_pos = q_des - q
    error_vel = qd_des - qd
    return stiffness * error_pos + damping * error_vel + tau_ff


def clip_effort(effort, effort_limit):
    """actuator_base.py:252-261."""
    return torch.clip(effort, min=-effort_limit, max=effort_limit)


def dc_motor_clip(effort, joint_vel, saturation_effort: float, velocity_limit, effort_limit):
    """actuator_pd.py:223-233: velocity dependent torque envelope, then clip."""
    zeros = torch.zeros_like(effort)
    max_effort = saturation_effort * (1.0 - joint_vel / velocity_limit)
    max_effort = torch.clip(max_effort, min=zeros, max=effort_limit)
    min_effort = saturation_effort * (-1.0 - joint_vel / velocity_limit)
    min_effort = torch.clip(min_effort, min=-effort_limit, max=zeros)
    return torch.clip(effort, min=min_effort, max=max_effort)


def ideal_pd(q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit):
    """IdealPDActuator.compute / ImplicitActuator.compute (actuator_pd.py:62-87, 131-145)."""
    computed = pd_torque(q_des, qd_des, tau_ff, q, qd, stiffness, damping)
    return computed, clip_effort(computed, effort_limit)


def dc_motor(q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit, velocity_limit, saturation_effort):
    """DCMotor.compute (actuator_pd.py:211-233)."""
    computed = pd_torque(q_des, qd_des, tau_ff, q, qd, stiffness, damping)
    return computed, dc_motor_clip(computed, qd, saturation_effort, velocity_limit, effort_limit)


def mlp_forward(x, weights, biases, activation: str):
    """Linear -> act -> ... -> Linear, the layout of the actuator nets (activation on all but the last layer)."""
    act = ACTIVATIONS[activation]
    h = x
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = torch.nn.functional.linear(h, w, b)
        if i + 1 < len(weights):
            h = act(h)
    return h


def actuator_net_mlp(
    q_des, q, qd, pos_err_hist, vel_hist, weights, biases, activation, input_idx, input_order, pos_scale, vel_scale,
    torque_scale, effort_limit, velocity_limit, saturation_effort,
):
    """ActuatorNetMLP.compute (actuator_net.py:147-188).

    ``pos_err_hist`` / ``vel_hist`` are (N, K, J) with slot 0 the newest; returns
    (computed, applied, new_pos_err_hist, new_vel_hist).
    """
    n, _, j = pos_err_hist.shape
    pe = pos_err_hist.roll(1, 1)
    pe[:, 0] = q_des - q
    vh = vel_hist.roll(1, 1)
    vh[:, 0] = qd
    pos_in = torch.cat([pe[:, i].unsqueeze(2) for i in input_idx], dim=2).reshape(n * j, -1)
    vel_in = torch.cat([vh[:, i].unsqueeze(2) for i in input_idx], dim=2).reshape(n * j, -1)
    if input_order == "pos_vel":
        net_in = torch.cat([pos_in * pos_scale, vel_in * vel_scale], dim=1)
    elif input_order == "vel_pos":
        net_in = torch.cat([vel_in * vel_scale, pos_in * pos_scale], dim=1)
    else:
        raise ValueError(f"Invalid input order for MLP actuator net: {input_order}. Must be 'pos_vel' or 'vel_pos'.")
    torques = mlp_forward(net_in, weights, biases, activation).reshape(n, j)
    computed = torques * torque_scale
    applied = dc_motor_clip(computed, qd, saturation_effort, velocity_limit, effort_limit)
    return computed, applied, pe, vh


def lstm_cell_stack(x, hidden, cell, w_ih, w_hh, b_ih, b_hh):
    """One time step of a multi-layer ``torch.nn.LSTM`` (the module ``ActuatorNetLSTM`` drives with a length-1 sequence,
    actuator_net.py:86-89), written out: per layer ``gates = W_ih x + b_ih + W_hh h + b_hh``, chunks (i, f, g, o),
    ``c' = sigmoid(f) c + sigmoid(i) tanh(g)``, ``h' = sigmoid(o) tanh(c')``; layer l + 1 takes h' of layer l.

    ``x`` (R, In); ``hidden`` / ``cell`` (L, R, H).  Returns (output of the last layer (R, H), new hidden, new cell).
    """
    new_h, new_c = [], []
    inp = x
    for l in range(hidden.shape[0]):
        gates = torch.nn.functional.linear(inp, w_ih[l], b_ih[l]) + torch.nn.functional.linear(hidden[l], w_hh[l], b_hh[l])
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(f) * cell[l] + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        new_h.append(h)
        new_c.append(c)
        inp = h
    return inp, torch.stack(new_h), torch.stack(new_c)


def actuator_net_lstm(q_des, q, qd, hidden, cell, lstm, head_weights, head_biases, head_activation, effort_limit,
                      velocity_limit, saturation_effort):
    """ActuatorNetLSTM.compute (actuator_net.py:76-99).

    Network input per (env, joint) row = (position error, velocity); ``hidden`` / ``cell`` are the (L, N*J, H) states the
    reference keeps on the actuator; ``lstm`` = (w_ih, w_hh, b_ih, b_hh) lists per layer; the head maps the last layer's
    output to one torque.  Returns (computed, applied, new_hidden, new_cell).
    """
    n, j = q.shape
    x = torch.stack([(q_des - q).flatten(), qd.flatten()], dim=1)
    out, h, c = lstm_cell_stack(x, hidden, cell, *lstm)
    computed = mlp_forward(out, head_weights, head_biases, head_activation).reshape(n, j)
    applied = dc_motor_clip(computed, qd, saturation_effort, velocity_limit, effort_limit)
    return computed, applied, h, c


class DelayLine:
    """DelayBuffer over a CircularBuffer (utils/buffers/delay_buffer.py:32-177, circular_buffer.py:93-175), restated:
    a ring of ``history_length + 1`` slots shared by all envs (one write pointer), per-env push counts and time lags.
    The first push after a reset fills every slot of that env; a read returns the entry ``min(lag, pushes - 1)`` back."""

    def __init__(self, history_length: int, batch_size: int):
        self.max_len = max(0, history_length) + 1
        self.pointer = -1
        self.buffer = None
        self.num_pushes = torch.zeros(batch_size, dtype=torch.long)
        self.time_lags = torch.zeros(batch_size, dtype=torch.int)

    def reset(self, ids):
        self.num_pushes[ids] = 0
        if self.buffer is not None:
            self.buffer[:, ids] = 0.0

    def compute(self, data):
        if self.buffer is None:
            self.buffer = torch.empty((self.max_len, *data.shape), dtype=data.dtype)
        self.pointer = (self.pointer + 1) % self.max_len
        self.buffer[self.pointer] = data
        fresh = self.num_pushes == 0
        self.buffer[:, fresh] = data[fresh]
        self.num_pushes += 1
        valid = torch.minimum(self.time_lags.long(), self.num_pushes - 1)
        index = torch.remainder(self.pointer - valid, self.max_len)
        return self.buffer[index, torch.arange(data.shape[0])].clone()


def linear_interpolation(x, y, q):
    """utils/interpolation/linear_interpolation.py:55-88: piecewise linear through (x, y), held constant outside."""
    q1 = q.reshape(-1)
    smaller = torch.sum(x.unsqueeze(1) < q1.unsqueeze(0), dim=0, dtype=torch.int)
    lo = torch.clamp(smaller - 1, min=0)
    hi = torch.clamp(smaller, max=x.numel() - 1)
    weight = (q1 - x[lo]) / (x[hi] - x[lo])
    weight[hi == lo] = 0.0
    return (y[lo] + weight * (y[hi] - y[lo])).reshape(q.shape)


def delayed_pd(lines, q_des, qd_des, tau_ff, q, qd, stiffness, damping, effort_limit, lookup=None):
    """DelayedPDActuator.compute (actuator_pd.py:283-291) and, with ``lookup`` = (angles, max torques),
    RemotizedPDActuator.compute (:347-363).  ``lines`` = the three DelayLine objects (positions, velocities, efforts)."""
    computed = pd_torque(lines[0].compute(q_des), lines[1].compute(qd_des), lines[2].compute(tau_ff), q, qd, stiffness, damping)
    applied = clip_effort(computed, effort_limit)
    if lookup is not None:
        lim = linear_interpolation(lookup[0], lookup[1], q)
        applied = torch.clamp(applied, min=-lim, max=lim)
    return computed, applied
