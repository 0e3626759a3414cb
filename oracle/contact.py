"""Oracle for the contact-sensor bookkeeping (test infrastructure, CPU only).

Follows ``source/isaaclab/isaaclab/sensors/sensor_base.py:197-205,284-293`` (update timing) and
``sensors/contact_sensor/contact_sensor.py:314-373`` (history ring, air / contact timers) of the reference.
All state tensors are updated in place, like the reference does.
"""

from __future__ import annotations

import torch

from . import mathx as M


def sensor_tick(timestamp, timestamp_last_update, is_outdated, dt: float, update_period: float):
    """sensor_base.py:199-200: advance the clock and flag sensors whose period elapsed."""
    timestamp += M.f32(dt)
    is_outdated |= (timestamp - timestamp_last_update + M.f32(1e-6)) >= M.f32(update_period)


def contact_update(
    net_forces_in, net_forces_w, history, current_air_time, last_air_time, current_contact_time, last_contact_time,
    timestamp, timestamp_last_update, is_outdated, dt: float, update_period: float, force_threshold: float,
):
    """One ``ContactSensor.update(dt)`` with ``history_length > 0`` and ``track_air_time=True``.

    ``net_forces_in`` (N,B,3) is what the physics view reports this substep.  ``history`` is (N,H,B,3) or None.
    Timer tensors may be None (``track_air_time=False``).
    """
    sensor_tick(timestamp, timestamp_last_update, is_outdated, dt, update_period)
    ids = is_outdated.nonzero().squeeze(-1)
    if len(ids) == 0:
        return
    net_forces_w[ids] = net_forces_in[ids]
    if history is not None:
        history[ids, 1:] = history[ids, :-1].clone()
        history[ids, 0] = net_forces_w[ids]
    if current_air_time is not None:
        elapsed = (timestamp[ids] - timestamp_last_update[ids]).unsqueeze(-1)
        f = net_forces_w[ids]
        is_contact = M.norm3(f[..., 0], f[..., 1], f[..., 2]) > float(M.f32(force_threshold))
        first_contact = (current_air_time[ids] > 0) & is_contact
        first_detached = (current_contact_time[ids] > 0) & ~is_contact
        last_air_time[ids] = torch.where(first_contact, current_air_time[ids] + elapsed, last_air_time[ids])
        current_air_time[ids] = torch.where(~is_contact, current_air_time[ids] + elapsed, torch.zeros(()))
        last_contact_time[ids] = torch.where(first_detached, current_contact_time[ids] + elapsed, last_contact_time[ids])
        current_contact_time[ids] = torch.where(is_contact, current_contact_time[ids] + elapsed, torch.zeros(()))
    timestamp_last_update[ids] = timestamp[ids]
    is_outdated[ids] = False


def compute_first_contact(current_contact_time, dt: float, abs_tol: float = 1.0e-8):
    """contact_sensor.py:169-203.  ``dt + abs_tol`` is evaluated in double on the host, then rounded once."""
    return (current_contact_time > 0.0) & (current_contact_time < float(M.f32(dt + abs_tol)))


def compute_first_air(current_air_time, dt: float, abs_tol: float = 1.0e-8):
    """contact_sensor.py:205-238."""
    return (current_air_time > 0.0) & (current_air_time < float(M.f32(dt + abs_tol)))


def contact_reset(env_ids, net_forces_w, history, current_air_time, last_air_time, current_contact_time,
                  last_contact_time, timestamp, timestamp_last_update, is_outdated):
    """sensor_base.py:182-195 + contact_sensor.py:136-155."""
    timestamp[env_ids] = 0.0
    timestamp_last_update[env_ids] = 0.0
    is_outdated[env_ids] = True
    net_forces_w[env_ids] = 0.0
    if history is not None:
        history[env_ids] = 0.0
    for t in (current_air_time, last_air_time, current_contact_time, last_contact_time):
        if t is not None:
            t[env_ids] = 0.0
