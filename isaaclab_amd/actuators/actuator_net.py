"""Actuator-network models on the fused HIP kernels ``ilx_actuator_net_mlp`` / ``ilx_actuator_net_lstm``
(interface of ``isaaclab/actuators/actuator_net.py``).

The reference loads arbitrary TorchScript modules; here the MLP is a stack of Linear layers with one activation
(softsign / elu / tanh / relu) and the LSTM net is a ``torch.nn.LSTM`` followed by such a stack, both given as tensors.
History roll / recurrent state update, the network and the DC-motor clip run in a single launch; an MLP of hidden
width 32 uses fp32 MFMA.
"""

from __future__ import annotations

import ctypes
from collections.abc import Sequence

import torch

from .. import _abi
from .actuator_pd import DCMotor

_ACT = {"softsign": 0, "elu": 1, "tanh": 2, "relu": 3, "identity": 4}


class MlpHandle:
    """Owns an ``ilx_mlp_t`` (device copy of the weights in kernel order)."""

    def __init__(self, weights: list[torch.Tensor], biases: list[torch.Tensor], activation: str):
        if activation not in _ACT:
            raise ValueError(f"unsupported activation '{activation}', expected one of {sorted(_ACT)}")
        ws = [w.detach().to("cpu", torch.float32).contiguous() for w in weights]
        bs = [b.detach().to("cpu", torch.float32).contiguous() for b in biases]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if w.shape != (dims[i + 1], dims[i]) or b.shape != (dims[i + 1],):
                raise ValueError(f"layer {i}: weight {tuple(w.shape)} / bias {tuple(b.shape)} do not chain")
        n = len(ws)
        c_dims = (ctypes.c_int32 * (n + 1))(*dims)
        c_w = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        c_b = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs])
        self.handle = ctypes.c_void_p()
        _abi.check(_abi.mlp_create(n, c_dims, _ACT[activation], c_w, c_b, ctypes.byref(self.handle)), "ilx_mlp_create")
        self.dims = dims
        self.activation = activation

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _abi is not None and getattr(_abi, "lib", None) is not None:  # module globals are gone at shutdown
            _abi.lib.ilx_mlp_destroy(h)
            self.handle = None


def load_network(cfg, device) -> tuple[list[torch.Tensor], list[torch.Tensor], str]:
    """Weights from ``cfg.network`` or from a state-dict file holding ``<k>.weight`` / ``<k>.bias`` pairs."""
    if getattr(cfg, "network", None) is not None:
        net = cfg.network
        return list(net["weights"]), list(net["biases"]), net.get("activation", "softsign")
    if not cfg.network_file:
        raise ValueError("ActuatorNetMLPCfg needs `network` or `network_file`")
    sd = torch.load(cfg.network_file, map_location="cpu", weights_only=True)
    keys = sorted({k.rsplit(".", 1)[0] for k in sd if k.endswith(".weight")}, key=lambda s: [int(t) if t.isdigit() else t for t in s.split(".")])
    return [sd[k + ".weight"] for k in keys], [sd[k + ".bias"] for k in keys], sd.get("activation", "softsign")


class LstmHandle:
    """Owns an ``ilx_lstm_t``: device copy of a ``torch.nn.LSTM``'s parameters and of the Linear head after it."""

    def __init__(self, params: dict[str, torch.Tensor], weights: list[torch.Tensor], biases: list[torch.Tensor], activation: str):
        layers = len([k for k in params if k.startswith("weight_ih_l")])
        if layers == 0 or len(params) != 4 * layers:
            raise ValueError("expected weight_ih_l*, weight_hh_l*, bias_ih_l*, bias_hh_l* of a unidirectional torch.nn.LSTM")
        if activation not in _ACT:
            raise ValueError(f"unsupported activation '{activation}', expected one of {sorted(_ACT)}")
        p = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in params.items()}
        self.hidden_dim = p["weight_hh_l0"].shape[1]
        self.input_size = p["weight_ih_l0"].shape[1]
        self.num_layers = layers
        for l in range(layers):
            in_l = self.input_size if l == 0 else self.hidden_dim
            want = {"weight_ih": (4 * self.hidden_dim, in_l), "weight_hh": (4 * self.hidden_dim, self.hidden_dim),
                    "bias_ih": (4 * self.hidden_dim,), "bias_hh": (4 * self.hidden_dim,)}
            for nm, shape in want.items():
                if tuple(p[f"{nm}_l{l}"].shape) != shape:
                    raise ValueError(f"{nm}_l{l}: shape {tuple(p[f'{nm}_l{l}'].shape)}, expected {shape}")
        ws = [w.detach().to("cpu", torch.float32).contiguous() for w in weights]
        bs = [b.detach().to("cpu", torch.float32).contiguous() for b in biases]
        dims = [self.hidden_dim] + [w.shape[0] for w in ws]
        for i, (w, b) in enumerate(zip(ws, bs)):
            if w.shape != (dims[i + 1], dims[i]) or b.shape != (dims[i + 1],):
                raise ValueError(f"head layer {i}: weight {tuple(w.shape)} / bias {tuple(b.shape)} do not chain")
        arr = lambda nm: (ctypes.c_void_p * layers)(*[p[f"{nm}_l{l}"].data_ptr() for l in range(layers)])
        n = len(ws)
        self.handle = ctypes.c_void_p()
        _abi.check(_abi.lstm_create(layers, self.input_size, self.hidden_dim, arr("weight_ih"), arr("weight_hh"), arr("bias_ih"),
                                    arr("bias_hh"), n, (ctypes.c_int32 * (n + 1))(*dims), _ACT[activation],
                                    (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws]),
                                    (ctypes.c_void_p * n)(*[b.data_ptr() for b in bs]), ctypes.byref(self.handle)),
                   "ilx_lstm_create")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and _abi is not None and getattr(_abi, "lib", None) is not None:
            _abi.lib.ilx_lstm_destroy(h)
            self.handle = None


def load_lstm_network(cfg):
    """(lstm parameter dict, head weights, head biases, head activation) from ``cfg.network`` or a state-dict file."""
    if getattr(cfg, "network", None) is not None:
        net = cfg.network
        return dict(net["lstm"]), list(net["weights"]), list(net["biases"]), net.get("activation", "softsign")
    if not cfg.network_file:
        raise ValueError("ActuatorNetLSTMCfg needs `network` or `network_file`")
    sd = torch.load(cfg.network_file, map_location="cpu", weights_only=True)
    lstm = {k.split("lstm.", 1)[1]: v for k, v in sd.items() if k.startswith("lstm.")}
    rest = {k: v for k, v in sd.items() if not k.startswith("lstm.") and torch.is_tensor(v)}
    keys = sorted({k.rsplit(".", 1)[0] for k in rest if k.endswith(".weight")}, key=lambda s: [int(t) if t.isdigit() else t for t in s.split(".")])
    return lstm, [rest[k + ".weight"] for k in keys], [rest[k + ".bias"] for k in keys], sd.get("activation", "softsign")


class ActuatorNetLSTM(DCMotor):
    """actuator_net.py:30-99."""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        lstm, weights, biases, activation = load_lstm_network(cfg)
        self.network = LstmHandle(lstm, weights, biases, activation)
        rows = self._num_envs * self.num_joints
        num_layers, hidden_dim = self.network.num_layers, self.network.hidden_dim
        self.sea_hidden_state = torch.zeros(num_layers, rows, hidden_dim, device=self._device)
        self.sea_cell_state = torch.zeros(num_layers, rows, hidden_dim, device=self._device)
        layer_shape_per_env = (num_layers, self._num_envs, self.num_joints, hidden_dim)
        self.sea_hidden_state_per_env = self.sea_hidden_state.view(layer_shape_per_env)
        self.sea_cell_state_per_env = self.sea_cell_state.view(layer_shape_per_env)

    def reset(self, env_ids: Sequence[int]):
        self.sea_hidden_state_per_env[:, env_ids] = 0.0
        self.sea_cell_state_per_env[:, env_ids] = 0.0

    def compute(self, control_action, joint_pos, joint_vel):
        q_des, q, qd = _abi.f32c(control_action.joint_positions), _abi.f32c(joint_pos), _abi.f32c(joint_vel)
        n, j = q.shape
        self._joint_vel = joint_vel
        _abi.check(_abi.actuator_net_lstm(
            self.network.handle, _abi.ptr(q_des), _abi.ptr(q), _abi.ptr(qd), _abi.ptr(self.sea_hidden_state),
            _abi.ptr(self.sea_cell_state), _abi.ptr(self.effort_limit), _abi.ptr(self.velocity_limit), 1,
            float(self._saturation_effort), _abi.ptr(self.computed_effort), _abi.ptr(self.applied_effort), n, j, _abi.stream()),
            "ilx_actuator_net_lstm")
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action


class ActuatorNetMLP(DCMotor):
    """actuator_net.py:102-188."""

    def __init__(self, cfg, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        weights, biases, activation = load_network(cfg, self._device)
        self.network = MlpHandle(weights, biases, activation)
        history_length = max(self.cfg.input_idx) + 1
        self._joint_pos_error_history = torch.zeros(self._num_envs, history_length, self.num_joints, device=self._device)
        self._joint_vel_history = torch.zeros(self._num_envs, history_length, self.num_joints, device=self._device)
        self._idx = (ctypes.c_int32 * len(self.cfg.input_idx))(*self.cfg.input_idx)
        if self.cfg.input_order not in ("pos_vel", "vel_pos"):
            raise ValueError(f"Invalid input order for MLP actuator net: {self.cfg.input_order}. Must be 'pos_vel' or 'vel_pos'.")
        self._flags = 0

    def reset(self, env_ids: Sequence[int]):
        self._joint_pos_error_history[env_ids] = 0.0
        self._joint_vel_history[env_ids] = 0.0

    def compute(self, control_action, joint_pos, joint_vel, scene_update=None, joint_action=None):
        """actuator_net.py:147-188.  ``scene_update`` (no counterpart in the reference): the descriptor of the scene update
        of the substep before (``InteractiveScene.describe_update``), which then travels with this launch.
        ``joint_action`` (likewise): the step's first substep -- ``(raw, action, prev_action, scale, offset, begin)`` of the
        JointPositionAction whose processing this launch does on the way (``control_action.joint_positions`` is then its
        output), ``begin`` = ``FusedMdp.begin_step_args()`` or None."""
        q_des, q, qd = _abi.f32c(control_action.joint_positions), _abi.f32c(joint_pos), _abi.f32c(joint_vel)
        n, j = q.shape
        self._joint_vel = joint_vel
        if joint_action is not None:
            raw, action, prev_action, scale, offset, begin = joint_action
            begin = begin if begin is not None else (None, 0, None)
            _abi.check(_abi.actuator_net_mlp_joint_action(
                self.network.handle, _abi.ptr(raw), _abi.ptr(action), _abi.ptr(prev_action), float(scale), _abi.ptr(offset),
                _abi.ptr(q_des), _abi.ptr(q), _abi.ptr(qd), _abi.ptr(self._joint_pos_error_history),
                _abi.ptr(self._joint_vel_history), self._joint_pos_error_history.shape[1], self._idx, len(self.cfg.input_idx),
                0 if self.cfg.input_order == "pos_vel" else 1, float(self.cfg.pos_scale), float(self.cfg.vel_scale),
                float(self.cfg.torque_scale), _abi.ptr(self.effort_limit), _abi.ptr(self.velocity_limit), 1,
                float(self._saturation_effort), _abi.ptr(self.computed_effort), _abi.ptr(self.applied_effort), n, j, self._flags,
                *begin, _abi.stream()), "ilx_actuator_net_mlp_joint_action")
            control_action.joint_efforts = self.applied_effort
            control_action.joint_positions = None
            control_action.joint_velocities = None
            return control_action
        args = (self.network.handle, _abi.ptr(q_des), _abi.ptr(q), _abi.ptr(qd), _abi.ptr(self._joint_pos_error_history),
                _abi.ptr(self._joint_vel_history), self._joint_pos_error_history.shape[1], self._idx, len(self.cfg.input_idx),
                0 if self.cfg.input_order == "pos_vel" else 1, float(self.cfg.pos_scale), float(self.cfg.vel_scale),
                float(self.cfg.torque_scale), _abi.ptr(self.effort_limit), _abi.ptr(self.velocity_limit), 1,
                float(self._saturation_effort), _abi.ptr(self.computed_effort), _abi.ptr(self.applied_effort), n, j, self._flags)
        if scene_update is not None:
            _abi.check(_abi.actuator_net_mlp_scene_update(*args, ctypes.byref(scene_update), _abi.stream()),
                       "ilx_actuator_net_mlp_scene_update")
        else:
            _abi.check(_abi.actuator_net_mlp(*args, _abi.stream()), "ilx_actuator_net_mlp")
        control_action.joint_efforts = self.applied_effort
        control_action.joint_positions = None
        control_action.joint_velocities = None
        return control_action
