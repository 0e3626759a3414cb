"""Host side of the fused MDP step (``ilx_mdp_step``): term tables -> one kernel launch.

:class:`Term` mirrors ``ilx_term`` and :class:`MdpDesc` mirrors ``ilx_mdp`` of ``include/isaaclab_amd_mdp.h``.
:class:`FusedMdp` owns a plan (term tables on the device) plus the descriptor and exposes ``step(phases)``.
Everything numeric that the reference evaluates in Python double precision before it meets an fp32 tensor
(``std**2``, ``n_max - n_min``, ``dt + abs_tol``, ``resampling_time / step_dt``) is evaluated here in double and
rounded once when stored into the descriptor.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_float, c_int32, c_int64, c_uint8, c_uint64, c_void_p

import torch

from .. import _abi
from ..utils import rng as _rng

# ---- enums (keep in sync with include/isaaclab_amd_mdp.h) ------------------------------------------
REW = {name: i for i, name in enumerate([
    "is_alive", "is_terminated", "is_terminated_term", "lin_vel_z_l2", "ang_vel_xy_l2", "flat_orientation_l2",
    "base_height_l2", "body_lin_acc_l2", "joint_torques_l2", "joint_vel_l1", "joint_vel_l2", "joint_acc_l2",
    "joint_deviation_l1", "joint_pos_limits", "joint_vel_limits", "applied_torque_limits", "action_rate_l2", "action_l2",
    "undesired_contacts", "contact_forces", "track_lin_vel_xy_exp", "track_ang_vel_z_exp", "feet_air_time",
    "feet_air_time_positive_biped", "feet_slide", "track_lin_vel_xy_yaw_frame_exp", "track_ang_vel_z_world_exp",
    "position_command_error", "position_command_error_tanh", "orientation_command_error", "joint_pos_target_l2"])}
TERM = {name: i for i, name in enumerate([
    "time_out", "command_resample", "bad_orientation", "root_height_below_minimum", "joint_pos_out_of_manual_limit",
    "joint_vel_out_of_limit", "joint_vel_out_of_manual_limit", "joint_effort_out_of_limit", "illegal_contact"])}
OBS = {name: i for i, name in enumerate([
    "base_pos_z", "base_lin_vel", "base_ang_vel", "projected_gravity", "root_pos_w", "root_quat_w", "root_lin_vel_w",
    "root_ang_vel_w", "joint_pos", "joint_pos_rel", "joint_pos_limit_normalized", "joint_vel", "joint_vel_rel",
    "height_scan", "last_action", "generated_commands", "external"])}
MAX_EXTERNAL_OBS = 4
OBS_NOISE, OBS_CLIP, OBS_SCALE, OBS_QUAT_UNIQUE = 1, 2, 4, 8
CMD_NONE, CMD_VELOCITY, CMD_POSE = 0, 1, 2
PHASE_TERMINATION, PHASE_REWARD, PHASE_RESET, PHASE_COMMAND, PHASE_OBSERVATION, PHASE_ALL = 1, 2, 4, 8, 16, 31


class Term(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("flags", c_int32), ("weight", c_float), ("p", c_float * 4), ("aux", c_float),
                ("ids_off", c_int32), ("ids_cnt", c_int32), ("ids2_off", c_int32), ("ids2_cnt", c_int32)]


F, U8, I64 = POINTER(c_float), POINTER(c_uint8), POINTER(c_int64)


class MdpDesc(ctypes.Structure):
    _fields_ = [
        ("num_envs", c_int64), ("num_joints", c_int32), ("num_bodies", c_int32), ("num_sensor_bodies", c_int32),
        ("history_len", c_int32), ("num_rays", c_int32), ("action_dim", c_int32), ("command_dim", c_int32),
        ("obs_dim", c_int32),
        ("root_state_w", c_void_p), ("joint_pos", c_void_p), ("joint_vel", c_void_p), ("joint_acc", c_void_p),
        ("applied_torque", c_void_p), ("computed_torque", c_void_p), ("default_joint_pos", c_void_p),
        ("default_joint_vel", c_void_p), ("soft_joint_pos_limits", c_void_p), ("soft_joint_vel_limits", c_void_p),
        ("body_state_w", c_void_p), ("body_lin_acc_w", c_void_p), ("env_origins", c_void_p), ("gravity_dir", c_float * 3),
        ("net_forces_w_history", c_void_p), ("current_air_time", c_void_p), ("last_air_time", c_void_p),
        ("current_contact_time", c_void_p), ("last_contact_time", c_void_p),
        ("ray_hits_w", c_void_p), ("sensor_pos_w", c_void_p),
        ("action", c_void_p), ("prev_action", c_void_p),
        ("episode_length_buf", c_void_p), ("max_episode_length", c_int64), ("step_dt", c_float),
        ("max_episode_length_s", c_float),
        ("num_terminations", c_int32), ("terminations", POINTER(Term)), ("term_dones", c_void_p),
        ("terminated", c_void_p), ("time_outs", c_void_p), ("reset_buf", c_void_p),
        ("num_rewards", c_int32), ("rewards", POINTER(Term)), ("reward_buf", c_void_p), ("episode_sums", c_void_p),
        ("step_reward", c_void_p),
        ("command_kind", c_int32), ("command", c_void_p), ("heading_target", c_void_p), ("is_heading_env", c_void_p),
        ("is_standing_env", c_void_p), ("time_left", c_void_p), ("command_counter", c_void_p),
        ("command_metrics", c_void_p), ("command_samples", c_void_p), ("reset_command_samples", c_void_p),
        ("cmd_range", (c_float * 2) * 7), ("rel_heading_envs", c_float), ("rel_standing_envs", c_float),
        ("heading_control_stiffness", c_float), ("heading_command", c_int32), ("pose_body_index", c_int32),
        ("make_quat_unique", c_int32), ("max_command_step", c_float),
        ("num_observations", c_int32), ("observations", POINTER(Term)), ("obs", c_void_p), ("noise_u", c_void_p),
        ("ids", POINTER(c_int32)), ("num_ids", c_int32),
        ("log_sums", c_void_p), ("seed", c_uint64), ("step_index", c_uint64), ("phases", c_int32),
        ("external_obs", c_void_p * 4), ("external_obs_dim", c_int32 * 4),
        ("step_index_device", c_void_p), ("height_scan", c_void_p),
    ]


_abi.lib.ilx_mdp_plan_create.argtypes = [POINTER(MdpDesc), POINTER(c_void_p)]
_abi.lib.ilx_mdp_plan_create.restype = ctypes.c_int
_abi.lib.ilx_mdp_plan_destroy.argtypes = [c_void_p]
_abi.lib.ilx_mdp_plan_destroy.restype = None
_abi.lib.ilx_mdp_step.argtypes = [c_void_p, POINTER(MdpDesc), c_void_p]
_abi.lib.ilx_mdp_step.restype = ctypes.c_int
_abi.lib.ilx_mdp_step_scene_update.argtypes = [c_void_p, POINTER(MdpDesc), POINTER(_abi.Scene), c_void_p]
_abi.lib.ilx_mdp_step_scene_update.restype = ctypes.c_int
_abi.lib.ilx_mdp_begin_step.argtypes = [c_void_p, ctypes.c_int32, c_void_p, c_void_p]
_abi.lib.ilx_mdp_begin_step.restype = ctypes.c_int
_abi.lib.ilx_mdp_plan_kernel.argtypes = [c_void_p]
_abi.lib.ilx_mdp_plan_kernel.restype = ctypes.c_char_p
_abi.lib.ilx_mdp_plan_select_kernel.argtypes = [c_void_p, ctypes.c_int]
_abi.lib.ilx_mdp_plan_select_kernel.restype = ctypes.c_int


class TermSpec:
    """One entry of a term table before it is packed: kind name, parameters and id lists."""

    def __init__(self, kind: str, weight: float = 1.0, p=(0.0, 0.0, 0.0, 0.0), ids=None, ids2=None, flags: int = 0,
                 aux: float = 0.0, name: str | None = None):
        self.kind, self.weight, self.p, self.ids, self.ids2 = kind, float(weight), tuple(p), ids, ids2
        self.flags, self.aux, self.name = flags, float(aux), name or kind


def _norm_ids(ids):
    if ids is None or (isinstance(ids, slice) and ids == slice(None)):
        return None
    if isinstance(ids, torch.Tensor):
        ids = ids.tolist()
    return [int(i) for i in ids]


def obs_term(kind: str, noise=None, clip=None, scale=None, ids=None, offset: float = 0.0, make_quat_unique=False, name=None):
    """Observation term with the post-processing of observation_manager.py:308-318 (uniform additive noise only)."""
    flags, p = 0, [0.0, 0.0, 0.0, 0.0]
    if noise is not None:
        n_min, n_max = noise
        flags |= OBS_NOISE
        p[0], p[1] = n_min, n_max - n_min  # the span is formed in double, as the reference does
    if clip is not None:
        flags |= OBS_CLIP
        p[2], p[3] = clip
    w = 1.0
    if scale is not None:
        flags |= OBS_SCALE
        w = scale
    if make_quat_unique:
        flags |= OBS_QUAT_UNIQUE
    return TermSpec(kind, weight=w, p=p, ids=ids, flags=flags, aux=offset, name=name)


class FusedMdp:
    """Packs term tables + tensors into an ``ilx_mdp`` descriptor and launches the fused kernel.

    ``tensors`` maps descriptor field names to GPU tensors (kept alive here); scalar fields are passed as keyword
    arguments.  Output buffers that are not supplied are allocated.

    The in-kernel generator (observation noise, command resampling) is keyed by ``(seed, step_index, env, column)``.
    ``seed`` None (what the managers pass): every launch takes the seed the environment installed and a draw index
    that no other launch of the process shares (:mod:`isaaclab_amd.utils.rng`), so two plans -- two observation groups,
    say -- never draw the same noise, and re-seeding the environment replays.  An explicit ``seed`` gives the plan a
    generator of its own whose ``step_index`` counts its launches from 0.
    """

    def __init__(self, num_envs: int, device, *, num_joints: int, num_bodies: int = 1, num_sensor_bodies: int = 0,
                 history_len: int = 0, num_rays: int = 0, action_dim: int = 0, command_dim: int = 3,
                 terminations=(), rewards=(), observations=(), step_dt: float = 0.02, max_episode_length: int = 1000,
                 max_episode_length_s: float | None = None, gravity=(0.0, 0.0, -1.0), command: dict | None = None,
                 seed: int | None = 0, external_obs_dims=()):
        self.n, self.device = num_envs, torch.device(device)
        self.terminations, self.rewards, self.observations = list(terminations), list(rewards), list(observations)
        d = self.desc = MdpDesc()
        d.num_envs, d.num_joints, d.num_bodies = num_envs, num_joints, num_bodies
        d.num_sensor_bodies, d.history_len, d.num_rays = num_sensor_bodies, history_len, num_rays
        d.action_dim, d.command_dim = action_dim, command_dim
        d.step_dt = step_dt
        d.max_episode_length = max_episode_length
        d.max_episode_length_s = max_episode_length * step_dt if max_episode_length_s is None else max_episode_length_s
        d.gravity_dir = (c_float * 3)(*gravity)
        self._own_generator = seed is not None
        self._holding = False   # the last launch left the step count as it was (see step)
        self._step_counter = None
        d.seed, d.step_index, d.phases = (seed if seed is not None else 0), 0, PHASE_ALL
        if len(external_obs_dims) > MAX_EXTERNAL_OBS:
            raise ValueError(f"a plan reads at most {MAX_EXTERNAL_OBS} external observation buffers, not {len(external_obs_dims)}")
        for slot, width in enumerate(external_obs_dims):   # widths of the buffers ILX_OBS_EXTERNAL terms copy (slot = the term's aux)
            d.external_obs_dim[slot] = int(width)
        self._external: dict[int, torch.Tensor] = {}
        # id lists
        ids: list[int] = []

        def pack(specs, table_enum):
            arr = (Term * max(len(specs), 1))()
            for i, s in enumerate(specs):
                t = arr[i]
                t.kind, t.flags, t.weight, t.aux = table_enum[s.kind], s.flags, s.weight, s.aux
                t.p = (c_float * 4)(*s.p)
                for attr_off, attr_cnt, lst in (("ids_off", "ids_cnt", _norm_ids(s.ids)), ("ids2_off", "ids2_cnt", _norm_ids(s.ids2))):
                    if lst is None:
                        setattr(t, attr_off, 0)
                        setattr(t, attr_cnt, -1)
                    else:
                        setattr(t, attr_off, len(ids))
                        setattr(t, attr_cnt, len(lst))
                        ids.extend(lst)
            return arr

        self._tt, self._tr, self._to = pack(self.terminations, TERM), pack(self.rewards, REW), pack(self.observations, OBS)
        self._ids = (c_int32 * max(len(ids), 1))(*ids)
        d.num_terminations, d.terminations = len(self.terminations), self._tt
        d.num_rewards, d.rewards = len(self.rewards), self._tr
        d.num_observations, d.observations = len(self.observations), self._to
        d.ids, d.num_ids = self._ids, len(ids)
        d.obs_dim = sum(self._obs_dim(s) for s in self.observations)
        self.obs_dim = d.obs_dim
        # command term
        self._cmd_cfg = command or {"kind": "none"}
        kind = self._cmd_cfg.get("kind", "none")
        d.command_kind = {"none": CMD_NONE, "velocity": CMD_VELOCITY, "pose": CMD_POSE}[kind]
        if kind != "none":
            c = self._cmd_cfg
            rng = [c["resampling_time_range"]] + list(c["ranges"])
            for i, (lo, hi) in enumerate(rng):
                d.cmd_range[i][0], d.cmd_range[i][1] = lo, hi
            d.rel_heading_envs = c.get("rel_heading_envs", 1.0)
            d.rel_standing_envs = c.get("rel_standing_envs", 0.0)
            d.heading_control_stiffness = c.get("heading_control_stiffness", 1.0)
            d.heading_command = int(c.get("heading_command", False))
            d.pose_body_index = int(c.get("body_index", 0))
            d.make_quat_unique = int(c.get("make_quat_unique", False))
            d.max_command_step = c["resampling_time_range"][1] / step_dt
        # buffers
        self.t: dict[str, torch.Tensor] = {}
        n, dev = num_envs, self.device
        nt, nr = len(self.terminations), len(self.rewards)
        self._own("term_dones", torch.zeros(max(nt, 1), n, dtype=torch.bool, device=dev))
        self._own("terminated", torch.zeros(n, dtype=torch.bool, device=dev))
        self._own("time_outs", torch.zeros(n, dtype=torch.bool, device=dev))
        self._own("reset_buf", torch.zeros(n, dtype=torch.bool, device=dev))
        self._own("reward_buf", torch.zeros(n, device=dev))
        self._own("episode_sums", torch.zeros(max(nr, 1), n, device=dev))
        self._own("step_reward", torch.zeros(n, max(nr, 1), device=dev))
        self._own("obs", torch.zeros(n, max(self.obs_dim, 1), device=dev))
        self._own("episode_length_buf", torch.zeros(n, dtype=torch.long, device=dev))
        self._own("log_sums", torch.zeros(nr + nt + 3, device=dev))
        if kind != "none":
            self._own("command", torch.zeros(n, command_dim, device=dev))
            if kind == "pose":
                self.t["command"][:, 3] = 1.0
            self._own("heading_target", torch.zeros(n, device=dev))
            self._own("is_heading_env", torch.zeros(n, dtype=torch.bool, device=dev))
            self._own("is_standing_env", torch.zeros(n, dtype=torch.bool, device=dev))
            self._own("time_left", torch.zeros(n, device=dev))
            self._own("command_counter", torch.zeros(n, dtype=torch.long, device=dev))
            self._own("command_metrics", torch.zeros(2, n, device=dev))
        self.plan = c_void_p()
        _abi.check(_abi.lib.ilx_mdp_plan_create(ctypes.byref(d), ctypes.byref(self.plan)), "ilx_mdp_plan_create")

    @property
    def kernel(self) -> str:
        """The kernel that serves full steps of this plan: "generic" (the plan-interpreting kernel) or the name of the
        kernel compiled for exactly this term list (``ilx_mdp_plan_kernel``)."""
        return _abi.lib.ilx_mdp_plan_kernel(self.plan).decode()

    def select_kernel(self, compiled: bool = True):
        """``compiled=False``: every step goes to the plan-interpreting kernel (``ilx_mdp_plan_select_kernel``)."""
        _abi.check(_abi.lib.ilx_mdp_plan_select_kernel(self.plan, int(compiled)), "ilx_mdp_plan_select_kernel")

    def _obs_dim(self, s: TermSpec) -> int:
        d = self.desc
        nj = d.num_joints if _norm_ids(s.ids) is None else len(_norm_ids(s.ids))
        return {"base_pos_z": 1, "root_quat_w": 4, "joint_pos": nj, "joint_pos_rel": nj, "joint_pos_limit_normalized": nj,
                "joint_vel": nj, "joint_vel_rel": nj, "height_scan": d.num_rays, "last_action": d.action_dim,
                "generated_commands": d.command_dim, "external": d.external_obs_dim[int(s.aux)] if s.kind == "external" else 0}.get(s.kind, 3)

    def _own(self, name: str, t: torch.Tensor):
        self.bind(**{name: t})

    def bind(self, **tensors):
        """Attach GPU tensors to descriptor fields (e.g. ``joint_pos=data.joint_pos``); ``None`` detaches."""
        for name, t in tensors.items():
            if t is None:
                self.t.pop(name, None)
                setattr(self.desc, name, None)
                continue
            if not t.is_cuda or not t.is_contiguous():
                raise _abi.IlxError(f"FusedMdp.bind({name}): need a contiguous GPU tensor")
            self.t[name] = t
            setattr(self.desc, name, t.data_ptr())

    def bind_external(self, slot: int, t: torch.Tensor):
        """Attach the (num_envs, width) buffer an ``external`` observation term copies."""
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise _abi.IlxError(f"FusedMdp.bind_external({slot}): need a contiguous float32 GPU tensor")
        if t.numel() != self.n * self.desc.external_obs_dim[slot]:
            raise ValueError(f"external observation {slot}: {tuple(t.shape)} is not ({self.n}, {self.desc.external_obs_dim[slot]})")
        self._external[slot] = t
        self.desc.external_obs[slot] = t.data_ptr()

    def step(self, phases: int = PHASE_ALL, step_index: int | None = None, hold_step_count: bool = False, scene_update=None):
        """``scene_update`` (``InteractiveScene.describe_update``): the scene update of the last physics substep, which then
        precedes the managers -- inside their launch where the term list allows (``ilx_mdp_step_scene_update``).
        One launch of the given phases; afterwards the generator's step count advances by one (on the host, or -- see
        :meth:`count_steps_on_device` -- by a one-word add on the stream).  An env step issued as several launches
        passes ``hold_step_count=True`` on all but the last, so that every launch of the step draws with the same
        step word; a caller that advances the device-side count itself (:meth:`advance_step_count`) passes it always."""
        self.desc.phases = phases
        if step_index is not None:
            self.desc.step_index = step_index
        elif not self._own_generator and not self._holding:
            self.desc.seed, self.desc.step_index = _rng.seed(), _rng.next_draw("fused_mdp")
        if scene_update is not None:
            _abi.check(_abi.lib.ilx_mdp_step_scene_update(self.plan, ctypes.byref(self.desc), ctypes.byref(scene_update), _abi.stream()),
                       "ilx_mdp_step_scene_update")
        else:
            _abi.check(_abi.lib.ilx_mdp_step(self.plan, ctypes.byref(self.desc), _abi.stream()), "ilx_mdp_step")
        self._holding = hold_step_count
        if not hold_step_count:
            self.advance_step_count()

    def begin_step(self, clear_log_sums: bool = True, advance_step_count: bool = True):
        """The statistics vector cleared and the device-side step count advanced, in one launch (``ilx_mdp_begin_step``);
        a host-side count just advances."""
        counter = self._step_counter if advance_step_count else None
        if advance_step_count and counter is None:
            self.desc.step_index += 1
        log = self.log_sums if clear_log_sums else None
        if log is None and counter is None:
            return
        _abi.check(_abi.lib.ilx_mdp_begin_step(_abi.ptr(log) if log is not None else None, 0 if log is None else log.numel(),
                                               _abi.ptr(counter) if counter is not None else None, _abi.stream()), "ilx_mdp_begin_step")

    def begin_step_args(self, clear_log_sums: bool = True, advance_step_count: bool = True):
        """:meth:`begin_step` for a caller whose first launch of the step can do it on the side
        (``ilx_joint_action_process_begin``): the (log_sums, count, step counter) arguments of that launch.  A host-side
        step count advances here."""
        counter = self._step_counter if advance_step_count else None
        if advance_step_count and counter is None:
            self.desc.step_index += 1
        log = self.log_sums if clear_log_sums else None
        return (_abi.ptr(log) if log is not None else None, 0 if log is None else log.numel(),
                _abi.ptr(counter) if counter is not None else None)

    def advance_step_count(self):
        """The generator's step count, one further: a host integer, or the device word's add on the current stream."""
        if self._step_counter is None:
            self.desc.step_index += 1
        else:
            self._step_counter.add_(1)

    def count_steps_on_device(self, enable: bool = True, start: int = 0):
        """Keep the generator's step count in device memory instead of on the host: the launch adds the device word to
        its step argument and the advance is a one-word add on the stream, so a step recorded in a HIP graph draws
        fresh random numbers at every replay.  ``start``: the word's first value (-1 for a caller that advances the
        count before the step's launches instead of after them)."""
        if enable and self._step_counter is None:
            self._step_counter = torch.full((1,), start, dtype=torch.int64, device=self.device)
            self.desc.step_index_device = self._step_counter.data_ptr()
        elif not enable:
            self._step_counter = None
            self.desc.step_index_device = None

    def __getattr__(self, name):
        t = self.__dict__.get("t")
        if t is not None and name in t:
            return t[name]
        raise AttributeError(name)

    def __del__(self):
        p = self.__dict__.get("plan")
        if p and _abi is not None and getattr(_abi, "lib", None) is not None:
            _abi.lib.ilx_mdp_plan_destroy(p)
            self.plan = None
