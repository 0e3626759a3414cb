"""Host-side logic that needs no GPU: the scene-wide reset event over several articulations, the joint action term's
parameter tables, the fork-priority guard of the locomotion step, and the argument checks of the ray-caster entry points
added beside ``ilx_raycaster_refresh``."""

import ctypes
import types

import pytest
import torch

from isaaclab_amd import _abi


class _RecordingAsset:
    """An articulation as ``reset_scene_to_default`` sees it: a data container and the three state writers."""

    def __init__(self, n, j):
        self.data = types.SimpleNamespace(
            root_state_w=torch.zeros(n, 13), default_root_state=torch.zeros(n, 13), default_joint_pos=torch.zeros(n, j),
            default_joint_vel=torch.zeros(n, j), joint_pos=torch.ones(n, j), joint_vel=torch.ones(n, j), _derived_timestamp=3.0)
        self.calls = []

    def write_root_pose_to_sim(self, pose, env_ids=None):
        self.calls.append(("pose", tuple(pose.shape)))

    def write_root_velocity_to_sim(self, vel, env_ids=None):
        self.calls.append(("velocity", tuple(vel.shape)))

    def write_joint_state_to_sim(self, pos, vel, env_ids=None):
        self.calls.append(("joints", tuple(pos.shape)))


@pytest.fixture
def events_without_gpu(monkeypatch):
    """The events module with its one kernel launch and the device-pointer helper replaced by recorders."""
    from isaaclab_amd.envs.mdp import events

    launches = []
    monkeypatch.setattr(events._abi, "event_reset_scene_to_default", lambda *a: launches.append(a) or 0)
    monkeypatch.setattr(events._abi, "ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(events._abi, "stream", lambda: None)
    return events, launches


def test_reset_scene_to_default_writes_every_articulation(events_without_gpu):
    # events.py:922-933 of the reference writes pose, velocity and joint state of EVERY articulation inside the loop
    events, launches = events_without_gpu
    a, b = _RecordingAsset(6, 3), _RecordingAsset(6, 5)
    env = types.SimpleNamespace(scene=types.SimpleNamespace(articulations={"a": a, "b": b}, env_origins=None))
    events.reset_scene_to_default(env, torch.tensor([1, 4]))
    assert len(launches) == 2
    for asset, joints in ((a, 3), (b, 5)):
        assert asset.calls == [("pose", (2, 7)), ("velocity", (2, 6)), ("joints", (2, joints))]
        assert asset.data._derived_timestamp == -1.0


def test_reset_scene_to_default_on_an_empty_scene(events_without_gpu):
    events, launches = events_without_gpu
    env = types.SimpleNamespace(scene=types.SimpleNamespace(articulations={}, env_origins=None))
    events.reset_scene_to_default(env, None)
    assert launches == []


def _joint_action(cfg_kwargs, joint_names=("hip", "knee", "ankle")):
    from isaaclab_amd.envs.mdp.actions import joint_actions as JA

    asset = types.SimpleNamespace(num_joints=len(joint_names),
                                  find_joints=lambda names, preserve_order=False: (list(range(len(joint_names))), list(joint_names)))
    env = types.SimpleNamespace(num_envs=4, device=torch.device("cpu"), scene={"robot": asset})
    cfg = JA.JointEffortActionCfg(asset_name="robot", **cfg_kwargs)
    return JA.JointEffortAction(cfg, env)


def test_joint_action_parameter_tables():
    # joint_actions.py:78-108: numbers stay numbers, {regex: value} tables become per-joint rows over the neutral element
    term = _joint_action(dict(scale=0.25, offset={"knee": 1.5, "h.*": -2.0}, clip={"ankle": (-1.0, 3.0)}))
    assert term._scale == 0.25 and isinstance(term._scale, float)
    assert term._offset.shape == (4, 3) and term._offset[2].tolist() == [-2.0, 1.5, 0.0]
    inf = float("inf")
    assert term._clip.shape == (4, 3, 2) and term._clip[0].tolist() == [[-inf, inf], [-inf, inf], [-1.0, 3.0]]
    assert _joint_action(dict(scale={"knee": 3.0}))._scale[1].tolist() == [1.0, 3.0, 1.0]


@pytest.mark.parametrize("kwargs, message", [
    (dict(scale="big"), "Unsupported scale type: <class 'str'>. Supported types are float and dict."),
    (dict(offset=[1.0]), "Unsupported offset type: <class 'list'>. Supported types are float and dict."),
    (dict(clip=2.0), "Unsupported clip type: <class 'float'>. Supported types are dict."),
])
def test_joint_action_rejects_other_types_like_the_reference(kwargs, message):
    with pytest.raises(ValueError) as err:
        _joint_action(kwargs)
    assert str(err.value) == message


def test_step_refuses_a_fork_between_streams_of_unequal_priority():
    from isaaclab_amd.envs.locomotion_step import LocomotionVelocityStep

    LocomotionVelocityStep.check_fork_priorities(0, 0)
    LocomotionVelocityStep.check_fork_priorities(-1, -1)
    with pytest.raises(_abi.IlxError, match="priority -1 onto one of priority 0"):
        LocomotionVelocityStep.check_fork_priorities(-1, 0)


def test_fork_stream_is_made_with_the_callers_priority(monkeypatch):
    from isaaclab_amd.envs import locomotion_step as LS

    made = []

    class FakeStream:
        def __init__(self, device=None, priority=0):
            self.priority = priority
            made.append(priority)

    monkeypatch.setattr(LS.torch.cuda, "Stream", FakeStream)
    step = LS.LocomotionVelocityStep.__new__(LS.LocomotionVelocityStep)
    step.device, step._side_streams = torch.device("cpu"), {}
    high, normal = FakeStream(priority=-1), FakeStream(priority=0)
    made.clear()
    assert step._fork_stream(high).priority == -1 and step._fork_stream(normal).priority == 0
    assert step._fork_stream(high) is step._fork_stream(high) and made == [-1, 0]


def test_static_refresh_refuses_a_preparation_it_cannot_have():
    # ILX_RAYS_PREPARED names records in the per-stream workspace; the entry point that allocates nothing reads its own
    # buffer, so the hint is an error there -- checked before any device is touched
    desc = _abi.RayCasterDesc(mesh=ctypes.c_void_p(0x1000), num_envs=8, num_rays=4, hints=_abi.RAYS_PREPARED | _abi.RAYS_Z_PARALLEL,
                              pose_pos=1, pose_quat=1, pose_pos_stride=3, pose_quat_stride=4, ray_starts_local=1,
                              ray_directions_local=1, pos_w=1, quat_w=1, ray_hits_w=1, timestamp=1, timestamp_last_update=1,
                              is_outdated=1)
    assert _abi.raycaster_refresh_static(ctypes.byref(desc), None) == -1
    assert "ILX_RAYS_PREPARED" in _abi.lib.ilx_last_error().decode()


def test_reserve_and_forget_check_their_arguments():
    assert _abi.raycaster_reserve(None, 16, 187, 2) == -1 and "mesh is NULL" in _abi.lib.ilx_last_error().decode()
    assert _abi.mesh_forget_patterns(None) == -1
