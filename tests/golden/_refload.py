"""Loader that imports individual files of the reference checkout by path.

Only used by ``gen_golden.py`` inside the build container, where the reference
sources are mounted at ``/root/reference``.  The reference package cannot be imported
as a whole (it needs the simulator, ``warp``, ``toml`` ...), so single files are
executed under their real dotted names and everything they import that is not
available is replaced by permissive placeholder modules.

Nothing in here is used by the product, by ``pytest``, or on the GPU box.
"""

from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = os.environ.get("ISAACLAB_REFERENCE", "/root/reference")
L = os.path.join(REF, "source/isaaclab/isaaclab")
T = os.path.join(REF, "source/isaaclab_tasks/isaaclab_tasks")


class Anything:
    """Placeholder object: any attribute, call or subscript yields another placeholder."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Anything()

    def __call__(self, *a, **k):
        return Anything()

    def __getitem__(self, key):
        return Anything()


class PlaceholderModule(types.ModuleType):
    """Module whose unknown attributes resolve to the :class:`Anything` class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Anything


def _attach(name: str, mod: types.ModuleType):
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, mod)


def placeholder(name: str, **attrs) -> types.ModuleType:
    """Register a permissive placeholder package under ``name`` (and its parents)."""
    parts = name.split(".")
    for i in range(1, len(parts)):
        parent = ".".join(parts[:i])
        if parent not in sys.modules:
            placeholder(parent)
    mod = sys.modules.get(name)
    if mod is None:
        mod = PlaceholderModule(name)
        mod.__path__ = []  # type: ignore[attr-defined]
        sys.modules[name] = mod
        _attach(name, mod)
    mod.__dict__.update(attrs)
    return mod


def load(name: str, path: str) -> types.ModuleType:
    """Execute the reference file ``path`` as module ``name``."""
    parts = name.split(".")
    for i in range(1, len(parts)):
        parent = ".".join(parts[:i])
        if parent not in sys.modules:
            placeholder(parent)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    _attach(name, mod)
    spec.loader.exec_module(mod)
    return mod


def available() -> bool:
    return os.path.isdir(L)


class SceneEntity:
    """Stand-in for ``SceneEntityCfg`` after resolution: name + resolved ids."""

    def __init__(self, name, joint_ids=slice(None), body_ids=slice(None)):
        self.name = name
        self.joint_ids = joint_ids
        self.body_ids = body_ids
        self.joint_names = None
        self.body_names = None
        self.preserve_order = False

    def resolve(self, scene):
        pass


_BOOTSTRAPPED = {}


def bootstrap():
    """Load the reference modules needed by the generator; returns a dict of modules."""
    if _BOOTSTRAPPED:
        return _BOOTSTRAPPED
    import logging

    # third-party / simulator packages that do not exist here
    for n in [
        "omni", "omni.log", "omni.kit", "omni.kit.app", "omni.physics", "omni.physics.tensors",
        "omni.physics.tensors.impl", "omni.physics.tensors.impl.api", "omni.usd", "pxr", "carb", "warp",
        "isaacsim", "isaacsim.core", "isaacsim.core.prims", "isaacsim.core.utils", "isaacsim.core.utils.torch",
        "prettytable", "trimesh", "toml", "gymnasium",
    ]:
        placeholder(n)
    sys.modules["omni.log"].info = lambda *a, **k: None
    sys.modules["omni.log"].warn = lambda *a, **k: None
    sys.modules["omni.log"].error = lambda *a, **k: None

    class _Tbl:
        def __init__(self, *a, **k):
            self.align = {}
            self.title = ""
            self.field_names = []

        def add_row(self, *a, **k):
            pass

        def get_string(self):
            return ""

    sys.modules["prettytable"].PrettyTable = _Tbl

    placeholder("isaaclab")
    placeholder("isaaclab.utils")
    m = {}
    m["math"] = load("isaaclab.utils.math", f"{L}/utils/math.py")
    m["string"] = load("isaaclab.utils.string", f"{L}/utils/string.py")
    m["types"] = load("isaaclab.utils.types", f"{L}/utils/types.py")
    m["array"] = load("isaaclab.utils.array", f"{L}/utils/array.py")
    m["dict"] = load("isaaclab.utils.dict", f"{L}/utils/dict.py")
    m["configclass"] = load("isaaclab.utils.configclass", f"{L}/utils/configclass.py")
    sys.modules["isaaclab.utils"].configclass = m["configclass"].configclass
    placeholder("isaaclab.utils.buffers")
    m["circular_buffer"] = load("isaaclab.utils.buffers.circular_buffer", f"{L}/utils/buffers/circular_buffer.py")
    m["timestamped_buffer"] = load("isaaclab.utils.buffers.timestamped_buffer", f"{L}/utils/buffers/timestamped_buffer.py")
    sys.modules["isaaclab.utils.buffers"].CircularBuffer = m["circular_buffer"].CircularBuffer
    sys.modules["isaaclab.utils.buffers"].TimestampedBuffer = m["timestamped_buffer"].TimestampedBuffer
    placeholder("isaaclab.utils.noise")
    # noise_cfg and noise_model import each other; model first (it needs the cfg only for typing)
    m["noise_model"] = load("isaaclab.utils.noise.noise_model", f"{L}/utils/noise/noise_model.py")
    m["noise_cfg"] = load("isaaclab.utils.noise.noise_cfg", f"{L}/utils/noise/noise_cfg.py")
    for k in ("NoiseCfg", "UniformNoiseCfg", "GaussianNoiseCfg", "ConstantNoiseCfg", "NoiseModelCfg"):
        setattr(sys.modules["isaaclab.utils.noise"], k, getattr(m["noise_cfg"], k))
    placeholder("isaaclab.utils.modifiers")
    placeholder("isaaclab.utils.assets", read_file=lambda p: p)

    placeholder("isaaclab.assets")
    placeholder("isaaclab.assets.articulation")
    placeholder("isaaclab.sensors")
    placeholder("isaaclab.scene")
    placeholder("isaaclab.sim")
    placeholder("isaaclab.sim.utils")
    placeholder("isaaclab.markers")
    placeholder("isaaclab.markers.config")
    placeholder("isaaclab.envs")
    placeholder("isaaclab.terrains")
    placeholder("isaaclab.terrains.trimesh")
    placeholder("isaaclab.terrains.trimesh.utils")
    placeholder("isaaclab.controllers")
    placeholder("isaaclab.controllers.operational_space")

    placeholder("isaaclab.managers")
    m["scene_entity_cfg"] = load("isaaclab.managers.scene_entity_cfg", f"{L}/managers/scene_entity_cfg.py")
    m["manager_term_cfg"] = load("isaaclab.managers.manager_term_cfg", f"{L}/managers/manager_term_cfg.py")
    sys.modules["isaaclab.managers"].SceneEntityCfg = m["scene_entity_cfg"].SceneEntityCfg
    for k in ("ObservationTermCfg", "ObservationGroupCfg", "RewardTermCfg", "TerminationTermCfg",
              "ManagerTermBaseCfg", "ActionTermCfg", "CommandTermCfg"):
        setattr(sys.modules["isaaclab.managers"], k, getattr(m["manager_term_cfg"], k))
    m["manager_base"] = load("isaaclab.managers.manager_base", f"{L}/managers/manager_base.py")
    sys.modules["isaaclab.managers"].ManagerTermBase = m["manager_base"].ManagerTermBase
    sys.modules["isaaclab.managers"].ManagerBase = m["manager_base"].ManagerBase
    m["reward_manager"] = load("isaaclab.managers.reward_manager", f"{L}/managers/reward_manager.py")
    m["termination_manager"] = load("isaaclab.managers.termination_manager", f"{L}/managers/termination_manager.py")
    m["observation_manager"] = load("isaaclab.managers.observation_manager", f"{L}/managers/observation_manager.py")

    placeholder("isaaclab.envs.mdp")
    m["rewards"] = load("isaaclab.envs.mdp.rewards", f"{L}/envs/mdp/rewards.py")
    m["terminations"] = load("isaaclab.envs.mdp.terminations", f"{L}/envs/mdp/terminations.py")
    m["observations"] = load("isaaclab.envs.mdp.observations", f"{L}/envs/mdp/observations.py")

    placeholder("isaaclab.actuators")
    m["actuator_base"] = load("isaaclab.actuators.actuator_base", f"{L}/actuators/actuator_base.py")
    m["actuator_pd"] = load("isaaclab.actuators.actuator_pd", f"{L}/actuators/actuator_pd.py")
    m["actuator_net"] = load("isaaclab.actuators.actuator_net", f"{L}/actuators/actuator_net.py")

    m["differential_ik"] = load("isaaclab.controllers.differential_ik", f"{L}/controllers/differential_ik.py")

    m["vel_rewards"] = load(
        "isaaclab_tasks.manager_based.locomotion.velocity.mdp.rewards",
        f"{T}/manager_based/locomotion/velocity/mdp/rewards.py",
    )
    m["reach_rewards"] = load(
        "isaaclab_tasks.manager_based.manipulation.reach.mdp.rewards",
        f"{T}/manager_based/manipulation/reach/mdp/rewards.py",
    )
    m["cartpole_rewards"] = load(
        "isaaclab_tasks.manager_based.classic.cartpole.mdp.rewards",
        f"{T}/manager_based/classic/cartpole/mdp/rewards.py",
    )
    logging.getLogger().setLevel(logging.WARNING)
    _BOOTSTRAPPED.update(m)
    return m
