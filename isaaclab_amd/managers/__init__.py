from .manager_term_cfg import (ManagerTermBaseCfg, ObservationGroupCfg, ObservationTermCfg, RewardTermCfg,  # noqa: F401
                               TerminationTermCfg, UniformNoiseCfg)
from .observation_manager import ObservationManager  # noqa: F401
from .reward_manager import RewardManager  # noqa: F401
from .scene_entity_cfg import SceneEntityCfg  # noqa: F401
from .termination_manager import TerminationManager  # noqa: F401
from .action_manager import ActionManager, ActionTerm, ActionTermCfg  # noqa: F401
from .command_manager import CommandManager, CommandTerm, CommandTermCfg  # noqa: F401
from .event_manager import EventManager, EventTermCfg, ManagerTermBase  # noqa: F401
