"""pyflyt_drone_amd -- MI355X-native vectorised fixed-wing env step.

The package is exactly the hot path of WdBlink/pyflyt-drone named by
BASELINE.json: the fixed-wing flight-dynamics step + reward + observation of
``envs/fixedwing_envs`` for thousands of envs in one HIP kernel, behind the
SB3 ``VecEnv`` surface the reference's training scripts consume.
"""
from . import config
from .config import (FwConfig, train_waypoints_v3_config, waypoints_config)
from .spaces import Box
from .vec_env import (FixedwingObjLockVecEnv, FixedwingVecEnv, FixedwingWaypointObjLockVecEnv,
                      FixedwingWaypointsVecEnv)
from . import rollout
from . import checkpoint, evaluate

__all__ = ["config", "FwConfig", "Box", "FixedwingVecEnv", "FixedwingWaypointsVecEnv", "FixedwingObjLockVecEnv", "FixedwingWaypointObjLockVecEnv",
           "waypoints_config", "train_waypoints_v3_config"]
