"""MI355X-native batched TSID + contact-dynamics hot path of UW-RoboSoccer/tsid_control.

Host surface: RobotConfig (ctrl/conf.py) and WalkController (ctrl/WalkController.py) with
reset()/step(); compute: hand-written HIP kernels behind the C-ABI in include/tsidb.h.
"""
from .conf import RobotConfig, op3_v0_conf  # noqa: F401


def __getattr__(name):
    if name in ("WalkController", "TrajectorySample", "map_tsid_to_mujoco"):
        from . import walk_controller
        return getattr(walk_controller, name)
    if name in ("WalkPlanner", "WalkSchedule", "op3_walking_conf", "op3_walking_posture", "op3_closed_loop_walking_conf"):
        from . import walk_planner
        return getattr(walk_planner, name)
    if name in ("Footstep", "Support", "FootstepPlanner"):
        from . import footstep_planner
        return getattr(footstep_planner, name)
    if name == "FootTrajectory":
        from .foot_trajectory import FootTrajectory
        return FootTrajectory
    if name in ("LIPM", "Trajectory"):
        from . import lipm
        return getattr(lipm, name)
    raise AttributeError(name)
