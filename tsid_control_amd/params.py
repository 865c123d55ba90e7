"""Flat parameter vector handed through the C-ABI (include/tsidb.h, TSIDB_P_*).

Packs the RobotConfig values the reference feeds to its task constructors
(ctrl/WalkController.py:55-184) into one float64 vector; index names mirror include/tsidb.h.
"""
import numpy as np

P_DT = 0
P_MU = 1
P_FMIN = 2
P_FMAX = 3
P_W_FORCEREF = 4
P_KP_CONTACT = 5
P_KD_CONTACT = 6
P_W_FOOT = 7
P_KP_FOOT = 8
P_KD_FOOT = 9
P_W_COM = 10
P_KP_COM = 11
P_KD_COM = 12
P_W_POSTURE = 13
P_HESS_REG = 14
P_QUIRKS = 15
P_NORMAL = 16
P_CPOINTS = 19
P_KP_POSTURE = 31
P_KD_POSTURE = 51
P_TAU_MAX = 71
P_V_MAX = 91
P_MAX_ITER = 111
P_SIM_ENABLED = 112
P_CLOSED_LOOP = 113
P_W_AM = 114
P_KP_AM = 115
P_REW_SIGMA = 118
P_REW_CTAU = 119
P_DONE_HEIGHT = 120
P_DONE_TILT = 121
P_SELF_COLLISION = 122
P_W_COP = 123
P_SIM_FLOSS_SCALE = 124
P_TSID_ARMATURE = 125
P_FRICTION_COMP = 126
P_PLANE_MESH = 127
P_COUNT = 128


def pack_params(conf, effort_limit, velocity_limit):
    """RobotConfig -> params[P_COUNT].  effort/velocity limits come from the model blob (URDF)."""
    if conf.w_contact >= 0.0:
        raise NotImplementedError(
            "soft contact motion tasks (w_contact >= 0, WalkController.py:82-83) are not built; "
            "the reference configuration uses hard contacts (conf.py:42)")
    if not np.all(np.asarray(conf.masks_posture) == 1):
        raise NotImplementedError("posture masks other than all-ones (conf.py:66) are not built")
    if not (conf.w_torque_bounds > 0.0 and conf.w_joint_bounds > 0.0):
        raise NotImplementedError("the bounds tasks are always in the stack (conf.py:71-72 > 0)")
    p = np.zeros(P_COUNT)
    p[P_DT] = conf.dt
    p[P_MU], p[P_FMIN], p[P_FMAX] = conf.mu, conf.fMin, conf.fMax
    p[P_W_FORCEREF] = conf.w_forceRef
    p[P_KP_CONTACT], p[P_KD_CONTACT] = conf.kp_contact, 2.0 * np.sqrt(conf.kp_contact)
    p[P_W_FOOT], p[P_KP_FOOT], p[P_KD_FOOT] = conf.w_foot, conf.kp_foot, 2.0 * np.sqrt(conf.kp_foot)
    p[P_W_COM], p[P_KP_COM], p[P_KD_COM] = conf.w_com, conf.kp_com, 2.0 * np.sqrt(conf.kp_com)
    p[P_W_POSTURE] = conf.w_posture
    p[P_HESS_REG] = getattr(conf, "hessian_regularization", 1e-8)
    p[P_QUIRKS] = 1.0 if getattr(conf, "reference_quirks", True) else 0.0
    p[P_NORMAL:P_NORMAL + 3] = conf.contactNormal
    # WalkController.py:55-57: 3x4 contact points, columns = points
    cp = np.ones((3, 4)) * (-conf.lz)
    cp[0, :] = [-conf.lxn, -conf.lxn, conf.lxp, conf.lxp]
    cp[1, :] = [-conf.lyn, conf.lyp, -conf.lyn, conf.lyp]
    p[P_CPOINTS:P_CPOINTS + 12] = cp.T.reshape(-1)
    na = len(effort_limit)                      # actuated joints of the blob's robot (20 for v1, 18 for v0; at most 20)
    if na > 20 or len(conf.gain_vector) < na:
        raise ValueError(f"the parameter vector holds 20 joint entries and conf.gain_vector must cover the robot's {na} joints")
    kp_post = conf.kp_posture * np.asarray(conf.gain_vector, dtype=np.float64)[:na]
    p[P_KP_POSTURE:P_KP_POSTURE + na] = kp_post
    p[P_KD_POSTURE:P_KD_POSTURE + na] = 2.0 * np.sqrt(kp_post)
    p[P_TAU_MAX:P_TAU_MAX + na] = conf.tau_max_scaling * np.asarray(effort_limit)
    p[P_V_MAX:P_V_MAX + na] = conf.v_max_scaling * np.asarray(velocity_limit)
    p[P_MAX_ITER] = getattr(conf, "qp_max_iter", 1000)
    p[P_SIM_ENABLED] = 1.0 if getattr(conf, "sim_enabled", True) else 0.0
    p[P_CLOSED_LOOP] = 1.0 if getattr(conf, "closed_loop", False) else 0.0
    # angular-momentum task of the legacy controller (legacy/biped.py:82-87); off in ctrl/WalkController.py
    p[P_W_AM] = getattr(conf, "w_am", 0.0)
    p[P_KP_AM:P_KP_AM + 3] = getattr(conf, "kp_am", 10.0) * np.asarray(getattr(conf, "mask_am", (1.0, 1.0, 0.0)))
    # reward / done outputs (no reference counterpart; SURVEY.md 8d write list)
    p[P_REW_SIGMA] = getattr(conf, "reward_com_sigma", 0.05)
    p[P_REW_CTAU] = getattr(conf, "reward_torque_cost", 1e-3)
    p[P_DONE_HEIGHT] = getattr(conf, "done_base_height", 0.2)
    p[P_DONE_TILT] = np.cos(np.deg2rad(getattr(conf, "done_tilt_deg", 45.0)))
    p[P_SELF_COLLISION] = 1.0 if getattr(conf, "self_collision", True) else 0.0
    p[P_W_COP] = getattr(conf, "w_cop", 0.0)
    p[P_SIM_FLOSS_SCALE] = getattr(conf, "sim_frictionloss_scale", 1.0)
    p[P_TSID_ARMATURE] = getattr(conf, "tsid_armature", 0.0)
    p[P_FRICTION_COMP] = getattr(conf, "friction_compensation", 0.0)
    pm = getattr(conf, "sim_plane_mesh", "mujoco")
    if pm not in ("all", "mujoco"):
        raise ValueError("conf.sim_plane_mesh must be 'all' or 'mujoco'")
    p[P_PLANE_MESH] = 1.0 if pm == "mujoco" else 0.0
    if p[P_CLOSED_LOOP] and not p[P_SIM_ENABLED]:
        raise ValueError("closed_loop needs the sim stage (sim_enabled=True)")
    return p
