"""RobotConfig - attribute-for-attribute mirror of the reference's ctrl/conf.py:5-75.

Every attribute name and value of the reference class is kept; the reference's
`import pinocchio` (conf.py:2-3) is dropped because nothing here needs it.  Batch-only knobs are
added at the bottom without renaming anything.
"""
import numpy as np


class RobotConfig:
    """Configuration parameters for the robot controller (ctrl/conf.py:5-75)."""

    # Robot configuration (conf.py:9-18).  The asset files themselves do not travel; the compiled
    # model blob below stands for robot_mod.urdf + robot.srdf + scene.xml.
    robot_path = "./robot/v1"
    root_urdf = f"{robot_path}/urdf"
    urdf = f"{robot_path}/urdf/robot_mod.urdf"
    pin_urdf = f"{root_urdf}/robot_mod.urdf"
    mjcf = f"{robot_path}/mujoco/scene.xml"
    srdf = f"{root_urdf}/robot.srdf"

    lf_fixed_joint = "left_sole_joint_fixed"
    rf_fixed_joint = "right_sole_joint_fixed"

    # Controller settings (conf.py:21)
    dt = 0.002

    # Step parameters (conf.py:24-28)
    step_height = 0.2
    step_width = 0.2
    step_length = 0.3
    step_duration = 0.5
    rise_ratio = 0.5

    # Frame dimensions (conf.py:31-35)
    lxn = 0.055
    lyn = 0.0275
    lxp = 0.055
    lyp = 0.0275
    lz = 0.0

    # Contact parameters (conf.py:38-44)
    mu = 0.5
    fMin = 10.0
    fMax = 1000.0
    contactNormal = np.array([0.0, 0.0, 1.0])
    w_contact = -1.0
    w_forceRef = 1e-5
    kp_contact = 10.0

    # Foot trajectory parameters (conf.py:47-48)
    w_foot = 1e-1
    kp_foot = 10.0

    # CoM parameters (conf.py:51-52)
    w_com = 1e-1
    kp_com = 10.0

    # Posture parameters (conf.py:55-63)
    w_posture = 1e-1
    kp_posture = 10.0
    gain_vector = np.array([
        100.0, 100.0,  # head
        10.0, 5.0, 5.0, 1.0, 1.0, 1.0,  # left leg
        10.0, 10.0, 10.0,  # left arm
        10.0, 5.0, 5.0, 1.0, 1.0, 1.0,  # right leg
        10.0, 10.0, 10.0,  # right arm
    ])

    # Masks for posture task (conf.py:66)
    masks_posture = np.ones(20)

    # Joint limit parameters (conf.py:69-72)
    tau_max_scaling = 5.0
    v_max_scaling = 10.0
    w_torque_bounds = 1e-2
    w_joint_bounds = 1e-2

    # conf.py:74-75
    visualizer = None

    # ---- batch-only additions (no reference counterpart)
    model_blob = None          # path to a compiled model blob; None -> packaged assets/op3_v1.tsidb
    num_envs = 1
    device = "cuda"
    dtype = "f64"              # arithmetic type of the HIP path: "f64" (reference precision) or "f32"
    reference_quirks = True    # reproduce SURVEY.md F6 (a), (e): quaternion slot copy, stale CoP frames
    sim_enabled = True         # run the MuJoCo-subset step (main.py:192-195) after each TSID tick
    closed_loop = False        # TSID reads the sim state each tick and the sim is driven by tau
    #                            (torque actuators) instead of the reference's teleport + position servos
    w_am = 0.0                 # angular-momentum task of legacy/biped.py:82-87 (legacy/op3_conf.py:15 uses 1e-3);
    kp_am = 10.0               #   0 = not in the stack, as in ctrl/WalkController.py.  Gains kp_am * mask_am
    mask_am = (1.0, 1.0, 0.0)  #   (legacy/biped.py:83), zero reference (legacy/biped.py:86-87)
    qp_max_iter = 1000         # eiquadprog-fast DEFAULT_MAX_ITER
    hessian_regularization = 1e-8  # tsid SolverHQuadProgFast default
    pipeline_sim_batch = 0     # WalkController.step_pipelined(): sim stages enqueued on their stream this many at a time, as ONE
                               # launch that steps every env that many times (at most 8); 0 = auto (8 for up to 1024 envs, else 1)
    sim_waves = 0              # wavefronts per env in the sim kernel: 1, 2 (collision phase beside the unconstrained dynamics: lower
                               # step latency for small batches; bit-identical), 0 = auto (2 up to 512 envs)
    self_collision = True      # sim stage collides the robot<->robot convex-hull pairs, as mj_step does (main.py:195);
    #                            False = floor contacts only (round-1 behaviour)
    qp_fast_equalities = -1    # the tick's QP: 1 = try the equality-constrained optimum by a small Cholesky first (float64; results to
                               # rounding, include/tsidb.h TSIDB_OPT_QP_FAST_EQ), 0 = always the QR, -1 = the library's default (1)
    sim_pack = -1              # sim kernel layout: 1 = two envs per wavefront (32 lanes each, tsidb_sim2.hpp; per env bit-identical to
                               # the one-env kernel), 0 = one env per wavefront, -1 = the library's choice for the batch size
    sim_plane_mesh = "mujoco"  # sim stage, floor <-> hull contacts (main.py:195).  "mujoco" = upstream's plane-mesh rule as far as it
    #                            is known here (mujoco itself is not available): the support vertex, then in hull-graph order at most
    #                            3 more of its neighbours within the margin, each >= 0.3 x the geom's bounding radius from the first;
    #                            "all" = every neighbour within the margin (rounds 1-2: up to 30 contacts under one flat sole)
    w_cop = 0.0                # CoP force task of legacy/biped.py:79-80 (legacy/op3_conf.py:14 uses 0); 0 = off
    sim_frictionloss_scale = 1.0   # closed-loop knobs, neutral = the reference's models: scale of the sim's joint frictionloss
    tsid_armature = 0.0            #   (robot.xml:8), rotor inertia on the diagonal of TSID's M (the sim has 0.005, the URDF none),
    friction_compensation = 0.0    #   Coulomb-friction feed-forward in tau [N m] (direction of the commanded joint velocity)
    reward_com_sigma = 0.05    # reward = exp(-|com - com_ref|^2 / sigma^2) - reward_torque_cost * |tau|^2   (per tick)
    reward_torque_cost = 1e-3
    done_base_height = 0.2     # done = failed QP, base lower than this [m], or tilted by more than done_tilt_deg
    done_tilt_deg = 45.0


def op3_v0_conf() -> RobotConfig:
    """The v0 robot (robot/v0/robot.urdf + robot.srdf: 18 actuated joints, no ankle roll; sim model robot/v0/robot.xml)
    with the values of its own configuration, legacy/op3_conf.py:11-62 - SURVEY.md 8f-4.  Its own blob
    (assets/op3_v0.tsidb) and its own build of the library (libtsidb_v0.so).  The stack is this package's
    (ctrl/WalkController.py's): the joint-bound rows stay in even though legacy/op3_conf.py:21 switches them off
    (w_joint_bounds = 0; at +-10 x the velocity limit they never activate)."""
    from pathlib import Path
    c = RobotConfig()
    c.model_blob = str(Path(__file__).parent / "assets" / "op3_v0.tsidb")
    c.robot_path, c.urdf, c.srdf, c.mjcf = "./robot", "./robot/robot.urdf", "./robot/robot.srdf", "./robot/robot.xml"   # op3_conf.py:56-62
    c.lf_fixed_joint, c.rf_fixed_joint = "leg_left_sole_joint_fixed", "leg_right_sole_joint_fixed"            # op3_conf.py:53-54
    c.step_length, c.step_height, c.step_width, c.step_duration = 0.1, 0.05, 0.1275, 0.7                         # op3_conf.py:8-11
    c.w_com, c.w_am, c.w_foot, c.w_posture, c.w_forceRef, c.w_cop = 1.0, 1e-3, 1e-1, 1e-1, 1e-5, 0.0               # op3_conf.py:13-20
    c.w_torque_bounds = 1e-1
    c.lxp = c.lxn = 0.0275                                                                                       # op3_conf.py:23-27
    c.lyp = c.lyn = 0.055
    c.mu, c.fMin, c.fMax = 0.5, 0.0, 1000.0
    c.tau_max_scaling, c.v_max_scaling = 3.0, 10.0                                                               # op3_conf.py:33-34
    c.kp_contact = c.kp_foot = c.kp_com = c.kp_am = 10.0
    c.kp_posture = 1.0
    c.masks_posture = np.ones(18)
    c.gain_vector = np.array([100.0, 100.0, 10.0, 5.0, 5.0, 1.0, 1.0, 10.0, 10.0, 10.0,
                              10.0, 5.0, 5.0, 1.0, 1.0, 10.0, 10.0, 10.0])                                         # op3_conf.py:42-48
    return c
