"""Walking wiring for config 3 (4096 LIPM walkers): footsteps -> swing trajectories -> per-tick
foot-task samples and contact flags for WalkController.update_tasks.

The reference only sketches this (ctrl/Walk_Planner.py:14-32 builds a FootTrajectory from footstep i
to footstep i+2 per step and cannot run: bad imports, `len(footsteps - 2)`, undefined `self.t`;
SURVEY.md F5) and never calls update_tasks (main.py:117 is commented out).  WalkPlanner.plan keeps
the reference's intent and signature; WalkSchedule is the batched table + evaluator the build
authors: polynomial coefficients per (env, step) held on the device, evaluated each tick as tensor
expressions (no host round trip).
"""
from typing import List

import numpy as np
import torch

from .conf import RobotConfig
from .foot_trajectory import FootTrajectory
from .footstep_planner import Footstep, FootstepPlanner, resample_path, unicycle_path
from .lipm import GRAVITY, eval_segment, segment


class WalkPlanner:
    def __init__(self, conf: RobotConfig = None):
        self.conf = conf or RobotConfig()
        self.t = 0.0

    def plan(self, footsteps: List[Footstep]):
        """Swing trajectory from footstep i to footstep i+2, one per step (Walk_Planner.py:23-32)."""
        c = self.conf
        swings = []
        for i in range(len(footsteps) - 2):
            start = [*footsteps[i].position, 0.0, footsteps[i].orientation[2]]
            target = [*footsteps[i + 2].position, 0.0, footsteps[i + 2].orientation[2]]
            swings.append(FootTrajectory([0.0, c.step_duration], start=start, target=target,
                                         step_height=c.step_height, rise_ratio=c.rise_ratio))
            self.t += c.step_duration
        return swings


def op3_walking_conf(conf: RobotConfig = None) -> RobotConfig:
    """Config-3 walking workload sized for the OP3.  The reference's step parameters (conf.py:24-28:
    0.3 m steps, 0.2 m apart and 0.2 m high) belong to a robot several times the OP3's size (leg length
    0.22 m, feet 0.085 m apart), and its task weights (conf.py:47-60, posture as heavy as the feet and the
    CoM) hold the standing posture - with them the TSID state does not follow a footstep plan at all and
    diverges after ~2500 ticks (tools/walk_trace.py reproduces it on the CPU).  The reference never ran
    its walking branch (main.py:117 is commented out), so these are the build's own values,
    in the proportions of tsid's biped walking example: tracking tasks dominate, posture regularises."""
    c = conf or RobotConfig()
    c.step_length, c.step_width, c.step_height, c.step_duration, c.rise_ratio = 0.05, 0.085, 0.03, 0.5, 0.5
    c.w_com, c.w_foot, c.w_posture = 1.0, 1.0, 1e-3
    c.kp_com, c.kp_foot = 20.0, 100.0
    # the reference's swing trajectories touch down at 4 h / T = 0.24 m/s (a parabola in z, a straight line in
    # x and y); with kp_contact = 10 the re-referenced contact overshoots by centimetres before it stops
    c.kp_contact = 900.0
    return c


def op3_closed_loop_walking_conf(conf: RobotConfig = None) -> RobotConfig:
    """Walking with the loop closed (TSID reads the sim state each tick, the sim is driven by tau).
    On top of op3_walking_conf, what a controller needs once its torques meet the sim's model instead of its own:
      - the contact rectangle INSIDE the real sole: the reference's rectangle (conf.py:31-34, +-0.055 in x,
        +-0.0275 in y) has x and y swapped against the collision mesh (0.055 wide in x, 0.13 long in y; SURVEY.md F6d,
        legacy/op3_conf.py:24-27 has it the right way round) - a centre of pressure TSID believes feasible would tip
        the real foot over its edge;
      - the sim's rotor inertia (armature 0.005, robot.xml:8) in TSID's mass matrix and a feed-forward for its joint
        Coulomb friction (frictionloss 0.1 N m): the URDF has neither;
      - stiffer contact and CoM gains.
    Use WalkSchedule(..., foot_press=0) with it (the swing foot is aimed AT the floor, not into it) and
    enable_touchdown_feedback().  64 envs walk 5000 ticks (18 steps) without a fall
    (tests/test_gpu_parity.py::test_closed_loop_walking_does_not_fall)."""
    c = op3_walking_conf(conf)
    c.reference_quirks = False
    c.closed_loop = True
    c.lxn = c.lxp = 0.02
    c.lyn = c.lyp = 0.05
    c.tsid_armature = 0.005
    c.friction_compensation = 0.1
    c.kp_contact, c.kp_com = 400.0, 40.0
    return c


def op3_walking_posture(bend=0.45):
    """Posture-task reference for walking: knees bent by 2*bend with the feet kept flat under the hips
    (TSID joint order: left leg hip pitch / knee / ankle pitch = joints 4, 5, 6, right leg 13, 14, 15; the
    ankle axes point the other way).  The SRDF standing pose (robot.srdf:5-25) has the legs straight, which
    is singular for lowering the hips; bend = 0.45 rad lowers them by 0.22 (1 - cos bend) = 2.2 cm."""
    p = np.zeros(20)
    p[4], p[5], p[6] = bend, -2.0 * bend, -bend
    p[13], p[14], p[15] = -bend, 2.0 * bend, bend
    return p


PLAN_NPARAMS = 16


def plan_params(conf: RobotConfig, t_start=1.0, com_drop=0.015, foot_press=0.002, resample_ds=None, unicycle=(0.5, 0.1, 0.1, 100),
                scale_range=(0.5, 1.0), seed=1):
    """The 16 numbers tsidb_walk_plan takes (include/tsidb.h): step_length, step_width, step_height, step_duration,
    rise_ratio (ctrl/conf.py:24-28), t_start, com_drop, foot_press, resample_ds (default step_length / 10; 0 = path
    vertices as given), the unicycle path v, w, dt, n (Footstep_Planner.py:131-141), the range the per-env path scale is
    drawn from, the seed of that draw."""
    ds = conf.step_length / 10 if resample_ds is None else resample_ds
    return np.array([conf.step_length, conf.step_width, conf.step_height, conf.step_duration, conf.rise_ratio, t_start, com_drop,
                     foot_press, ds, *unicycle, *scale_range, float(seed)], dtype=np.float64)


class WalkSchedule:
    """Per-env walking tables, evaluated each tick on the device (no host round trip).

    Timeline: [0, t_start) both feet down, the CoM sinks by com_drop and the divergent component of
    motion (DCM) is brought to the first step's start value; step k occupies
    [t_start + k T, t_start + (k+1) T) in single support on the foot the plan put down last; after the
    last step both feet are down again.

      coef  [N, K, 4, 4]   swing polynomials x, y, z, yaw in time since step start (a12)
      side  [N, K]         0 = left foot swings;  nsteps [N]
      rest  [N, K+1, 2, 4] (x, y, yaw, z) of [left, right] foot before step k
      com   [N, K+2, 2, 3] LIPM segment (zmp, d, c) per planar axis for phase 0 (start), 1..ns (steps),
                           ns+1 (final stand):  x(s) = zmp + d/2 e^{w s} + c e^{-w s}  (a13; the closed
                           form of LIPM.py:44-49's recurrence about a fixed ZMP), s = time in the phase.
    Yaw is relative to the robot's initial heading."""

    def __init__(self, plans: List[List[Footstep]], conf: RobotConfig, device, dtype, com0=None, heading=0.0,
                 t_start=1.0, com_drop=0.015, foot_press=0.002):
        """foot_press: every swing is aimed this far below the height the foot started the episode at, so
        that the slave sim (base teleported to the TSID pose each step, main.py:192) finds the stance foot
        pressed into the floor instead of hovering a tracking error above it."""
        self.conf = conf
        N = len(plans)
        K = max(len(p) - 2 for p in plans)
        T = conf.step_duration
        com0 = np.asarray(com0 if com0 is not None else [0.0, 0.0, 0.24], dtype=np.float64)
        self.z0, self.dz, self.t_start = float(com0[2]), float(com_drop), float(t_start)
        self.omega = float(np.sqrt(GRAVITY / (self.z0 - self.dz)))  # LIPM.py:15
        w = self.omega
        coef = np.zeros((N, K, 4, 4))
        side = np.zeros((N, K), dtype=np.int64)
        nsteps = np.zeros(N, dtype=np.int64)
        rest = np.zeros((N, K + 1, 2, 4))
        com = np.zeros((N, K + 2, 2, 3))
        done = {}
        for e, steps in enumerate(plans):
            if id(steps) in done:  # envs sharing a plan share its tables
                s = done[id(steps)]
                coef[e], side[e], nsteps[e], rest[e], com[e] = coef[s], side[s], nsteps[s], rest[s], com[s]
                continue
            done[id(steps)] = e
            ns = nsteps[e] = max(len(steps) - 2, 0)
            cur = {}
            for s in steps[:2]:
                cur[int(bool(s.side))] = np.array([s.position[0], s.position[1], s.orientation[2] - heading, 0.0])
            for k in range(K + 1):
                rest[e, k, 0], rest[e, k, 1] = cur[0], cur[1]
                if k < ns:
                    # swing from footstep k to footstep k+2 (Walk_Planner.py:23-31)
                    sd = int(bool(steps[k].side))
                    tgt = steps[k + 2]
                    nxt = np.array([tgt.position[0], tgt.position[1], tgt.orientation[2] - heading, -foot_press])
                    c0 = cur[sd]
                    coef[e, k] = FootTrajectory([0.0, T], start=[c0[0], c0[1], c0[3], c0[2]], target=[nxt[0], nxt[1], nxt[3], nxt[2]],
                                                step_height=conf.step_height, rise_ratio=conf.rise_ratio).coefficients()
                    side[e, k] = sd
                    cur[sd] = nxt
            # DCM end points backwards from the final stand (ZMP of step k = the stance foot = steps[k+1])
            final = 0.5 * (steps[ns].position + steps[ns + 1].position) if ns > 0 else com0[:2]
            xi = np.zeros((ns + 1, 2))
            xi[ns] = final
            for k in range(ns - 1, -1, -1):
                z = steps[k + 1].position
                xi[k] = z + (xi[k + 1] - z) * np.exp(-w * T)
            # phase 0: constant ZMP that carries the DCM from the initial CoM (at rest) to xi[0] in t_start
            x = com0[:2].copy()
            E0 = np.exp(w * t_start)
            z = (xi[0] - x * E0) / (1.0 - E0)
            d, c = segment(w, z, x, dcm0=x)
            com[e, 0, :, 0], com[e, 0, :, 1], com[e, 0, :, 2] = z, d, c
            x = eval_segment(w, z, d, c, t_start)[0]
            for k in range(ns):
                z = steps[k + 1].position
                d, c = segment(w, z, x, dcm0=xi[k])
                com[e, k + 1, :, 0], com[e, k + 1, :, 1], com[e, k + 1, :, 2] = z, d, c
                x = eval_segment(w, z, d, c, T)[0]
            com[e, ns + 1:, :, 0] = final
            com[e, ns + 1:, :, 2] = x - final
        t = lambda a, dt=dtype: torch.as_tensor(a, device=device).to(dt)
        self.coef, self.rest, self.com = t(coef), t(rest), t(com)
        self.side, self.nsteps = t(side, torch.long), t(nsteps, torch.long)
        self.N, self.K = N, K
        self.device, self.dtype = device, dtype
        self.t_offset = None  # [N] per-env start delay (set_phase_offsets); None = every env on the same clock
        self.td_latch, self.td_fraction = None, 0.6  # contact-timing feedback (enable_touchdown_feedback)

    @classmethod
    def on_device(cls, wc, conf: RobotConfig = None, K=104, t_start=1.0, com_drop=0.015, foot_press=0.002, seed=1,
                  scale_range=(0.5, 1.0), unicycle=(0.5, 0.1, 0.1, 100), resample_ds=None, scale=None, plan=True):
        """A schedule whose tables live on the device from the start and are BUILT there (tsidb_walk_plan): K = capacity
        in steps per env; every env walks the reference's demo unicycle path (Footstep_Planner.py:131-141) scaled by
        scale[e] if given, else by U(scale_range) drawn on the device from hash(seed, env, episode[e]).  plan(...) /
        restart via WalkController.reset(env_ids, sched=...) / reset_done(sched) replan single envs without touching
        the host.  (Vertex spacing resample_ds = step_length / 10 by default: a path whose pieces add up to exactly one
        step_length - scales that are multiples of 0.1 - leaves `travelled >= step_length`, Footstep_Planner.py:106, to
        rounding, and two implementations may then differ by one vertex in where a step lands.)"""
        self = cls.__new__(cls)
        conf = conf if conf is not None else wc.conf
        self.conf, self.N, self.K = conf, wc.num_envs, int(K)
        self.device, self.dtype = wc.device, wc.dtype
        N, dev = self.N, wc.device
        z = lambda *sh, dt=wc.dtype: torch.zeros(*sh, dtype=dt, device=dev)
        self.coef, self.rest, self.com = z(N, K, 4, 4), z(N, K + 1, 2, 4), z(N, K + 2, 2, 3)
        self.side, self.nsteps = z(N, K, dt=torch.int32), z(N, dt=torch.int32)
        self.steps, self.flags = z(N, K + 2, 4, dt=torch.float64), z(N, dt=torch.int32)
        self.episode = z(N, dt=torch.int32)
        self.scale = None if scale is None else torch.as_tensor(scale, dtype=torch.float64, device=dev).contiguous()
        if self.scale is not None and self.scale.numel() != N:
            raise ValueError("WalkSchedule.on_device: scale must hold one value per env")
        self.t_offset = z(N)
        self.td_latch, self.td_fraction = None, 0.6
        self._side32, self._nsteps32, self._coef_c, self._rest_c, self._com_c = self.side, self.nsteps, self.coef, self.rest, self.com
        self.pp = plan_params(conf, t_start, com_drop, foot_press, resample_ds, unicycle, scale_range, seed)
        self.t_start, self.dz = float(t_start), float(com_drop)
        self.z0 = float(wc.com_ref[0, 2])   # (standing CoM height: the same for every env right after a reset)
        self.omega = float(np.sqrt(GRAVITY / max(self.z0 - self.dz, 1e-3)))   # (k_plan's guard: such envs are flagged and stand)
        if plan:
            self.plan(wc)
        return self

    def plan(self, wc, env_ids=None, t=0.0, done_only=False, new_paths=False, path=None, npts=None, t_device=None):
        """(Re)build the plan of the selected envs ON THE DEVICE and restart their clocks at time t (env time = t' - t
        from then on; the touch-down latch is cleared): env_ids = list / tensor (None = all), done_only = only envs
        whose done flag is set in wc.rows (no host sync), new_paths = advance the env's episode counter first, i.e. draw
        a new path scale.  path [N,P,2] float64 + npts [N]: explicit polylines (world coordinates) instead of the
        unicycle path.  Needs the envs' standing state in wc.cop_frames / wc.com_ref (a reset puts it there)."""
        import ctypes as C
        from . import _lib
        if not hasattr(self, "pp"):
            raise _lib.TsidbError("this schedule was built on the host (WalkSchedule(...)): use WalkSchedule.on_device")
        ids, n_ids = None, 0
        if env_ids is not None:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device).contiguous()
            n_ids = ids.numel()
            if n_ids == 0:
                return
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        if path is not None:
            path = torch.as_tensor(path, dtype=torch.float64, device=self.device).contiguous()
            npts = torch.as_tensor(npts, dtype=torch.int32, device=self.device).contiguous()
            if path.dim() != 3 or path.shape[0] != self.N or path.shape[2] != 2 or path.shape[1] < 2 or npts.numel() != self.N:
                raise _lib.TsidbError("WalkSchedule.plan: path must be [N, P >= 2, 2] and npts [N] (rows are indexed by env id, "
                                      "also when only some envs are replanned)")
            self._path_keep = (path, npts)
        use_rng = path is None and self.scale is None
        with torch.cuda.device(wc.device):
            rc = wc._L.tsidb_walk_plan(wc._h, p(ids), n_ids, p(wc.rows) if done_only else None, wc.NROW,
                                       self.pp.ctypes.data_as(C.c_void_p), PLAN_NPARAMS, p(path), p(npts),
                                       int(path.shape[1]) if path is not None else 0, p(self.scale), p(self.episode) if use_rng else None,
                                       int(bool(new_paths)), self.K, p(self.steps), p(self.coef), p(self.side), p(self.nsteps),
                                       p(self.rest), p(self.com), p(self.flags), p(self.t_offset), p(self.td_latch), float(t),
                                       p(t_device), wc._stream())
        _lib.check(wc._L, wc._h, rc, "tsidb_walk_plan")

    def enable_touchdown_feedback(self, fraction=0.6):
        """Closed loop: take a step's touch-down as soon as the sim reports the swing foot on the floor after
        `fraction` of its swing (instead of at the scheduled time); device path (apply) only."""
        self.td_latch = torch.full((self.N,), -1, dtype=torch.int32, device=self.device)
        self.td_fraction = float(fraction)

    def set_phase_offsets(self, t_offset):
        """De-phase the batch: env e follows its timeline on the clock max(t - t_offset[e], 0), so that single-
        and double-support ticks (38- / 50-variable QPs) mix inside one launch.  Device path (apply) only."""
        self.t_offset = None if t_offset is None else torch.as_tensor(t_offset, device=self.device).to(self.dtype).contiguous()

    @classmethod
    def from_demo_paths(cls, num_envs, conf: RobotConfig, device, dtype, seed=1, q0_feet=None, com0=None, **kw):
        """Config-3 workload: the reference's demo unicycle path
        (Footstep_Planner.py:131-141) scaled per env by U(0.5, 1.0), turned into the robot's initial
        heading (the direction the left foot is to the left of) and started between its feet."""
        rng = np.random.default_rng(seed)
        planner = FootstepPlanner(conf.step_width, conf.step_length)
        lf = np.asarray(q0_feet[0] if q0_feet is not None else [0.0, 0.1], dtype=np.float64)
        rf = np.asarray(q0_feet[1] if q0_feet is not None else [0.0, -0.1], dtype=np.float64)
        left = lf - rf
        heading = float(np.arctan2(-left[0], left[1]))
        ch, sh = np.cos(heading), np.sin(heading)
        Rh = np.array([[ch, -sh], [sh, ch]])
        plans = []
        scales = rng.uniform(0.5, 1.0, size=num_envs)
        cache = {}
        for sc in scales:
            key = round(float(sc), 2)  # 51 distinct paths keep the host-side planning cheap
            if key not in cache:
                path = resample_path([Rh @ p + 0.5 * (lf + rf) for p in unicycle_path(scale=key)], conf.step_length / 10)
                init = [Footstep(lf, np.array([0.0, 0.0, heading]), 0), Footstep(rf, np.array([0.0, 0.0, heading]), 1)]
                cache[key] = planner.plan(path, init)
            plans.append(cache[key])
        if com0 is None:
            com0 = np.array([*(0.5 * (lf + rf)), 0.24])
        return cls(plans, conf, device, dtype, com0=com0, heading=heading, **kw)

    def _phase(self, t: float):
        """(step index k, time in the step) for t >= t_start; k = -1 before the first step."""
        T = self.conf.step_duration
        if t < self.t_start:
            return -1, t
        k = int(np.floor((t - self.t_start) / T))
        return k, (t - self.t_start) - k * T

    def sample(self, t: float):
        """Foot-task samples and contact flags at time t: (sampleLF [N,24], sampleRF [N,24],
        contact_LF [N] bool, contact_RF [N] bool).  Sample layout = tsid SE3 TrajectorySample:
        p(3), R column-major(9), v(6), a(6), twists world-aligned."""
        k, s = self._phase(t)
        N = self.N
        kk = torch.full((N,), max(k, 0), dtype=torch.long, device=self.device)
        walking = (kk < self.nsteps) & (k >= 0)
        kc = torch.minimum(kk, torch.clamp(self.nsteps - 1, min=0))
        ar = torch.arange(N, device=self.device)
        c = self.coef[ar, kc]                         # [N,4,4]
        pw = torch.tensor([1.0, s, s * s, s ** 3], dtype=self.dtype, device=self.device)
        d1 = torch.tensor([0.0, 1.0, 2 * s, 3 * s * s], dtype=self.dtype, device=self.device)
        d2 = torch.tensor([0.0, 0.0, 2.0, 6 * s], dtype=self.dtype, device=self.device)
        pos, vel, acc = c @ pw, c @ d1, c @ d2        # [N,4] x y z yaw
        sd = self.side[ar, kc]
        rest = self.rest[ar, torch.minimum(kk, self.nsteps)]  # [N,2,4]
        out = []
        for f in (0, 1):
            swing = walking & (sd == f)
            x = torch.where(swing, pos[:, 0], rest[:, f, 0])
            y = torch.where(swing, pos[:, 1], rest[:, f, 1])
            z = torch.where(swing, pos[:, 2], rest[:, f, 3])
            yaw = torch.where(swing, pos[:, 3], rest[:, f, 2])
            cy, sy = torch.cos(yaw), torch.sin(yaw)
            zero, one = torch.zeros_like(x), torch.ones_like(x)
            Rcm = torch.stack([cy, sy, zero, -sy, cy, zero, zero, zero, one], dim=-1)  # column-major Rz(yaw)
            m = swing.to(self.dtype)[:, None]
            v6 = torch.stack([vel[:, 0], vel[:, 1], vel[:, 2], zero, zero, vel[:, 3]], dim=-1) * m
            a6 = torch.stack([acc[:, 0], acc[:, 1], acc[:, 2], zero, zero, acc[:, 3]], dim=-1) * m
            out.append((torch.cat([torch.stack([x, y, z], dim=-1), Rcm, v6, a6], dim=-1), ~swing))
        return out[0][0], out[1][0], out[0][1], out[1][1]

    def com_ref(self, t: float):
        """CoM task reference [N, 9] = position, velocity, acceleration: LIPM segments in the plane,
        a quintic descent by com_drop during [0, t_start) in height."""
        k, s = self._phase(t)
        ph = torch.minimum(torch.full((self.N,), k + 1, dtype=torch.long, device=self.device), self.nsteps + 1)
        T = self.conf.step_duration
        # time inside the final stand keeps running from the end of the last step
        s_t = torch.where(ph > self.nsteps, (t - self.t_start) - self.nsteps.to(self.dtype) * T,
                          torch.full((self.N,), s, dtype=self.dtype, device=self.device)) if k >= 0 else \
            torch.full((self.N,), s, dtype=self.dtype, device=self.device)
        seg = self.com[torch.arange(self.N, device=self.device), ph]  # [N,2,3]
        w = self.omega
        ep, em = torch.exp(w * s_t)[:, None], torch.exp(-w * s_t)[:, None]
        u = 0.5 * seg[:, :, 1] * ep + seg[:, :, 2] * em
        pos = seg[:, :, 0] + u
        vel = w * (0.5 * seg[:, :, 1] * ep - seg[:, :, 2] * em)
        acc = w * w * u
        a = min(t / self.t_start, 1.0) if self.t_start > 0 else 1.0
        sz = a * a * a * (10 - 15 * a + 6 * a * a)
        dsz = 30 * a * a * (1 - a) ** 2 / self.t_start if self.t_start > 0 else 0.0
        ddsz = 60 * a * (1 - a) * (1 - 2 * a) / self.t_start ** 2 if self.t_start > 0 else 0.0
        out = torch.zeros(self.N, 9, dtype=self.dtype, device=self.device)
        out[:, 0:2], out[:, 3:5], out[:, 6:8] = pos, vel, acc
        out[:, 2], out[:, 5], out[:, 8] = self.z0 - self.dz * sz, -self.dz * dsz, -self.dz * ddsz
        return out

    def args(self, wc, t: float, t_device=None):
        """This tick's reference update as the argument block of tsidb_tick_walk (WalkController.tick(walk=...) /
        step_pipelined(walk=...)): the update then runs in the tick kernel's prologue instead of a launch of its own."""
        from . import _lib
        if not hasattr(self, "_side32"):   # (host-built schedule: int32 / contiguous copies for the kernel)
            self._side32 = self.side.to(torch.int32).contiguous()
            self._nsteps32 = self.nsteps.to(torch.int32).contiguous()
            self._coef_c, self._rest_c, self._com_c = self.coef.contiguous(), self.rest.contiguous(), self.com.contiguous()
        p = lambda x: x.data_ptr() if x is not None else None
        fb = self.td_latch is not None
        if fb:
            wc.sync_sim()  # the kernel reads the last sim step's contact list: nothing of it may still be in flight
        return _lib.WalkArgs(p(self._coef_c), p(self._side32), p(self._nsteps32), p(self._rest_c), p(self._com_c), self.K, float(t),
                             float(self.conf.step_duration), float(self.t_start), float(self.omega), float(self.z0), float(self.dz),
                             p(wc.frames), p(self.t_offset), p(wc.ncon) if fb else None, p(wc.con_pairs) if fb else None,
                             p(self.td_latch) if fb else None, float(self.td_fraction), p(t_device))

    def apply(self, wc, t: float, t_device=None):
        """Device path of one tick's reference update: foot samples, contact switching and the CoM
        reference for every env in one kernel (tsidb_walk_update); equivalent to
        wc.update_tasks(*self.sample(t)) followed by wc.com_ref[:] = self.com_ref(t).  t_device: a one-element
        float64 tensor holding the time - read by the kernel instead of `t` (graph capture)."""
        import ctypes as C
        from . import _lib
        if not hasattr(self, "_side32"):   # (host-built schedule: int32 / contiguous copies for the kernel)
            self._side32 = self.side.to(torch.int32).contiguous()
            self._nsteps32 = self.nsteps.to(torch.int32).contiguous()
            self._coef_c, self._rest_c, self._com_c = self.coef.contiguous(), self.rest.contiguous(), self.com.contiguous()
        p = lambda x: C.c_void_p(x.data_ptr())
        if self.td_latch is not None:
            wc.sync_sim()  # the kernel reads the last sim step's contact list: nothing of it may still be in flight
        with torch.cuda.device(wc.device):
            rc = wc._L.tsidb_walk_update(wc._h, p(self._coef_c), p(self._side32), p(self._nsteps32), p(self._rest_c),
                                         p(self._com_c), self.K, float(t), float(self.conf.step_duration),
                                         float(self.t_start), float(self.omega), float(self.z0), float(self.dz),
                                         p(wc.frames), p(self.t_offset) if self.t_offset is not None else None,
                                         p(wc.ncon) if self.td_latch is not None else None,
                                         p(wc.con_pairs) if self.td_latch is not None else None,
                                         p(self.td_latch) if self.td_latch is not None else None, float(self.td_fraction),
                                         p(t_device) if t_device is not None else None, wc._stream())
        _lib.check(wc._L, wc._h, rc, "tsidb_walk_update")
