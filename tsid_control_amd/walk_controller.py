"""WalkController - batched, MI355X-resident counterpart of the reference's ctrl/WalkController.py.

Same constructor argument (a RobotConfig), same attribute and method names
(ctrl/WalkController.py:11-295), but every quantity carries a leading env axis and lives in a
torch-ROCm tensor that the HIP kernels update in place through the C-ABI (include/tsidb.h).
`reset()` and `step()` are new: they encapsulate WalkController.py:22-26,72-79 + main.py:57-64 and
main.py:119-129,192-195 respectively (the reference writes that loop inline).

The task stack (what the reference builds with tsid calls at WalkController.py:54-187) is fixed
inside the kernels: 2x Contact6d (hard), 2x TaskSE3Equality, TaskComEquality, TaskJointPosture,
TaskActuationBounds, TaskJointBounds, SolverHQuadProgFast.  The `formulation`, `solver`, `robot`
objects of the reference have no counterpart - their work happens inside `step()`.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .conf import RobotConfig
from .model import ModelBlob
from .params import P_COUNT, pack_params

NQ, NV, NA, NOBS, NROW, MAXCON = 27, 26, 20, 65, 67, 32


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class TrajectorySample:
    """Minimal stand-in for tsid.TrajectorySample for SE3 tasks: pos [N,12] (p, R column-major),
    vel [N,6], acc [N,6] (WalkController.py:195-196 passes such samples to setReference)."""

    def __init__(self, pos, vel=None, acc=None):
        self.pos = pos
        self.vel = vel if vel is not None else torch.zeros(pos.shape[0], 6, dtype=pos.dtype, device=pos.device)
        self.acc = acc if acc is not None else torch.zeros(pos.shape[0], 6, dtype=pos.dtype, device=pos.device)


class WalkController:
    def __init__(self, conf: RobotConfig = None, num_envs: int = None, device=None):
        self.conf = conf = conf if conf is not None else RobotConfig()
        self.num_envs = N = int(num_envs if num_envs is not None else getattr(conf, "num_envs", 1))
        self.device = torch.device(device if device is not None else getattr(conf, "device", "cuda"))
        if self.device.type != "cuda":
            raise _lib.TsidbError("WalkController needs a ROCm device: the hot path has no CPU implementation")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        dt_name = getattr(conf, "dtype", "f64")
        self.dtype = {"f64": torch.float64, "f32": torch.float32}[dt_name]
        self.model = ModelBlob(getattr(conf, "model_blob", None))
        self.params = pack_params(conf, self.model.effort_limit, self.model.velocity_limit)
        # one library per robot: pick the build whose dimensions are the blob's (libtsidb.so = v1, libtsidb_v0.so = robot/v0)
        self._L = L = _lib.load_for(self.model["model_dims"])
        NJ_, NQ, NV, NA, _nb, self.has_sim = _lib.dims(L)
        NOBS, NROW = NQ + NV + 12, NQ + NV + 14
        self.NQ, self.NV, self.NA, self.NOBS, self.NROW = NQ, NV, NA, NOBS, NROW
        self._h = C.c_void_p()
        raw = self.model.raw
        rc = L.tsidb_create(raw, len(raw), self.params.ctypes.data_as(C.c_void_p), P_COUNT, N, self.device.index,
                            0 if dt_name == "f64" else 1, C.byref(self._h))
        _lib.check(L, self._h, rc, "tsidb_create")

        z = lambda *s, dt=self.dtype: torch.zeros(*s, dtype=dt, device=self.device)
        # TSID state (WalkController.py:23-24) and sim state (main.py:51,64)
        self.q, self.v = z(N, NQ), z(N, NV)
        self.qpos, self.qvel, self.qacc_warmstart = z(N, NQ), z(N, NV), z(N, NV)
        # task references
        self.com_ref, self.posture_ref = z(N, 9), z(N, NA)
        self.foot_ref, self.contact_ref, self.cop_frames = z(N, 2, 24), z(N, 2, 12), z(N, 2, 12)
        self.contact_active = torch.ones(N, 2, dtype=torch.uint8, device=self.device)
        # outputs
        self.tau, self.dv, self.f = z(N, NA), z(N, NV), z(N, 24)
        self.status = z(N, dt=torch.int32)
        # one contiguous row per env = obs[65] + reward + done: what the all-gather sends
        self.rows = z(N, NROW)
        self.obs, self.reward, self.done = self.rows[:, :NOBS], self.rows[:, NOBS], self.rows[:, NOBS + 1]
        self.gather_width = NROW
        self.frames = z(N, 2, 12)
        self.ncon, self.con_pairs = z(N, dt=torch.int32), z(N, MAXCON, dt=torch.int32)
        self.info = z(N, 4, dt=torch.int32)
        self.env_params = None
        self.terrain = None
        rc = L.tsidb_set_refs(self._h, _ptr(self.com_ref), _ptr(self.posture_ref), _ptr(self.foot_ref),
                              _ptr(self.contact_ref), _ptr(self.contact_active), _ptr(self.cop_frames))
        _lib.check(L, self._h, rc, "tsidb_set_refs")
        sw = int(getattr(conf, "sim_waves", 0))   # 0 = the library's choice (2 wavefronts per env up to 512 envs, else 1)
        if sw:
            _lib.check(L, self._h, L.tsidb_set_option(self._h, 1, sw), "tsidb_set_option(sim_waves)")
        sp = int(getattr(conf, "sim_pack", -1))   # -1 = the library's choice; 1 = two envs per wavefront in the sim kernel (tsidb_sim2.hpp)
        if os.environ.get("TSIDB_SIM_PACK"):      # diagnostic override (A/B runs of bench.py)
            sp = int(os.environ["TSIDB_SIM_PACK"])
        if sp >= 0:
            _lib.check(L, self._h, L.tsidb_set_option(self._h, 4, sp), "tsidb_set_option(sim_pack)")
        fe = int(getattr(conf, "qp_fast_equalities", -1))   # -1 = the library's default (on)
        if os.environ.get("TSIDB_QP_FAST_EQ"):    # diagnostic override (A/B runs)
            fe = int(os.environ["TSIDB_QP_FAST_EQ"])
        if fe >= 0:
            _lib.check(L, self._h, L.tsidb_set_option(self._h, 5, fe), "tsidb_set_option(qp_fast_eq)")
        if os.environ.get("TSIDB_LDS_PAD"):       # diagnostic (occupancy measurements): unused LDS per workgroup
            _lib.check(L, self._h, L.tsidb_set_option(self._h, 2, int(os.environ["TSIDB_LDS_PAD"])), "tsidb_set_option(lds_pad)")
        self.cop_ref = z(N, 3)   # reference of the CoP force task (legacy/biped.py:79-80; conf.w_cop)
        _lib.check(L, self._h, L.tsidb_set_cop_ref(self._h, _ptr(self.cop_ref)), "tsidb_set_cop_ref")

        # WalkController.py:168-169,179-180
        self.tau_max = conf.tau_max_scaling * self.model.effort_limit
        self.tau_min = -self.tau_max
        self.v_max = conf.v_max_scaling * self.model.velocity_limit
        self.v_min = -self.v_max
        self.LF_frame, self.RF_frame = 0, 1
        self.posture_bias = None
        self.t = 0.0
        b = int(getattr(conf, "pipeline_sim_batch", 0))
        # 0 = auto: small batches are latency bound, there the barrier packets of the per-step cross-stream handshake are
        # 15-20 % of a step (measured: 512 envs +7 %, 1024 +5 %, 2048 and up nothing / noise)
        self.sim_batch = min(8, b) if b > 0 else (8 if self.num_envs <= 1024 else 1)   # step_pipelined(): sim stages enqueued this many at a time (one launch: tsidb_sim_batch)
        self.reset()
        self.q0 = self.q.clone()  # WalkController.py:23 (after the z shift of :74, which aliases q0)

    def __del__(self):
        try:
            if self._h:
                import sys
                self._pipe = None    # (its reference to the sim stream's wrapper is ours, not a caller's)
                for ext, hs in getattr(self, "_streams", {}).values():
                    # a caller (or a captured graph's keep list) may still hold the ExternalStream wrapper of a library
                    # stream: then the HIP stream is left alive (a leaked stream is harmless, a dangling one is not)
                    if hs is not None and sys.getrefcount(ext) <= 3:
                        self._L.tsidb_stream_destroy(self._h, hs)
                self._streams = {}
                self._L.tsidb_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def contactLF_active(self):
        return self.contact_active[:, 0].bool()

    @property
    def contactRF_active(self):
        return self.contact_active[:, 1].bool()

    def set_params(self):
        """Re-read self.conf (edited RobotConfig values) into the device-side constants."""
        self.sync_sim()  # sim stages step_pipelined() has not launched yet belong to the OLD constants: run them first
        self.params = pack_params(self.conf, self.model.effort_limit, self.model.velocity_limit)
        rc = self._L.tsidb_set_params(self._h, self.params.ctypes.data_as(C.c_void_p), P_COUNT)
        _lib.check(self._L, self._h, rc, "tsidb_set_params")

    def set_env_params(self, mass_scale=None, friction=None, floor_normal=None, floor_offset=None, terrain=None):
        """Per-env randomisation of the sim stage (BASELINE config 5; no reference counterpart): any
        of mass_scale [N], friction [N] (floor contacts), floor_normal [N,3] (normalised here), floor_offset [N],
        terrain = dict(direction [N,2], phase [N], step_length [N] or float, heights [N,16]): a stepped floor -
        the surface is raised along its normal by heights[cell & 15], cell = floor((direction . x_world_xy - phase)
        / step_length).  Calling with no argument restores the nominal model."""
        self.sync_sim()  # a sim stage left in flight by step_pipelined() may still read the old table
        N = self.num_envs
        if mass_scale is None and friction is None and floor_normal is None and floor_offset is None:
            self.env_params = None
        else:
            ep = torch.zeros(N, 8, dtype=self.dtype, device=self.device)
            ep[:, 0], ep[:, 1], ep[:, 4] = 1.0, 1.0, 1.0
            if mass_scale is not None:
                ep[:, 0] = torch.as_tensor(mass_scale, dtype=self.dtype, device=self.device)
            if friction is not None:
                ep[:, 1] = torch.as_tensor(friction, dtype=self.dtype, device=self.device)
            if floor_normal is not None:
                nrm = torch.as_tensor(floor_normal, dtype=self.dtype, device=self.device).reshape(N, 3)
                ep[:, 2:5] = nrm / nrm.norm(dim=1, keepdim=True)
            if floor_offset is not None:
                ep[:, 5] = torch.as_tensor(floor_offset, dtype=self.dtype, device=self.device)
            self.env_params = ep
        if terrain is None:
            self.terrain = None
        else:
            tr = torch.zeros(N, 20, dtype=torch.float64)
            d = torch.as_tensor(terrain["direction"], dtype=torch.float64).reshape(N, 2)
            tr[:, 0:2] = d / d.norm(dim=1, keepdim=True)
            tr[:, 2] = torch.as_tensor(terrain.get("phase", 0.0), dtype=torch.float64)
            tr[:, 3] = 1.0 / torch.as_tensor(terrain["step_length"], dtype=torch.float64)
            tr[:, 4:] = torch.as_tensor(terrain["heights"], dtype=torch.float64).reshape(N, 16)
            self.terrain = tr.to(self.device, self.dtype).contiguous()
        rc = self._L.tsidb_set_env_params(self._h, _ptr(self.env_params), _ptr(self.terrain))
        _lib.check(self._L, self._h, rc, "tsidb_set_env_params")

    def randomize(self, seed=2, mass=(0.8, 1.2), friction=(0.4, 1.0), tilt_deg=5.0, step_height=0.01, step_length=(0.04, 0.12)):
        """BASELINE config 5 workload: body-mass scale U(mass), contact friction
        U(friction), floor = random plane through the origin tilted by at most tilt_deg, with terrain steps of
        step_height (1 cm): strips of width U(step_length) across a random horizontal direction, each strip raised
        by 0 or step_height at random (period 16 strips; the strip under the robot's start is level).
        step_height = 0 leaves the floor a plane."""
        g = torch.Generator().manual_seed(seed)
        N = self.num_envs
        u = lambda lo, hi: lo + (hi - lo) * torch.rand(N, generator=g, dtype=torch.float64)
        tilt = torch.deg2rad(u(0.0, tilt_deg))
        az = u(0.0, 2 * np.pi)
        nrm = torch.stack([torch.sin(tilt) * torch.cos(az), torch.sin(tilt) * torch.sin(az), torch.cos(tilt)], dim=1)
        terrain = None
        if step_height > 0:
            ang = u(0.0, 2 * np.pi)
            length = u(*step_length)
            heights = step_height * torch.randint(0, 2, (N, 16), generator=g).to(torch.float64)
            heights[:, 0] = 0.0
            heights[:, 15] = 0.0
            # cell 0 is centred on the world origin (where every env's robot starts): phase = -length / 2
            terrain = dict(direction=torch.stack([torch.cos(ang), torch.sin(ang)], dim=1), phase=-0.5 * length,
                           step_length=length, heights=heights)
        self.set_env_params(mass_scale=u(*mass), friction=u(*friction), floor_normal=nrm,
                            floor_offset=torch.zeros(N, dtype=torch.float64), terrain=terrain)

    # ------------------------------------------------------------------ reset / step
    def set_posture_bias(self, bias):
        """[NA] offsets every reset adds to the posture reference it captures (ctrl/WalkController.py:164-165 takes q0's
        joints; a walking workload keeps its knees bent: walk_planner.op3_walking_posture()).  Also applied to the
        current posture references; None removes it."""
        old = self.posture_bias
        new = None if bias is None else torch.as_tensor(np.asarray(bias), device=self.device).to(self.dtype).contiguous()
        if old is not None:
            self.posture_ref -= old
        if new is not None:
            self.posture_ref += new
        self.posture_bias = new
        _lib.check(self._L, self._h, self._L.tsidb_set_posture_bias(self._h, _ptr(new)), "tsidb_set_posture_bias")

    def reset_done(self, sched=None, t=None, new_paths=True):
        """Episode lifecycle on the device: reset every env whose done flag (self.done, written by the last tick) is set -
        standing state, references, sim state - and, with a WalkSchedule.on_device schedule, rebuild its plan (a new path
        when new_paths) and restart its clock at time t (default self.t, the time of the next tick).  Nothing comes back
        to the host; envs that are not done are untouched."""
        self.sync_sim()
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_reset_done(self._h, _ptr(self.rows), self.NROW, _ptr(self.q), _ptr(self.v), _ptr(self.qpos),
                                          _ptr(self.qvel), _ptr(self.qacc_warmstart), _ptr(self.frames), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_reset_done")
        if sched is not None:
            sched.plan(self, t=self.t if t is None else t, done_only=True, new_paths=new_paths)

    def reset(self, env_ids=None, sched=None, t=None, new_paths=False):
        """Standing state with the soles on z = 0 and all references re-captured
        (WalkController.py:22-26,72-79,81,122,151-152,164-165; main.py:57-64).  sched (a WalkSchedule.on_device
        schedule): the reset envs' plans are rebuilt on the device (new paths when new_paths) and their clocks restart at
        time t (default: self.t after the reset, i.e. 0 for a full reset)."""
        self.sync_sim()  # step_pipelined() may have left a sim stage running on the side stream
        ids = None
        n_ids = 0
        if env_ids is not None:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device).contiguous()
            n_ids = ids.numel()
            if n_ids == 0:
                return  # nothing to reset (the C entry point reads a NULL id list as "every env")
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_reset(self._h, _ptr(ids), n_ids, _ptr(self.q), _ptr(self.v), _ptr(self.qpos),
                                     _ptr(self.qvel), _ptr(self.qacc_warmstart), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_reset")
        if env_ids is None:
            self.frames.copy_(self.cop_frames)
            self.t = 0.0
        else:
            self.frames[ids.long()] = self.cop_frames[ids.long()]
        if sched is not None:
            sched.plan(self, env_ids=ids, t=self.t if t is None else t, new_paths=new_paths)

    def step(self, n_substeps: int = 1):
        """One env step for every env: TSID tick (main.py:119-129) then, if conf.sim_enabled, base
        teleport + joint targets + sim step (main.py:192-195).  Returns (tau, q, v, status, obs);
        all are views of the controller's tensors, updated in place."""
        self.sync_sim()
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_step(self._h, _ptr(self.q), _ptr(self.v), _ptr(self.qpos), _ptr(self.qvel),
                                    _ptr(self.qacc_warmstart), _ptr(self.tau), _ptr(self.dv), _ptr(self.f),
                                    _ptr(self.status), _ptr(self.rows), self.NROW, _ptr(self.frames), _ptr(self.ncon),
                                    _ptr(self.con_pairs), _ptr(self.info), int(n_substeps), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_step")
        self.t += n_substeps * self.conf.dt
        return self.tau, self.q, self.v, self.status, self.obs

    def step_pipelined(self, events=None, walk=None):
        """One env step with the sim stage left running on a second HIP stream, so that it overlaps with
        what the caller enqueues next on the current stream - normally the TSID tick of the NEXT step.  The reference
        couples the two stages one way (the sim never feeds back into TSID, main.py:119-129 vs :192-195), so sim(t) and
        tick(t+1) are independent; the TSID state is handed to the sim through a ring of snapshot slots the tick writes.
        walk = (schedule, t) runs that tick's WalkSchedule.apply inside the tick's launch (tsidb_tick_walk).
        tau, q, v, status, obs are valid on the current stream as after step(); the sim state (qpos, qvel,
        qacc_warmstart, ncon, con_pairs, info[:, 2:4]) is valid after sync_sim() ONLY: with conf.pipeline_sim_batch > 1
        (the default for up to 1024 envs is 8) the last few sim stages are not even launched until the batch is full, so a
        device / stream synchronize does not make the sim state current - sync_sim() launches them and makes the current
        stream wait.  Every entry point of this class that reads or rewrites sim-side data (step, sim_step, reset,
        reset_done, set_params, set_env_params, capture_steps, WalkSchedule.apply with touch-down feedback) calls it.
        `events` = four torch.cuda.Event recorded around the tick (current stream) and around the sim (sim stream), for
        timing."""
        if getattr(self.conf, "closed_loop", False) or not getattr(self.conf, "sim_enabled", True):
            raise _lib.TsidbError("step_pipelined needs the open-loop sim stage (closed loop: the tick reads the sim state)")
        cur = torch.cuda.current_stream(self.device)
        self._ensure_pipe()
        P = self._pipe
        par = P["par"]
        P["par"] = (par + 1) % len(P["q"])
        ev = P["done"][par]                    # the sim batch that read this slot 2 * sim_batch steps ago
        lw = P.get("last_wait")
        if ev is not None and not (lw is not None and lw[0] == cur.cuda_stream and lw[1] is ev):
            # (once per batch: the slots of one batch share its event, and a cross-stream wait is a barrier packet that costs
            #  the tick stream ~10 us each - at 512 envs a fifth of the step when it was issued before every tick)
            if torch.cuda.is_current_stream_capturing() or not ev.query():   # (already complete: no packet)
                cur.wait_event(ev)
            P["last_wait"] = (cur.cuda_stream, ev)   # (the reference keeps the event alive: no id reuse)
        if events:
            events[0].record(cur)
        # the tick writes the TSID state it ends on into the slot as well (two copy kernels less on this stream); walk =
        # (schedule, t): the walking reference update of this tick in the same launch
        self.tick(walk=walk, _snap=(P["q"][par], P["v"][par]))
        if events:
            events[1].record(cur)
        P["pending"].append(par)
        # conf.pipeline_sim_batch > 1 enqueues the sim stages that many at a time, as one launch (one cross-stream wait and one
        # record per batch instead of per step, no launch gaps; the sim state then lags the tick by up to that many steps
        # until sync_sim()).  Measured (DESIGN.md section 5 "Streams"): no gain from 2048 envs on; 512 / 1024 walkers +25-30 %
        # together with the fused tick launch (8 at a time, the default for up to 1024 envs).
        if len(P["pending"]) >= self.sim_batch or events:
            self._flush_sims(events)
        self.t += self.conf.dt
        return self.tau, self.q, self.v, self.status, self.obs

    def _streams_overlap(self, sa, sb):
        """True if work on the two streams really runs concurrently.  HIP multiplexes its streams onto a few hardware queues
        (GPU_MAX_HW_QUEUES, 4 by default) and two streams - even two created one after the other - can share one, in which
        case tick and sim run one after the other and the pipelined step loses its overlap without any error
        (tools/stream_overlap_probe.py).  Probe: a short device-side spin on each, timed together against one alone."""
        if os.environ.get("TSIDB_NO_STREAM_PROBE") == "1" or not hasattr(torch.cuda, "_sleep") or torch.cuda.is_current_stream_capturing():
            return True
        import time
        try:
            def spin(streams, cycles=600000):   # ~0.25 ms at 2.4 GHz
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for st in streams:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(cycles)
                for st in streams:
                    st.synchronize()
                return time.perf_counter() - t0
            spin([sa, sb], 1000)                 # (first use of the kernel on these streams)
            one = min(spin([sa]), spin([sb]))
            both = min(spin([sa, sb]), spin([sa, sb]))
            return both < 1.6 * one
        except Exception:
            return True

    def _lib_stream(self, role):
        """tsidb_stream_create: the stream the library recommends for the tick (role 0) / the sim (role 1) of the pipelined
        step - on disjoint halves of the CUs for up to 512 envs (include/tsidb.h), plain streams above"""
        st = getattr(self, "_streams", None)
        if st is None:
            st = self._streams = {}
        if role not in st:
            split = C.c_int(0)
            _lib.check(self._L, self._h, self._L.tsidb_get_option(self._h, 3, C.byref(split)), "tsidb_get_option(cu_split)")
            for r in (0, 1):   # (both at once)
                if split.value:
                    hs = C.c_void_p()
                    _lib.check(self._L, self._h, self._L.tsidb_stream_create(self._h, r, C.byref(hs)), "tsidb_stream_create")
                    st[r] = (torch.cuda.ExternalStream(hs.value, device=self.device), hs)
                else:
                    st[r] = (torch.cuda.Stream(device=self.device), None)   # no CU split for this batch size: torch's pool
            ok = False
            for _ in range(6):                   # a pair that shares a hardware queue would serialise tick and sim
                if self._streams_overlap(st[0][0], st[1][0]):
                    ok = True
                    break
                if st[0][1] is not None or st[1][1] is not None:
                    # the CU-masked pair does not overlap: give up the split for BOTH roles (a tick confined to half the
                    # CUs beside a sim that spans all of them is a silent regression) and say so
                    import warnings
                    warnings.warn("tsid_control_amd: the CU-masked tick / sim streams do not run concurrently on this device; "
                                  "using ordinary streams for both (no CU split)")
                    for r in (0, 1):
                        if st[r][1] is not None:
                            self._L.tsidb_stream_destroy(self._h, st[r][1])
                        st[r] = (torch.cuda.Stream(device=self.device), None)
                else:
                    st[1] = (torch.cuda.Stream(device=self.device), None)
            if not ok and not self._streams_overlap(st[0][0], st[1][0]):
                import warnings
                warnings.warn("tsid_control_amd: no pair of HIP streams that runs concurrently was found (they share a hardware "
                              "queue): the pipelined step will run tick and sim one after the other")
        return st[role][0]

    @property
    def tick_stream(self):
        """The stream to run the pipelined loop on: `with torch.cuda.stream(wc.tick_stream): wc.step_pipelined(...)`.  For up
        to 512 envs it is restricted to one half of the CUs and the sim stream to the other (the two kernels slow each other
        down by a quarter when they share CU groups: +11-14 % env-steps/s at 256 / 512 envs); otherwise an ordinary stream.
        Optional - step_pipelined() works on any current stream; the sim stream is paired with this one only if the FIRST
        step_pipelined() runs on it."""
        return self._lib_stream(0)

    def _ensure_pipe(self):
        """the second stream and the ring of snapshot slots step_pipelined() hands the TSID state to the sim stages through;
        (re)built when missing or too small for the current sim batch"""
        need = min(16, max(4, 2 * self.sim_batch, int(os.environ.get("TSIDB_RING_SLOTS", "0"))))
        P = getattr(self, "_pipe", None)
        if P is not None and len(P["q"]) >= need:
            return
        if P is not None:
            self.sync_sim()                     # nothing may be pending in the ring that is replaced
        # ring of snapshot slots: the tick writes its slot itself, so it must wait for the sim that read the slot
        # K steps ago BEFORE it starts - with only two slots that wait held tick(t) back until sim(t - 2) was done
        # and cost 12 % at 4096 envs; four slots and the tick stream runs ahead as before
        # (never fewer than two batches of slots: a tick must not overwrite a snapshot whose sim is still pending; the
        #  library numbers slots 0 .. 15)
        K = need
        qring = torch.empty(K, *self.q.shape, dtype=self.dtype, device=self.device)   # one allocation: a batch of sim
        vring = torch.empty(K, *self.v.shape, dtype=self.dtype, device=self.device)   # stages names its slots by number
        # the sim stream: the library's (on the other half of the CUs for up to 512 envs) when the loop runs on
        # self.tick_stream, an ordinary one otherwise - a CU-masked stream is a BLOCKING stream (hipExtStreamCreateWithCUMask
        # takes no flags), and beside work on the legacy default stream it would serialise with it
        ts = getattr(self, "_streams", {}).get(0)
        on_tick = ts is not None and torch.cuda.current_stream(self.device).cuda_stream == ts[0].cuda_stream
        if on_tick:
            sim_stream = self._lib_stream(1)
        else:
            cur = torch.cuda.current_stream(self.device)
            for _ in range(6):                   # (a stream that shares the current one's hardware queue would serialise)
                sim_stream = torch.cuda.Stream(device=self.device)
                if self._streams_overlap(cur, sim_stream):
                    break
        self._pipe = dict(stream=sim_stream,
                          par=0, done=[None] * K, pending=[], qring=qring, vring=vring,
                          q=[qring[k] for k in range(K)], v=[vring[k] for k in range(K)])

    def gather_rows(self, out=None):
        """[N, 67] = obs, reward, done of the last tick: the per-env row the multi-GPU all-gather carries."""
        if out is None:
            return self.rows
        out.copy_(self.rows)
        return out

    def capture_steps(self, n_steps: int, sched=None, sim_batch: int = None):
        """Capture n_steps pipelined env steps (walking reference update, TSID tick, sim step on the second stream)
        in ONE HIP graph and return it; graph.replay() then enqueues all of them with a single launch instead of
        ~8 host calls per step - what bounds small batches (512-1024 envs per GPU: the strong split of 4096 walkers
        over 4-8 GPUs).  The schedule's clock lives on the device (self.t_device, advanced inside the graph); results
        are bit-identical to the same number of step_pipelined() calls with the same number of sim steps per launch (float32:
        a different number of sim steps per launch agrees to rounding only, include/tsidb.h tsidb_sim_batch).  The sim state
        is valid after sync_sim().
        sim_batch = sim steps per launch INSIDE the graph (default: as in eager mode).  A graph ends with a join, so the
        sim batch still to run after the last tick runs alone; measured (tools/r03_graph.sh, 512 envs): eager 6.64 M
        env-steps/s; 16 steps per graph 5.2 / 5.5 / 5.2 M with 1 / 2 / 8 sim steps per launch, 64 steps per graph 5.8 / 6.0 /
        6.4 M - replaying a graph never beats the eager pipeline here."""
        if getattr(self.conf, "closed_loop", False) or not getattr(self.conf, "sim_enabled", True):
            raise _lib.TsidbError("capture_steps uses the open-loop pipeline (step_pipelined)")
        dt = self.conf.dt
        self.sync_sim()   # the state saved below must include the sim stage a previous step_pipelined() left in flight
        self._ensure_pipe()   # (the ring sized for the eager batch: it is not rebuilt, and the graph's pointers stay valid, afterwards)
        batch_keep = self.sim_batch
        if sim_batch is not None:
            self.sim_batch = max(1, min(int(sim_batch), len(self._pipe["q"]) // 2))
        self.t_device = torch.full((1,), self.t, dtype=torch.float64, device=self.device)   # float64 whatever the path's dtype
        # warm up outside the capture (lazy kernel loads, cached contiguous tables), then rewind the state
        keep = {k: getattr(self, k).clone() for k in ("q", "v", "qpos", "qvel", "qacc_warmstart", "com_ref", "posture_ref",
                                                      "foot_ref", "contact_ref", "contact_active", "frames", "rows", "tau", "dv", "f",
                                                      "status", "ncon", "con_pairs", "info", "cop_ref")}
        latch_keep = sched.td_latch.clone() if sched is not None and sched.td_latch is not None else None
        t_keep = self.t
        if sched is not None:
            sched.apply(self, self.t, t_device=self.t_device)
        self.step_pipelined()
        self.t_device += dt
        self.sync_sim()
        torch.cuda.synchronize(self.device)
        for k, v in keep.items():
            getattr(self, k).copy_(v)
        if latch_keep is not None:
            sched.td_latch.copy_(latch_keep)
        self.t = t_keep
        self.t_device.fill_(self.t)
        self._pipe["done"] = [None] * len(self._pipe["q"])   # no event from outside the capture may be waited on inside it
        self._pipe["par"] = 0
        self._pipe["last_wait"] = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n_steps):
                self.step_pipelined(walk=(sched, 0.0, self.t_device) if sched is not None else None)
                self.t_device += dt
            self.sync_sim()                 # join the sim stream: the graph ends with every kernel done
        self._pipe["done"] = [None] * len(self._pipe["q"])
        self._pipe["last_wait"] = None
        self.sim_batch = batch_keep
        self.t = t_keep                     # the capture advanced the host clock without running anything

        outer = self

        class _Graph:
            steps = n_steps
            keep = (self._pipe["qring"], self._pipe["vring"], self._pipe["stream"])   # what the captured kernels point at

            def replay(self_inner):
                g.replay()
                for _ in range(n_steps):   # the same additions the device clock and step_pipelined() make
                    outer.t += dt

        return _Graph()

    def _flush_sims(self, events=None):
        """enqueue the sim stages of the ticks step_pipelined() has run since the last flush (oldest first)"""
        P = self._pipe
        if not P["pending"]:
            return
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(P["stream"]):
            P["stream"].wait_event(ready)
            pend = list(P["pending"])
            if events:                       # timing events bracket the last sim step alone
                head, pend = pend[:-1], pend[-1:]
                if head:
                    self._sim_batch(head)
                events[2].record(P["stream"])
            self._sim_batch(pend)
            if events:
                events[3].record(P["stream"])
            done = torch.cuda.Event()
            done.record(P["stream"])
        for slot in P["pending"]:
            P["done"][slot] = done
        P["pending"] = []

    def _sim_batch(self, slots):
        """the sim stages of several ticks in ONE launch (tsidb_sim_batch): each env steps len(slots) times, teleporting to
        the snapshot of its tick each time - no launch gaps between the steps"""
        B = len(slots)
        if B == 1:
            self.sim_step(q_tsid=self._pipe["q"][slots[0]], v_tsid=self._pipe["v"][slots[0]], _from_pipe=True)
            return
        sl = (C.c_int32 * B)(*slots)
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_sim_batch(self._h, B, _ptr(self._pipe["qring"]), _ptr(self._pipe["vring"]), sl, _ptr(self.qpos),
                                         _ptr(self.qvel), _ptr(self.qacc_warmstart), None, _ptr(self.ncon), _ptr(self.con_pairs),
                                         _ptr(self.info), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_sim_batch")

    def sync_sim(self):
        """Make the current stream wait for the sim stages step_pipelined() left in flight (or not yet enqueued)."""
        P = getattr(self, "_pipe", None)
        if P is not None:
            self._flush_sims()
            torch.cuda.current_stream(self.device).wait_stream(P["stream"])

    def tick(self, walk=None, _snap=None):
        """TSID stage only (main.py:119-129).  walk = (schedule, t): that tick's walking reference update
        (WalkSchedule.apply(self, t)) runs in the same launch, ahead of the tick (tsidb_tick_walk) - same results."""
        if walk is None and _snap is None:
            with torch.cuda.device(self.device):
                rc = self._L.tsidb_tick(self._h, _ptr(self.q), _ptr(self.v), _ptr(self.tau), _ptr(self.dv), _ptr(self.f),
                                        _ptr(self.status), _ptr(self.rows), self.NROW, _ptr(self.frames), _ptr(self.info), self._stream())
            _lib.check(self._L, self._h, rc, "tsidb_tick")
            return self.tau, self.q, self.v, self.status, self.obs
        wa = walk[0].args(self, walk[1], *walk[2:]) if walk is not None else None
        qs, vs = _snap if _snap is not None else (None, None)
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_tick_walk(self._h, C.byref(wa) if wa is not None else None, _ptr(self.q), _ptr(self.v), _ptr(self.tau),
                                         _ptr(self.dv), _ptr(self.f), _ptr(self.status), _ptr(self.rows), self.NROW, _ptr(self.frames),
                                         _ptr(self.info), _ptr(qs), _ptr(vs), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_tick_walk")
        return self.tau, self.q, self.v, self.status, self.obs

    def sim_step(self, teleport=True, q_tsid=None, v_tsid=None, _from_pipe=False):
        """Sim stage only (main.py:192-195); teleport=False steps the sim state on its own; q_tsid / v_tsid
        override the TSID state the base pose / joint targets (and, with reference_quirks=False, the base
        velocity) are taken from (snapshots of self.q / self.v when the sim stage runs on another stream
        than the tick)."""
        if not _from_pipe:
            self.sync_sim()
        src = q_tsid if q_tsid is not None else self.q
        srcv = v_tsid if v_tsid is not None else self.v
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_sim(self._h, _ptr(src) if teleport else None, _ptr(srcv) if teleport else None,
                                   _ptr(self.qpos), _ptr(self.qvel),
                                   _ptr(self.qacc_warmstart), None, _ptr(self.ncon), _ptr(self.con_pairs),
                                   _ptr(self.info), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_sim")
        return self.qpos, self.qvel

    def rbd_terms(self, q=None, v=None):
        """Rigid-body terms of computeProblemData (main.py:119) for inspection/tests."""
        q = self.q if q is None else q
        v = self.v if v is None else v
        N = self.num_envs
        z = lambda *s: torch.zeros(*s, dtype=self.dtype, device=self.device)
        NV = self.NV
        out = dict(M=z(N, NV, NV), h=z(N, NV), Jcom=z(N, 3, NV), Jf=z(N, 2, 6, NV), oMf=z(N, 2, 12), com=z(N, 3))
        with torch.cuda.device(self.device):
            rc = self._L.tsidb_rbd_terms(self._h, _ptr(q), _ptr(v), _ptr(out["M"]), _ptr(out["h"]), _ptr(out["Jcom"]),
                                         _ptr(out["Jf"]), _ptr(out["oMf"]), _ptr(out["com"]), self._stream())
        _lib.check(self._L, self._h, rc, "tsidb_rbd_terms")
        return out

    # ------------------------------------------------------------------ reference method surface
    @staticmethod
    def _frames_to_se3vec(fr):
        """[.., 12] R row-major + p  ->  [.., 12] p + R column-major (tsid SE3ToVector)."""
        R = fr[..., :9].reshape(*fr.shape[:-1], 3, 3)
        return torch.cat([fr[..., 9:], R.transpose(-1, -2).reshape(*fr.shape[:-1], 9)], dim=-1)

    def _mask(self, flag):
        if isinstance(flag, torch.Tensor):
            return flag.to(self.device).bool().reshape(-1)
        return torch.full((self.num_envs,), bool(flag), dtype=torch.bool, device=self.device)

    def _sample24(self, s):
        if isinstance(s, torch.Tensor):
            return s.to(self.device, self.dtype).reshape(self.num_envs, 24)
        return torch.cat([s.pos, s.vel, s.acc], dim=-1).to(self.device, self.dtype).reshape(self.num_envs, 24)

    def update_tasks(self, sampleLF, sampleRF, contact_LF, contact_RF):
        """WalkController.py:189-209, batched: set both foot-task references, then switch contacts
        on the edges of the (per-env) contact flags."""
        self.foot_ref[:, 0] = self._sample24(sampleLF)
        self.foot_ref[:, 1] = self._sample24(sampleRF)
        cLF, cRF = self._mask(contact_LF), self._mask(contact_RF)
        aLF, aRF = self.contactLF_active, self.contactRF_active
        self.add_contact(left_foot=cLF & ~aLF, right_foot=cRF & ~aRF)
        self.remove_contact(left_foot=~cLF & aLF, right_foot=~cRF & aRF)

    def display(self, q):
        """WalkController.py:211-213: viewer hook; no viewer exists here."""
        return None

    def remove_contact(self, left_foot=True, right_foot=True):
        """WalkController.py:215-232 (as legacy/biped.py:168-189 makes it work): re-reference the foot
        task at the current placement, drop the rigid contact."""
        cur = self._frames_to_se3vec(self.frames)
        for f, flag in ((0, left_foot), (1, right_foot)):
            m = self._mask(flag) & self.contact_active[:, f].bool()
            ref = torch.cat([cur[:, f], torch.zeros(self.num_envs, 12, dtype=self.dtype, device=self.device)], dim=-1)
            self.foot_ref[:, f] = torch.where(m[:, None], ref, self.foot_ref[:, f])
            self.contact_active[:, f] = torch.where(m, torch.zeros_like(self.contact_active[:, f]), self.contact_active[:, f])

    def add_contact(self, left_foot=True, right_foot=True):
        """WalkController.py:234-253: re-reference the contact at the current placement, add it back."""
        cur = self._frames_to_se3vec(self.frames)
        for f, flag in ((0, left_foot), (1, right_foot)):
            m = self._mask(flag) & ~self.contact_active[:, f].bool()
            self.contact_ref[:, f] = torch.where(m[:, None], cur[:, f], self.contact_ref[:, f])
            self.contact_active[:, f] = torch.where(m, torch.ones_like(self.contact_active[:, f]), self.contact_active[:, f])

    def get_cop(self, sol=None):
        """WalkController.py:255-289: centre of pressure of the last tick's contact forces, [N,3];
        rows are NaN where the reference would return None (not both feet in contact)."""
        o = self.NQ + self.NV
        cop = self.obs[:, o + 3:o + 6].clone()
        both = self.contactLF_active & self.contactRF_active
        cop[~both] = float("nan")
        return cop

    def compute_capture_point(self, com=None, dcom=None, w=None):
        """legacy/biped.py:224-227, batched: cp = com + dcom / w with cp_z = 0; defaults to the last tick's
        CoM / CoM velocity (obs) and the LIPM frequency of the current CoM height."""
        com = self.obs[:, self.NQ + self.NV:self.NQ + self.NV + 3] if com is None else com
        if dcom is None:
            raise ValueError("dcom (CoM velocity [N,3]) is required: the observation vector carries positions only")
        if w is None:
            w = torch.sqrt(9.80665 / com[:, 2:3])
        cp = com + dcom / w
        cp[:, 2] = 0
        return cp

    def compute_support_polygon(self):
        """legacy/biped.py:229-234, batched: the two sole positions in the plane, [N, 2 (LF, RF), 2]."""
        return self.frames[:, :, 9:11].clone()

    def integrate_dv(self, q, v, dv, dt):
        """WalkController.py:291-295 for caller-held tensors: v updated in place, new q returned.
        (step() integrates inside the kernel; this exists for callers that drive the pieces.)"""
        v_mean = v + 0.5 * dt * dv
        v += dt * dv
        d = dt * v_mean
        w = d[:, 3:6]
        th = torch.linalg.norm(w, dim=-1, keepdim=True)
        th2 = th * th
        small = th < 1e-8
        ths = torch.where(small, torch.ones_like(th), th)
        b = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(ths)) / (ths * ths))
        c = torch.where(small, 1.0 / 6 - th2 / 120, (ths - torch.sin(ths)) / (ths ** 3))
        sh = torch.where(small, 0.5 - th2 / 48, torch.sin(0.5 * ths) / ths)
        ch = torch.cos(0.5 * th)
        wxv = torch.cross(w, d[:, :3], dim=-1)
        pd = d[:, :3] + b * wxv + c * torch.cross(w, wxv, dim=-1)
        x, y, z_, w_ = q[:, 3], q[:, 4], q[:, 5], q[:, 6]
        R = torch.stack([1 - 2 * (y * y + z_ * z_), 2 * (x * y - w_ * z_), 2 * (x * z_ + w_ * y),
                         2 * (x * y + w_ * z_), 1 - 2 * (x * x + z_ * z_), 2 * (y * z_ - w_ * x),
                         2 * (x * z_ - w_ * y), 2 * (y * z_ + w_ * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
        qn = q.clone()
        qn[:, :3] = q[:, :3] + (R @ pd[:, :, None])[:, :, 0]
        dq = torch.cat([sh * w, ch], dim=-1)
        a = q[:, 3:7]
        r = torch.stack([
            a[:, 3] * dq[:, 0] + a[:, 0] * dq[:, 3] + a[:, 1] * dq[:, 2] - a[:, 2] * dq[:, 1],
            a[:, 3] * dq[:, 1] - a[:, 0] * dq[:, 2] + a[:, 1] * dq[:, 3] + a[:, 2] * dq[:, 0],
            a[:, 3] * dq[:, 2] + a[:, 0] * dq[:, 1] - a[:, 1] * dq[:, 0] + a[:, 2] * dq[:, 3],
            a[:, 3] * dq[:, 3] - a[:, 0] * dq[:, 0] - a[:, 1] * dq[:, 1] - a[:, 2] * dq[:, 2]], dim=-1)
        qn[:, 3:7] = r / torch.linalg.norm(r, dim=-1, keepdim=True)
        qn[:, 7:] = q[:, 7:] + d[:, 6:]
        return qn, v


def map_tsid_to_mujoco(q_tsid, model: ModelBlob = None):
    """main.py:11-44, batched: joint-angle targets in the sim's actuator order from a TSID q."""
    model = model or ModelBlob()
    idx = torch.as_tensor(np.asarray(model["mj_ctrl_qidx"]), dtype=torch.long, device=q_tsid.device)
    return q_tsid[..., idx]
