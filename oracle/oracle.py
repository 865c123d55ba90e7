"""ctypes binding of the CPU float64 oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (tsid_control_amd/) never does.  PARITY UNPINNED for the native stages - see oracle.h.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).parent
NQ, NV, NA, NVAR, NEQ, NIN, MAXCON, MAXEFC, NOBS = 27, 26, 20, 50, 18, 160, 32, 20 + 4 * 32, 65


def _structs(NQ, NV, NA, condim=3):
    """ctypes mirrors of oracle.h's structs for a robot's dimensions (one liboracle*.so per robot)"""
    NVAR, NEQ, NIN, MAXEFC = NV + 24, 18, 68 + 2 * NA + 2 * NV, NA + 2 * (condim - 1) * MAXCON

    class OrTerms(C.Structure):
        _fields_ = [("M", C.c_double * (NV * NV)), ("h", C.c_double * NV), ("com", C.c_double * 3),
                    ("vcom", C.c_double * 3), ("acom", C.c_double * 3), ("Jcom", C.c_double * (3 * NV)),
                    ("oMf", C.c_double * 24), ("Jf", C.c_double * (2 * 6 * NV)), ("vf", C.c_double * 12),
                    ("af", C.c_double * 12), ("mass", C.c_double), ("Aam", C.c_double * (3 * NV)), ("Lam", C.c_double * 3),
                    ("dLam", C.c_double * 3)]

    class OrQP(C.Structure):
        _fields_ = [("nvar", C.c_int), ("neq", C.c_int), ("nin", C.c_int),
                    ("H", C.c_double * (NVAR * NVAR)), ("g", C.c_double * NVAR),
                    ("CE", C.c_double * (NEQ * NVAR)), ("ce0", C.c_double * NEQ),
                    ("CI", C.c_double * (NIN * NVAR)), ("ci0", C.c_double * NIN), ("slot_foot", C.c_int * 2)]

    class OrQPSol(C.Structure):
        _fields_ = [("x", C.c_double * NVAR), ("u", C.c_double * (NEQ + NIN)), ("A", C.c_int * (NEQ + NIN)),
                    ("iq", C.c_int), ("iter", C.c_int), ("status", C.c_int), ("f_value", C.c_double)]

    class OrSimInfo(C.Structure):
        _fields_ = [("ncon", C.c_int), ("nefc", C.c_int), ("solver_iter", C.c_int),
                    ("con_geom", C.c_int * MAXCON), ("con_vert", C.c_int * MAXCON),
                    ("con_dist", C.c_double * MAXCON), ("con_pos", C.c_double * (3 * MAXCON)),
                    ("efc_force", C.c_double * MAXEFC), ("qacc", C.c_double * NV), ("qacc_smooth", C.c_double * NV),
                    ("qfrc_bias", C.c_double * NV), ("qfrc_actuator", C.c_double * NV), ("M", C.c_double * (NV * NV)),
                    ("con_body1", C.c_int * MAXCON), ("con_frame", C.c_double * (3 * MAXCON)), ("flags", C.c_int),
                    ("con_body2", C.c_int * MAXCON), ("newton_full", C.c_int), ("newton_rank1", C.c_int)]

    import types
    return types.SimpleNamespace(OrTerms=OrTerms, OrQP=OrQP, OrQPSol=OrQPSol, OrSimInfo=OrSimInfo, NQ=NQ, NV=NV, NA=NA,
                                 NVAR=NVAR, NEQ=NEQ, NIN=NIN, MAXEFC=MAXEFC, NOBS=NQ + NV + 12)


_V1 = _structs(NQ, NV, NA)
OrTerms, OrQP, OrQPSol, OrSimInfo = _V1.OrTerms, _V1.OrQP, _V1.OrQPSol, _V1.OrSimInfo


class OrWalkTables(C.Structure):
    _fields_ = [("coef", C.c_void_p), ("rest", C.c_void_p), ("com", C.c_void_p), ("t_off", C.c_void_p),
                ("side", C.c_void_p), ("nsteps", C.c_void_p), ("K", C.c_int), ("t", C.c_double), ("T", C.c_double),
                ("t_start", C.c_double), ("omega", C.c_double), ("z0", C.c_double), ("dz", C.c_double),
                ("td_latch", C.c_void_p), ("td_frac", C.c_double)]


class WalkTables:
    """numpy copies of a WalkSchedule's tables (first n envs) in the layout or_env_step_batch_walk reads"""

    def __init__(self, sched, n=None):
        n = sched.N if n is None else n
        g = lambda x: np.ascontiguousarray(x[:n].detach().cpu().numpy(), dtype=np.float64)
        self.keep = dict(coef=g(sched.coef), rest=g(sched.rest), com=g(sched.com),
                         side=np.ascontiguousarray(sched.side[:n].cpu().numpy(), dtype=np.int32),
                         nsteps=np.ascontiguousarray(sched.nsteps[:n].cpu().numpy(), dtype=np.int32),
                         t_off=g(sched.t_offset) if getattr(sched, "t_offset", None) is not None else None)
        k = self.keep
        self.c = OrWalkTables(k["coef"].ctypes.data, k["rest"].ctypes.data, k["com"].ctypes.data,
                              k["t_off"].ctypes.data if k["t_off"] is not None else None, k["side"].ctypes.data,
                              k["nsteps"].ctypes.data, sched.K, 0.0, float(sched.conf.step_duration),
                              float(sched.t_start), float(sched.omega), float(sched.z0), float(sched.dz), None, 0.0)
        if getattr(sched, "td_latch", None) is not None:   # contact-timing feedback: the oracle keeps its own latch
            k["td_latch"] = np.ascontiguousarray(sched.td_latch[:n].cpu().numpy(), dtype=np.int32)
            self.c.td_latch, self.c.td_frac = k["td_latch"].ctypes.data, float(sched.td_fraction)

    def at(self, t):
        self.c.t = float(t)
        return self


def build(force=False):
    so, so0 = HERE / "liboracle.so", HERE / "liboracle_v0.so"
    srcs = list(HERE.glob("*.c")) + [HERE / "oracle.h"]
    for lib in (so, so0):
        if force or not lib.exists() or any(s.stat().st_mtime > lib.stat().st_mtime for s in srcs):
            subprocess.run(["make", "-C", str(HERE), lib.name], check=True, capture_output=True)
    return so


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    def __init__(self, blob_bytes: bytes, lib_path=None):
        import os
        import struct
        # which robot is the blob for?  (model_dims section: NJ NQ NV NA NB has_sim) -> liboracle.so or liboracle_v0.so
        nsec = struct.unpack_from("<I", blob_bytes, 8)[0]
        md = None
        for i in range(nsec):
            off = 16 + 40 * i
            if blob_bytes[off:off + 24].split(b"\0")[0] == b"model_dims":
                _, cnt, o = struct.unpack_from("<IIQ", blob_bytes, off + 24)
                md = struct.unpack_from("<9i", blob_bytes, o)   # NJ NQ NV NA NB has_sim NG condim eulerdamp
        if md is None:
            raise RuntimeError("oracle: the blob has no model_dims section")
        v1 = md[0] == 21
        default = HERE / ("liboracle.so" if v1 else "liboracle_v0.so")
        so = Path(lib_path or (os.environ.get("TSIDB_ORACLE_LIB") if v1 else None) or default)  # env: sanitizer build (v1)
        if not so.exists():
            build()
        self.lib = L = C.CDLL(str(so))
        self.S = _structs(md[1], md[2], md[3], md[7])
        self.dims = md
        L.or_model_load.restype = C.c_void_p
        L.or_model_load.argtypes = [C.c_char_p, C.c_size_t]
        for name in ("or_model_free", "or_rbd_terms", "or_rnea", "or_integrate", "or_log6", "or_tsid_assemble", "or_walk_update"):
            getattr(L, name).restype = None
        for name in ("or_qp_solve", "or_tsid_tick", "or_sim_step", "or_sim_step_env", "or_env_step_batch", "or_env_step_batch_env",
                     "or_env_step_batch_walk"):
            getattr(L, name).restype = C.c_int
        self.m = C.c_void_p(L.or_model_load(blob_bytes, len(blob_bytes)))
        if not self.m:
            raise RuntimeError("oracle: model blob rejected")

    def __del__(self):
        try:
            self.lib.or_model_free(self.m)
        except Exception:
            pass

    # ---- rigid-body terms
    def terms(self, q, v):
        S = self.S; NV = S.NV
        t = S.OrTerms()
        q, v = _f64(q), _f64(v)
        self.lib.or_rbd_terms(self.m, _p(q), _p(v), C.byref(t))
        g = lambda f, *s: np.array(f, dtype=np.float64).reshape(*s)
        return dict(M=g(t.M, NV, NV), h=g(t.h, NV), com=g(t.com, 3), vcom=g(t.vcom, 3), acom=g(t.acom, 3),
                    Jcom=g(t.Jcom, 3, NV), oMf=g(t.oMf, 2, 12), Jf=g(t.Jf, 2, 6, NV), vf=g(t.vf, 2, 6),
                    af=g(t.af, 2, 6), mass=t.mass, Aam=g(t.Aam, 3, NV), Lam=g(t.Lam, 3), dLam=g(t.dLam, 3), _raw=t)

    def rnea(self, q, v, a):
        q, v, a = _f64(q), _f64(v), _f64(a)
        tau = np.zeros(self.S.NV)
        self.lib.or_rnea(self.m, _p(q), _p(v), _p(a), _p(tau))
        return tau

    def integrate(self, q, vdt):
        q, vdt = _f64(q), _f64(vdt)
        out = np.zeros(self.S.NQ)
        self.lib.or_integrate(_p(q), _p(vdt), _p(out))
        return out

    def log6(self, R, p):
        m = _f64(np.concatenate([np.asarray(R).reshape(-1), np.asarray(p).reshape(-1)]))
        out = np.zeros(6)
        self.lib.or_log6(_p(m), _p(out))
        return out

    # ---- TSID problem
    def assemble(self, params, q, v, com_ref, posture_ref, foot_ref, contact_ref, contact_active, cop_ref=None):
        t = self.terms(q, v)["_raw"]
        S = self.S; NVAR, NEQ, NIN = S.NVAR, S.NEQ, S.NIN
        qp = S.OrQP()
        a = [_f64(x) for x in (params, q, v, com_ref, posture_ref, foot_ref, contact_ref)]
        ca = np.ascontiguousarray(contact_active, dtype=np.uint8)
        cr = _f64(cop_ref) if cop_ref is not None else None
        self.lib.or_tsid_assemble_cop.restype = None
        self.lib.or_tsid_assemble_cop(self.m, _p(a[0]), C.byref(t), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                      _p(a[6]), _p(ca), _p(cr) if cr is not None else None, C.byref(qp))
        n, ne, ni = qp.nvar, qp.neq, qp.nin
        H = np.array(qp.H).reshape(NVAR, NVAR)[:n, :n]
        CE = np.array(qp.CE).reshape(NEQ, NVAR)[:ne, :n]
        CI = np.array(qp.CI).reshape(NIN, NVAR)[:ni, :n]
        return dict(H=H, g=np.array(qp.g)[:n], CE=CE, ce0=np.array(qp.ce0)[:ne], CI=CI, ci0=np.array(qp.ci0)[:ni],
                    slot_foot=list(qp.slot_foot), _raw=qp)

    def qp_solve(self, qp_raw, max_iter=1000):
        sol = self.S.OrQPSol()
        st = self.lib.or_qp_solve(C.byref(qp_raw), max_iter, C.byref(sol))
        n = qp_raw.nvar
        return dict(status=st, x=np.array(sol.x)[:n], u=np.array(sol.u)[:sol.iq], A=np.array(sol.A)[:sol.iq],
                    iq=sol.iq, iter=sol.iter, f_value=sol.f_value)

    def tsid_tick(self, params, q, v, com_ref, posture_ref, foot_ref, contact_ref, contact_active, cop_frames=None, cop_ref=None):
        """In-place on q, v (float64 arrays). Returns dict(tau, dv, f, obs, status, iters)."""
        assert q.dtype == np.float64 and v.dtype == np.float64
        a = [_f64(x) for x in (params, com_ref, posture_ref, foot_ref, contact_ref)]
        ca = np.ascontiguousarray(contact_active, dtype=np.uint8)
        cf = _f64(cop_frames) if cop_frames is not None else None
        S = self.S
        tau, dv, f, obs = np.zeros(S.NA), np.zeros(S.NV), np.zeros(24), np.zeros(S.NOBS)
        it = C.c_int(0)
        cr = _f64(cop_ref) if cop_ref is not None else None
        self.lib.or_tsid_tick_cop.restype = C.c_int
        st = self.lib.or_tsid_tick_cop(self.m, _p(a[0]), _p(q), _p(v), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(ca),
                                       _p(cf) if cf is not None else None, _p(cr) if cr is not None else None,
                                       _p(tau), _p(dv), _p(f), _p(obs), C.byref(it))
        return dict(tau=tau, dv=dv, f=f, obs=obs, status=st, iters=it.value)

    # ---- sim
    def sim_step(self, qpos, qvel, ctrl, qacc_ws, envp=None, terrain=None, self_collision=True, plane_mesh="mujoco"):
        assert all(x.dtype == np.float64 for x in (qpos, qvel, qacc_ws))
        self.lib.or_model_set_plane_mesh.argtypes = [C.c_void_p, C.c_int]
        self.lib.or_model_set_plane_mesh(self.m, int(plane_mesh == "mujoco"))
        ctrl = _f64(ctrl)
        info = self.S.OrSimInfo()
        NV = self.S.NV
        ep = _f64(envp) if envp is not None else None
        tr = _f64(terrain) if terrain is not None else None
        self.lib.or_sim_step_ext.restype = C.c_int
        rc = self.lib.or_sim_step_ext(self.m, _p(qpos), _p(qvel), _p(ctrl), None, _p(qacc_ws), _p(ep) if ep is not None else None,
                                      _p(tr) if tr is not None else None, int(bool(self_collision)), C.byref(info))
        nc, ne = info.ncon, info.nefc
        return dict(rc=rc, ncon=nc, nefc=ne, iters=info.solver_iter, con_geom=np.array(info.con_geom)[:nc],
                    con_body1=np.array(info.con_body1)[:nc], con_body2=np.array(info.con_body2)[:nc], con_frame=np.array(info.con_frame).reshape(MAXCON, 3)[:nc],
                    flags=info.flags,
                    con_vert=np.array(info.con_vert)[:nc], con_dist=np.array(info.con_dist)[:nc],
                    con_pos=np.array(info.con_pos).reshape(MAXCON, 3)[:nc], efc_force=np.array(info.efc_force)[:ne],
                    qacc=np.array(info.qacc), qacc_smooth=np.array(info.qacc_smooth),
                    qfrc_bias=np.array(info.qfrc_bias), qfrc_actuator=np.array(info.qfrc_actuator),
                    M=np.array(info.M).reshape(NV, NV))

    def mpr(self, a, Ra, pa, b, Rb, pb, margin=0.0):
        """or_mpr_penetration of geom a's hull (its body placed at Ra, pa) and geom b's: None or (depth, dir a->b, pos).
        margin: both hulls grown by margin / 2 (the caller's dist = margin - depth)."""
        Ra, pa, Rb, pb = (_f64(np.asarray(x).reshape(-1)) for x in (Ra, pa, Rb, pb))
        depth = C.c_double(0.0)
        d, p = np.zeros(3), np.zeros(3)
        self.lib.or_mpr_penetration_margin.restype = C.c_int
        hit = self.lib.or_mpr_penetration_margin(self.m, int(a), _p(Ra), _p(pa), int(b), _p(Rb), _p(pb), C.c_double(margin),
                                                 C.byref(depth), _p(d), _p(p))
        return (depth.value, d, p) if hit else None

    # ---- batch env step (all arrays float64, env-major, updated in place)
    def env_step_batch(self, params, st, nthreads=1, walk=None):
        """walk = WalkTables.at(t): the walking reference update (or_walk.c) runs per env before its tick, with
        st["frames"] [n,2,12] carrying the sole placements between ticks (the config-3 workload)."""
        n = st["q"].shape[0]
        params = _f64(params)
        cf = st.get("cop_frames")
        ep = st.get("env_params")
        fr = st.get("frames")
        rd = st.get("rewdone")
        tr = st.get("terrain")
        cr = st.get("cop_ref")
        if walk is not None and fr is None:
            raise ValueError("walking env step needs st['frames']")
        self.lib.or_env_step_batch_walk(
            self.m, _p(params), n, _p(st["q"]), _p(st["v"]), _p(st["qpos"]), _p(st["qvel"]), _p(st["qacc_ws"]),
            _p(st["com_ref"]), _p(st["posture_ref"]), _p(st["foot_ref"]), _p(st["contact_ref"]),
            _p(st["contact_active"]), _p(cf) if cf is not None else None, _p(ep) if ep is not None else None,
            _p(st["tau"]), _p(st["dv"]), _p(st["f"]),
            _p(st["status"]), _p(st["obs"]), _p(st["ncon"]), _p(st["con_geom"]), int(nthreads),
            C.byref(walk.c) if walk is not None else None, _p(fr) if fr is not None else None,
            _p(rd) if rd is not None else None, _p(tr) if tr is not None else None,
            _p(cr) if cr is not None else None)


def walk_update(lib, sched, t, frames, foot_ref, contact_ref, contact_active, com_ref):
    """or_walk_update on numpy copies of a WalkSchedule's tables; the four reference arrays are updated in place."""
    g = lambda x: _f64(x.detach().cpu().numpy())
    coef, rest, com = g(sched.coef), g(sched.rest), g(sched.com)
    side = np.ascontiguousarray(sched.side.cpu().numpy(), dtype=np.int32)
    nsteps = np.ascontiguousarray(sched.nsteps.cpu().numpy(), dtype=np.int32)
    toff = g(sched.t_offset) if getattr(sched, "t_offset", None) is not None else None
    frames = _f64(frames)
    lib.or_walk_update.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                   C.c_void_p] + [C.c_double] * 5 + [C.c_void_p] * 5
    lib.or_walk_update(sched.N, _p(coef), _p(side), _p(nsteps), _p(rest), _p(com), sched.K, float(t),
                       _p(toff) if toff is not None else None, float(sched.conf.step_duration), float(sched.t_start),
                       float(sched.omega), float(sched.z0), float(sched.dz), _p(frames), _p(foot_ref), _p(contact_ref),
                       _p(contact_active), _p(com_ref))


def walk_plan(lib, pp, cop_frames, com_ref, K, path=None, npts=None, scale=None, episode=None):
    """or_walk_plan: the episode plan tables for n envs (layouts in or_walk.c).  pp = the 16 plan parameters
    (tsid_control_amd.walk_planner.plan_params)."""
    cop_frames, com_ref, pp = _f64(cop_frames), _f64(com_ref), _f64(pp)
    n = cop_frames.shape[0]
    P = 0
    if path is not None:
        path = _f64(path)
        P = path.shape[1]
        npts = np.ascontiguousarray(npts, dtype=np.int32)
    scale = _f64(scale) if scale is not None else None
    episode = np.ascontiguousarray(episode, dtype=np.int32) if episode is not None else None
    out = dict(steps=np.zeros((n, K + 2, 4)), coef=np.zeros((n, K, 4, 4)), side=np.zeros((n, K), dtype=np.int32),
               nsteps=np.zeros(n, dtype=np.int32), rest=np.zeros((n, K + 1, 2, 4)), com=np.zeros((n, K + 2, 2, 3)),
               flags=np.zeros(n, dtype=np.int32))
    o = lambda x: _p(x) if x is not None else None
    lib.or_walk_plan.restype = None
    lib.or_walk_plan.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
    lib.or_walk_plan(n, _p(pp), _p(cop_frames), _p(com_ref), o(path), o(npts), P, o(scale), o(episode), K, _p(out["steps"]),
                     _p(out["coef"]), _p(out["side"]), _p(out["nsteps"]), _p(out["rest"]), _p(out["com"]), _p(out["flags"]))
    return out


def new_state(n, dims=None):
    """Zeroed env-major float64 state/IO arrays in the layout or_env_step_batch expects (dims = (NQ, NV, NA) of the
    robot; default the v1 robot's)."""
    NQ, NV, NA = dims if dims is not None else (_V1.NQ, _V1.NV, _V1.NA)
    NOBS = NQ + NV + 12
    z = lambda *s: np.zeros(s, dtype=np.float64)
    return dict(q=z(n, NQ), v=z(n, NV), qpos=z(n, NQ), qvel=z(n, NV), qacc_ws=z(n, NV), com_ref=z(n, 9),
                posture_ref=z(n, NA), foot_ref=z(n, 2, 24), contact_ref=z(n, 2, 12),
                contact_active=np.ones((n, 2), dtype=np.uint8), cop_frames=z(n, 2, 12), tau=z(n, NA), dv=z(n, NV),
                f=z(n, 24), status=np.zeros(n, dtype=np.int32), obs=z(n, NOBS), ncon=np.zeros(n, dtype=np.int32),
                con_geom=np.zeros((n, MAXCON), dtype=np.int32), rewdone=z(n, 2))
