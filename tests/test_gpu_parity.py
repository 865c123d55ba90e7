"""GPU parity: the HIP path (through the C-ABI, via the WalkController facade) against the CPU oracle
on the same seeded inputs.  Tolerances:
  f64 path  - tau/dv/f-wrench 1e-7 abs, next q/v/qpos 1e-9, qvel 1e-6; status, contact pairs bit-exact
  f32 path  - one tick / one env step from identical inputs: test_single_tick_f32_within_tolerance (64 envs) and
              test_one_env_step_f32_against_section5 (256 envs; states what holds as written and what only in an amended form)
The oracle itself is unpinned against tsid/pinocchio/mujoco (none available)."""
import numpy as np
import pytest
import torch

from conftest import oracle_state  # noqa: F401

pytestmark = pytest.mark.gpu
NQ, NV = 27, 26


def make(n, dtype="f64", **conf_over):
    from tsid_control_amd import RobotConfig, WalkController
    conf = RobotConfig()
    conf.dtype = dtype
    if conf_over.pop("walking", False):
        from tsid_control_amd.walk_planner import op3_walking_conf
        op3_walking_conf(conf)
    for k, v in conf_over.items():
        setattr(conf, k, v)
    return WalkController(conf, num_envs=n, device="cuda:0")


def perturb(wc, seed=0, dq=0.05, dv=0.05):
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = wc.num_envs
    wc.q[:, 7:] += ((torch.rand(n, 20, generator=g, dtype=torch.float64) - 0.5) * 2 * dq).to(wc.device, wc.dtype)
    wc.v[:] = (torch.randn(n, 26, generator=g, dtype=torch.float64) * dv).to(wc.device, wc.dtype)


def mirror(wc):
    """oracle state arrays holding exactly what the controller's tensors hold"""
    from oracle.oracle import new_state
    st = new_state(wc.num_envs)
    for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames", "contact_active"):
        st[k][...] = getattr(wc, k).cpu().numpy().reshape(st[k].shape)
    st["qacc_ws"][...] = wc.qacc_warmstart.double().cpu().numpy()
    return st


def diff(t, a):
    return float(np.abs(t.double().cpu().numpy().reshape(a.shape) - a).max())


def wrench(f, params):
    cp = params[19:31].reshape(4, 3)
    T = np.zeros((6, 12))
    for i in range(4):
        T[:3, 3 * i:3 * i + 3] = np.eye(3)
        T[3:, 3 * i:3 * i + 3] = np.array([[0, -cp[i, 2], cp[i, 1]], [cp[i, 2], 0, -cp[i, 0]], [-cp[i, 1], cp[i, 0], 0]])
    f = np.asarray(f).reshape(-1, 2, 12)
    return np.einsum("ij,nfj->nfi", T, f)


def test_reset_matches_reference_init(oracle, standing):
    wc = make(5)
    assert diff(wc.q, np.tile(standing["q"], (5, 1))) < 1e-14
    assert diff(wc.qpos, np.tile(standing["q"], (5, 1))) < 1e-14               # main.py:64
    assert diff(wc.com_ref, np.tile(standing["com_ref"], (5, 1))) < 1e-14
    assert diff(wc.contact_ref, np.tile(standing["contact_ref"], (5, 1, 1))) < 1e-14
    assert diff(wc.foot_ref, np.tile(standing["foot_ref"], (5, 1, 1))) == 0
    assert diff(wc.cop_frames, np.tile(standing["cop_frames"], (5, 1, 1))) < 1e-14
    assert wc.contact_active.cpu().numpy().tolist() == [[1, 1]] * 5
    assert abs(float(wc.q[0, 2]) - 0.331968) < 1e-6
    # partial reset touches only the listed envs
    wc.q += 1.0
    wc.reset(env_ids=[1, 3])
    assert diff(wc.q[[1, 3]], np.tile(standing["q"], (2, 1))) < 1e-14 and float((wc.q[[0, 2, 4]] - 1.0 - wc.q[[1, 3]][0]).abs().max()) < 1e-14


def test_rbd_terms_f64(oracle):
    wc = make(16)
    perturb(wc, 1, dq=0.5, dv=1.0)
    g = torch.Generator().manual_seed(2)
    quat = torch.randn(16, 4, generator=g, dtype=torch.float64)
    wc.q[:, 3:7] = (quat / quat.norm(dim=1, keepdim=True)).to(wc.device)
    wc.q[:, :3] += torch.randn(16, 3, generator=g, dtype=torch.float64).to(wc.device)
    t = wc.rbd_terms()
    q, v = wc.q.cpu().numpy(), wc.v.cpu().numpy()
    ref = [oracle.terms(q[e], v[e]) for e in range(16)]
    for key in ("M", "h", "Jcom", "Jf", "oMf", "com"):
        o = np.stack([r[key] for r in ref])
        assert diff(t[key], o) < 1e-12 * max(1.0, np.abs(o).max()), key
    M = t["M"].cpu().numpy()
    assert np.abs(M - M.transpose(0, 2, 1)).max() == 0


def test_rbd_terms_f32(oracle):
    wc = make(16, "f32")
    perturb(wc, 1, dq=0.5, dv=1.0)
    wc.q[:, :3] += 3.0  # a few metres from the origin: base-centred spatial vectors keep fp32 accurate
    t = wc.rbd_terms()
    q, v = wc.q.double().cpu().numpy(), wc.v.double().cpu().numpy()
    ref = [oracle.terms(q[e], v[e]) for e in range(16)]
    for key, tol in (("M", 2e-6), ("h", 5e-5), ("Jcom", 1e-6), ("Jf", 2e-6), ("com", 2e-6)):
        o = np.stack([r[key] for r in ref])
        assert diff(t[key], o) < tol * max(1.0, np.abs(o).max()), key


@pytest.mark.parametrize("sim", [True, False])
def test_env_loop_f64_matches_oracle(oracle, sim):
    """Config 2 in small: perturbed standing envs, 40 env steps, every per-env output every step."""
    wc = make(48, sim_enabled=sim)
    perturb(wc, 3)
    st = mirror(wc)
    for i in range(40):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
        assert diff(wc.tau, st["tau"]) < 1e-7 and diff(wc.dv, st["dv"]) < 1e-7, i
        assert np.abs(wrench(wc.f.cpu().numpy(), wc.params) - wrench(st["f"], wc.params)).max() < 1e-7, i
        assert diff(wc.q, st["q"]) < 1e-9 and diff(wc.v, st["v"]) < 1e-9, i
        assert diff(wc.obs, st["obs"]) < 1e-7, i
        assert diff(wc.rows[:, 65:], st["rewdone"]) < 1e-9, i                       # reward, done
        if sim:
            assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
            assert np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i      # integer contact indexing bit-exact
            assert diff(wc.qpos, st["qpos"]) < 1e-9 and diff(wc.qvel, st["qvel"]) < 1e-6, i
    assert int((wc.info[:, 1] > 18).sum()) > 0     # inequality constraints were active somewhere
    if sim:
        assert int(wc.ncon.max()) > 0               # contacts formed somewhere


def test_single_tick_f32_within_tolerance(oracle):
    wc = make(64, "f32")
    perturb(wc, 4)
    st = mirror(wc)
    wc.tick()
    for e in range(64):
        out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                               st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
        st["tau"][e], st["dv"][e], st["f"][e], st["status"][e] = out["tau"], out["dv"], out["f"], out["status"]
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    tau, dv = wc.tau.double().cpu().numpy(), wc.dv.double().cpu().numpy()
    # rtol 1e-3, atol 1e-4 for tau, dv and the per-foot wrench, as written.  (Until the equality QR
    # took the contact-motion columns first - DESIGN.md section 7 - the wrench met it only with atol 1e-3: 9.7e-4 on 20 N
    # components; with that order it is 1.4e-6 of the normal force.)
    assert np.allclose(tau, st["tau"], rtol=1e-3, atol=1e-4)
    assert np.allclose(dv, st["dv"], rtol=1e-3, atol=1e-4)
    assert np.allclose(wrench(wc.f.double().cpu().numpy(), wc.params), wrench(st["f"], wc.params), rtol=1e-3, atol=1e-4)
    assert diff(wc.q, st["q"]) < 1e-5 and diff(wc.v, st["v"]) < 1e-4


def test_one_env_step_f32_against_section5(oracle):
    """The f32 path against BASELINE.md section 5 on ONE env step from identical inputs, 256 perturbed standing envs (VERDICT r2
    item 8) - what holds as written, and what holds only in an amended form, with the reason:
      tau, dv, per-foot wrench,  as written (rtol 1e-3 / atol 1e-4; atol 1e-5): observed ratios to the tolerance 0.05, 0.06, 0.03
      next q, next v             (wrench error 1.4e-6 of the normal force), 3e-8, 4e-8.  (tau and the wrench needed amended
                                 tolerances - 9.4e-5 of f_z - until the equality QR took the contact-motion columns first,
                                 DESIGN.md section 7: that order is the better-conditioned one in float32.)
      contact (geom, vertex) ids bit-exact where float32 can tell the lowest sole vertex from its neighbours: ties are taken
                                 within 2e-6 m in float32 (1e-9 in float64), so on near-flat soles the
                                 support vertex - and with it the neighbour list - may differ: >= 85 % of the envs identical
      next qpos, qvel            on the envs with identical contact lists: qpos atol 1e-5 as written (observed 1e-7); qvel atol
                                 5e-5, 99 % within 1e-5 - the contact rows are stiff (D ~ 1e6 against M ~ 1e-3 in H = M + J^T D J),
                                 and dt x qacc carries float32's share of that conditioning (observed 1.9e-5 in two envs)"""
    n = 256
    wc = make(n, "f32")
    perturb(wc, 4)
    st = mirror(wc)
    wc.step()
    oracle.env_step_batch(wc.params, st, nthreads=8)
    g = lambda k: getattr(wc, k).double().cpu().numpy().reshape(n, -1)
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    assert np.allclose(g("dv"), st["dv"], rtol=1e-3, atol=1e-4)
    assert np.abs(g("q") - st["q"]).max() < 1e-5 and np.abs(g("v") - st["v"]).max() < 1e-5
    w, w0 = wrench(g("f"), wc.params), wrench(st["f"], wc.params)
    assert np.allclose(w, w0, rtol=1e-3, atol=1e-4)
    assert np.allclose(g("tau"), st["tau"], rtol=1e-3, atol=1e-4)
    same = (wc.con_pairs.cpu().numpy() == st["con_geom"]).all(axis=1)
    assert same.mean() >= 0.85 and (wc.ncon.cpu().numpy() == st["ncon"]).mean() >= 0.95
    dq, dqv = np.abs(g("qpos") - st["qpos"])[same], np.abs(g("qvel") - st["qvel"])[same]
    assert dq.max() < 1e-5 and dqv.max() < 5e-5 and (dqv.max(axis=1) <= 1e-5).mean() >= 0.99


def test_sim_only_settles_on_the_floor_f64(oracle):
    """Contact-solver stress without the teleport: drop 0.5 mm into the floor and let the sim settle."""
    wc = make(4)
    wc.qpos[:, 3:7] = torch.tensor([1.0, 0, 0, 0], dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 2] -= 0.0005
    wc.qpos[1, 7] = 0.05
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc.qpos, wc.qvel, wc.qacc_warmstart))
    for i in range(300):
        wc.sim_step(teleport=False)
        for e in range(4):
            oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e])
    assert diff(wc.qpos, qpos) < 1e-8 and diff(wc.qvel, qvel) < 1e-5
    assert float(wc.qvel.abs().max()) < 0.1 and 0.3315 < float(wc.qpos[0, 2]) < 0.3325
    assert int(wc.ncon.min()) >= 2


def test_contact_switching_and_walking_refs_f64(oracle):
    """Config 3 in small: footstep schedule drives update_tasks (contact on/off edges, swing
    references); the oracle receives the same reference arrays each tick."""
    from tsid_control_amd.walk_planner import WalkSchedule
    from tsid_control_amd.walk_planner import op3_walking_posture
    n = 12
    wc = make(n, sim_enabled=False, walking=True)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf = wc.frames[0, 0, 9:11].cpu().numpy()
    rf = wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), t_start=0.1,
                                         com0=wc.com_ref[0, :3].cpu().numpy())
    st = mirror(wc)
    seen_single = False
    for i in range(300):
        t = i * wc.conf.dt
        sLF, sRF, cLF, cRF = sched.sample(t)
        wc.update_tasks(sLF, sRF, cLF, cRF)
        wc.com_ref[:] = sched.com_ref(t)
        for k in ("foot_ref", "contact_ref", "contact_active", "com_ref"):
            st[k][...] = getattr(wc, k).cpu().numpy().reshape(st[k].shape)
        seen_single |= bool((wc.contact_active.sum(dim=1) == 1).any())
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
        ok = st["status"] == 0
        assert diff(wc.q[ok], st["q"][ok]) < 1e-8 and diff(wc.tau[ok], st["tau"][ok]) < 1e-6, i
    assert seen_single


def test_active_set_stress_f64(oracle):
    """Large velocity/posture perturbations: many inequality rows become active and get dropped again
    (the register-resident active-set loop's add, partial-step/drop and re-add paths), single and double
    support mixed.  One tick per env from identical inputs; every env must agree with the oracle."""
    n = 256
    wc = make(n, sim_enabled=False)
    perturb(wc, 21, dq=0.25, dv=1.5)
    wc.contact_active[::3, 0] = 0          # every third env in right single support
    wc.contact_active[1::7, 1] = 0         # some in left single support / flight
    wc.contact_active[(wc.contact_active.sum(dim=1) == 0), 0] = 1
    st = mirror(wc)
    wc.tick()
    iters_o = np.zeros(n, int)
    for e in range(n):
        out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                               st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
        st["tau"][e], st["dv"][e], st["f"][e], st["status"][e], iters_o[e] = out["tau"], out["dv"], out["f"], out["status"], out["iters"]
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    ok = st["status"] == 0
    assert ok.sum() > n // 2
    # accelerations reach ~1e2 rad/s^2 here; H carries a 1e-8 regulariser, so compare relatively
    assert np.allclose(wc.dv.cpu().numpy()[ok], st["dv"][ok], rtol=1e-6, atol=1e-6)
    assert np.allclose(wc.tau.cpu().numpy()[ok], st["tau"][ok], rtol=1e-6, atol=1e-6)
    assert np.allclose(wrench(wc.f.cpu().numpy(), wc.params)[ok], wrench(st["f"], wc.params)[ok], rtol=1e-6, atol=1e-5)
    assert diff(wc.q[ok], st["q"][ok]) < 1e-9
    it_g = wc.info[:, 0].cpu().numpy()
    assert it_g.max() >= 8 and (wc.info[:, 1].cpu().numpy() - np.where(st["contact_active"].sum(1) == 2, 18, 12)).max() >= 6
    # the two solvers walk the same active-set path (same number of outer iterations) on most envs; on
    # paths of 30-45 iterations with 20+ active rows rounding reorders near-equal violations, the optimum
    # (checked above) is the same
    assert (it_g[ok] == iters_o[ok]).mean() > 0.7 and np.abs(it_g[ok] - iters_o[ok]).max() <= 16


def test_angular_momentum_task_f64(oracle):
    """the legacy controller's angular-momentum task (legacy/biped.py:82-87) as three more cost
    rows; light (legacy/op3_conf.py:15) and heavy weights, double and single support, moving states."""
    for w_am in (1e-3, 5.0):
        n = 48
        wc = make(n, sim_enabled=False, w_am=w_am)
        perturb(wc, 41, dq=0.1, dv=0.6)
        wc.contact_active[::3, 0] = 0
        wc.contact_active[1::5, 1] = 0
        wc.contact_active[(wc.contact_active.sum(dim=1) == 0), 1] = 1
        st = mirror(wc)
        wc.tick()
        for e in range(n):
            out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                                   st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
            st["tau"][e], st["dv"][e], st["f"][e], st["status"][e] = out["tau"], out["dv"], out["f"], out["status"]
        assert np.array_equal(wc.status.cpu().numpy(), st["status"])
        ok = st["status"] == 0
        assert ok.sum() > n // 2
        assert np.allclose(wc.dv.cpu().numpy()[ok], st["dv"][ok], rtol=1e-7, atol=1e-7)
        assert np.allclose(wc.tau.cpu().numpy()[ok], st["tau"][ok], rtol=1e-7, atol=1e-7)
        assert diff(wc.q[ok], st["q"][ok]) < 1e-9
    # and the task does something: same states without it give different accelerations
    ref = make(n, sim_enabled=False)
    perturb(ref, 41, dq=0.1, dv=0.6)
    ref.tick()
    assert float((ref.dv[:, :6] - wc.dv[:, :6]).abs().max()) > 1e-2


def test_legacy_capture_point_and_support_polygon():
    """legacy/biped.py:224-234, batched."""
    wc = make(4, sim_enabled=False)
    wc.tick()
    com = wc.obs[:, 53:56].clone()
    dcom = torch.full((4, 3), 0.2, dtype=wc.dtype, device=wc.device)
    cp = wc.compute_capture_point(com, dcom, w=3.0)
    assert torch.allclose(cp[:, :2], com[:, :2] + 0.2 / 3.0) and float(cp[:, 2].abs().max()) == 0.0
    sp = wc.compute_support_polygon()
    assert sp.shape == (4, 2, 2) and torch.equal(sp[:, 0], wc.frames[:, 0, 9:11]) and torch.equal(sp[:, 1], wc.frames[:, 1, 9:11])
    assert float(sp[0, 0, 0]) > 0 > float(sp[0, 1, 0])            # left sole at +x, right at -x


def test_flight_and_single_support_ticks(oracle):
    """All four contact configurations (both, left only, right only, none) in one batch."""
    n = 64
    wc = make(n, sim_enabled=False)
    perturb(wc, 33, dq=0.1, dv=0.5)
    wc.contact_active[:, 0] = (torch.arange(n, device=wc.device) % 4 < 2).to(torch.uint8)
    wc.contact_active[:, 1] = (torch.arange(n, device=wc.device) % 2 == 0).to(torch.uint8)
    st = mirror(wc)
    wc.tick()
    for e in range(n):
        out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                               st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
        st["tau"][e], st["dv"][e], st["f"][e], st["status"][e] = out["tau"], out["dv"], out["f"], out["status"]
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    ok = st["status"] == 0
    assert ok.sum() >= n // 2
    assert np.allclose(wc.dv.cpu().numpy()[ok], st["dv"][ok], rtol=1e-7, atol=1e-7)
    assert np.allclose(wc.tau.cpu().numpy()[ok], st["tau"][ok], rtol=1e-7, atol=1e-7)
    f = wc.f.cpu().numpy()
    ca = st["contact_active"]
    assert np.all(f[ca[:, 0] == 0, :12] == 0) and np.all(f[ca[:, 1] == 0, 12:] == 0)   # no force on a lifted foot
    assert (wc.info[:, 1].cpu().numpy()[ok] >= 6 + 6 * ca.sum(1)[ok]).all()


def test_max_iter_status_on_gpu(oracle):
    wc = make(32, sim_enabled=False, qp_max_iter=3)
    perturb(wc, 5, dq=0.2, dv=1.0)
    st = mirror(wc)
    q0 = wc.q.clone()
    wc.tick()
    for e in range(32):
        out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                               st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
        st["status"][e] = out["status"]
    sg = wc.status.cpu().numpy()
    assert np.array_equal(sg, st["status"]) and (sg == 3).any()          # HQP_STATUS_MAX_ITER_REACHED
    assert torch.equal(wc.q[torch.as_tensor(sg == 3)], q0[torch.as_tensor(sg == 3)])


def test_env_loop_f32_tracks_oracle(oracle):
    """f32 path over 25 env steps: states stay within tolerance of the float64 oracle."""
    wc = make(32, "f32")
    perturb(wc, 9, dq=0.03, dv=0.03)
    st = mirror(wc)
    for i in range(25):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    assert diff(wc.q, st["q"]) < 2e-4 and diff(wc.v, st["v"]) < 2e-2
    assert diff(wc.qpos[:, :7], st["qpos"][:, :7]) < 2e-4


def test_walk_update_kernel_equals_host_path():
    """tsidb_walk_update (one kernel) == update_tasks(sample(t)) + com_ref(t) (tensor expressions)."""
    from tsid_control_amd.walk_planner import WalkSchedule
    n = 40
    a, b = make(n, sim_enabled=False, walking=True), make(n, sim_enabled=False, walking=True)
    lf, rf = a.frames[0, 0, 9:11].cpu().numpy(), a.frames[0, 1, 9:11].cpu().numpy()
    sa = WalkSchedule.from_demo_paths(n, a.conf, a.device, a.dtype, seed=3, q0_feet=(lf, rf), t_start=0.3)
    sb = WalkSchedule.from_demo_paths(n, b.conf, b.device, b.dtype, seed=3, q0_feet=(lf, rf), t_start=0.3)
    for i in range(0, 700, 7):
        t = i * a.conf.dt
        sa.apply(a, t)
        sLF, sRF, cLF, cRF = sb.sample(t)
        b.update_tasks(sLF, sRF, cLF, cRF)
        b.com_ref[:] = sb.com_ref(t)
        assert torch.equal(a.contact_active, b.contact_active), i
        assert float((a.foot_ref - b.foot_ref).abs().max()) < 1e-10, i
        assert float((a.contact_ref - b.contact_ref).abs().max()) < 1e-10, i  # the two runs drift apart at rounding level
        assert float((a.com_ref - b.com_ref).abs().max()) < 1e-10, i
        a.tick(); b.tick()
    assert int((a.contact_active.sum(dim=1) == 1).sum()) > 0


def test_walking_workload_stays_on_its_plan():
    """Config 3 over its full length in small: 5000 ticks (10 s, 18 steps) of the bench's walking workload,
    TSID + sim, never a failed QP, the CoM on its LIPM reference and the feet on their footsteps."""
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    n = 64
    wc = make(n, walking=True, reference_quirks=False)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf),
                                         com0=wc.com_ref[0, :3].cpu().numpy())
    bad = torch.zeros(n, dtype=torch.bool, device=wc.device)
    worst_com = 0.0
    for i in range(5000):
        sched.apply(wc, i * wc.conf.dt)
        wc.step()
        bad |= wc.status != 0
        if i % 50 == 0:
            worst_com = max(worst_com, float((wc.obs[:, 53:55] - wc.com_ref[:, :2]).abs().max()))
    assert not bool(bad.any())
    assert worst_com < 0.01                                     # CoM within 1 cm of the LIPM reference
    assert float(wc.q[:, 2].min()) > 0.28 and float(wc.q[:, 2].max()) < 0.34   # nobody fell (base height)
    travelled = (wc.q[:, :2] - torch.as_tensor(0.5 * (lf + rf), device=wc.device)).norm(dim=1)
    assert float(travelled.min()) > 0.5                         # 18 steps of 5 cm
    assert int(wc.ncon.min()) >= 1 and int(wc.ncon.max()) <= 32  # the slave sim keeps its feet on the floor


def test_infeasible_envs_are_flagged_not_fatal(oracle):
    wc = make(6)
    wc.conf.fMin, wc.conf.fMax = 500.0, 400.0
    wc.set_params()
    q_before = wc.q.clone()
    wc.step()
    assert wc.status.cpu().numpy().tolist() == [1] * 6            # tsid HQP_STATUS_INFEASIBLE
    assert torch.equal(wc.q, q_before)                            # a failed env does not integrate
    wc.conf.fMin, wc.conf.fMax = 10.0, 1000.0
    wc.set_params()
    wc.step()
    assert wc.status.cpu().numpy().tolist() == [0] * 6


def test_randomised_envs_config5_f64(oracle):
    """BASELINE config 5 in small: per-env mass scale, contact friction and tilted floor (contact-solver
    stress); sim driven by the TSID loop for 40 steps, plus a settle run without the teleport."""
    n = 32
    wc = make(n)
    perturb(wc, 13)
    wc.randomize(seed=2, step_height=0.0)   # the plane-only half; terrain steps: test_rough_terrain_config5_f64
    st = _mirror_env(wc, mirror(wc))
    for i in range(40):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
        assert np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i
        assert diff(wc.qpos, st["qpos"]) < 1e-9 and diff(wc.qvel, st["qvel"]) < 1e-6, i
    assert int(wc.ncon.max()) > 0
    # settle on the tilted floors without the teleport
    wc2 = make(8)
    wc2.randomize(seed=3, tilt_deg=5.0, step_height=0.0)
    wc2.qpos[:, 3:7] = torch.tensor([1.0, 0, 0, 0], dtype=wc2.dtype, device=wc2.device)
    wc2.qpos[:, 2] -= 0.002
    ep = wc2.env_params.cpu().numpy()
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc2.qpos, wc2.qvel, wc2.qacc_warmstart))
    for i in range(150):
        wc2.sim_step(teleport=False)
        for e in range(8):
            oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e], envp=ep[e])
    assert diff(wc2.qpos, qpos) < 1e-7 and diff(wc2.qvel, qvel) < 1e-4
    assert int(wc2.ncon.min()) >= 1
    wc2.set_env_params()   # back to the nominal model
    assert wc2.env_params is None


def test_closed_loop_matches_oracle_and_stands(oracle):
    """TSID reads the sim state, the sim is driven by tau.  120 steps vs the oracle, and the
    closed-loop robot keeps standing on its contacts."""
    n = 16
    wc = make(n, closed_loop=True)
    assert float((wc.qpos[:, 3] - 1.0).abs().max()) == 0          # reset wrote a proper wxyz quaternion
    g = torch.Generator().manual_seed(5)
    wc.qpos[:, 7:] += ((torch.rand(n, 20, generator=g, dtype=torch.float64) - 0.5) * 0.02).to(wc.device)
    st = mirror(wc)
    for i in range(120):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
        assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
        assert diff(wc.tau, st["tau"]) < 1e-6 and diff(wc.qpos, st["qpos"]) < 1e-8 and diff(wc.qvel, st["qvel"]) < 1e-5, i
    assert int(wc.status.abs().sum()) == 0 and int(wc.ncon.min()) >= 2
    assert 0.325 < float(wc.qpos[:, 2].min()) and float(wc.qpos[:, 2].max()) < 0.335
    assert float(wc.qvel.abs().max()) < 0.5


def test_shard_invariance_bitwise():
    """1 vs 2 shards give bit-identical per-env results (what 1/2/4/8 GPUs must reproduce)."""
    full = make(32)
    perturb(full, 7)
    a, b = make(16), make(16)
    a.q.copy_(full.q[:16]); a.v.copy_(full.v[:16]); b.q.copy_(full.q[16:]); b.v.copy_(full.v[16:])
    for _ in range(10):
        full.step(); a.step(); b.step()
    for name in ("q", "v", "qpos", "qvel", "tau", "obs"):
        assert torch.equal(getattr(full, name), torch.cat([getattr(a, name), getattr(b, name)])), name


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_shard_invariance_across_kernel_shapes(dtype):
    """a strong split of one batch over 1 / 2 / 8 GPUs crosses the thresholds at which the library changes the
    shape of the sim kernel - one step per launch above 1024 envs, eight per launch (a separately compiled instantiation) up
    to 1024, two wavefronts per env up to 512 - all with the DEFAULT options and the pipelined step the bench uses.  Walkers
    (start phase, lift-off, a touch-down: 800 ticks) on one controller of 2048 envs against 2 x 1024 and against the first and
    the last of 8 x 256: bit-identical, in float32 as well (the library is built with -ffp-contract=on: every instantiation
    rounds alike by construction)."""
    n, dt, ticks = 2048, 0.002, 800
    g = torch.Generator(device="cpu").manual_seed(17)
    scale = 0.5 + 0.47 * torch.rand(n, generator=g, dtype=torch.float64)
    def run(lo, hi):
        wc = make(hi - lo, dtype, walking=True, reference_quirks=False)
        from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
        wc.set_posture_bias(op3_walking_posture())
        sched = WalkSchedule.on_device(wc, scale=scale[lo:hi])
        for i in range(ticks):
            wc.step_pipelined(walk=(sched, i * dt))
        wc.sync_sim()
        torch.cuda.synchronize()
        return wc
    full = run(0, n)
    import ctypes as C
    opt = lambda w, o: (lambda v: (w._L.tsidb_get_option(w._h, o, C.byref(v)), v.value)[1])(C.c_int(0))
    assert opt(full, 1) == 1 and full.sim_batch == 1
    assert int(full.status.abs().sum()) == 0 and int((full.contact_active.sum(dim=1) == 1).sum()) > n // 2   # walking, single support
    for lo, hi in ((0, 1024), (1024, 2048), (0, 256), (1792, 2048)):
        part = run(lo, hi)
        assert part.sim_batch == 8 and opt(part, 1) == (2 if hi - lo <= 512 else 1)
        for k in ("q", "v", "tau", "dv", "f", "status", "rows", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "contact_active"):
            assert torch.equal(getattr(full, k)[lo:hi], getattr(part, k)), (dtype, lo, hi, k)
        del part


def test_f32_three_wavefront_sim_build_is_bit_identical():
    """float32, 3072 envs and more, open loop: tsidb_sim runs the build of the sim kernel that is register-allocated for three
    wavefronts per SIMD (so that the next tick runs beside it; +24 % pipelined at 4096 walkers).  Another register allocation
    of the same arithmetic: a 512-env slice, which gets the two-wavefront build, is bit-identical."""
    n = 3200
    g = torch.Generator(device="cpu").manual_seed(29)
    dq = ((torch.rand(n, 20, generator=g, dtype=torch.float64) - 0.5) * 0.08)
    dv = torch.randn(n, 26, generator=g, dtype=torch.float64) * 0.05
    def run(lo, hi):
        wc = make(hi - lo, "f32")
        wc.q[:, 7:] += dq[lo:hi].to(wc.device, wc.dtype)
        wc.v[:] = dv[lo:hi].to(wc.device, wc.dtype)
        for _ in range(40):
            wc.step()
        torch.cuda.synchronize()
        return wc
    full, part = run(0, n), run(1024, 1536)
    for k in ("q", "v", "tau", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info"):
        assert torch.equal(getattr(full, k)[1024:1536], getattr(part, k)), k
    assert int(full.ncon.max()) >= 8 and int(full.status.abs().sum()) == 0


def test_fast_equality_solve_agrees_with_the_qr_path(oracle):
    """conf.qp_fast_equalities (TSIDB_OPT_QP_FAST_EQ, on by default): the tick first takes the equality-constrained optimum by
    a Cholesky of B^T B and sweeps the inequalities there; only envs with a violated inequality go on to the QR + active set.
    Against the same envs with the switch off (always the QR): statuses and iteration counts identical, solution to rounding -
    standing and single support, perturbed states, envs WITH active inequalities (tight torque bounds) mixed in; and against
    the oracle at the tolerances of the env-loop test."""
    n = 256
    mk = lambda fe: make(n, qp_fast_equalities=fe, tau_max_scaling=0.35)
    a, b = mk(1), mk(0)
    for w in (a, b):
        perturb(w, 13, dq=0.06, dv=0.08)
        w.contact_active[::3, 0] = 0          # a third of the envs in single support (38 variables)
    st = mirror(a)
    worst = 0.0
    for i in range(12):
        a.tick()
        b.tick()
        torch.cuda.synchronize()
        assert torch.equal(a.status, b.status) and torch.equal(a.info[:, :2], b.info[:, :2]), i
        for k in ("tau", "dv", "q", "v"):
            worst = max(worst, float((getattr(a, k) - getattr(b, k)).abs().max()))
        w0, w1 = wrench(a.f.cpu().numpy(), a.params), wrench(b.f.cpu().numpy(), b.params)
        worst = max(worst, float(np.abs(w0 - w1).max()))
        b.q.copy_(a.q); b.v.copy_(a.v)      # (same inputs every tick: the comparison is per tick)
    assert worst < 1e-9, worst
    assert int((a.info[:, 0] > 1).sum()) > n // 20 and int((a.info[:, 0] == 1).sum()) > n // 4   # both kinds of env were there
    for i in range(12):
        oracle.env_step_batch(a.params, st, nthreads=8)   # (a was ticked only: compare its TSID side)
    # the oracle ran tick + sim; the TSID state does not depend on the sim (open loop)
    assert np.array_equal(a.status.cpu().numpy(), st["status"])
    assert diff(a.tau, st["tau"]) < 1e-7 and diff(a.dv, st["dv"]) < 1e-7 and diff(a.q, st["q"]) < 1e-9 and diff(a.v, st["v"]) < 1e-9


def test_full_size_properties():
    """BASELINE config sizes: 4096 envs (one GPU) - size-independent properties instead of the oracle."""
    wc = make(4096)
    for _ in range(5):
        wc.step()
    assert int(wc.status.abs().sum()) == 0
    # identical envs stay bit-identical; standing: sum f_z = m g, zero acceleration, CoP between the feet
    for name in ("q", "tau", "f", "qpos"):
        t = getattr(wc, name)
        assert torch.equal(t, t[0:1].expand_as(t)), name
    fz = wc.f.reshape(4096, 8, 3)[:, :, 2].sum(dim=1)
    assert float((fz - 2.893639 * 9.81).abs().max()) < 1e-3
    assert float(wc.dv.abs().max()) < 1e-3
    assert torch.isfinite(wc.obs).all()
    perturb(wc, 11)
    for _ in range(20):
        wc.step()
    assert int(wc.status.abs().sum()) == 0 and torch.isfinite(wc.q).all() and torch.isfinite(wc.qvel).all()
    quat = wc.q[:, 3:7]
    assert float((quat.norm(dim=1) - 1).abs().max()) < 1e-12


def test_api_error_behaviour():
    import ctypes as C
    from tsid_control_amd import _lib
    from tsid_control_amd.model import ModelBlob
    from tsid_control_amd.params import P_COUNT
    L = _lib.load()
    h = C.c_void_p()
    p = np.zeros(P_COUNT)
    rc = L.tsidb_create(b"not a blob at all....", 21, p.ctypes.data_as(C.c_void_p), P_COUNT, 4, 0, 0, C.byref(h))
    assert rc != 0 and b"TSIDBM01" in L.tsidb_last_error(h)
    L.tsidb_destroy(h)
    wc = make(2)
    # truncated / corrupt blobs are rejected before anything is read past the buffer
    raw = wc.model.raw
    for cut in (20, 16 + 40 * 3, len(raw) // 2, len(raw) - 8):
        rc = L.tsidb_create(raw[:cut], cut, wc.params.ctypes.data_as(C.c_void_p), P_COUNT, 4, 0, 0, C.byref(h))
        assert rc != 0 and b"model blob" in L.tsidb_last_error(h), cut
        L.tsidb_destroy(h)
    bad = bytearray(raw)
    bad[16 + 24 + 4:16 + 24 + 8] = (0x7fffffff).to_bytes(4, "little")  # first section's count
    rc = L.tsidb_create(bytes(bad), len(bad), wc.params.ctypes.data_as(C.c_void_p), P_COUNT, 4, 0, 0, C.byref(h))
    assert rc != 0 and b"bad section" in L.tsidb_last_error(h)
    L.tsidb_destroy(h)
    # address tables that would index device memory out of range or alias the contact-id flag bit
    import struct
    def patched(section, index, value):
        b = bytearray(raw)
        nsec = struct.unpack_from("<I", raw, 8)[0]
        for i in range(nsec):
            off = 16 + 40 * i
            if raw[off:off + 24].split(b"\0")[0].decode() == section:
                o = struct.unpack_from("<Q", raw, off + 32)[0]
                struct.pack_into("<i", b, o + 4 * index, value)
                return bytes(b)
        raise KeyError(section)
    nvert = len(wc.model["mj_hull_vert"]) // 3
    for sec, idx, val, msg in (("mj_hull_adr", 3, 10 ** 6, b"hull"), ("mj_hull_adr", 21, nvert + 64, b"span"), ("mj_chunk_adr", 2, 0, b"increasing"),
                               ("mj_hull_eadr", 5, 10 ** 7, b"hull graph"), ("mj_hull_edge", 0, 40000, b"leaves its hull")):
        blob = patched(sec, idx, val)
        rc = L.tsidb_create(blob, len(blob), wc.params.ctypes.data_as(C.c_void_p), P_COUNT, 4, 0, 0, C.byref(h))
        assert rc != 0 and msg in L.tsidb_last_error(h), (sec, L.tsidb_last_error(h))
        L.tsidb_destroy(h)
    rc = L.tsidb_create(wc.model.raw, len(wc.model.raw), wc.params.ctypes.data_as(C.c_void_p), P_COUNT, 4, 99, 0, C.byref(h))
    assert rc != 0 and b"device" in L.tsidb_last_error(h)
    L.tsidb_destroy(h)
    rc = L.tsidb_create(wc.model.raw, len(wc.model.raw), wc.params.ctypes.data_as(C.c_void_p), P_COUNT, 4, 0, 0, C.byref(h))
    assert rc == 0
    rc = L.tsidb_tick(h, None, None, None, None, None, None, None, 65, None, None, None)
    assert rc != 0 and b"tsidb_set_refs" in L.tsidb_last_error(h)   # references never registered
    L.tsidb_destroy(h)


def test_odd_batch_sizes_and_partial_reset(oracle):
    """num_envs = 1, 3, 65 (no multiple of anything); reset(env_ids) touches only those envs; an empty id
    list is a no-op."""
    for n in (1, 3, 65):
        wc = make(n)
        perturb(wc, 100 + n)
        st = mirror(wc)
        for _ in range(3):
            wc.step()
            oracle.env_step_batch(wc.params, st, nthreads=4)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"])
        assert diff(wc.q, st["q"]) < 1e-9 and diff(wc.qpos, st["qpos"]) < 1e-9 and diff(wc.tau, st["tau"]) < 1e-7
        assert np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"])
    wc = make(6)
    q_init, qpos_init = wc.q.clone(), wc.qpos.clone()
    perturb(wc, 3)
    for _ in range(4):
        wc.step()
    moved = wc.q.clone()
    wc.reset(env_ids=[])
    assert torch.equal(wc.q, moved)
    wc.reset(env_ids=[1, 4])
    for e in range(6):
        if e in (1, 4):
            assert torch.equal(wc.q[e], q_init[e]) and torch.equal(wc.qpos[e], qpos_init[e]) and float(wc.v[e].abs().max()) == 0
        else:
            assert torch.equal(wc.q[e], moved[e])


def test_non_finite_inputs_are_contained(oracle):
    """A NaN / Inf state or reference in one env: that env reports HQP_STATUS_ERROR (4) and is left as it
    was, its neighbours are unaffected (same results as a batch without the poisoned envs); the oracle
    restates the same guard."""
    n = 8
    wc, ref = make(n), make(n)
    perturb(wc, 77); perturb(ref, 77)
    st = mirror(wc)
    wc.step(); ref.step()                       # a good tick first: tau, dv, f of every env are non-zero now
    oracle.env_step_batch(wc.params, st, nthreads=2)
    assert float(wc.tau[2].abs().max()) > 0 and float(wc.f[5].abs().max()) > 0
    wc.q[2, 9] = float("nan")
    wc.com_ref[5, 1] = float("inf")
    st["q"][2, 9] = np.nan
    st["com_ref"][5, 1] = np.inf
    q_before, qpos_before = wc.q.clone(), wc.qpos.clone()
    wc.step(); ref.step()
    oracle.env_step_batch(wc.params, st, nthreads=2)
    s = wc.status.cpu().numpy()
    assert s[2] == 4 and s[5] == 4 and (np.delete(s, [2, 5]) == 0).all()
    assert np.array_equal(s & 0xff, st["status"] & 0xff)
    # nothing of the previous tick is handed on: tau = dv = f = 0, on the device as in the oracle
    for e in (2, 5):
        assert float(wc.tau[e].abs().max()) == 0 and float(wc.dv[e].abs().max()) == 0 and float(wc.f[e].abs().max()) == 0
        assert np.abs(st["tau"][e]).max() == 0 and np.abs(st["dv"][e]).max() == 0 and np.abs(st["f"][e]).max() == 0
    good = [0, 1, 3, 4, 6, 7]
    assert torch.equal(wc.q[good], ref.q[good]) and torch.equal(wc.tau[good], ref.tau[good]) and torch.equal(wc.qpos[good], ref.qpos[good])
    # poisoned envs: TSID state untouched; the sim of env 2 sees a NaN target and skips its step (info[3] bit 4)
    assert torch.equal(wc.q[5], q_before[5]) and torch.isnan(wc.q[2, 9]) and torch.equal(wc.q[2, :9], q_before[2, :9])
    assert int(wc.info[2, 3]) == 4 and torch.equal(wc.qpos[2], qpos_before[2])
    assert diff(wc.qpos[good], st["qpos"][good]) < 1e-9


def test_non_finite_reference_goes_limp_in_the_closed_loop():
    """closed loop: an env with a finite sim state but a NaN reference is flagged every tick and its motors get
    tau = 0 (not the last good tick's torques) while its sim keeps stepping"""
    wc = make(4, closed_loop=True, reference_quirks=False)
    perturb(wc, 78)
    for _ in range(3):
        wc.step()
    assert float(wc.tau[1].abs().max()) > 0
    wc.com_ref[1, 0] = float("nan")
    qpos0 = wc.qpos[1].clone()
    for _ in range(5):
        wc.step()
        assert int(wc.status[1]) == 4 and float(wc.tau[1].abs().max()) == 0
    assert not torch.equal(wc.qpos[1], qpos0) and int(wc.info[1, 3]) == 0    # its sim keeps stepping (limp), unflagged
    assert int((wc.status[[0, 2, 3]] != 0).sum()) == 0


@pytest.mark.parametrize("n", [1, 7, 13, 67, 257])
def test_every_env_is_stepped_for_any_batch_size(oracle, n):
    """the workgroup -> env table gives each of the 8 XCDs a contiguous env range (csrc/tsidb_common.hpp: env_of_block);
    batch sizes that are not multiples of 8 - and smaller than 8 - must still step every env exactly once"""
    wc = make(n)
    perturb(wc, 90 + n)
    st = mirror(wc)
    for _ in range(3):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
    assert np.array_equal(wc.status.cpu().numpy(), st["status"]) and int((wc.status != 0).sum()) == 0
    assert diff(wc.tau, st["tau"]) < 1e-7 and diff(wc.q, st["q"]) < 1e-10 and diff(wc.qpos, st["qpos"]) < 1e-9
    assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]) and np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"])


def test_diverged_sim_state_is_contained(oracle):
    """A finite but absurd sim state (the reference's own standing loop drives its teleported sim to 1e150 within 700
    ticks; products of such values overflow to inf / NaN inside the step, and a NaN placement used to index the hull
    arrays out of bounds - a GPU memory fault): the env is skipped and flagged (info[3] bit 4) like a non-finite one,
    on the device as in the oracle; states just inside the bound are stepped and agree."""
    n = 8
    wc = make(n)
    wc.qpos[:, 3:7] = torch.tensor([1.0, 0, 0, 0], dtype=wc.dtype, device=wc.device)
    wc.qvel[1, 9] = 1e150
    wc.qvel[2, 2] = -2e6
    wc.qpos[3, 0] = 1e200
    wc.qvel[4, 2] = -9e5          # inside the bound
    wc.qvel[5, 12] = 3e4          # a joint spinning absurdly fast, inside the bound
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc.qpos, wc.qvel, wc.qacc_warmstart))
    before = (qpos.copy(), qvel.copy())
    wc.sim_step(teleport=False)
    info = wc.info.cpu().numpy()
    for e in range(n):
        r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e])
        assert (r["rc"] == 4) == (e in (1, 2, 3)) == bool(info[e, 3] & 4), e
    for e in (1, 2, 3):
        assert np.array_equal(wc.qpos[e].cpu().numpy(), before[0][e]) and np.array_equal(wc.qvel[e].cpu().numpy(), before[1][e])
        assert int(wc.ncon[e]) == 0
    ok = [0, 4, 5, 6, 7]
    assert bool(torch.isfinite(wc.qpos[ok]).all())
    assert diff(wc.qpos[ok], qpos[ok]) < 1e-6 * max(1.0, float(np.abs(qpos[ok]).max())) and diff(wc.qvel[ok], qvel[ok]) < 1e-6 * 9e5


def test_fallen_robot_hits_the_contact_cap(oracle):
    """A robot lying on the floor: more hull vertices touch than the 32-contact cap; the sim keeps the
    first 32 in body order like the oracle (bit-exact pairs) and stays finite."""
    n = 4
    wc = make(n)
    st = mirror(wc)
    # lay the sim robot on its back / side at a few heights (sim state only; no teleport)
    quats = torch.tensor([[0.7071068, 0.7071068, 0, 0], [0.7071068, 0, 0.7071068, 0], [0.5, 0.5, 0.5, 0.5], [0.9238795, 0.3826834, 0, 0]],
                         dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 3:7] = quats
    wc.qpos[:, 2] = torch.tensor([0.05, 0.06, 0.05, 0.12], dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 7:] += 0.3
    qpos, qvel, ws = wc.qpos.cpu().numpy().copy(), wc.qvel.cpu().numpy().copy(), wc.qacc_warmstart.cpu().numpy().copy()
    ncon_max = 0
    for _ in range(20):
        wc.sim_step(teleport=False)
        for e in range(n):
            r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e])
            got = wc.con_pairs[e].cpu().numpy()
            want = np.full(32, -1, dtype=np.int32)
            want[:r["ncon"]] = (r["con_geom"] << 16) | r["con_vert"]
            assert np.array_equal(got, want)
            ncon_max = max(ncon_max, r["ncon"])
        assert diff(wc.qpos, qpos) < 1e-7 and diff(wc.qvel, qvel) < 1e-4
    assert ncon_max == 32 and bool(torch.isfinite(wc.qpos).all())


@pytest.mark.parametrize("batch,dtype", [(1, "f64"), (3, "f64"), (8, "f64"), (3, "f32"), (8, "f32")])
def test_step_pipelined_equals_step(batch, dtype):
    """sim(t) on a second stream overlapping tick(t+1): same results, bit for bit, as the serial step() - also with the
    sim stages enqueued several at a time (conf.pipeline_sim_batch; 25 steps leave one of them for sync_sim to flush).
    (float32: holds on these 25 standing steps; the multi-step sim kernel is a separate compilation and agrees with the
    single-step one to rounding in general - include/tsidb.h, tools/dbg_batch_f32.py.)"""
    a, b = make(96, dtype, reference_quirks=False), make(96, dtype, reference_quirks=False, pipeline_sim_batch=batch)
    perturb(a, 5); perturb(b, 5)
    for _ in range(25):
        a.step()
        b.step_pipelined()
    b.sync_sim()
    torch.cuda.synchronize()
    for k in ("q", "v", "tau", "dv", "f", "status", "obs", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    c = make(4, closed_loop=True)
    with pytest.raises(Exception, match="step_pipelined"):
        c.step_pipelined()


def test_pipeline_on_the_library_streams():
    """WalkController.tick_stream / the sim stream from tsidb_stream_create (for up to 512 envs: disjoint halves of the CUs,
    hipExtStreamCreateWithCUMask): the pipelined loop on them gives the serial step()'s results bit for bit; the option
    TSIDB_OPT_CU_SPLIT switches the split off."""
    a, b, c = make(96, reference_quirks=False), make(96, reference_quirks=False), make(96, reference_quirks=False)
    from tsid_control_amd import _lib
    _lib.check(c._L, c._h, c._L.tsidb_set_option(c._h, 3, 0), "tsidb_set_option(cu_split)")   # plain streams
    for w in (a, b, c):
        perturb(w, 5)
    for w in (b, c):
        with torch.cuda.stream(w.tick_stream):
            for _ in range(25):
                w.step_pipelined()
            w.sync_sim()
        w.tick_stream.synchronize()
    for _ in range(25):
        a.step()
    torch.cuda.synchronize()
    assert b.tick_stream.cuda_stream != b._pipe["stream"].cuda_stream
    for k in ("q", "v", "tau", "dv", "f", "status", "obs", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info"):
        assert torch.equal(getattr(a, k), getattr(b, k)) and torch.equal(getattr(a, k), getattr(c, k)), k
    with pytest.raises(Exception, match="role"):
        _lib.check(a._L, a._h, a._L.tsidb_stream_create(a._h, 7, __import__("ctypes").byref(__import__("ctypes").c_void_p())), "tsidb_stream_create")


def test_step_pipelined_mixed_with_reset_and_env_params():
    """reset() / step() / sim_step() / set_env_params() wait for the sim stage that
    step_pipelined() left on the side stream - mixing the entry points (the RL pattern: partial reset
    between pipelined steps) gives the serial path's results bit for bit."""
    a, b = make(96, reference_quirks=False), make(96, reference_quirks=False)
    perturb(a, 7); perturb(b, 7)
    ids = [3, 17, 64, 95]
    fr = torch.linspace(0.5, 1.0, 96, dtype=torch.float64)
    for k in range(30):
        a.step()
        b.step_pipelined()
        if k % 7 == 6:
            a.reset(env_ids=ids)
            b.reset(env_ids=ids)
        if k == 12:
            a.set_env_params(friction=fr)
            b.set_env_params(friction=fr)
        if k == 20:
            a.step()
            b.step()  # a serial step right after a pipelined one
    b.sync_sim()
    torch.cuda.synchronize()
    for k in ("q", "v", "tau", "dv", "f", "status", "obs", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_set_params_in_the_middle_of_a_sim_batch():
    """with the sim stages enqueued four at a time, up to three of them are not even launched when set_params()
    is called; they must run with the constants of THEIR step (set_params flushes them first), so the pipelined path
    stays bit-identical to step() across a RobotConfig edit (dt, self_collision, frictionloss scale)"""
    a, b = make(64, reference_quirks=False), make(64, reference_quirks=False, pipeline_sim_batch=4)
    perturb(a, 9); perturb(b, 9)
    for k in range(22):
        a.step()
        b.step_pipelined()
        if k == 9:        # 10 steps done: 8 sims flushed, 2 pending
            for w in (a, b):
                w.conf.dt = 0.003
                w.conf.self_collision = False
                w.conf.sim_frictionloss_scale = 0.5
                w.set_params()
    b.sync_sim()
    torch.cuda.synchronize()
    for k in ("q", "v", "tau", "dv", "f", "status", "obs", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


@pytest.mark.parametrize("dephase", [False, True])
def test_walk_update_kernel_matches_oracle_restatement(oracle, dephase):
    """f-2's independent check: tsidb_walk_update (k_walk) against oracle/or_walk.c - a C restatement of
    the same reference semantics (Foot_Trajectory.py:21-27 polynomials, Walk_Planner.py:23-31 swings,
    WalkController.py:189-253 contact edges, LIPM.py:34-49 in closed form) that shares no code with the
    kernel or with WalkSchedule.sample().  Contact flags bit-exact, references to 1e-12, over the start
    phase, every contact edge of the first steps and the final stand; also with per-env start delays."""
    from oracle.oracle import walk_update
    from tsid_control_amd.walk_planner import WalkSchedule
    n = 24
    wc = make(n, sim_enabled=False, walking=True)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=5, q0_feet=(lf, rf), t_start=0.25)
    if dephase:
        sched.set_phase_offsets(torch.linspace(0.0, 0.9, n, dtype=torch.float64))
    npy = lambda x: np.ascontiguousarray(x.cpu().numpy())
    ticks = list(range(0, 900, 3)) + list(range(5000, 5400, 9))   # the second block is past every env's last step
    for i in ticks:
        t = i * wc.conf.dt
        fr, cr, ca, cm = npy(wc.foot_ref), npy(wc.contact_ref), npy(wc.contact_active), npy(wc.com_ref)
        walk_update(oracle.lib, sched, t, npy(wc.frames), fr, cr, ca, cm)
        sched.apply(wc, t)
        assert np.array_equal(npy(wc.contact_active), ca), i
        assert diff(wc.foot_ref, fr) < 1e-12 and diff(wc.contact_ref, cr) < 1e-12 and diff(wc.com_ref, cm) < 1e-12, i
        wc.tick()   # moves the feet, so the next edge re-references at a new placement
    assert int((wc.status != 0).sum()) == 0


def test_golden_foot_trajectories_through_the_device_kernel():
    """Pins the DEVICE evaluation of a12 to the reference itself: the five FootTrajectory cases of
    tests/golden/planners.json (generated by importing /root/reference/ctrl/Foot_Trajectory.py, see
    make_planner_golden.py) go through tsidb_walk_update's polynomial path as one-step schedules; the
    foot-task reference the kernel writes equals the reference's get_position() to 1e-11, and its
    acceleration slot equals the reference's get_velocity() (which is the 2nd derivative, quirk F6f)."""
    import json
    from pathlib import Path
    from tsid_control_amd.foot_trajectory import FootTrajectory
    from tsid_control_amd.walk_planner import WalkSchedule
    gold = json.loads((Path(__file__).parent / "golden" / "planners.json").read_text())["foot_traj"]
    for case in gold:
        p = case["params"]
        t0, t1 = p["t"]
        wc = make(1, sim_enabled=False)
        start, target = list(map(float, p["start"])), list(map(float, p["target"]))
        tr = FootTrajectory([t0, t1], start, target, p["h"], p["rise"])
        # a hand-made one-step table: step 0 swings the left foot along this trajectory
        sched = WalkSchedule.__new__(WalkSchedule)
        sched.conf = type("C", (), {"step_duration": t1 - t0})()
        sched.N, sched.K, sched.device, sched.dtype, sched.t_offset = 1, 1, wc.device, wc.dtype, None
        sched.td_latch, sched.td_fraction = None, 0.6
        sched.t_start, sched.omega, sched.z0, sched.dz = 0.0, 3.0, 0.24, 0.0
        dev = lambda a, dt=wc.dtype: torch.as_tensor(np.asarray(a), device=wc.device).to(dt)
        sched.coef = dev(tr.coefficients()[None, None])
        sched.side, sched.nsteps = dev([[0]], torch.long), dev([1], torch.long)
        sched.rest = dev(np.zeros((1, 2, 2, 4)))
        sched.com = dev(np.zeros((1, 3, 2, 3)))
        for i, t in enumerate(case["ts"]):
            if t >= t1:
                continue   # the step is over at t1: the foot is back on its rest pose
            sched.apply(wc, t - t0)
            fr = wc.foot_ref[0, 0].cpu().numpy()
            if i == 0:
                continue   # the tick that lifts the foot re-references at the current placement (remove_contact)
            assert np.allclose(fr[:3], case["pos"][i], atol=1e-11), (p, t)
            assert np.allclose(fr[18:21], case["vel"][i], atol=1e-8), (p, t)   # reference "velocity" = 2nd derivative
            if "yaw" in case:
                assert abs(np.arctan2(fr[4], fr[3]) - case["yaw"][i]) < 1e-11


def test_walking_env_loop_matches_oracle_walk(oracle):
    """Config 3 in small, end to end against the oracle's OWN walking update (or_walk.c inside
    or_env_step_batch_walk): k_walk + k_tick + k_sim on the device, walk update + tick + sim step on the CPU,
    each side producing its own references from the same tables - start phase, first lift-off, several
    single-support steps with touch-down / lift-off edges; de-phased so that 50- and 38-variable QPs mix."""
    from oracle.oracle import WalkTables
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    n = 12
    wc = make(n, walking=True, reference_quirks=False)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=4, q0_feet=(lf, rf),
                                         com0=wc.com_ref[0, :3].cpu().numpy(), t_start=0.4)
    sched.set_phase_offsets(torch.linspace(0.0, 0.5, n, dtype=torch.float64))
    st = mirror(wc)
    st["frames"] = wc.frames.cpu().numpy().copy()
    tables = WalkTables(sched)
    mixed = 0
    for i in range(900):
        t = i * wc.conf.dt
        sched.apply(wc, t)
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8, walk=tables.at(t))
        assert np.array_equal(wc.contact_active.cpu().numpy(), st["contact_active"]), i
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
        ns = wc.contact_active.sum(dim=1)
        mixed += int(ns.min() != ns.max())
        if i % 25 == 0 or i == 899:
            assert diff(wc.tau, st["tau"]) < 1e-6 and diff(wc.dv, st["dv"]) < 1e-5, i
            assert diff(wc.q, st["q"]) < 1e-8 and diff(wc.v, st["v"]) < 1e-7, i
            assert diff(wc.com_ref, st["com_ref"]) < 1e-12 and diff(wc.foot_ref, st["foot_ref"]) < 1e-8, i
            assert diff(wc.qpos, st["qpos"]) < 1e-8, i
            assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
    assert mixed > 100 and int((wc.status != 0).sum()) == 0


def test_wide_batch_walking_parity(oracle):
    """the walking env loop against the oracle's on a WIDE batch: 1024 de-phased walkers (random start delays, so that
    start phase, lift-offs, touch-downs and single support are all present at every tick) through 700 ticks - statuses,
    contact flags and contact lists bit-exact on every env, states to the tolerances of the 12-env test"""
    from oracle.oracle import WalkTables
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    n = 1024
    wc = make(n, walking=True, reference_quirks=False)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=9, q0_feet=(lf, rf),
                                         com0=wc.com_ref[0, :3].cpu().numpy(), t_start=0.4)
    g = torch.Generator().manual_seed(21)
    sched.set_phase_offsets(torch.rand(n, generator=g, dtype=torch.float64) * 0.9)
    st = mirror(wc)
    st["frames"] = wc.frames.cpu().numpy().copy()
    tables = WalkTables(sched)
    mixed = 0
    for i in range(700):
        t = i * wc.conf.dt
        sched.apply(wc, t)
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=16, walk=tables.at(t))
        if i % 50 == 49 or i == 699:
            ns = wc.contact_active.sum(dim=1)
            mixed += int(ns.min() != ns.max())
            assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
            assert np.array_equal(wc.contact_active.cpu().numpy(), st["contact_active"]), i
            assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]) and np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i
            assert diff(wc.tau, st["tau"]) < 1e-6 and diff(wc.q, st["q"]) < 1e-8 and diff(wc.qpos, st["qpos"]) < 1e-8, i
    assert mixed >= 5 and int((wc.status != 0).sum()) == 0   # 50- and 38-variable QPs side by side in most of the checks


def test_reward_and_done_outputs(oracle):
    """reward[N], done[N] (SURVEY.md 8d write list; no reference counterpart) ride in columns 65, 66 of the
    per-env row: tracking reward minus torque cost; done = failed QP / non-finite input / base too low / tilted."""
    n = 8
    wc = make(n, sim_enabled=False)
    perturb(wc, 11)
    wc.q[1, 2] = 0.15                                        # base below done_base_height (0.2 m)
    ang = np.deg2rad(60.0)                                   # tilted by 60 deg about x (limit 45 deg)
    wc.q[2, 3:7] = torch.tensor([np.sin(ang / 2), 0.0, 0.0, np.cos(ang / 2)], dtype=wc.dtype, device=wc.device)
    wc.q[3, 9] = float("nan")
    st = mirror(wc)
    wc.step()
    oracle.env_step_batch(wc.params, st, nthreads=4)
    rd = wc.rows[:, 65:].cpu().numpy()
    assert np.array_equal(wc.status.cpu().numpy(), st["status"])
    assert np.allclose(rd, st["rewdone"], atol=1e-9)
    assert rd[1, 1] == 1 and rd[2, 1] == 1 and rd[3, 1] == 1 and rd[3, 0] == 0
    ok = [0, 4, 5, 6, 7]
    assert np.all(rd[ok, 1] == 0) and np.all(rd[ok, 0] > 0.5) and np.all(rd[ok, 0] < 1.0)
    assert torch.equal(wc.reward, wc.rows[:, 65]) and torch.equal(wc.done, wc.rows[:, 66])
    assert wc.gather_rows().shape == (n, 67) and wc.gather_rows().is_contiguous()


def _mirror_env(wc, st):
    if wc.env_params is not None:
        st["env_params"] = wc.env_params.double().cpu().numpy().copy()
    if wc.terrain is not None:
        st["terrain"] = wc.terrain.double().cpu().numpy().copy()
    return st


def test_rough_terrain_config5_f64(oracle):
    """The rough-terrain half of BASELINE configs[4]: per-env stepped floor (1 cm steps over a tilted plane) on top
    of the mass / friction randomisation.  TSID-driven sim for 60 steps - contact (body, vertex) pairs bit-exact,
    state to 1e-9 - then a free settle onto the steps."""
    n = 32
    wc = make(n)
    perturb(wc, 17)
    wc.randomize(seed=5, step_length=(0.02, 0.06))    # narrow strips: every sole straddles a step edge
    assert wc.terrain is not None and float(wc.terrain[:, 4:].max()) == 0.01
    st = _mirror_env(wc, mirror(wc))
    seen = set()
    for i in range(60):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
        assert np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i
        assert diff(wc.qpos, st["qpos"]) < 1e-9 and diff(wc.qvel, st["qvel"]) < 1e-6, i
        seen.update(np.unique(st["ncon"]).tolist())
    assert max(seen) >= 4 and len(seen) >= 3
    # the same robots dropped onto their terrain, no teleport
    wc2 = make(8)
    wc2.randomize(seed=6, step_length=(0.02, 0.06))
    wc2.qpos[:, 3:7] = torch.tensor([1.0, 0, 0, 0], dtype=wc2.dtype, device=wc2.device)
    wc2.qpos[:, 2] += 0.004
    ep, tr = wc2.env_params.cpu().numpy(), wc2.terrain.cpu().numpy()
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc2.qpos, wc2.qvel, wc2.qacc_warmstart))
    for i in range(120):
        wc2.sim_step(teleport=False)
        for e in range(8):
            r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e], envp=ep[e], terrain=tr[e])
            want = np.full(32, -1, dtype=np.int32)
            want[:r["ncon"]] = (r["con_geom"] << 16) | r["con_vert"]
            assert np.array_equal(wc2.con_pairs[e].cpu().numpy(), want), (i, e)
    assert diff(wc2.qpos, qpos) < 1e-7 and diff(wc2.qvel, qvel) < 1e-4
    assert int(wc2.ncon.min()) >= 1


def _self_collision_poses(n, seed):
    """sim joint angles U(-0.6, 0.6): a few robot<->robot hull pairs penetrate in most of them"""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 20, generator=g, dtype=torch.float64) - 0.5) * 1.2


def test_robot_robot_hull_contacts_f64(oracle):
    """a9's missing piece: the 170 robot<->robot candidate pairs (robot.xml:13-15 after the excludes of :18-52 and the
    parent-child filter) collided as convex hulls (MPR, one contact per pair).  Robots held in the air in
    self-penetrating poses: the (geom1 body, geom2 body) pair list bit-exact against the oracle, contact
    geometry and the state after each step to the tolerance of two independent MPR runs."""
    n = 24
    wc = make(n, self_collision=True)
    wc.qpos[:, 7:] = _self_collision_poses(n, 3).to(wc.device, wc.dtype)
    wc.qpos[:, 3:7] = torch.tensor([1.0, 0, 0, 0], dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 2] = 1.0
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc.qpos, wc.qvel, wc.qacc_warmstart))
    n_hh = 0
    cross = False
    par = wc.model["mj_parent"]
    anc = lambda b: {b} | (anc(int(par[b])) if par[b] >= 0 else set())
    for i in range(12):
        wc.sim_step(teleport=False)
        for e in range(n):
            r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e], self_collision=True)
            want = np.full(32, -1, dtype=np.int32)
            want[:r["ncon"]] = (r["con_geom"] << 16) | r["con_vert"]
            assert np.array_equal(wc.con_pairs[e].cpu().numpy(), want), (i, e)       # pair list bit-exact
            hh = r["con_body1"] >= 0
            n_hh += int(hh.sum())
            cross |= any(a not in anc(int(b)) and b not in anc(int(a)) for a, b in zip(r["con_body1"][hh], r["con_geom"][hh]))
            assert r["flags"] == int(wc.info[e, 3]) & (8 | 16 | 32)
        assert diff(wc.qpos, qpos) < 1e-8 and diff(wc.qvel, qvel) < 1e-5, i
    assert n_hh > 100 and cross     # both the tree-sparse and the dense Newton Hessian paths ran
    assert bool(torch.isfinite(wc.qpos).all())
    # the same poses without self-collision: no contact at all in the air
    wc0 = make(n, self_collision=False)
    wc0.qpos.copy_(torch.as_tensor(qpos, device=wc0.device))
    wc0.qpos[:, 2] = 1.0
    wc0.sim_step(teleport=False)
    assert int(wc0.ncon.max()) == 0


def test_fallen_robot_with_self_collision(oracle):
    """the fallen-robot case: floor contacts up to the cap plus robot<->robot contacts, in the TSID-driven loop"""
    n = 6
    wc = make(n, self_collision=True)
    quats = torch.tensor([[0.7071068, 0.7071068, 0, 0], [0.7071068, 0, 0.7071068, 0], [0.5, 0.5, 0.5, 0.5],
                          [0.9238795, 0.3826834, 0, 0], [0.7071068, -0.7071068, 0, 0], [0.0, 1.0, 0, 0]], dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 3:7] = quats
    wc.qpos[:, 2] = 0.09
    wc.qpos[:, 7:] = _self_collision_poses(n, 9).to(wc.device, wc.dtype)
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc.qpos, wc.qvel, wc.qacc_warmstart))
    both = 0
    for i in range(25):
        wc.sim_step(teleport=False)
        for e in range(n):
            r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e], self_collision=True)
            want = np.full(32, -1, dtype=np.int32)
            want[:r["ncon"]] = (r["con_geom"] << 16) | r["con_vert"]
            assert np.array_equal(wc.con_pairs[e].cpu().numpy(), want), (i, e)
            both += int((r["con_body1"] >= 0).any() and (r["con_body1"] < 0).any())
        assert diff(wc.qpos, qpos) < 1e-7 and diff(wc.qvel, qvel) < 1e-4, i
    assert both > 20 and bool(torch.isfinite(wc.qpos).all())


def test_config3_full_size_walking_properties():
    """BASELINE configs[2] at its own size: 4096 walkers, plans built on the device, the pipelined step
    the bench runs, through the start phase, the first lift-off (tick 500) and the first touch-down / second lift-off
    (tick 750) - size-independent properties: no QP fails at any tick (accumulated on the device every step), no sim env is
    flagged, states finite and unit-norm, contact lists well formed, both contact phases present, and a 512-env slice
    stepped on its own controller is bit-identical (the slice runs another shape of the sim kernel: 8 steps per launch)."""
    n, dt, ticks = 4096, 0.002, 800
    g = torch.Generator(device="cpu").manual_seed(23)
    scale = 0.5 + 0.47 * torch.rand(n, generator=g, dtype=torch.float64)
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    def run(lo, hi, check):
        wc = make(hi - lo, walking=True, reference_quirks=False)
        wc.set_posture_bias(op3_walking_posture())
        sched = WalkSchedule.on_device(wc, scale=scale[lo:hi])
        failed = torch.zeros(hi - lo, dtype=torch.bool, device=wc.device)
        flagged = torch.zeros(hi - lo, dtype=torch.bool, device=wc.device)
        seen = torch.zeros(3, dtype=torch.int64, device=wc.device)
        for i in range(ticks):
            wc.step_pipelined(walk=(sched, i * dt))
            if check:
                failed |= wc.status != 0
                if i % 10 == 9:
                    wc.sync_sim()
                    flagged |= (wc.info[:, 3] & (1 | 2 | 4 | 32)) != 0
                    ns = wc.contact_active.sum(dim=1)
                    seen += torch.stack([(ns == 2).sum(), (ns == 1).sum(), (wc.ncon > 0).sum()])
        wc.sync_sim()
        torch.cuda.synchronize()
        return wc, failed, flagged, seen
    wc, failed, flagged, seen = run(0, n, True)
    assert int(failed.sum()) == 0 and int(flagged.sum()) == 0
    assert int(seen[0]) > 40 * n and int(seen[1]) > 25 * n and int(seen[2]) > 60 * n   # double support, single support, floor contacts
    for t in (wc.q, wc.v, wc.qpos, wc.qvel, wc.tau, wc.rows, wc.f):
        assert bool(torch.isfinite(t).all())
    assert float((wc.q[:, 3:7].norm(dim=1) - 1).abs().max()) < 1e-12 and float((wc.qpos[:, 3:7].norm(dim=1) - 1).abs().max()) < 1e-12
    nc, cp = wc.ncon, wc.con_pairs
    assert int(nc.min()) >= 0 and int(nc.max()) <= 32
    valid = torch.arange(32, device=wc.device)[None, :] < nc[:, None]
    assert bool((cp[~valid] == -1).all()) and bool((cp[valid] >= 0).all())
    assert bool(((cp >> 16)[valid] < 21).all()) and bool((((cp & 0xffff)[valid] < 2823) | ((cp & 0xffff)[valid] >= 0x8000)).all())
    assert int(wc.done.sum()) == 0 and float(wc.q[:, 2].min()) > 0.2 and float((wc.obs[:, 53:56] - wc.com_ref[:, :3]).abs().max()) < 0.02
    assert int((wc.contact_active.sum(dim=1) == 1).sum()) == n    # tick 800: everyone in the second single-support phase
    lo, hi = 2048, 2560
    sub, _, _, _ = run(lo, hi, False)
    assert sub.sim_batch == 8 and wc.sim_batch == 1
    for k in ("q", "v", "tau", "dv", "f", "status", "rows", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "contact_active"):
        assert torch.equal(getattr(wc, k)[lo:hi], getattr(sub, k)), k


def test_walking_f32_against_section5_every_tick(oracle):
    """The float32 path through WALKING against BASELINE.md section 5, whose rows are "one tick / one step
    from identical inputs": every tick the float32 controller is handed the float64 oracle's state and references, both
    produce this tick's walking references from their own tables (float32 on the device, float64 on the host), take one
    env step, and are compared - 620 ticks: start phase, lift-off, single support, touch-down, the next step
    (tools/f32_walk_err.py prints the per-tick ratios; profiles/r04_f32_walking_vs_oracle.txt).
    As written: status and contact flags bit-exact on every tick; next q atol 1e-5 (observed 1e-7); tau on 99 % of the ticks
    (the five ticks after the touch-down reach 1.7 x the tolerance); the per-foot wrench on 97 % (4.8 x on the touch-down
    tick itself, where the QP takes 11 active-set iterations).
    Amended, with the reason: dv - atol 1e-4 x kp_contact / 10.  Section 5's tolerance belongs to the reference's gains
    (kp_contact 10, conf.py:44); the walking workload needs kp_contact 900 (walk_planner.op3_walking_conf), and the
    acceleration a PD task asks for carries float32's rounding of the forward kinematics (1e-7 m on 0.3 m) times its gain:
    observed 3.5e-4 in the median, within that tolerance on 99 % of the ticks and 1e-2 on the tick before the touch-down (six
    active-set iterations).  next v: its own 1e-5 plus dt x that dv tolerance.  Sim state, as in
    test_one_env_step_f32_against_section5, on the envs whose contact list float32 reproduces exactly (ties among the
    near-coplanar sole vertices of a flat foot are taken within 2e-6 m in float32: 75 % of the env-ticks here): qpos atol 1e-5
    as written, qvel atol 5e-5 on 98 % of the ticks and 2e-4 always (stiff contact rows, D ~ 1e6 against M ~ 1e-3)."""
    from oracle.oracle import WalkTables
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    n, ticks = 16, 620   # t_start = 0.5 s: lift-off at tick 250, touch-down / next lift-off at tick 500
    wc = make(n, "f32", walking=True, reference_quirks=False)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
    lf, rf = wc.frames[0, 0, 9:11].double().cpu().numpy(), wc.frames[0, 1, 9:11].double().cpu().numpy()
    com0 = wc.com_ref[0, :3].double().cpu().numpy()
    # (t_start and the step duration are binary fractions: the phase boundaries t_start + k T are then exact in float32 as
    #  well.  With t_start = 0.4, float32 puts a boundary that falls exactly on a tick one tick later than float64 does.)
    mk = lambda dt_: WalkSchedule.from_demo_paths(n, wc.conf, wc.device, dt_, seed=4, q0_feet=(lf, rf), com0=com0, t_start=0.5)
    s32, s64 = mk(torch.float32), mk(torch.float64)
    st = mirror(wc)
    for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames"):
        st[k][...] = getattr(wc, k).double().cpu().numpy().reshape(st[k].shape)
    st["frames"] = wc.frames.double().cpu().numpy().copy()
    tables = WalkTables(s64)
    push = lambda name, arr: getattr(wc, name).copy_(torch.as_tensor(arr, device=wc.device).reshape(getattr(wc, name).shape).to(getattr(wc, name).dtype))
    ratio = lambda a, b, rtol, atol: float((np.abs(a - b) / (atol + rtol * np.abs(b))).max())
    dv_atol = 1e-4 * wc.conf.kp_contact / 10.0
    dt = wc.conf.dt
    R = {k: [] for k in ("tau", "dv", "w", "q", "v", "qpos", "qvel", "same")}
    phases = set()
    for i in range(ticks):
        t = i * dt
        for k in ("q", "v", "qpos", "qvel", "com_ref", "foot_ref", "contact_ref", "contact_active", "frames"):
            push(k, st[k])
        push("qacc_warmstart", st["qacc_ws"])
        s32.apply(wc, t)
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8, walk=tables.at(t))
        g = lambda k: getattr(wc, k).double().cpu().numpy().reshape(n, -1)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]) and int(wc.status.abs().sum()) == 0, i
        assert np.array_equal(wc.contact_active.cpu().numpy(), st["contact_active"]), i
        phases.add(int(wc.contact_active[0].sum()))
        R["tau"].append(ratio(g("tau"), st["tau"], 1e-3, 1e-4))
        R["dv"].append(ratio(g("dv"), st["dv"], 1e-3, dv_atol))
        R["w"].append(ratio(wrench(g("f"), wc.params), wrench(st["f"], wc.params), 1e-3, 1e-4))
        R["q"].append(float(np.abs(g("q") - st["q"]).max()) / 1e-5)
        R["v"].append(float((np.abs(g("v") - st["v"]) / (1e-5 + dt * (dv_atol + 1e-3 * np.abs(st["dv"])))).max()))
        same = (wc.con_pairs.cpu().numpy() == st["con_geom"]).all(axis=1)
        R["same"].append(same.mean())
        R["qpos"].append(float(np.abs(g("qpos") - st["qpos"])[same].max()) / 1e-5 if same.any() else 0.0)
        R["qvel"].append(float(np.abs(g("qvel") - st["qvel"])[same].max()) / 5e-5 if same.any() else 0.0)
    R = {k: np.asarray(v) for k, v in R.items()}
    summary = {k: (float(np.percentile(v, 50)), float(np.percentile(v, 99)), float(v.max())) for k, v in R.items()}
    assert phases == {1, 2}, phases
    assert (R["dv"] <= 1).mean() >= 0.99 and R["dv"].max() <= 2 and R["q"].max() <= 1 and R["v"].max() <= 1 and R["qpos"].max() <= 1, summary
    assert (R["tau"] <= 1).mean() >= 0.99 and R["tau"].max() <= 2, summary
    assert (R["w"] <= 1).mean() >= 0.97 and R["w"].max() <= 6, summary
    assert (R["qvel"] <= 1).mean() >= 0.98 and R["qvel"].max() <= 4, summary
    assert R["same"].mean() >= 0.7, summary


def test_closed_loop_standing_f32_tracks_oracle(oracle):
    """float32 with the loop closed: TSID reads the sim state, the sim is driven by tau - 120 free-running
    steps of perturbed standing against the float64 oracle: statuses identical, the robot keeps standing on its contacts,
    torques and states track within float32's accumulated rounding (no teacher forcing here: tolerances are those of a
    120-step trajectory, not of BASELINE.md section 5's single step)."""
    n = 16
    wc = make(n, "f32", closed_loop=True)
    g = torch.Generator().manual_seed(5)
    wc.qpos[:, 7:] += ((torch.rand(n, 20, generator=g, dtype=torch.float64) - 0.5) * 0.02).to(wc.device, wc.dtype)
    st = mirror(wc)
    for i in range(120):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
    assert int(wc.status.abs().sum()) == 0 and int(wc.ncon.min()) >= 2
    assert diff(wc.tau, st["tau"]) < 3e-2 and diff(wc.qpos, st["qpos"]) < 5e-4 and diff(wc.qvel, st["qvel"]) < 5e-2   # (observed 8e-3, 2e-4, 2.3e-2)
    assert 0.325 < float(wc.qpos[:, 2].min()) and float(wc.qpos[:, 2].max()) < 0.335
    assert float(wc.qvel.abs().max()) < 0.5


def test_config5_full_size_properties():
    """BASELINE configs[4] at its full size - 65 536 envs with per-env mass / friction / floor tilt / 1 cm terrain
    steps - through size-independent properties: nothing fails or is flagged, states stay finite and unit-norm, the
    contact lists are well formed, and a slice re-run on its own
    controller is bit-identical (what the sharded run relies on)."""
    n = 65536
    wc = make(n)
    perturb(wc, 19, dq=0.03, dv=0.03)
    wc.randomize(seed=2)
    q0, v0 = wc.q.clone(), wc.v.clone()
    for _ in range(30):
        wc.step()
    # no failed QP, no solver failure / skipped step (bits 1, 2, 4).  The contact caps are reported since round 3 (silent
    # before) and a few envs in 65 536 do hit them: the base is teleported to the TSID pose every step (main.py:192), and on a
    # floor tilted up under it a sole is pushed in far enough for more than TSIDB_MAXCON vertices to be inside the margin
    # (bit 8), or the support vertex is one of the 18 (of 11 335) with more than 63 hull-graph neighbours (bit 16)
    assert int((wc.status != 0).sum()) == 0 and int(((wc.info[:, 3] & (7 | 32)) != 0).sum()) == 0
    assert int((wc.info[:, 3] != 0).sum()) < n // 1000
    for t in (wc.q, wc.v, wc.qpos, wc.qvel, wc.tau, wc.rows):
        assert bool(torch.isfinite(t).all())
    assert float((wc.qpos[:, 3:7].norm(dim=1) - 1).abs().max()) < 1e-12
    nc = wc.ncon
    # (the sim's base is teleported to the TSID pose every step, main.py:192: on a floor tilted away under it a robot
    # hovers - most envs touch, none exceeds the cap)
    assert int(nc.min()) >= 0 and int(nc.max()) <= 32 and float((nc > 0).float().mean()) > 0.5
    cp = wc.con_pairs
    valid = torch.arange(32, device=wc.device)[None, :] < nc[:, None]
    assert bool((cp[~valid] == -1).all()) and bool((cp[valid] >= 0).all())
    body, vert = cp >> 16, cp & 0xffff
    assert bool((body[valid] < 21).all()) and bool(((vert[valid] < 2823) | (vert[valid] >= 0x8000)).all())
    assert int(wc.done.sum()) == 0 and float(wc.reward.min()) > 0
    # a slice stepped on its own controller with the same per-env parameters: bit-identical
    lo, hi = 40000, 40512
    sub = make(hi - lo)
    sub.q.copy_(q0[lo:hi]); sub.v.copy_(v0[lo:hi])
    sub.env_params = wc.env_params[lo:hi].clone()
    sub.terrain = wc.terrain[lo:hi].clone()
    rc = sub._L.tsidb_set_env_params(sub._h, sub.env_params.data_ptr(), sub.terrain.data_ptr())
    assert rc == 0
    for _ in range(30):
        sub.step()
    for name in ("q", "v", "qpos", "qvel", "tau", "rows", "ncon", "con_pairs"):
        assert torch.equal(getattr(sub, name), getattr(wc, name)[lo:hi]), name


def test_cop_force_task_f64(oracle):
    """SURVEY 8f-3's last piece: the legacy controller's CoP force task (legacy/biped.py:79-80) as a rank-2 term on the
    force block of the Hessian (the k_tick<T, COP=true> variant); light and heavy weights, double and single
    support, moving states, references off the support centre."""
    for w_cop in (1e-2, 50.0):
        n = 48
        wc = make(n, sim_enabled=False, w_cop=w_cop)
        perturb(wc, 43, dq=0.1, dv=0.5)
        wc.contact_active[::3, 0] = 0
        wc.contact_active[1::5, 1] = 0
        wc.contact_active[(wc.contact_active.sum(dim=1) == 0), 1] = 1
        g = torch.Generator().manual_seed(3)
        wc.cop_ref[:, :2] += ((torch.rand(n, 2, generator=g, dtype=torch.float64) - 0.5) * 0.04).to(wc.device)
        st = mirror(wc)
        cr = wc.cop_ref.cpu().numpy()
        wc.tick()
        for e in range(n):
            out = oracle.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e],
                                   st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e], cop_ref=cr[e])
            st["tau"][e], st["dv"][e], st["f"][e], st["status"][e] = out["tau"], out["dv"], out["f"], out["status"]
        assert np.array_equal(wc.status.cpu().numpy(), st["status"])
        ok = st["status"] == 0
        assert ok.sum() > n // 2
        assert np.allclose(wc.dv.cpu().numpy()[ok], st["dv"][ok], rtol=1e-7, atol=1e-7)
        assert np.allclose(wc.tau.cpu().numpy()[ok], st["tau"][ok], rtol=1e-7, atol=1e-7)
        assert np.abs(wrench(wc.f.cpu().numpy(), wc.params)[ok] - wrench(st["f"], wc.params)[ok]).max() < 1e-6
        assert diff(wc.q[ok], st["q"][ok]) < 1e-9
    # the task does something: the same states without it give different contact forces
    ref = make(n, sim_enabled=False)
    perturb(ref, 43, dq=0.1, dv=0.5)
    ref.contact_active.copy_(wc.contact_active)
    ref.tick()
    assert float((ref.f - wc.f).abs().max()) > 1e-2
    # reset writes the CoP reference between the soles, on the floor
    fresh = make(3, w_cop=1.0)
    mid = 0.5 * (fresh.frames[:, 0, 9:11] + fresh.frames[:, 1, 9:11])
    assert float((fresh.cop_ref[:, :2] - mid).abs().max()) < 1e-15 and float(fresh.cop_ref[:, 2].abs().max()) == 0
    for _ in range(20):
        fresh.step()
    assert int((fresh.status != 0).sum()) == 0


def _closed_loop_walker(n, t_start=1.0, seed=1, feedback=0.6):
    from tsid_control_amd import RobotConfig, WalkController
    from tsid_control_amd.walk_planner import WalkSchedule, op3_closed_loop_walking_conf, op3_walking_posture
    conf = op3_closed_loop_walking_conf(RobotConfig())
    wc = WalkController(conf, num_envs=n, device="cuda:0")
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=seed, q0_feet=(lf, rf),
                                         com0=wc.com_ref[0, :3].cpu().numpy(), foot_press=0.0, t_start=t_start)
    if feedback:
        sched.enable_touchdown_feedback(feedback)
    return wc, sched, 0.5 * (lf + rf)


def test_closed_loop_walking_matches_oracle(oracle):
    """f-1 completed: the loop closed WHILE walking - every tick reads the sim state (main.py:126-129 turned around),
    the sim is driven by tau, the schedule's touch-downs follow the sim's contact list - against the oracle doing the
    same (or_walk_update_fb + tick on the sim state + sim step with motor torques) over the start phase and more than
    two steps, every contact edge included."""
    from oracle.oracle import WalkTables
    n = 6
    wc, sched, _ = _closed_loop_walker(n, t_start=0.3, seed=4)
    st = mirror(wc)
    st["frames"] = wc.frames.cpu().numpy().copy()
    tables = WalkTables(sched)
    edges = 0
    prev = wc.contact_active.clone()
    for i in range(720):
        t = i * wc.conf.dt
        sched.apply(wc, t)
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=6, walk=tables.at(t))
        assert np.array_equal(wc.contact_active.cpu().numpy(), st["contact_active"]), i
        assert np.array_equal(wc.status.cpu().numpy(), st["status"]), i
        assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]), i
        edges += int((wc.contact_active != prev).sum())
        prev = wc.contact_active.clone()
        if i % 20 == 0 or i == 719:
            assert diff(wc.tau, st["tau"]) < 1e-5 and diff(wc.qpos, st["qpos"]) < 1e-7 and diff(wc.qvel, st["qvel"]) < 1e-5, i
            assert np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i
    assert edges >= 4 * n and int((sched.td_latch >= 0).sum()) > 0
    assert np.array_equal(sched.td_latch.cpu().numpy(), tables.keep["td_latch"])
    assert int((wc.status != 0).sum()) == 0 and float(wc.qpos[:, 2].min()) > 0.29


def test_closed_loop_walking_does_not_fall():
    """64 walkers, 5000 ticks (10 s, 18 steps) with the loop closed: nobody falls, no QP fails, the CoM stays on its
    LIPM reference, the robots arrive where their footstep plans end."""
    n = 64
    wc, sched, start = _closed_loop_walker(n)
    bad = torch.zeros(n, dtype=torch.bool, device=wc.device)
    done = torch.zeros(n, dtype=wc.dtype, device=wc.device)
    worst_tilt = worst_com = 0.0
    for i in range(5000):
        sched.apply(wc, i * wc.conf.dt)
        wc.step()
        bad |= wc.status != 0
        done += wc.done
        if i % 25 == 0:
            worst_tilt = max(worst_tilt, float(2 * wc.qpos[:, 4:6].norm(dim=1).max()))
            worst_com = max(worst_com, float((wc.obs[:, 53:55] - wc.com_ref[:, :2]).abs().max()))
    assert not bool(bad.any()) and float(done.sum()) == 0
    assert float(wc.qpos[:, 2].min()) > 0.29 and worst_tilt < 0.3 and worst_com < 0.02
    travelled = (wc.qpos[:, :2] - torch.as_tensor(start, device=wc.device)).norm(dim=1)
    assert float(travelled.min()) > 0.5
    assert int((sched.td_latch >= 0).sum()) == n       # every env took at least one touch-down from the sim


@pytest.mark.parametrize("dtype,replays", [("f64", 30), ("f32", 130)])
def test_captured_graph_equals_eager_steps(dtype, replays):
    """capture_steps(): k_walk + k_tick + k_sim of several pipelined steps in one HIP graph with the clock on the
    device - replaying it gives, bit for bit, what the same number of eager pipelined steps give.  float32 over 1040
    ticks: the device clock is float64 whatever the path's type (a float32 clock advanced by dt per tick
    drifts away from the host's), so the kernel sees the very double the eager path passes."""
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture

    def walker():
        wc = make(64, dtype, walking=True, reference_quirks=False)
        wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
        lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
        sched = WalkSchedule.from_demo_paths(64, wc.conf, wc.device, wc.dtype, seed=2, q0_feet=(lf, rf),
                                             com0=wc.com_ref[0, :3].double().cpu().numpy(), t_start=0.2)
        sched.set_phase_offsets(torch.linspace(0.0, 0.3, 64, dtype=torch.float64))
        return wc, sched

    a, sa = walker()
    b, sb = walker()
    for i in range(40):                                # some eager steps first, on both
        sa.apply(a, i * a.conf.dt); a.step_pipelined()
        sb.apply(b, i * b.conf.dt); b.step_pipelined()
    graph = b.capture_steps(8, sb)
    for r in range(replays):                           # 240 / 1040 steps: through lift-off and touch-down edges
        for k in range(8):
            sa.apply(a, a.t); a.step_pipelined()
        graph.replay()
    a.sync_sim(); b.sync_sim()
    torch.cuda.synchronize()
    assert abs(a.t - b.t) < 1e-12 and float(b.t_device) == b.t and b.t_device.dtype == torch.float64
    for k in ("q", "v", "tau", "dv", "f", "status", "rows", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "contact_active",
              "foot_ref", "com_ref"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert int((b.contact_active.sum(dim=1) == 1).sum()) > 0


# ---------------------------------------------------------------------------- episode lifecycle on the device (f-2)
def _walker(n, dtype="f64", seed=5, plan=True, **kw):
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
    wc = make(n, dtype, walking=True, reference_quirks=False)
    wc.set_posture_bias(op3_walking_posture())
    return wc, WalkSchedule.on_device(wc, seed=seed, plan=plan, **kw)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_plan_kernel_matches_restatement_and_reference_footsteps(oracle, dtype):
    """tsidb_walk_plan (k_plan) against oracle/or_walk.c:or_walk_plan - the unicycle path with drawn scales, every table -
    and, through explicit polylines, against the reference's own footsteps (tests/golden/planners.json: PINNED)."""
    import json
    from pathlib import Path
    from oracle.oracle import walk_plan
    from tsid_control_amd.walk_planner import WalkSchedule, plan_params
    tol = 1e-12 if dtype == "f64" else 2e-6
    n = 48
    wc, sched = _walker(n, dtype, seed=9, scale_range=(0.5, 0.97))
    torch.cuda.synchronize()
    ref = walk_plan(oracle.lib, sched.pp, wc.cop_frames.double().cpu().numpy(), wc.com_ref.double().cpu().numpy(), sched.K,
                    episode=np.zeros(n, np.int32))
    assert np.array_equal(sched.nsteps.cpu().numpy(), ref["nsteps"]) and int(sched.flags.sum()) == 0 and len(set(ref["nsteps"].tolist())) > 4
    assert np.array_equal(sched.side.cpu().numpy(), ref["side"])
    assert diff(sched.steps, ref["steps"]) < 1e-12          # footsteps are float64 whatever the path's type
    for k in ("coef", "rest", "com"):
        assert diff(getattr(sched, k), ref[k]) < tol, k
    # golden footsteps through the device kernel: explicit paths, feet where the reference's demo puts them
    gold = json.loads((Path(__file__).parent / "golden" / "planners.json").read_text())["footsteps"]
    P = max(len(c["path"]) for c in gold)
    path, npts = np.zeros((n, P, 2)), np.full(n, 2, np.int32)
    path[:, 1, 0] = 1.0
    for i, c in enumerate(gold):
        path[i, :len(c["path"])], npts[i] = np.array(c["path"]), len(c["path"])
    wc.cop_frames[:, 0, 9:11] = torch.tensor([0.0, 0.1], dtype=wc.dtype, device=wc.device)
    wc.cop_frames[:, 1, 9:11] = torch.tensor([0.0, -0.1], dtype=wc.dtype, device=wc.device)
    for i, c in enumerate(gold):
        wc.conf.step_length, wc.conf.step_width = c["params"]["step_length"], c["params"]["step_width"]
        sched.pp = plan_params(wc.conf, resample_ds=0.0)
        sched.plan(wc, env_ids=[i], path=path, npts=npts)
        want = np.array([[s["pos"][0], s["pos"][1], s["yaw"], int(s["side"])] for s in c["steps"]])
        assert int(sched.nsteps[i]) + 2 == len(want)
        assert diff(sched.steps[i, :len(want)], want) < (1e-12 if dtype == "f64" else 1e-7)   # (f32: the feet positions are read as float32)
    # paths without a direction plan no step and flag it (bit 1); npts beyond the row length is clamped - same as the twin
    path[:] = 0.0
    path[0, :4] = [[0, 0], [0, 0.3], [0, 0.6], [0, 0.9]]
    path[1, :4] = [[0.1, 0.2]] * 4
    npts[:] = 1
    npts[0], npts[1] = P + 5, 4
    path[0, 4:] = path[0, 3]                      # (the clamp reads the whole row: keep it a valid polyline)
    sched.pp = plan_params(wc.conf, resample_ds=0.0)
    sched.plan(wc, env_ids=[0, 1, 2], path=path, npts=npts)
    twin = walk_plan(oracle.lib, sched.pp, wc.cop_frames.double().cpu().numpy()[:3], wc.com_ref.double().cpu().numpy()[:3], sched.K,
                     path=path[:3], npts=np.minimum(npts[:3], P))
    assert sched.flags[:3].cpu().tolist() == [0, 2, 2] == twin["flags"].tolist()
    assert sched.nsteps[:3].cpu().tolist() == twin["nsteps"].tolist() and int(sched.nsteps[1]) == 0
    for k in ("steps", "coef", "rest", "com"):
        assert torch.isfinite(getattr(sched, k)[:3]).all() and diff(getattr(sched, k)[:3], twin[k]) < (tol if k != "steps" else 1e-7), k


def test_plan_kernel_guards_its_loops_and_tables():
    """tsidb_walk_plan runs one thread per env over parameter-sized loops - parameters that would stall the GPU
    or put NaN into the tables are rejected by the call or flagged per env: a resample step of step_length / 1e6, a
    unicycle path of 1e9 vertices, non-finite parameters (errors); com_drop above the standing CoM height (flag bits 1 and
    3: no step planned, every table finite, the env keeps standing through 50 ticks)."""
    from tsid_control_amd import _lib
    from tsid_control_amd.walk_planner import WalkSchedule
    n = 8
    wc, sched = _walker(n, plan=False)
    for bad in (dict(resample_ds=wc.conf.step_length * 1e-6), dict(unicycle=(0.5, 0.1, 0.1, 1e9)), dict(unicycle=(0.5, 0.1, 0.0, 100)),
                dict(com_drop=float("nan")), dict(scale_range=(0.0, 1.0))):
        with pytest.raises(_lib.TsidbError):
            WalkSchedule.on_device(wc, **bad)
    sched = WalkSchedule.on_device(wc, com_drop=1.0)
    torch.cuda.synchronize()
    assert sched.flags.tolist() == [10] * n and int(sched.nsteps.abs().sum()) == 0
    for t in (sched.coef, sched.rest, sched.com, sched.steps):
        assert bool(torch.isfinite(t).all())


def test_episode_lifecycle_on_the_device():
    """64 walkers; some are made to fall at chosen ticks (base pushed below done_base_height): the tick reports done,
    reset_done() resets exactly those envs and replans them with a NEW path on the device (no host sync), and each then walks
    again - bit for bit like a fresh controller whose schedule starts at that tick with that episode's path.  Envs that never
    fell are bit-identical to a run without any reset."""
    n, F = 64, 640
    dt = 0.002
    falls = {3: 150, 17: 150, 40: 420, 41: 420, 63: 420, 8: 421}
    A, sa = _walker(n)
    for i in range(F):
        sa.apply(A, i * dt)
        for e, r in falls.items():
            if r == i:
                A.q[e, 2] = 0.05
        A.step()
        if i in falls.values():
            assert sorted(torch.nonzero(A.done).flatten().tolist()) == sorted(e for e, r in falls.items() if r == i)
        A.reset_done(sa, t=(i + 1) * dt, new_paths=True)
    torch.cuda.synchronize()
    ep = sa.episode.cpu().numpy()
    assert sorted(np.nonzero(ep)[0].tolist()) == sorted(falls) and ep.max() == 1 and int(A.done.sum()) == 0
    assert float(A.q[:, 2].min()) > 0.25                     # everybody is up and walking
    keys = ("q", "v", "tau", "dv", "f", "qpos", "qvel", "qacc_warmstart", "rows", "contact_active", "foot_ref", "com_ref", "contact_ref")
    # never fell: identical to a run that never resets anything
    C_, sc = _walker(n)
    for i in range(F):
        sc.apply(C_, i * dt)
        C_.step()
    stay = [e for e in range(n) if e not in falls]
    for k in keys:
        assert torch.equal(getattr(A, k)[stay], getattr(C_, k)[stay]), k
    assert torch.equal(sa.coef[stay], sc.coef[stay]) and not torch.equal(sa.coef[3], sc.coef[3])   # the fallen ones got new paths
    # fell at tick r: identical to a fresh controller started at tick r + 1 on episode 1's path
    for r in sorted(set(falls.values())):
        B, sb = _walker(n, plan=False)
        sb.episode[:] = 1
        sb.plan(B, t=(r + 1) * dt)
        for i in range(r + 1, F):
            sb.apply(B, i * dt)
            B.step()
        es = [e for e, rr in falls.items() if rr == r]
        for k in keys:
            assert torch.equal(getattr(A, k)[es], getattr(B, k)[es]), (k, r)
        assert torch.equal(sa.coef[es], sb.coef[es]) and torch.equal(sa.com[es], sb.com[es])


def test_reset_with_schedule_restarts_the_walk():
    """WalkController.reset(env_ids, sched=...): the host-initiated form - the listed envs get their standing state back, their
    plan rebuilt (same path unless new_paths) and their clock restarted; afterwards they repeat the batch's first ticks"""
    n, dt = 16, 0.002
    A, sa = _walker(n, seed=2)
    first = {}
    for i in range(300):
        sa.apply(A, i * dt)
        A.step()
        if i < 100:
            first[i] = (A.q[5].clone(), A.tau[5].clone(), A.qpos[5].clone())
    A.reset(env_ids=[5, 9], sched=sa, t=300 * dt)
    assert int(sa.episode[5]) == 0
    for i in range(300, 400):
        sa.apply(A, i * dt)
        A.step()
        q, tau, qpos = first[i - 300]
        # (env time = i dt - 300 dt is not bit-equal to (i - 300) dt: compare to rounding, not bit for bit)
        assert float((A.q[5] - q).abs().max()) < 1e-9 and float((A.tau[5] - tau).abs().max()) < 1e-6 and float((A.qpos[5] - qpos).abs().max()) < 1e-8
    A.reset(env_ids=[9], sched=sa, t=400 * dt, new_paths=True)
    assert int(sa.episode[9]) == 1 and int(sa.episode[5]) == 0


@pytest.mark.parametrize("rule", ["all", "mujoco"])
def test_plane_mesh_rule_matches_oracle(oracle, rule):
    """a9 fidelity switch (conf.sim_plane_mesh): both plane <-> mesh multi-contact rules, HIP vs oracle, pair lists bit-exact -
    perturbed standing envs pressed 2 mm into the floor so that many sole vertices are inside the margin"""
    n = 24
    wc = make(n, sim_plane_mesh=rule)
    perturb(wc, 31, dq=0.02, dv=0.02)
    wc.q[:, 2] -= 0.002
    st = mirror(wc)
    nmax = 0
    for i in range(20):
        wc.step()
        oracle.env_step_batch(wc.params, st, nthreads=8)
        assert np.array_equal(wc.ncon.cpu().numpy(), st["ncon"]) and np.array_equal(wc.con_pairs.cpu().numpy(), st["con_geom"]), i
        assert diff(wc.qpos, st["qpos"]) < 1e-9 and diff(wc.qvel, st["qvel"]) < 1e-6, i
        nmax = max(nmax, int(wc.ncon.max()))
    per_geom = max(int((wc.con_pairs[e, :int(wc.ncon[e])] >> 16).bincount().max()) for e in range(n) if int(wc.ncon[e]) > 0)
    assert (per_geom <= 4) if rule == "mujoco" else (nmax > 8)
    assert int((wc.info[:, 3] != 0).sum()) == 0


def test_contact_caps_are_flagged(oracle):
    """silent caps are not silent: a robot lying on the floor makes more floor contacts than TSIDB_MAXCON - bit 8 of info[:, 3],
    on the device as in the oracle"""
    n = 4
    wc = make(n, self_collision=True, sim_plane_mesh="all")      # (every neighbour in the margin: the rule that can overflow)
    quats = torch.tensor([[0.7071068, 0.7071068, 0, 0], [0.7071068, 0, 0.7071068, 0], [0.5, 0.5, 0.5, 0.5], [1.0, 0, 0, 0]],
                         dtype=wc.dtype, device=wc.device)
    wc.qpos[:, 3:7] = quats
    wc.qpos[:3, 2] = 0.03
    qpos, qvel, ws = (x.cpu().numpy().copy() for x in (wc.qpos, wc.qvel, wc.qacc_warmstart))
    wc.sim_step(teleport=False)
    flagged = 0
    for e in range(n):
        r = oracle.sim_step(qpos[e], qvel[e], np.zeros(20), ws[e], self_collision=True, plane_mesh="all")
        assert r["flags"] == int(wc.info[e, 3]) & (8 | 16 | 32), e
        assert r["ncon"] == int(wc.ncon[e])
        flagged += int(bool(r["flags"] & 8))
        if r["flags"] & 8:
            assert r["ncon"] == 32
    assert flagged >= 1 and int(wc.info[3, 3]) == 0     # the standing env is not flagged


def test_two_wavefront_sim_is_bit_identical():
    """conf.sim_waves: the small-batch shape of the sim kernel (collision phase on a second wavefront beside the
    unconstrained dynamics; the library's default up to 512 envs, tsidb_create) against one wavefront per env - same operations on the same data,
    bit for bit: perturbed standing, randomised floors with terrain steps, self-colliding poses"""
    n = 96
    a, b = make(n, sim_waves=1), make(n, sim_waves=2)
    for w in (a, b):
        perturb(w, 41, dq=0.04, dv=0.05)
        w.randomize(seed=3)
        w.qpos[: n // 3, 7:] = _self_collision_poses(n // 3, 5).to(w.device, w.dtype)
    for i in range(30):
        a.step()
        b.step()
    for i in range(20):
        a.sim_step(teleport=False)
        b.sim_step(teleport=False)
    torch.cuda.synchronize()
    for k in ("q", "v", "tau", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info", "rows"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert int(a.ncon.max()) > 4 and int((a.con_pairs & 0x8000).bool().sum()) > 0     # floor and robot<->robot contacts were there


@pytest.mark.parametrize("dtype,n", [("f64", 96), ("f64", 33), ("f32", 64)])
def test_packed_sim_is_bit_identical(dtype, n):
    """conf.sim_pack: the sim kernel with TWO envs per wavefront (tsidb_sim2.hpp: 32 lanes per env, DPP broadcasts instead of
    v_readlane, per-env divergent control flow) against one env per wavefront - same operations on the same data in the same
    order, bit for bit: perturbed standing, randomised floors with terrain steps, self-colliding poses (MPR, dense Newton
    factor), an odd number of envs (the last wavefront holds one env), envs paired with a diverged neighbour"""
    a, b = make(n, dtype, sim_waves=1, sim_pack=0), make(n, dtype, sim_waves=1, sim_pack=1)
    for w in (a, b):
        perturb(w, 41, dq=0.04, dv=0.05)
        w.randomize(seed=3)
        w.qpos[: n // 3, 7:] = _self_collision_poses(n // 3, 5).to(w.device, w.dtype)
        w.qvel[n - 2, :] = 1e9     # a diverged env (its step is skipped, failure bit 4) beside a healthy one
    for i in range(30):
        a.step()
        b.step()
    for i in range(20):
        a.sim_step(teleport=False)
        b.sim_step(teleport=False)
    torch.cuda.synchronize()
    for k in ("q", "v", "tau", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info", "rows"):
        assert torch.equal(getattr(a, k), getattr(b, k)), (k, (getattr(a, k) != getattr(b, k)).nonzero()[:8].tolist())
    assert int(a.ncon.max()) > 4 and int((a.con_pairs & 0x8000).bool().sum()) > 0     # floor and robot<->robot contacts were there
    assert int(a.info[n - 2, 3]) == 4 and int(a.info[n - 1, 3]) != 4


def test_fused_walk_tick_and_device_episodes_in_the_pipelined_loop():
    """the loop INTEGRATION.md shows - step_pipelined(walk=(sched, t)) (reference update + tick in one launch, the TSID state
    snapshot written by the tick, sim on the second stream) with reset_done(sched) every step - against the plain sequence
    apply(); step(); reset_done(): bit-identical, including an env that falls and restarts on a new path"""
    n, dt = 32, 0.002
    A, sa = _walker(n, seed=7)
    B, sb = _walker(n, seed=7)
    for i in range(260):
        if i == 120:
            A.q[5, 2] = 0.05
            B.q[5, 2] = 0.05
        sa.apply(A, i * dt)
        A.step()
        A.reset_done(sa, t=(i + 1) * dt)
        B.step_pipelined(walk=(sb, i * dt))
        B.reset_done(sb, t=(i + 1) * dt)
    B.sync_sim()
    torch.cuda.synchronize()
    assert int(sa.episode[5]) == 1 and int(sb.episode[5]) == 1 and int(sa.episode.sum()) == 1
    for k in ("q", "v", "tau", "dv", "f", "status", "rows", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "contact_active",
              "foot_ref", "com_ref", "contact_ref", "frames"):
        assert torch.equal(getattr(A, k), getattr(B, k)), k
    assert torch.equal(sa.coef, sb.coef) and torch.equal(sa.t_offset, sb.t_offset)
