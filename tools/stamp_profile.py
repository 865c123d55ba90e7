"""Phase breakdown of k_tick / k_sim from in-kernel shader-clock stamps (diagnostic build only).

    hipcc ... -DTSIDB_STAMPS -o gpurun_out/libtsidb_stamps.so tsidb_api.hip
    TSIDB_LIB_PATH=gpurun_out/libtsidb_stamps.so python tools/stamp_profile.py f64 4096
Reads SHARES, not run time (stamps fence the schedule)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController, _lib

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
conf = RobotConfig(); conf.dtype = dtype
WALK = len(sys.argv) > 3 and sys.argv[3] == "walk"
if WALK:
    from tsid_control_amd.walk_planner import op3_walking_conf, op3_walking_posture
    op3_walking_conf(conf); conf.reference_quirks = False
import os
if os.environ.get("TSIDB_TAU_MAX_SCALING"):    # (bench.py --tau-max-scaling: torque bounds tight enough to be active)
    conf.tau_max_scaling = float(os.environ["TSIDB_TAU_MAX_SCALING"])
wc = WalkController(conf, num_envs=N)
torch.manual_seed(0)
if len(sys.argv) > 3 and sys.argv[3] == "walk":
    from tsid_control_amd.walk_planner import WalkSchedule
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
    sched = WalkSchedule.from_demo_paths(N, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].double().cpu().numpy())
    if os.environ.get("TSIDB_DEPHASE"):        # (bench.py --dephase: per-env start delays)
        sched.set_phase_offsets(torch.rand(N, generator=torch.Generator().manual_seed(7), dtype=torch.float64) * float(os.environ["TSIDB_DEPHASE"]))
    for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 100):
        sched.apply(wc, i * conf.dt); wc.step()
else:
    wc.q[:, 7:] += (torch.rand(N, 20, dtype=wc.dtype, device=wc.device) - 0.5) * 0.1
    wc.v[:] = torch.randn(N, 26, dtype=wc.dtype, device=wc.device) * 0.05
    for _ in range(5): wc.step()
torch.cuda.synchronize()
L = _lib.load()
n = min(N, 8192)
buf = np.zeros((n, 32), dtype=np.uint64)
L.tsidb_debug_stamps.argtypes = [C.c_void_p, C.c_int]
rc = L.tsidb_debug_stamps(buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0
b = buf.astype(np.int64)
names_t = ["rbd_terms", "rhs+Dyn", "H assemble", "cholesky", "fwdsub+inv+J write", "equalities(18)", "(unused)", "inequality loop", "decode+integrate"]
idx_t = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), None, (6, 8), (8, 9)]
print(f"k_tick {dtype} N={N}: qp iters mean {wc.info[:,0].float().mean():.2f} max {int(wc.info[:,0].max())}; iq mean {wc.info[:,1].float().mean():.1f}")
tot = np.median(b[:, 9] - b[:, 0])
for nm, ix in zip(names_t, idx_t):
    if ix is None: continue
    d = np.median(b[:, ix[1]] - b[:, ix[0]])
    print(f"  {nm:24s} {d:10.0f} cyc  {100*d/tot:5.1f}%")
print(f"  {'total':24s} {tot:10.0f} cyc")
bb = np.arange(n)                      # stamp row = workgroup; its env = env_of_block (tsidb_common.hpp: XCD x owns a contiguous env range)
env_of_block = (bb % 8) * (N // 8) + np.minimum(bb % 8, N % 8) + bb // 8
it = wc.info[:, 0].cpu().numpy()[env_of_block]
tt = b[:, 9] - b[:, 0]
for lo_, hi_ in ((1, 1), (2, 3), (4, 7), (8, 15), (16, 999)):
    sel = (it >= lo_) & (it <= hi_)
    if sel.any():
        print(f"  envs with {lo_}-{hi_} qp iterations: {int(sel.sum()):5d}  median total {np.median(tt[sel]):9.0f} cyc  inequality loop {np.median((b[:, 8] - b[:, 6])[sel]):9.0f}")
fast = it == 1
if fast.any():   # envs that ended in the fast equality solve: stamps 5, 10, 11, 12, 6 bracket its pieces
    for nm, (s0_, s1_) in (("  fast eq: B^T B", (5, 10)), ("  fast eq: Cholesky", (10, 11)), ("  fast eq: substitutions", (11, 12)), ("  fast eq: z, x", (12, 6)), ("  fast eq: sweep", (6, 8))):
        print(f"{nm:26s} {np.median((b[:, s1_] - b[:, s0_])[fast]):10.0f} cyc")
for nm, k in (("  as: sweep (row values)", 10), ("  as: select+np", 11), ("  as: d = J^T np", 12), ("  as: householder+z", 13), ("  as: r, steps, add", 14)):
    print(f"{nm:26s} {np.median(b[:, k]):10.0f} cyc (sum over iterations)")
for nm, k in (("  rbd: setup+sincos", 15), ("  rbd: depth loop", 23), ("  rbd: inertia+force", 29), ("  rbd: subtree gather", 30), ("  rbd: per-dof h,M", 31)):
    print(f"{nm:26s} {np.median(b[:, k]):10.0f} cyc")
names_s = ["kin+bias+M", "qacc_smooth chol", "collision", "rows+warmstart", "newton loop", "euler+write"]
idx_s = [(16, 17), (17, 18), (18, 19), (19, 20), (20, 21), (21, 22)]
if int(__import__("os").environ.get("TSIDB_SIM_PACK", "0")):   # packed sim kernel: one workgroup (= one stamp row) per PAIR of envs
    b = b[: (n + 1) // 2]
tot = np.median(b[:, 22] - b[:, 16])
print(f"k_sim: ncon mean {wc.ncon.float().mean():.1f}; newton iters mean {wc.info[:,2].float().mean():.2f}")
for nm, ix in zip(names_s, idx_s):
    d = np.median(b[:, ix[1]] - b[:, ix[0]])
    print(f"  {nm:24s} {d:10.0f} cyc  {100*d/tot:5.1f}%")
print(f"  {'total':24s} {tot:10.0f} cyc")
for nm, k in (("  newton: eval+grad", 24), ("  newton: Hessian build", 25), ("  newton: chol+solve", 26), ("  newton: line search+move", 27)):
    print(f"{nm:26s} {np.median(b[:, k]):10.0f} cyc (sum over iterations)")
