"""Quick GPU-vs-oracle comparison used while bringing the kernels up (not a test)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.oracle import Oracle, new_state
from tsid_control_amd import RobotConfig, WalkController

def main(dtype="f64", N=64, ticks=50):
    conf = RobotConfig(); conf.dtype = dtype
    wc = WalkController(conf, num_envs=N)
    orc = Oracle(wc.model.raw)
    torch.manual_seed(0)
    wc.q[:, 7:] += (torch.rand(N, 20, dtype=wc.dtype, device=wc.device) - 0.5) * 0.1
    wc.v[:] = torch.randn(N, 26, dtype=wc.dtype, device=wc.device) * 0.05
    st = new_state(N)
    for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames"):
        src = getattr(wc, k if k != "qacc_ws" else "qacc_warmstart")
        st[k][...] = src.double().cpu().numpy().reshape(st[k].shape)
    # rbd terms
    t = wc.rbd_terms()
    to = [orc.terms(st["q"][e], st["v"][e]) for e in range(N)]
    for key, okey in (("M", "M"), ("h", "h"), ("Jcom", "Jcom"), ("Jf", "Jf"), ("com", "com")):
        g = t[key].double().cpu().numpy(); o = np.stack([x[okey] for x in to])
        print(f"rbd {key}: max abs diff {np.abs(g - o).max():.3e}  (scale {np.abs(o).max():.3e})")
    g = t["oMf"].double().cpu().numpy(); o = np.stack([x["oMf"] for x in to]); print(f"rbd oMf: {np.abs(g-o).max():.3e}")
    t0 = time.time()
    for i in range(ticks):
        wc.step()
        orc.env_step_batch(wc.params, st, nthreads=8)
        if i in (0, 1, 2, 5, 10, 20, ticks - 1):
            torch.cuda.synchronize()
            d = lambda a, b: np.abs(a.double().cpu().numpy().reshape(b.shape) - b).max()
            print(f"tick {i}: status gpu {wc.status.unique().tolist()} orc {np.unique(st['status']).tolist()} "
                  f"dtau {d(wc.tau, st['tau']):.2e} ddv {d(wc.dv, st['dv']):.2e} df {d(wc.f, st['f']):.2e} "
                  f"dq {d(wc.q, st['q']):.2e} dv {d(wc.v, st['v']):.2e} dqpos {d(wc.qpos, st['qpos']):.2e} "
                  f"dqvel {d(wc.qvel, st['qvel']):.2e} dobs {d(wc.obs, st['obs']):.2e} "
                  f"ncon eq {bool((wc.ncon.cpu().numpy() == st['ncon']).all())} pairs eq {bool((wc.con_pairs.cpu().numpy() == st['con_geom']).all())} "
                  f"ncon max {int(wc.ncon.max())} qp iters max {int(wc.info[:,0].max())} iq max {int(wc.info[:,1].max())} newton max {int(wc.info[:,2].max())} fail {int(wc.info[:,3].max())}")
    print("elapsed", time.time() - t0)
    # timing
    for _ in range(3): wc.step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): wc.step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print(f"{dtype} N={N}: {dt*1e3:.3f} ms/step -> {N/dt:.0f} env-steps/s")

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "f64", int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 50)
