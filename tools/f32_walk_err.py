"""float32 walking, teacher-forced against the float64 oracle (the loop of tests/test_gpu_parity.py::
test_walking_f32_against_section5_every_tick): per-tick error ratios to BASELINE.md section 5's tolerances, worst ticks.
    python tools/f32_walk_err.py > gpurun_out/f32_walk_err.txt"""
import sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_parity import make, mirror, wrench
from oracle.oracle import Oracle, WalkTables
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
n, ticks = 16, 620
wc = make(n, "f32", walking=True, reference_quirks=False)
oracle = Oracle(wc.model.raw)
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
lf, rf = wc.frames[0, 0, 9:11].double().cpu().numpy(), wc.frames[0, 1, 9:11].double().cpu().numpy()
com0 = wc.com_ref[0, :3].double().cpu().numpy()
mk = lambda dt_: WalkSchedule.from_demo_paths(n, wc.conf, wc.device, dt_, seed=4, q0_feet=(lf, rf), com0=com0, t_start=0.5)
s32, s64 = mk(torch.float32), mk(torch.float64)
st = mirror(wc)
for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames"):
    st[k][...] = getattr(wc, k).double().cpu().numpy().reshape(st[k].shape)
st["frames"] = wc.frames.double().cpu().numpy().copy()
tables = WalkTables(s64)
push = lambda name, arr: getattr(wc, name).copy_(torch.as_tensor(arr, device=wc.device).reshape(getattr(wc, name).shape).to(getattr(wc, name).dtype))
ratio = lambda a, b, rtol, atol: (np.abs(a - b) / (atol + rtol * np.abs(b)))
rows = []
for i in range(ticks):
    t = i * wc.conf.dt
    for k in ("q", "v", "qpos", "qvel", "com_ref", "foot_ref", "contact_ref", "contact_active", "frames"):
        push(k, st[k])
    push("qacc_warmstart", st["qacc_ws"])
    s32.apply(wc, t)
    wc.step()
    oracle.env_step_batch(wc.params, st, nthreads=8, walk=tables.at(t))
    g = lambda k: getattr(wc, k).double().cpu().numpy().reshape(n, -1)
    same = (wc.con_pairs.cpu().numpy() == st["con_geom"]).all(axis=1)
    r = dict(i=i, ns=int(wc.contact_active[0].sum()), iters=int(wc.info[:, 0].max()), status=int((wc.status.cpu().numpy() != st["status"]).sum()),
             tau=ratio(g("tau"), st["tau"], 1e-3, 1e-4).max(), dv=ratio(g("dv"), st["dv"], 1e-3, 1e-4).max(),
             w=ratio(wrench(g("f"), wc.params), wrench(st["f"], wc.params), 1e-3, 1e-4).max(),
             q=np.abs(g("q") - st["q"]).max() / 1e-5, v=np.abs(g("v") - st["v"]).max() / 1e-5, same=same.mean(),
             qpos=(np.abs(g("qpos") - st["qpos"])[same].max() / 1e-5) if same.any() else 0, qvel=(np.abs(g("qvel") - st["qvel"])[same].max() / 5e-5) if same.any() else 0,
             fref=np.abs(g("foot_ref") - st["foot_ref"].reshape(n, -1)).max(), cref=np.abs(g("com_ref") - st["com_ref"].reshape(n, -1)).max(),
             dv_env=int(ratio(g("dv"), st["dv"], 1e-3, 1e-4).max(axis=1).argmax()), dv_idx=int(ratio(g("dv"), st["dv"], 1e-3, 1e-4).max(axis=0).argmax()))
    rows.append(r)
for key in ("tau", "dv", "w", "q", "v", "qpos", "qvel"):
    top = sorted(rows, key=lambda r: -r[key])[:6]
    print(key, "worst ticks:", [(r["i"], round(float(r[key]), 3), "ns", r["ns"], "it", r["iters"]) for r in top])
print("ticks over tolerance per key:", {k: sum(1 for r in rows if r[k] > 1) for k in ("tau", "dv", "w", "q", "v", "qpos", "qvel")})
print("same-contact-list fraction mean", np.mean([r["same"] for r in rows]), "min", min(r["same"] for r in rows))
print("max foot_ref diff", max(r["fref"] for r in rows), "max com_ref diff", max(r["cref"] for r in rows))
for r in rows:
    if r["dv"] > 1 or r["qvel"] > 1:
        print({k: (round(float(v), 3) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()})
