import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_posture
def walker():
    wc = T.make(64, walking=True, reference_quirks=False)
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(64, wc.conf, wc.device, wc.dtype, seed=2, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy(), t_start=0.2)
    sched.set_phase_offsets(torch.linspace(0.0, 0.3, 64, dtype=torch.float64))
    return wc, sched
keys = ("q", "v", "tau", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info")
res = {}
for name in ("eager_nosync", "eager_sync", "graph_nosync", "graph_sync", "serial"):
    wc, s = walker()
    for i in range(40):
        s.apply(wc, i * wc.conf.dt); wc.step_pipelined()
    g = wc.capture_steps(8, s) if name.startswith("graph") else None
    for r in range(30):
        if g: g.replay()
        else:
            for k in range(8):
                s.apply(wc, wc.t)
                if name == "serial": wc.step()
                else: wc.step_pipelined()
                if name == "eager_sync": wc.sync_sim(); torch.cuda.synchronize()
        if name == "graph_sync": torch.cuda.synchronize()
    wc.sync_sim(); torch.cuda.synchronize()
    res[name] = {k: getattr(wc, k).clone() for k in keys}
base = res["serial"]
for name, r in res.items():
    print(name, "vs serial:", {k: float((r[k].double() - base[k].double()).abs().max()) for k in keys if not torch.equal(r[k], base[k])})
