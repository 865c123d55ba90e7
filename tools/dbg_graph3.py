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
keys = ("q", "qpos", "qvel", "ncon")
mode = sys.argv[1]
a, sa = walker(); b, sb = walker()
for i in range(40):
    sa.apply(a, i * a.conf.dt); a.step()
    sb.apply(b, i * b.conf.dt); b.step_pipelined()
g = b.capture_steps(8, sb)
for r in range(30):
    for k in range(8):
        sa.apply(a, a.t); a.step()
    g.replay()
    if mode == "syncsim": b.sync_sim()
    if mode == "peek":
        torch.cuda.synchronize()
        bad = {k: float((getattr(a, k).double() - getattr(b, k).double()).abs().max()) for k in keys if not torch.equal(getattr(a, k), getattr(b, k))}
        if bad: print("replay", r, bad); break
torch.cuda.synchronize()
print(mode, {k: float((getattr(a, k).double() - getattr(b, k).double()).abs().max()) for k in keys if not torch.equal(getattr(a, k), getattr(b, k))})
