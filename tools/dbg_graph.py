import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
def walker():
    if "testmake" in sys.argv:
        import test_gpu_parity as T
        wc = T.make(64, walking=True, reference_quirks=False)
    else:
        conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
        wc = WalkController(conf, num_envs=64, device="cuda:0")
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(64, wc.conf, wc.device, wc.dtype, seed=2, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy(), t_start=0.2)
    if "off" in sys.argv: sched.set_phase_offsets(torch.linspace(0.0, 0.3, 64, dtype=torch.float64))
    return wc, sched
keys = ("q", "v", "tau", "dv", "f", "status", "rows", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs", "info", "contact_active", "foot_ref", "com_ref")
def cmp(a, b, tag):
    a.sync_sim(); b.sync_sim(); torch.cuda.synchronize()
    bad = [k for k in keys if not torch.equal(getattr(a, k), getattr(b, k))]
    print(tag, "differs:", bad, [float((getattr(a, k).double() - getattr(b, k).double()).abs().max()) for k in bad])
a, sa = walker(); b, sb = walker()
for i in range(40):
    sa.apply(a, i * a.conf.dt); a.step_pipelined()
    sb.apply(b, i * b.conf.dt); b.step_pipelined()
cmp(a, b, "after 40 eager")
g = b.capture_steps(int(sys.argv[1]) if len(sys.argv) > 1 else 8, sb)
cmp(a, b, "after capture")
for r in range(30):
    for k in range(g.steps):
        sa.apply(a, a.t); a.step_pipelined()
    g.replay()
    cmp(a, b, f"after replay {r}")
