import sys, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
n = 1024
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
hist = torch.zeros(64, dtype=torch.long, device=wc.device)
for i in range(5600):
    sched.apply(wc, i * conf.dt); wc.step()
    if i >= 600:
        hist += torch.bincount(wc.info[:, 0].clamp(max=63), minlength=64)
h = hist.cpu().double(); h /= h.sum()
print("qp outer iterations over ticks 600..5600:", {k: round(float(v), 4) for k, v in enumerate(h) if v > 0})
