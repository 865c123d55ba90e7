"""Diagnostic: does the walking controller survive closed loop (TSID on the sim state, sim driven by tau)?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False; conf.closed_loop = True
for kv in sys.argv[2:]:
    k, v = kv.split("="); setattr(conf, k, float(v))
n = 16
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
walk = len(sys.argv) > 1 and sys.argv[1] == "walk"
for i in range(3000):
    if walk:
        sched.apply(wc, i * conf.dt)
    wc.step()
    if i % 100 == 0:
        print(i, "status", wc.status[:4].tolist(), "ncon", wc.ncon[:4].tolist(), "base z", [round(float(x), 3) for x in wc.qpos[:4, 2]],
              "tilt", [round(float(2 * torch.linalg.norm(wc.qpos[e, 4:6])), 3) for e in range(2)], "act", wc.contact_active[0].tolist())
    if float(wc.qpos[:, 2].max()) < 0.15:
        print("all fell at", i); break
