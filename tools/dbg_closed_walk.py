"""Diagnostic: does the walking controller survive closed loop (TSID on the sim state, sim driven by tau)?
    python tools/dbg_closed_walk.py walk|stand [conf_attr=value ...] [press=0.0] [n=16] [ticks=3000]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_closed_loop_walking_conf, op3_walking_posture
conf = op3_closed_loop_walking_conf(RobotConfig())
press, n, ticks, fb = 0.0, 16, 3000, 0.6
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "press": press = float(v)
    elif k == "n": n = int(v)
    elif k == "ticks": ticks = int(v)
    elif k == "fb": fb = float(v)
    else: setattr(conf, k, float(v))
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy(), foot_press=press)
if fb > 0: sched.enable_touchdown_feedback(fb)
walk = len(sys.argv) > 1 and sys.argv[1] == "walk"
fell_at = None
for i in range(ticks):
    if walk:
        sched.apply(wc, i * conf.dt)
    wc.step()
    if i % 250 == 0:
        print(i, "bad", int((wc.status != 0).sum()), "ncon", wc.ncon[:3].tolist(), "base z", [round(float(x), 3) for x in wc.qpos[:3, 2]],
              "tilt", [round(float(2 * torch.linalg.norm(wc.qpos[e, 4:6])), 3) for e in range(2)], "act", wc.contact_active[0].tolist(),
              "com err", round(float((wc.obs[:, 53:55] - wc.com_ref[:, :2]).abs().max()), 4),
              "latched", int((sched.td_latch >= 0).sum()) if sched.td_latch is not None else -1)
    if float(wc.qpos[:, 2].min()) < 0.2 and fell_at is None:
        fell_at = i
    if float(wc.qpos[:, 2].max()) < 0.15:
        break
print("RESULT", " ".join(sys.argv[1:]), "first fall at", fell_at, "fallen", int((wc.qpos[:, 2] < 0.2).sum()), "of", n, "after", i + 1, "ticks; travelled",
      round(float((wc.qpos[:, :2] - torch.as_tensor(0.5 * (lf + rf), device=wc.device)).norm(dim=1).mean()), 3))
