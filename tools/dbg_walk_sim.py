"""Diagnostic: slave-sim state while the TSID side walks (GPU)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
from tsid_control_amd.walk_controller import map_tsid_to_mujoco
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
n = 8
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1500):
    sched.apply(wc, i * conf.dt)
    wc.step()
    if i % 50 == 0:
        ctrl = map_tsid_to_mujoco(wc.q)
        print(i, "act", wc.contact_active[0].tolist(), "ncon", wc.ncon[:4].tolist(), "fz", [round(float(x), 4) for x in wc.frames[0, :, 11]],
              "qpos z", round(float(wc.qpos[0, 2]), 4), "q z", round(float(wc.q[0, 2]), 4), "qvel6", [round(float(x), 3) for x in wc.qvel[0, :6]],
              "jerr", round(float((wc.qpos[0, 7:] - ctrl[0]).abs().max()), 4))
