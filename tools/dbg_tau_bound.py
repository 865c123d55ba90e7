import sys
import torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.12
n = 256
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False; conf.tau_max_scaling = sc
wc = WalkController(conf, num_envs=n, device="cuda:0")
print("tau_max", wc.tau_max[:3], wc.params[71:74])
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
for i in range(900):
    sched.apply(wc, i * conf.dt); wc.step()
    if i % 60 == 0:
        print(i, "iters>1 frac", float((wc.info[:, 0] > 1).float().mean()), "max|tau|", float(wc.tau.abs().max()), "argmax joint", int(wc.tau.abs().max(dim=0).values.argmax()),
              "status!=0", int((wc.status != 0).sum()), "z", float(wc.q[:, 2].mean()), "nact", float(wc.info[:, 1].float().mean()))
