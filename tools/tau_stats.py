"""GPU box: distribution of the joint torques of the walking workload (to size the tightened torque bound of
bench.py's active-set secondary run)."""
import sys
import torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture

n = 1024
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
wc = WalkController(conf, num_envs=n, device="cuda:0")
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
mx = torch.zeros(n, device=wc.device, dtype=wc.dtype)
for i in range(900):
    sched.apply(wc, i * conf.dt); wc.step()
    if i > 600: mx = torch.maximum(mx, wc.tau.abs().max(dim=1).values)
q = torch.quantile(mx, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], device=wc.device, dtype=wc.dtype))
print("max |tau| per env over ticks 600..900, quantiles 0/10/50/90/100 %:", q.tolist())
for sc in (0.2, 0.15, 0.12, 0.1, 0.08, 0.06):
    print(f"tau_max_scaling {sc}: bound {sc * 10:.2f} N m, envs exceeding it {(mx > sc * 10).float().mean().item():.3f}")
