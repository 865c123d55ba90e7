import sys, time, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].cpu().numpy())
for i in range(700):
    sched.apply(wc, i * conf.dt); wc.step_pipelined()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(700, 2700):
    sched.apply(wc, i * conf.dt); wc.step_pipelined()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={n}: enqueue {1e6*(t1-t0)/2000:.1f} us/step, total {1e6*(t2-t0)/2000:.1f} us/step")
