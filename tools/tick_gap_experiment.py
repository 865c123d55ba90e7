"""What separates consecutive tick launches at small batches: tick-only loop, pipelined loop, pipelined without the ring wait.
   python tools/tick_gap_experiment.py <envs>"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture

n = int(sys.argv[1])
steps = 1200
conf = op3_walking_conf(RobotConfig())
conf.reference_quirks = False
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].double().cpu().numpy())


def timed(f, label):
    for i in range(100):
        f()
    wc.sync_sim(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f()
    wc.sync_sim(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"envs {n} {label}: {el / steps * 1e3:.4f} ms per step")


for i in range(620):
    wc.step_pipelined(walk=(sched, wc.t))
wc.sync_sim()


def tick_only():
    wc.tick(walk=(sched, wc.t)); wc.t += conf.dt


def host_only():   # the host cost of one pipelined step: everything enqueued, nothing waited for
    wc.step_pipelined(walk=(sched, wc.t))


timed(tick_only, "tick-only loop (one stream, no sim)")
timed(host_only, "pipelined")
t0 = time.perf_counter()
for i in range(400):
    wc.step_pipelined(walk=(sched, wc.t))
el = time.perf_counter() - t0
wc.sync_sim(); torch.cuda.synchronize()
print(f"envs {n} host enqueue cost of a pipelined step (no sync): {el / 400 * 1e3:.4f} ms")
t0 = time.perf_counter()
for i in range(400):
    wc.tick(walk=(sched, wc.t)); wc.t += conf.dt
el = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"envs {n} host enqueue cost of a tick: {el / 400 * 1e3:.4f} ms")
