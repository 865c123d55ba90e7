"""Soak: N walkers with device-side episodes (plan on the device, reset_done every step) for many steps; counts resets, failed
QPs (status of EVERY tick, accumulated on the device), flagged sim envs (every 50th step), non-finite states.
    python tools/soak_episodes.py [envs] [steps] [closed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_closed_loop_walking_conf, op3_walking_posture

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
closed = len(sys.argv) > 3 and sys.argv[3] == "closed"
conf = op3_closed_loop_walking_conf(RobotConfig()) if closed else op3_walking_conf(RobotConfig())
conf.reference_quirks = False
conf.closed_loop = closed
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.set_posture_bias(op3_walking_posture())
sched = WalkSchedule.on_device(wc, seed=3, K=40, scale_range=(0.1, 0.3), **({"foot_press": 0.0} if closed else {}))   # short paths: episodes end often
if closed:
    sched.enable_touchdown_feedback()
resets = torch.zeros((), dtype=torch.int64, device=wc.device)
failed = torch.zeros((), dtype=torch.int64, device=wc.device)
flagged = torch.zeros((), dtype=torch.int64, device=wc.device)
failed_every = torch.zeros((), dtype=torch.int64, device=wc.device)
t0 = time.perf_counter()
with torch.cuda.stream(wc.tick_stream):
    for i in range(steps):
        if closed:
            sched.apply(wc, wc.t); wc.step()
        else:
            wc.step_pipelined(walk=(sched, wc.t))
        failed_every.add_((wc.status != 0).sum())   # every tick (two tiny kernels on the tick stream, no host sync)
        if i % 50 == 49:
            # an episode also ends when its plan is walked to the end: mark those envs done (t > t_start + nsteps T + 1 s)
            tl = wc.t - sched.t_offset.double()
            over = tl > (sched.t_start + sched.nsteps.double() * conf.step_duration + 1.0)
            wc.rows[:, wc.NOBS + 1] = torch.maximum(wc.rows[:, wc.NOBS + 1], over.to(wc.dtype))
            resets += (wc.rows[:, wc.NOBS + 1] != 0).sum()
            failed += (wc.status != 0).sum()
            wc.sync_sim()
            flagged += ((wc.info[:, 3] & (1 | 2 | 4 | 32)) != 0).sum()
            wc.reset_done(sched, t=wc.t)
    wc.sync_sim()
torch.cuda.synchronize()
el = time.perf_counter() - t0
ok = bool(torch.isfinite(wc.q).all() and torch.isfinite(wc.qpos).all() and torch.isfinite(wc.qvel).all())
print(f"{'closed' if closed else 'open'} loop, {n} envs x {steps} steps in {el:.1f} s ({n * steps / el / 1e6:.2f} M env-steps/s): episode resets {int(resets)}, "
      f"failed QPs over ALL ticks {int(failed_every)} (in the samples every 50th step: {int(failed)}), flagged-sim samples {int(flagged)}, states finite {ok}, episodes max {int(sched.episode.max())}, "
      f"base height min {float(wc.q[:, 2].min()):.3f}")
