"""How long the pipelined step takes to settle after a device synchronisation (the driver's 20-step window starts at one):
per-step period (tick start to next tick start) and per-kernel durations over the first steps after the sync.
    python tools/pipeline_transient.py [envs] [steps] > gpurun_out/pipeline_transient.txt"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 120
conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False
wc = WalkController(conf, num_envs=N)
dev = wc.device
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=dev).to(wc.dtype)
sched = WalkSchedule.from_demo_paths(N, conf, dev, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].double().cpu().numpy())
with torch.cuda.stream(wc.tick_stream):
    for i in range(605):
        wc.step_pipelined(walk=(sched, i * conf.dt))
    wc.sync_sim(); torch.cuda.synchronize()
    for rep in range(3):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
        base = 605 + rep * K
        for k in range(K):
            wc.step_pipelined(events=ev[k], walk=(sched, (base + k) * conf.dt))
        wc.sync_sim(); torch.cuda.synchronize()
        t0 = ev[0][0]
        ts = np.array([[t0.elapsed_time(e) for e in row] for row in ev])   # ms since the first tick's start
        per = np.diff(ts[:, 0])
        print(f"rep {rep}: steps after the sync -> period ms (tick start to tick start), k_tick ms, k_sim ms")
        for lo in range(0, K - 1, 10):
            hi = min(lo + 10, K - 1)
            print(f"  steps {lo:3d}-{hi:3d}: period {per[lo:hi].mean():.4f}  tick {np.mean(ts[lo:hi, 1] - ts[lo:hi, 0]):.4f}  sim {np.mean(ts[lo:hi, 3] - ts[lo:hi, 2]):.4f}  sim start - tick start {np.mean(ts[lo:hi, 2] - ts[lo:hi, 0]):.4f}")
        print(f"  whole: {K} steps in {ts[-1, 3]:.3f} ms = {N * K / ts[-1, 3] / 1e3:.2f} M env-steps/s; first 20: {N * 20 / ts[19, 3] / 1e3:.2f} M")
