"""does a CU-masked stream pair (512-env controller) slow down a controller created afterwards in the same process?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture


def run(n, steps=800, keep=None):
    conf = op3_walking_conf(RobotConfig())
    conf.reference_quirks = False
    wc = WalkController(conf, num_envs=n, device="cuda:0")
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].double().cpu().numpy())
    with torch.cuda.stream(wc.tick_stream):
        for i in range(620):
            wc.step_pipelined(walk=(sched, wc.t))
        wc.sync_sim(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            wc.step_pipelined(walk=(sched, wc.t))
        wc.sync_sim(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"envs {n}: {n * steps / el / 1e6:.3f} M env-steps/s", flush=True)
    if keep is not None:
        keep.append(wc)


mode = sys.argv[1] if len(sys.argv) > 1 else "destroy"
run(1024)
held = [] if mode == "keep" else None
run(512, keep=held)
gc.collect()
run(1024)
run(2048)
run(512)
run(1024)
