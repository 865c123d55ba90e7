"""eager pipeline, sim batch 1 against 8 (and one against two wavefronts per env), walking envs: first step at which the
sim states differ bit for bit.   python tools/dbg_batch_f32.py [f32|f64] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1400


def walker(batch, waves):
    conf = op3_walking_conf(RobotConfig())
    conf.dtype, conf.reference_quirks, conf.pipeline_sim_batch, conf.sim_waves = dtype, False, batch, waves
    wc = WalkController(conf, num_envs=64, device="cuda:0")
    wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
    lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
    sched = WalkSchedule.from_demo_paths(64, wc.conf, wc.device, wc.dtype, seed=2, q0_feet=(lf, rf),
                                         com0=wc.com_ref[0, :3].double().cpu().numpy(), t_start=0.2)
    sched.set_phase_offsets(torch.linspace(0.0, 0.3, 64, dtype=torch.float64))
    return wc, sched


for (ba, wa), (bb, wb) in (((1, 2), (8, 2)), ((1, 1), (8, 1)), ((1, 1), (1, 2))):
    a, sa = walker(ba, wa)
    b, sb = walker(bb, wb)
    first = None
    for i in range(steps):
        sa.apply(a, a.t); a.step_pipelined()
        sb.apply(b, b.t); b.step_pipelined()
        if i % 8 == 7:
            a.sync_sim(); b.sync_sim()
            torch.cuda.synchronize()
            bad = [k for k in ("q", "v", "tau", "qpos", "qvel", "qacc_warmstart", "ncon", "con_pairs") if not torch.equal(getattr(a, k), getattr(b, k))]
            if bad:
                first = (i, bad)
                d = (a.qvel - b.qvel).abs().max(dim=1).values
                e = int(d.argmax())
                print(f"batch/waves {ba}/{wa} vs {bb}/{wb}: first difference at step {i}: {bad}; env {e} |dqvel| {float(d[e]):.3e} ncon {int(a.ncon[e])} {int(b.ncon[e])} "
                      f"info {a.info[e].tolist()} {b.info[e].tolist()}")
                break
    if first is None:
        print(f"batch/waves {ba}/{wa} vs {bb}/{wb}: identical over {steps} steps")
