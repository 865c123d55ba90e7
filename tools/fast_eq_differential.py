"""Differential soak of the tick's fast equality solve (TSIDB_OPT_QP_FAST_EQ) against the QR-only path: N walkers with tight-ish
torque bounds, both controllers fed the SAME state every tick (the fast one's), through lift-offs and touch-downs: status /
iteration-count mismatches and the worst difference of tau, dv, f, q, v.
    python tools/fast_eq_differential.py [envs] [ticks] > gpurun_out/fast_eq_differential.txt"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
def mk(fe, tms):
    conf = op3_walking_conf(RobotConfig()); conf.reference_quirks = False; conf.qp_fast_equalities = fe; conf.tau_max_scaling = tms
    wc = WalkController(conf, num_envs=n)
    wc.set_posture_bias(op3_walking_posture())
    return wc, WalkSchedule.on_device(wc, seed=5, scale_range=(0.3, 1.0))
for tms in (5.0, 0.2):
    a, sa = mk(1, tms); b, sb = mk(0, tms)
    worst = dict(tau=0.0, dv=0.0, f=0.0, q=0.0, v=0.0); mism = 0; iters_mism = 0; slow = 0; fastn = 0
    for i in range(ticks):
        t = i * 0.002
        a.tick(walk=(sa, t)); b.tick(walk=(sb, t))
        mism += int((a.status != b.status).sum()); iters_mism += int((a.info[:, 0] != b.info[:, 0]).sum())
        slow += int((a.info[:, 0] > 1).sum()); fastn += int((a.info[:, 0] == 1).sum())
        for k in worst:
            worst[k] = max(worst[k], float((getattr(a, k) - getattr(b, k)).abs().max()))
        b.q.copy_(a.q); b.v.copy_(a.v)
    torch.cuda.synchronize()
    print(f"tau_max_scaling {tms}: {n} envs x {ticks} ticks; env-ticks decided by the fast solve {fastn}, with active-set iterations {slow}; "
          f"status mismatches {mism}, iteration-count mismatches {iters_mism}; worst |fast - QR|: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items())
          + f"; failed QPs {int((a.status != 0).sum())} (last tick)")
