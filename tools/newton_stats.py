"""CPU diagnostic (oracle only): the config-3 walking workload through or_env_step_batch_walk for a few envs; per window of
ticks the histogram of rows whose state changed between Newton iterations - what decides between rank-1 updates of the
factor and a rebuild (OR_NEWTON_INCR_MAX).  Also checks that the state does not depend on that constant beyond rounding.

    python tools/newton_stats.py [envs] [ticks]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle.oracle import Oracle, WalkTables, new_state  # noqa: E402
from tsid_control_amd.conf import RobotConfig  # noqa: E402
from tsid_control_amd.model import ModelBlob  # noqa: E402
from tsid_control_amd.params import pack_params  # noqa: E402
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture  # noqa: E402


def run(n, ticks, incr_max, windows):
    conf = op3_walking_conf(RobotConfig())
    conf.reference_quirks = False
    mb = ModelBlob(conf.model_blob)
    orc = Oracle(mb.raw)
    C.c_int.in_dll(orc.lib, "or_newton_incr_max").value = incr_max
    params = pack_params(conf, mb.effort_limit, mb.velocity_limit)
    q0 = np.array(mb.q0, dtype=np.float64)
    t0 = orc.terms(q0, np.zeros(26))
    q0[2] -= t0["oMf"][0, 11]
    t0 = orc.terms(q0, np.zeros(26))
    fr = t0["oMf"].copy()
    lf, rf = fr[0, 9:11], fr[1, 9:11]
    sched = WalkSchedule.from_demo_paths(n, conf, "cpu", torch.float64, seed=1, q0_feet=(lf, rf), com0=t0["com"])
    st = new_state(n)
    to_se3 = lambda f: np.concatenate([f[9:12], f[:9].reshape(3, 3).T.reshape(-1)])
    st["q"][:] = q0
    st["qpos"][:] = np.concatenate([q0[:3], q0[[6, 3, 4, 5]], q0[7:][np.asarray(mb["mj_ctrl_qidx"]) - 7]])
    st["com_ref"][:, :3] = t0["com"]
    st["posture_ref"][:] = q0[7:] + op3_walking_posture()
    st["contact_ref"][:] = np.stack([to_se3(fr[0]), to_se3(fr[1])]).reshape(st["contact_ref"].shape[1:])
    st["cop_frames"][:] = fr.reshape(st["cop_frames"].shape[1:])
    st["foot_ref"][:] = 0
    st["frames"] = np.tile(fr.reshape(1, 2, 12), (n, 1, 1)).copy()
    tab = WalkTables(sched, n)
    hist = (C.c_long * 200)()
    out = {}
    lo = 0
    for name, hi in windows:
        orc.lib.or_newton_hist_get(hist, 1)
        for i in range(lo, min(hi, ticks)):
            orc.env_step_batch(params, st, nthreads=8, walk=tab.at(i * conf.dt))
        orc.lib.or_newton_hist_get(hist, 1)
        out[name] = np.array(hist[:40])
        lo = hi
    return st, out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    wins = [("start 0-500", 500), ("500-600", 600), ("600-640 (touch-down window of the driver's bench)", 640), ("640-760", 760), ("760-900", 900)]
    a, ha = run(n, ticks, 0, wins)
    b, hb = run(n, ticks, 8, wins)
    for name, _ in wins:
        h = hb[name]
        tot = h.sum()
        if tot == 0:
            continue
        print(f"{name}: Newton iterations after the first {tot}; rows changed 0: {h[0]}, 1-2: {h[1:3].sum()}, 3-4: {h[3:5].sum()}, "
              f"5-8: {h[5:9].sum()}, 9-16: {h[9:17].sum()}, >16: {h[17:].sum()}   mean {np.dot(h, np.arange(len(h))) / tot:.2f}")
    for k in ("q", "qpos", "qvel", "tau"):
        print(k, "max |rebuild-always - incremental|", np.abs(a[k] - b[k]).max())
