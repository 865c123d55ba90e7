import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from oracle.oracle import Oracle
from tsid_control_amd.model import ModelBlob
orc = Oracle(ModelBlob().raw)
wc = T.make(64, "f32"); T.perturb(wc, 4); st = T.mirror(wc); wc.tick()
for e in range(64):
    out = orc.tsid_tick(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e], st["contact_ref"][e], st["contact_active"][e], st["cop_frames"][e])
    st["tau"][e], st["dv"][e], st["f"][e] = out["tau"], out["dv"], out["f"]
tau, dv = wc.tau.double().cpu().numpy(), wc.dv.double().cpu().numpy()
w, w0 = T.wrench(wc.f.double().cpu().numpy(), wc.params), T.wrench(st["f"], wc.params)
for name, a, b in (("tau", tau, st["tau"]), ("dv", dv, st["dv"]), ("wrench", w, w0)):
    d = np.abs(a - b)
    print(name, "max abs err", d.max(), "max |ref|", np.abs(b).max(), "max err/(1e-4+1e-3|ref|)", (d / (1e-4 + 1e-3 * np.abs(b))).max())
