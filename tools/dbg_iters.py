import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_parity import make, perturb, mirror
from oracle.oracle import Oracle
n = 256
wc = make(n, sim_enabled=False)
orc = Oracle(wc.model.raw)
perturb(wc, 21, dq=0.25, dv=1.5)
wc.contact_active[::3, 0] = 0; wc.contact_active[1::7, 1] = 0
wc.contact_active[(wc.contact_active.sum(dim=1) == 0), 0] = 1
st = mirror(wc)
wc.tick()
it_g = wc.info[:, 0].cpu().numpy(); iq_g = wc.info[:, 1].cpu().numpy()
rows = []
for e in range(n):
    qp = orc.assemble(wc.params, st["q"][e], st["v"][e], st["com_ref"][e], st["posture_ref"][e], st["foot_ref"][e], st["contact_ref"][e], st["contact_active"][e])
    sol = orc.qp_solve(qp["_raw"])
    if sol["status"] == 0 and (sol["iter"] != it_g[e]):
        rows.append((e, int(st["contact_active"][e].sum()), it_g[e], sol["iter"], iq_g[e], sol["iq"], float(np.abs(sol["x"][:26] - wc.dv[e].cpu().numpy()).max())))
print(len(rows), "mismatching")
for r in rows[:30]: print(r)
