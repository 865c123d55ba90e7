"""time one launch of B sim steps (tsidb_sim_batch) against B launches of one step, same snapshots (diagnostic)"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
for n in (512, 1024, 4096):
    for B in (1, 2, 4, 8):
        wc = T.make(n, walking=True, reference_quirks=False, pipeline_sim_batch=B)
        T.perturb(wc, 3, dq=0.02, dv=0.02)
        for _ in range(40):
            wc.step_pipelined()
        wc.sync_sim()
        P = wc._pipe
        slots = list(range(B))
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = 50
        e0.record()
        for _ in range(reps):
            wc._sim_batch(slots)
        e1.record()
        for _ in range(reps):
            for s in slots:
                wc._sim_batch([s])
        e2.record()
        torch.cuda.synchronize()
        print(f"envs {n} B {B}: one launch {e0.elapsed_time(e1) / reps / B * 1e3:.1f} us per step, {B} launches {e1.elapsed_time(e2) / reps / B * 1e3:.1f} us per step")
