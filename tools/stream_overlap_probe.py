"""Which pairs of HIP streams really run concurrently?  For successive pairs (torch pool streams, then hipStreamCreateWithFlags
streams, again after a HIP graph launch): 150 ticks on one and 150 sim steps on the other, enqueued together - the time
against the two run one after the other.   python tools/stream_overlap_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController

conf = RobotConfig()
conf.reference_quirks = False
wc = WalkController(conf, num_envs=512, device="cuda:0")
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1]); break


def hip_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0   # hipStreamNonBlocking
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


def probe(sa, sb, label):
    def ticks():
        with torch.cuda.stream(sa):
            for _ in range(150):
                wc.tick()
    def sims():
        with torch.cuda.stream(sb):
            for _ in range(150):
                wc.sim_step(_from_pipe=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ticks(); torch.cuda.synchronize(); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); sims(); torch.cuda.synchronize(); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); ticks(); sims(); torch.cuda.synchronize(); tab = time.perf_counter() - t0
    verdict = "CONCURRENT" if tab < 0.8 * (ta + tb) else "serialised"
    print(f"{label:34s} ticks {ta*1e3:6.2f} ms  sims {tb*1e3:6.2f} ms  together {tab*1e3:6.2f} ms  -> {verdict}", flush=True)


probe(torch.cuda.current_stream(), torch.cuda.current_stream(), "same stream (default)")
pool = [torch.cuda.Stream() for _ in range(10)]
for i in range(0, 10, 2):
    probe(pool[i], pool[i + 1], f"torch pool streams {i},{i+1}")
probe(pool[0], pool[4], "torch pool streams 0,4")
probe(pool[1], pool[5], "torch pool streams 1,5")
hs = [hip_stream() for _ in range(8)]
for i in range(0, 8, 2):
    probe(hs[i], hs[i + 1], f"hipStreamCreate streams {i},{i+1}")
probe(hs[0], hs[4], "hipStreamCreate streams 0,4")
# a HIP graph launch, then again
g = torch.cuda.CUDAGraph()
x = torch.zeros(1024, device="cuda:0")
with torch.cuda.graph(g):
    y = x + 1
g.replay(); torch.cuda.synchronize()
print("-- after a HIP graph launch")
probe(pool[0], pool[1], "torch pool streams 0,1")
hs2 = [hip_stream() for _ in range(6)]
for i in range(0, 6, 2):
    probe(hs2[i], hs2[i + 1], f"new hipStreamCreate streams {i},{i+1}")
pool2 = [torch.cuda.Stream() for _ in range(4)]
probe(pool2[0], pool2[1], "more torch pool streams a,b")
probe(pool2[2], pool2[3], "more torch pool streams c,d")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
