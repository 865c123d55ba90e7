"""Tick and sim streams on disjoint CU sets (hipExtStreamCreateWithCUMask): does the tick stop slowing down beside the sim at
small batches?   python tools/cu_mask_experiment.py <envs> <layout: none|halves|evenodd|pairs|quads> [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsid_control_amd import RobotConfig, WalkController
from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture

n, layout = int(sys.argv[1]), sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
torch.cuda.init()
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1])
        break
assert hip is not None


def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


pat = {"halves": None, "evenodd": (0x55555555, 0xAAAAAAAA), "pairs": (0x33333333, 0xCCCCCCCC), "quads": (0x0F0F0F0F, 0xF0F0F0F0),
       "bytes": (0x00FF00FF, 0xFF00FF00), "words": (0x0000FFFF, 0xFFFF0000)}
if layout == "none":
    s_tick, s_sim = torch.cuda.Stream(), None
elif layout == "prio":       # tick stream at high priority
    s_tick, s_sim = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
elif layout == "prio_sim":   # sim stream at high priority
    s_tick, s_sim = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
elif layout == "halves":
    s_tick, s_sim = masked_stream([0xFFFFFFFF] * 4 + [0] * 4), masked_stream([0] * 4 + [0xFFFFFFFF] * 4)
elif layout.startswith("a") and layout[1:].isdigit():   # the first k CUs for the tick, the rest for the sim
    k = int(layout[1:])
    def bits(lo, hi):
        w = [0] * 8
        for c in range(lo, hi):
            w[c // 32] |= 1 << (c % 32)
        return w
    s_tick, s_sim = masked_stream(bits(0, k)), masked_stream(bits(k, 256))
elif layout == "alt32":
    s_tick, s_sim = masked_stream([0xFFFFFFFF, 0] * 4), masked_stream([0, 0xFFFFFFFF] * 4)
else:
    a, b = pat[layout]
    s_tick, s_sim = masked_stream([a] * 8), masked_stream([b] * 8)

conf = op3_walking_conf(RobotConfig())
conf.reference_quirks = False
wc = WalkController(conf, num_envs=n, device="cuda:0")
wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=wc.device).to(wc.dtype)
lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
sched = WalkSchedule.from_demo_paths(n, wc.conf, wc.device, wc.dtype, seed=1, q0_feet=(lf, rf), com0=wc.com_ref[0, :3].double().cpu().numpy())
with torch.cuda.stream(s_tick):
    wc._ensure_pipe()
    if s_sim is not None:
        wc._pipe["stream"] = s_sim
    for i in range(620):
        wc.step_pipelined(walk=(sched, wc.t))
    wc.sync_sim(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        wc.step_pipelined(walk=(sched, wc.t))
    wc.sync_sim(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"envs {n} layout {layout}: {n * steps / el / 1e6:.3f} M env-steps/s, {el / steps * 1e3:.4f} ms per step; failed {int((wc.status != 0).sum())}")
