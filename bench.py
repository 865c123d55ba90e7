#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched TSID + contact-dynamics hot path on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: for every env one TSID tick (main.py:119-129)
and one sim step (main.py:192-195), preceded by the walking reference update that config 3 needs
(footstep schedule -> update_tasks) and, for N > 1, followed by the RCCL all-gather of the
observations.  Workload = BASELINE.json configs[2] "4096 OP3 LIPM walking, 1 MI355X" per GPU (weak
scaling: configs[3] = 8 x 4096 at N = 8).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# SURVEY.md 8(d) accounting, split per kernel (words of the arithmetic type per env per launch):
#   k_tick reads q 27, v 26, CoM ref 9, posture ref 20, foot refs 48, contact refs 24, flags 2 (156)
#          writes q 27, v 26, tau 20, dv 26, f 24, status 1, obs 65 (189)
#   k_sim  reads TSID q 27, qpos 27, qvel 26, qacc_warmstart 26 (106); writes qpos 27, qvel 26, qacc_warmstart 26 (79)
TICK_WORDS = 156 + 189
SIM_WORDS = 106 + 79
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}
NOMINAL_FLOP_PER_ENV_STEP = 0.5e6  # SURVEY.md 8(d), structure-exploiting estimate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000, help="timed steps; 5000 ticks = the 10 s of SURVEY.md 8(d) cfg 3")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=600,
                    help="walk workload only: untimed steps of state initialisation before the warm-up, so that the "
                         "timed window starts in steady walking (the plan's first second is a double-support start)")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--workload", choices=["walk", "stand"], default="walk")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables it")
    ap.add_argument("--cpu-sample", type=int, default=256)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --envs-per-gpu is the WHOLE job's env count, split evenly over the ranks "
                         "(default is weak scaling: that many envs on every GPU)")
    ap.add_argument("--randomize", action="store_true",
                    help="BASELINE configs[4] (not the headline): per-env mass scale U(0.8,1.2), friction U(0.4,1.0) and a "
                         "floor plane tilted by up to 5 degrees in the sim (seed 2); use with --envs-per-gpu 65536")
    ap.add_argument("--event-every", type=int, default=int(os.environ.get("TSIDB_EVENT_EVERY", "8")),
                    help="bracket the kernels with HIP timing events on every k-th timed step only")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: run the obs all-gather on the tick stream instead of a side stream")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run tick and sim back to back on one stream instead of overlapping sim(t) with tick(t+1)")
    return ap.parse_args()


def host_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box hands each 1-GPU job a 16-CPU share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cap = int(os.environ.get("TSIDB_CPU_THREADS", "16"))
    return max(1, min(n, cap))


def cpu_baseline(wc, seconds, sample):
    """The oracle (a CPU port, float64) timed on this box's host cores, on a bounded sample of the
    same workload: the first `sample` envs' state and references as they stand after warm-up."""
    from oracle.oracle import Oracle, new_state
    n = min(sample, wc.num_envs)
    orc = Oracle(wc.model.raw)
    st = new_state(n)
    for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames", "contact_active"):
        st[k][...] = getattr(wc, k)[:n].double().cpu().numpy().reshape(st[k].shape) if k != "contact_active" \
            else wc.contact_active[:n].cpu().numpy()
    st["qacc_ws"][...] = wc.qacc_warmstart[:n].double().cpu().numpy()
    cores = host_cores()
    orc.env_step_batch(wc.params, st, nthreads=cores)  # warm
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        orc.env_step_batch(wc.params, st, nthreads=cores)
        reps += 1
    el = time.perf_counter() - t0
    # single-thread figure on a smaller slice
    st1 = {k: np.ascontiguousarray(v[:32]) for k, v in st.items()}
    t1 = time.perf_counter()
    r1 = 0
    while time.perf_counter() - t1 < min(3.0, seconds / 4):
        orc.env_step_batch(wc.params, st1, nthreads=1)
        r1 += 1
    el1 = time.perf_counter() - t1
    return dict(value=n * reps / el, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{n} envs (state + references captured after warm-up, references frozen) x {reps} env steps, "
                       f"oracle/liboracle.so float64, OpenMP over envs",
                value_1thread=32 * r1 / el1)


def main():
    args = parse()
    from tsid_control_amd import RobotConfig, WalkController
    from tsid_control_amd.sharding import ObsGather, init_distributed
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
    import torch.distributed as dist

    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # TSIDB_BENCH_ONE_DEVICE=1 (+ TSIDB_DIST_BACKEND=gloo) rehearses the N > 1 plumbing on a 1-GPU box
    one_dev = os.environ.get("TSIDB_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda", local if (world > 1 and not one_dev) else 0)
    torch.cuda.set_device(dev)
    n = args.envs_per_gpu
    if args.strong:
        from tsid_control_amd.sharding import shard_range
        lo, hi = shard_range(args.envs_per_gpu, rank, world)
        if (hi - lo) * world != args.envs_per_gpu:
            raise SystemExit("--strong needs an env count divisible by the number of GPUs")
        n = hi - lo
    conf = RobotConfig()
    conf.dtype = args.dtype
    if args.workload == "walk":
        # OP3-sized steps and walking task weights (walk_planner.op3_walking_conf explains why conf.py's
        # own values cannot walk); the sim follows the true base orientation (quirk F6a would tip it over
        # as soon as the path turns)
        op3_walking_conf(conf)
        conf.reference_quirks = False
    wc = WalkController(conf, num_envs=n, device=dev)
    torch.manual_seed(1 + rank)
    if args.randomize:
        wc.randomize(seed=2 + rank)
    sched = None
    if args.workload == "walk":
        lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
        wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=dev).to(wc.dtype)
        sched = WalkSchedule.from_demo_paths(n, conf, dev, wc.dtype, seed=1 + rank, q0_feet=(lf, rf),
                                             com0=wc.com_ref[0, :3].double().cpu().numpy())
    else:  # config 2: perturbed standing
        wc.q[:, 7:] += (torch.rand(n, 20, dtype=wc.dtype, device=dev) - 0.5) * 0.1
        wc.v[:] = torch.randn(n, 26, dtype=wc.dtype, device=dev) * 0.05
    gather = ObsGather(n, 65, world, wc.dtype, dev)
    total = args.warmup + args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] if k % args.event_every == 0 else None
          for k in range(args.steps)]

    # The sim stage of step t only needs the TSID state tick t produced, and tick t+1 does not depend on
    # sim t (the reference couples them one way, main.py:192-195): WalkController.step_pipelined() leaves
    # sim(t) running on a second HIP stream while tick(t+1) runs on the first.
    overlap = not args.no_overlap
    s_tick = torch.cuda.current_stream(dev)
    # N > 1: the all-gather of step t's observations runs on a third stream from a two-slot snapshot, so the
    # collective's latency is off the tick stream's critical path (the next tick overwrites wc.obs)
    side_gather = world > 1 and not args.sync_gather
    s_comm = torch.cuda.Stream(device=dev) if side_gather else None
    obs_snap = [torch.empty_like(wc.obs), torch.empty_like(wc.obs)] if side_gather else None
    comm_done = [None, None]

    def one_step(i, timed_idx=None):
        par = i & 1
        if sched is not None:
            sched.apply(wc, i * conf.dt)
        e = ev[timed_idx] if timed_idx is not None else None
        if overlap:
            wc.step_pipelined(events=e)
        else:
            if e: e[0].record(s_tick)
            wc.tick()
            if e: e[1].record(s_tick)
            if e: e[2].record(s_tick)
            wc.sim_step()
            if e: e[3].record(s_tick)
        if side_gather:
            if comm_done[par] is not None:
                s_tick.wait_event(comm_done[par])
            obs_snap[par].copy_(wc.obs)
            snap = torch.cuda.Event()
            snap.record(s_tick)
            with torch.cuda.stream(s_comm):
                s_comm.wait_event(snap)
                gather(obs_snap[par])
                comm_done[par] = torch.cuda.Event()
                comm_done[par].record(s_comm)
        else:
            gather(wc.obs)

    failed_any = torch.zeros(n, dtype=torch.bool, device=dev)
    pre = args.preroll if args.workload == "walk" else 0
    for i in range(pre + args.warmup):
        one_step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(pre + args.warmup + k, k)
        if k % 64 == 63:
            failed_any |= wc.status != 0   # sampled every 64th step: one tiny kernel, not per step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    elt = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
    el = float(elt.item())

    evs = [e for e in ev if e is not None]
    tick_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
    sim_ms = sum(e[2].elapsed_time(e[3]) for e in evs) / len(evs)
    wsz = 8 if args.dtype == "f64" else 4
    dom, dom_ms, dom_words = ("k_tick", tick_ms, TICK_WORDS) if tick_ms >= sim_ms else ("k_sim", sim_ms, SIM_WORDS)
    alg_bytes = n * dom_words * wsz
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    n_bad = int((wc.status != 0).sum().item())
    qp_it = wc.info[:, 0].float()
    stats = {"qp_iters_mean": float(qp_it.mean()), "qp_iters_max": int(qp_it.max()),
             "frac_envs_in_active_set_loop": float((qp_it > 1).float().mean()),
             "active_rows_mean": float(wc.info[:, 1].float().mean()), "ncon_mean": float(wc.ncon.float().mean()),
             "newton_iters_mean": float(wc.info[:, 2].float().mean()),
             "single_support_frac": float((wc.contact_active.sum(dim=1) == 1).float().mean()),
             "envs_with_a_failed_qp_in_sampled_steps": int(failed_any.sum().item()),
             "com_tracking_err_max_m": float((wc.obs[:, 53:56] - wc.com_ref[:, :3]).abs().max()),
             "base_height_min_m": float(wc.q[:, 2].min())}

    traffic = traffic_x2 = None
    tf = ROOT / "profiles" / "pmc_traffic.json"
    if tf.exists() and args.dtype == "f64" and n == 4096:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_profile.sh), committed under profiles/
        rec = json.loads(tf.read_text()).get(dom, {})
        traffic, traffic_x2 = rec.get("bytes_per_launch"), rec.get("bytes_per_launch_fetch_x2")

    if rank == 0:
        value = world * n * args.steps / el
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"cfg3: {n} OP3 LIPM walking per GPU (footstep plan along the demo path, LIPM/DCM CoM "
                                    "reference + swing trajectories -> update_tasks each tick; TSID tick + sim step)"
                                    + ("; cfg5 randomised mass / friction / floor tilt" if args.randomize else "")
                                    if args.workload == "walk" else
                                    "cfg2: perturbed stand/balance per GPU"),
                       "envs_per_gpu": n, "global_envs": n * world, "preroll_steps": pre, "parallelism": f"env-sharded x{world}, obs all-gather" + (" on a side stream" if side_gather else ""),
                       "streams": "sim(t) overlapped with tick(t+1) on a second HIP stream" if overlap else "single stream",
                       "qp_failed_envs_last_step": n_bad, "last_step_stats": stats},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_fetch_x2": traffic_x2,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                         "k_tick_ms": tick_ms, "k_sim_ms": sim_ms,
                         "valu_frac_nominal": (n * NOMINAL_FLOP_PER_ENV_STEP / ((tick_ms + sim_ms) * 1e-3)) / (VALU_PEAK_TFLOPS[args.dtype] * 1e12)},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(wc, args.cpu_seconds, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
