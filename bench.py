#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched TSID + contact-dynamics hot path on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: for every env one TSID tick (main.py:119-129)
and one sim step (main.py:192-195), preceded by the walking reference update that config 3 needs
(footstep schedule -> update_tasks) and, for N > 1, followed by the RCCL all-gather of the
observations, reward and done flags.

Workload = BASELINE.json's metric: "env-steps/sec (whole node), 4096 OP3 walkers at 1/2/4/8 MI355X" -
configs[2] (4096 OP3 LIPM walking) at N = 1, and for N > 1 the SAME 4096 walkers split evenly over the
ranks (strong scaling: 2048 / 1024 / 512 per GPU).  `--weak` keeps --envs walkers on EVERY
GPU instead (configs[3]: 8 x 4096 = 32768 at N = 8).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line; at N = 1 it also carries `secondary`: the unfavourable paths
measured beside the headline (cfg2 stand at 1024 and 4096 envs, a de-phased walking batch that mixes
50- and 38-variable QPs in one launch, and a walking batch with tightened torque bounds that drives
envs into the dual active-set loop).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# SURVEY.md 8(d) accounting, split per kernel (words of the arithmetic type per env per launch):
#   k_tick reads q 27, v 26, CoM ref 9, posture ref 20, foot refs 48, contact refs 24, flags 2 (156)
#          writes q 27, v 26, tau 20, dv 26, f 24, status 1, obs 65, reward 1, done 1 (191)
#   k_sim  reads TSID q 27, qpos 27, qvel 26, qacc_warmstart 26 (106); writes qpos 27, qvel 26, qacc_warmstart 26 (79)
TICK_WORDS = 156 + 191
SIM_WORDS = 106 + 79
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}
NOMINAL_FLOP_PER_ENV_STEP = 0.5e6  # SURVEY.md 8(d), structure-exploiting estimate
USEFUL_LANES = 26              # dofs of the v1 robot on a 64-lane wavefront (the matrix phases' active lanes)
WAVES_PER_SIMD = {"f64": 2, "f32": 2}   # residency of k_tick / k_sim (profiles/*kernel_resource_usage.txt: 256 VGPRs, 20 KB LDS in
                                        # float64; the float32 k_sim is compiled for 2 as well, its k_tick alone reaches 3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000, help="timed steps; 5000 ticks = the 10 s of SURVEY.md 8(d) cfg 3")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=600,
                    help="walk workload only: untimed steps of state initialisation before the warm-up, so that the "
                         "timed window starts in steady walking (the plan's first second is a double-support start)")
    ap.add_argument("--envs", type=int, default=4096,
                    help="env count of the WHOLE job, split evenly over the ranks (strong scaling, the metric's definition)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: --envs walkers on EVERY GPU (BASELINE configs[3] = 8 x 4096)")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="same as --weak --envs N")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--workload", choices=["walk", "stand"], default="walk")
    ap.add_argument("--closed-loop", action="store_true",
                    help="walk workload with the loop closed (SURVEY 8f-1): the tick reads the sim state, the sim is driven by "
                         "tau, touch-downs follow the sim's contact list (walk_planner.op3_closed_loop_walking_conf); tick and sim "
                         "cannot overlap then")
    ap.add_argument("--robot", choices=["v1", "v0"], default="v1",
                    help="v0: the reference's second robot (robot/v0, libtsidb_v0.so; SURVEY 8f-4) - standing workload only "
                         "(the reference has no walking configuration for it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables it")
    ap.add_argument("--cpu-sample", type=int, default=256)
    ap.add_argument("--randomize", action="store_true",
                    help="BASELINE configs[4] (not the headline): per-env mass scale U(0.8,1.2), friction U(0.4,1.0), a floor "
                         "plane tilted by up to 5 degrees and 1 cm terrain steps in the sim (seed 2); use with --envs 65536")
    ap.add_argument("--dephase", type=float, default=0.0,
                    help="walk workload: per-env start delays U(0, DEPHASE) seconds, so that double- and single-support "
                         "ticks mix in one launch")
    ap.add_argument("--tau-max-scaling", type=float, default=None,
                    help="override conf.tau_max_scaling (ctrl/conf.py:69; 5.0): small values make torque bounds active")
    ap.add_argument("--graph", type=int, default=0,
                    help="N = 1, walk workload: capture this many pipelined steps in ONE HIP graph (WalkController.capture_steps) "
                         "and replay it - one launch instead of ~8 host calls per step; per-kernel event timing is not "
                         "available inside a graph")
    ap.add_argument("--sim-batch", type=int, default=0,
                    help="conf.pipeline_sim_batch: sim stages handed to the second stream this many at a time, in one launch "
                         "(0 = the default: 8 up to 1024 envs, else 1 - WalkController.sim_batch)")
    ap.add_argument("--device-plan", action="store_true",
                    help="walk workload: the episode plans are BUILT ON THE DEVICE (tsidb_walk_plan / k_plan: footsteps, swing "
                         "polynomials, DCM / LIPM CoM plan per env, path scales U(0.5, 1) drawn per env) and the episode lifecycle "
                         "runs there as well (tsidb_reset_done + replan every 50 steps: envs that fell restart on a new path) - "
                         "the path an RL user runs; default: the host-planned schedule (WalkSchedule.from_demo_paths)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (N = 1 only)")
    ap.add_argument("--secondary-steps", type=int, default=200)
    ap.add_argument("--event-every", type=int, default=int(os.environ.get("TSIDB_EVENT_EVERY", "8")),
                    help="bracket the kernels with HIP timing events on every k-th timed step only")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: run the obs all-gather on the tick stream instead of a side stream")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run tick and sim back to back on one stream instead of overlapping sim(t) with tick(t+1)")
    ap.add_argument("--self-collision", type=int, default=1,
                    help="1 = robot<->robot hull pairs collided in the sim (mj_step's behaviour), 0 = floor contacts only")
    a = ap.parse_args()
    if a.envs_per_gpu is not None:
        a.weak, a.envs = True, a.envs_per_gpu
    return a


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


def cpu_baseline(wc, sched, t_now, seconds, sample):
    """The oracle (a CPU port, float64) timed on this box's host cores, on a bounded sample of the same
    workload: the first `sample` envs continue from the state the GPU run left them in - for the walking
    workload WITH the per-tick walking reference update (oracle/or_walk.c), i.e. the schedule keeps
    stepping; for the standing workload the references are constant anyway."""
    from oracle.oracle import Oracle, WalkTables, new_state
    n = min(sample, wc.num_envs)
    orc = Oracle(wc.model.raw)
    st = new_state(n, (wc.NQ, wc.NV, wc.NA))
    for k in ("q", "v", "qpos", "qvel", "com_ref", "posture_ref", "foot_ref", "contact_ref", "cop_frames", "contact_active"):
        st[k][...] = getattr(wc, k)[:n].double().cpu().numpy().reshape(st[k].shape) if k != "contact_active" \
            else wc.contact_active[:n].cpu().numpy()
    st["qacc_ws"][...] = wc.qacc_warmstart[:n].double().cpu().numpy()
    st["frames"] = np.ascontiguousarray(wc.frames[:n].double().cpu().numpy())
    if wc.env_params is not None:
        st["env_params"] = np.ascontiguousarray(wc.env_params[:n].double().cpu().numpy())
    tables = WalkTables(sched, n) if sched is not None else None
    cores = host_cores()
    dt = wc.conf.dt
    tt = [t_now]

    def one(state, nthreads, tab):
        orc.env_step_batch(wc.params, state, nthreads=nthreads, walk=tab.at(tt[0]) if tab is not None else None)
        tt[0] += dt

    one(st, cores, tables)  # warm
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        one(st, cores, tables)
        reps += 1
    el = time.perf_counter() - t0
    # single-thread figure on a smaller slice
    st1 = {k: np.ascontiguousarray(v[:32]) for k, v in st.items()}
    tab1 = WalkTables(sched, min(32, n)) if sched is not None else None
    t1 = time.perf_counter()
    r1 = 0
    while time.perf_counter() - t1 < min(3.0, seconds / 4):
        one(st1, 1, tab1)
        r1 += 1
    el1 = time.perf_counter() - t1
    what = ("walking reference update + TSID tick + sim step per env step, the schedule keeps stepping"
            if sched is not None else "TSID tick + sim step per env step")
    return dict(value=n * reps / el, unit="env-steps/s", cores=cores, kind="port",
                sample=f"first {n} envs of the GPU run's final state x {reps} env steps ({what}); "
                       f"oracle/liboracle.so float64, OpenMP over envs",
                value_1thread=min(32, n) * r1 / el1)


def run_workload(a, dev, rank, world, n, with_gather=True):
    """Build the workload `a` describes on `n` local envs, run preroll + warm-up untimed and a.steps timed.
    Returns (result dict, wc, sched, time of the next tick)."""
    from tsid_control_amd import RobotConfig, WalkController
    from tsid_control_amd.sharding import ObsGather
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
    import torch.distributed as dist

    robot = getattr(a, "robot", "v1")
    if robot == "v0":
        from tsid_control_amd import op3_v0_conf
        if a.workload != "stand":
            raise SystemExit("--robot v0 runs the standing workload only: add --workload stand")
        conf = op3_v0_conf()
    else:
        conf = RobotConfig()
    conf.dtype = a.dtype
    conf.self_collision = bool(a.self_collision)
    if a.workload == "walk":
        # OP3-sized steps and walking task weights (walk_planner.op3_walking_conf explains why conf.py's
        # own values cannot walk); the sim follows the true base orientation (quirk F6a would tip it over
        # as soon as the path turns)
        if getattr(a, "closed_loop", False):
            from tsid_control_amd.walk_planner import op3_closed_loop_walking_conf
            op3_closed_loop_walking_conf(conf)
        else:
            op3_walking_conf(conf)
        conf.reference_quirks = False
    if a.tau_max_scaling is not None:
        conf.tau_max_scaling = a.tau_max_scaling
    conf.pipeline_sim_batch = int(getattr(a, "sim_batch", 0))
    conf.sim_waves = int(os.environ.get("TSIDB_SIM_WAVES", "0"))   # (diagnostic override; 0 = the library's choice)
    wc = WalkController(conf, num_envs=n, device=dev)
    torch.manual_seed(1 + rank)
    if a.randomize:
        wc.randomize(seed=2 + rank)
    sched = None
    if a.workload == "walk":
        lf, rf = wc.frames[0, 0, 9:11].cpu().numpy(), wc.frames[0, 1, 9:11].cpu().numpy()
        if getattr(a, "device_plan", False):
            if getattr(a, "closed_loop", False) or a.dephase > 0:
                raise SystemExit("--device-plan runs the open-loop cfg3 workload without start delays")
            wc.set_posture_bias(op3_walking_posture())   # (reset_done restores the walking posture of an env it restarts)
            sched = WalkSchedule.on_device(wc, seed=1 + rank)
        else:
            wc.posture_ref += torch.as_tensor(op3_walking_posture(), device=dev).to(wc.dtype)
            sched = WalkSchedule.from_demo_paths(n, conf, dev, wc.dtype, seed=1 + rank, q0_feet=(lf, rf),
                                                 com0=wc.com_ref[0, :3].double().cpu().numpy(),
                                                 **({"foot_press": 0.0} if getattr(a, "closed_loop", False) else {}))
        if getattr(a, "closed_loop", False):
            sched.enable_touchdown_feedback()
        if a.dephase > 0:
            g = torch.Generator().manual_seed(7 + rank)
            sched.set_phase_offsets(torch.rand(n, generator=g, dtype=torch.float64) * a.dephase)
    else:  # config 2: perturbed standing
        wc.q[:, 7:] += (torch.rand(n, wc.NA, dtype=wc.dtype, device=dev) - 0.5) * 0.1
        wc.v[:] = torch.randn(n, wc.NV, dtype=wc.dtype, device=dev) * 0.05
    # the gathered row per env: obs[65] + reward + done (SURVEY.md 8e "obs (+reward, done)")
    gather = ObsGather(n, wc.gather_width, world, wc.dtype, dev) if with_gather else None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] if k % a.event_every == 0 else None
          for k in range(a.steps)]

    # The sim stage of step t only needs the TSID state tick t produced, and tick t+1 does not depend on
    # sim t (the reference couples them one way, main.py:192-195): WalkController.step_pipelined() leaves
    # sim(t) running on a second HIP stream while tick(t+1) runs on the first.
    overlap = not a.no_overlap and not getattr(a, "closed_loop", False)
    # the pipelined loop runs on the stream the library recommends (WalkController.tick_stream: for up to 512 envs the tick and
    # the sim stream sit on disjoint halves of the CUs); everything of this workload from here on is enqueued on it
    use_lib_stream = overlap and not getattr(a, "graph", 0) and os.environ.get("TSIDB_BENCH_TICK_STREAM", "1") == "1"
    s_tick = wc.tick_stream if use_lib_stream else torch.cuda.current_stream(dev)   # (graph capture brings its own stream)
    torch.cuda.current_stream(dev).synchronize()
    _stream_ctx = torch.cuda.stream(s_tick)
    _stream_ctx.__enter__()
    # N > 1: the all-gather of step t's rows runs on a third stream from a two-slot snapshot, so the
    # collective's latency is off the tick stream's critical path (the next tick overwrites the rows)
    side_gather = with_gather and world > 1 and not a.sync_gather
    s_comm = None
    if side_gather:
        # a third stream for the collective that shares a hardware queue with neither of the other two (HIP multiplexes
        # streams onto a few queues; WalkController._streams_overlap)
        if overlap:
            wc._ensure_pipe()
        others = [s_tick] + ([wc._pipe["stream"]] if getattr(wc, "_pipe", None) else [])
        for _ in range(6):
            s_comm = torch.cuda.Stream(device=dev)
            if all(wc._streams_overlap(o, s_comm) for o in others):
                break
    snap_buf = [torch.empty(n, wc.gather_width, dtype=wc.dtype, device=dev) for _ in range(2)] if side_gather else None
    comm_done = [None, None]

    dev_plan = bool(getattr(a, "device_plan", False)) and sched is not None
    if dev_plan and not overlap:
        raise SystemExit("--device-plan runs the pipelined step (drop --no-overlap)")

    def one_step(i, timed_idx=None):
        par = i & 1
        e = ev[timed_idx] if timed_idx is not None else None
        if overlap:
            # the walking reference update of this tick runs in the tick's own launch (tsidb_tick_walk)
            wc.step_pipelined(events=e, walk=(sched, i * conf.dt) if sched is not None else None)
            if dev_plan and i % 50 == 49:   # episode lifecycle on the device: done envs restart on a new plan, no host sync
                wc.reset_done(sched, t=(i + 1) * conf.dt)
        else:
            if sched is not None:
                sched.apply(wc, i * conf.dt)
            if getattr(a, "closed_loop", False):   # one call: the tick reads the sim state, the sim takes tau
                if e: e[0].record(s_tick)
                wc.step()
                if e: e[1].record(s_tick); e[2].record(s_tick); e[3].record(s_tick)   # (k_tick_ms = tick + sim of that call)
            else:
                if e: e[0].record(s_tick)
                wc.tick()
                if e: e[1].record(s_tick)
                if e: e[2].record(s_tick)
                wc.sim_step()
                if e: e[3].record(s_tick)
        if gather is None:
            return
        if side_gather:
            if comm_done[par] is not None:
                s_tick.wait_event(comm_done[par])
            wc.gather_rows(out=snap_buf[par])
            snap = torch.cuda.Event()
            snap.record(s_tick)
            with torch.cuda.stream(s_comm):
                s_comm.wait_event(snap)
                gather(snap_buf[par])
                comm_done[par] = torch.cuda.Event()
                comm_done[par].record(s_comm)
        else:
            gather(wc.gather_rows())

    graph_steps = getattr(a, "graph", 0) if (gather is None or world == 1) and overlap and sched is not None else 0
    failed_any = torch.zeros(n, dtype=torch.bool, device=dev)
    loop_frac = torch.zeros((), dtype=torch.float32, device=dev)   # envs in the dual active-set loop, mean over samples
    ds_frac = torch.zeros((), dtype=torch.float32, device=dev)
    n_samples = n_stat = 0
    stride = max(1, min(64, a.steps))
    pre = a.preroll if a.workload == "walk" else 0
    for i in range(pre + a.warmup):
        one_step(i)
    # first use of the sampling expressions loads their kernels: do that outside the timed region
    failed_any |= wc.status != 0
    loop_frac += (wc.info[:, 0] > 1).float().mean()
    ds_frac += (wc.contact_active.sum(dim=1) == 2).float().mean()
    failed_any.zero_(); loop_frac.zero_(); ds_frac.zero_()
    wc.sync_sim()          # nothing of the warm-up is left for the timed region
    if world > 1 and with_gather:
        dist.barrier()
    torch.cuda.synchronize()
    if graph_steps:
        if a.steps % graph_steps:
            raise SystemExit("--steps must be a multiple of --graph")
        wc.t = (pre + a.warmup) * conf.dt
        gsb = os.environ.get("TSIDB_GRAPH_SIM_BATCH")
        graph = wc.capture_steps(graph_steps, sched, sim_batch=int(gsb) if gsb else None)
        graph.replay()                     # first launch uploads the graph: outside the timed region
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps // graph_steps):
            graph.replay()
            if gather is not None:
                gather(wc.gather_rows())
            failed_any |= wc.status != 0
            n_samples += 1
            loop_frac += (wc.info[:, 0] > 1).float().mean()
            ds_frac += (wc.contact_active.sum(dim=1) == 2).float().mean()
            n_stat += 1
    else:
        t0 = time.perf_counter()
        for k in range(a.steps):
            one_step(pre + a.warmup + k, k)
            if k % stride == stride - 1 or k == a.steps - 1:
                failed_any |= wc.status != 0   # sampled: a few tiny kernels every 64th step, device-side accumulation, no host sync
                n_samples += 1
                loop_frac += (wc.info[:, 0] > 1).float().mean()
                ds_frac += (wc.contact_active.sum(dim=1) == 2).float().mean()
                n_stat += 1
    wc.sync_sim()          # (sim stages step_pipelined() has not enqueued yet belong to the timed steps)
    torch.cuda.synchronize()
    if world > 1 and with_gather:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1 and with_gather:
        elt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
        el = float(elt.item())
    wc.sync_sim()
    torch.cuda.synchronize()

    evs = [e for e in ev if e is not None]
    if graph_steps:
        tick_ms = sim_ms = float("nan")    # no HIP events inside a captured graph
    else:
        tick_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
        sim_ms = sum(e[2].elapsed_time(e[3]) for e in evs) / len(evs)
    qp_it = wc.info[:, 0].float()
    stats = {"qp_iters_mean": float(qp_it.mean()), "qp_iters_max": int(qp_it.max()),
             "frac_envs_in_active_set_loop": float((qp_it > 1).float().mean()),
             "frac_envs_in_active_set_loop_window_mean": float(loop_frac) / max(n_stat, 1),
             "double_support_frac_window_mean": float(ds_frac) / max(n_stat, 1),
             "active_rows_mean": float(wc.info[:, 1].float().mean()), "ncon_mean": float(wc.ncon.float().mean()),
             "newton_iters_mean": float(wc.info[:, 2].float().mean()),
             "single_support_frac": float((wc.contact_active.sum(dim=1) == 1).float().mean()),
             "double_support_frac": float((wc.contact_active.sum(dim=1) == 2).float().mean()),
             "envs_with_a_failed_qp_in_sampled_steps": int(failed_any.sum().item()), "failed_qp_samples": n_samples,
             "qp_failed_envs_last_step": int((wc.status != 0).sum().item()),
             "sim_flagged_envs_last_step": int((wc.info[:, 3] != 0).sum().item()),
             "done_envs_last_step": int(wc.done.sum().item()),
             "com_tracking_err_max_m": float((wc.obs[:, 53:56] - wc.com_ref[:, :3]).abs().max()),
             "base_height_min_m": float(wc.q[:, 2].min())}
    res = dict(el=el, tick_ms=tick_ms, sim_ms=sim_ms, stats=stats, pre=pre, overlap=overlap, side_gather=side_gather,
               graph_steps=graph_steps)
    torch.cuda.synchronize()
    _stream_ctx.__exit__(None, None, None)
    return res, wc, sched, (pre + a.warmup + a.steps) * conf.dt


def workload_name(a, n):
    if a.workload == "walk":
        s = (f"cfg3: {n} OP3 LIPM walking per GPU (footstep plan along the demo path, LIPM/DCM CoM reference + swing "
             "trajectories -> update_tasks each tick; TSID tick + sim step")
        s += ", robot<->robot hull pairs collided)" if a.self_collision else ", floor contacts only)"
        if getattr(a, "closed_loop", False):
            s += "; LOOP CLOSED: the tick reads the sim state, the sim is driven by tau, touch-down feedback"
        if a.randomize:
            s += "; cfg5 randomised mass / friction / floor tilt + 1 cm terrain steps"
        if a.dephase > 0:
            s += f"; per-env start delays U(0, {a.dephase} s)"
        if a.tau_max_scaling is not None:
            s += f"; tau_max_scaling {a.tau_max_scaling}"
        if getattr(a, "device_plan", False):
            s += "; plans built and episodes restarted ON THE DEVICE (tsidb_walk_plan + tsidb_reset_done every 50 steps)"
        return s
    if getattr(a, "robot", "v1") == "v0":
        return (f"cfg2 on the v0 robot (robot/v0: 18 actuated joints, 52 collision meshes, condim 4, joint damping): {n} "
                f"perturbed stand/balance per GPU")
    return f"cfg2: {n} perturbed stand/balance per GPU"


def kernel_words(wc):
    """DESIGN.md section 4 "Algorithmic bytes", with the robot's dimensions (v1: 347 and 185 words per env per launch)"""
    tick_words = (wc.NQ + wc.NV + 9 + wc.NA + 48 + 24 + 2) + (wc.NQ + wc.NV + wc.NA + wc.NV + 24 + 1 + wc.NOBS + 2)
    sim_words = (wc.NQ + wc.NQ + 2 * wc.NV) + (wc.NQ + 2 * wc.NV)
    return tick_words, sim_words


def secondary_runs(a, dev):
    """The unfavourable paths beside the headline, each a short run of its own."""
    out = {}
    base = dict(dtype=a.dtype, randomize=False, dephase=0.0, tau_max_scaling=None, graph=0, steps=a.secondary_steps, warmup=20,
                preroll=600, event_every=4, no_overlap=a.no_overlap, sync_gather=False, self_collision=a.self_collision,
                robot="v1", closed_loop=False, sim_batch=0, device_plan=False)
    cases = [
        ("cfg2_stand_1024", dict(workload="stand", steps=max(a.secondary_steps, 400), warmup=200), 1024),
        ("cfg2_stand_4096", dict(workload="stand", steps=max(a.secondary_steps, 400), warmup=200), 4096),
        # window centred on t = 1.5 s with start delays U(0, 1 s): about half the envs are still in the
        # double-support start (50-variable QP), the rest in single support (38 variables)
        ("cfg3_walk_4096_dephased", dict(workload="walk", dephase=1.0, preroll=700 - a.secondary_steps // 2), 4096),
        # torque bounds at 1.2 N m (the gait needs up to 2.3 N m) with the envs spread over one step period: at any
        # tick a good part of the batch is in the dual active-set loop (frac_envs_in_active_set_loop_window_mean)
        ("cfg3_walk_4096_tight_torque_bounds", dict(workload="walk", tau_max_scaling=0.12, dephase=0.5, preroll=800), 4096),
        # the per-GPU share of the 4096 walkers at 8 GPUs (strong split): a step is one wavefront's latency.  (Replaying the
        # steps from a HIP graph - WalkController.capture_steps, a convenience API for callers that want one launch per K
        # steps - is slower than this eager pipeline, DESIGN.md section 5 "Streams", and is no longer benched: --graph K.)
        # the headline workload with the plans built and the episodes restarted ON THE DEVICE (bench.py --device-plan): same
        # path scales U(0.5, 1) and table capacity as the host-planned headline, same 20-step window
        ("cfg3_walk_4096_device_plan_driver_window", dict(workload="walk", device_plan=True, steps=20, warmup=5, preroll=600, event_every=8), 4096),
        ("cfg3_walk_4096_device_plan", dict(workload="walk", device_plan=True, steps=max(a.secondary_steps * 5, 1000), preroll=600, event_every=8), 4096),
        ("cfg3_walk_4096_closed_loop", dict(workload="walk", closed_loop=True, steps=max(a.secondary_steps, 400), preroll=600), 4096),
        ("cfg3_walk_512_eager", dict(workload="walk", steps=800), 512),
        # the per-GPU shares of the 4096 walkers at 4 and 2 GPUs (strong split)
        ("cfg3_walk_1024_eager", dict(workload="walk", steps=800), 1024),
        ("cfg3_walk_2048_eager", dict(workload="walk", steps=600), 2048),
        # BASELINE configs[4]: 65536 envs, randomised mass / friction / floor tilt + 1 cm terrain steps
        ("cfg5_65536_randomized", dict(workload="walk", randomize=True, steps=max(a.secondary_steps // 2, 100), event_every=8), 65536),
        # the reference's second robot (robot/v0: 52 collision meshes, condim 4, joint damping), perturbed standing
        ("v0_stand_4096", dict(workload="stand", robot="v0", steps=max(a.secondary_steps, 200), warmup=100), 4096),
    ]
    wsz = 8 if a.dtype == "f64" else 4
    for name, over, n in cases:
        b = SimpleNamespace(**{**base, **over})
        res, wc, _, _ = run_workload(b, dev, 0, 1, n, with_gather=False)
        closed = bool(getattr(b, "closed_loop", False))
        tick_ms = None if res["graph_steps"] else res["tick_ms"]
        # closed loop: tick and sim are ONE call (tsidb_step) - the events bracket both, there is no separate sim time
        sim_ms = None if (res["graph_steps"] or closed) else res["sim_ms"]
        roof = None
        if tick_ms is not None:
            tw, sw = kernel_words(wc)
            if b.randomize:
                sw += 8 + 20
            if closed:
                dom, dms, words = "k_tick + k_sim (one tsidb_step call)", tick_ms, tw + sw
            elif sim_ms >= tick_ms:
                dom, dms, words = "k_sim", sim_ms, sw
            else:
                dom, dms, words = "k_tick", tick_ms, tw
            ach = n * words * wsz / (dms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": n * words * wsz, "avg_launch_ms": dms}
        out[name] = {"workload": workload_name(b, n), "value": n * b.steps / res["el"], "unit": "env-steps/s",
                     "steps": b.steps, "ms_per_step": 1e3 * res["el"] / b.steps,
                     "k_tick_ms": tick_ms, "k_sim_ms": sim_ms, "roofline": roof,
                     "hip_graph_steps_per_launch": res["graph_steps"], "last_step_stats": res["stats"]}
        del wc
        torch.cuda.empty_cache()
    if a.dtype == "f64":
        out["cfg3_walk_4096_device_episodes"] = device_episodes_run(dev, 4096, max(a.secondary_steps * 20, 4000))
    return out


def device_episodes_run(dev, n, steps):
    """The pipelined walking loop with the episode lifecycle on the device: plans built by tsidb_walk_plan
    (short paths, so that episodes end often), every 50 steps the envs whose plan is walked to the end are marked done and
    tsidb_reset_done + a new plan restart them - no host synchronisation inside the timed region."""
    from tsid_control_amd import RobotConfig, WalkController
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf, op3_walking_posture
    conf = op3_walking_conf(RobotConfig())
    conf.reference_quirks = False
    wc = WalkController(conf, num_envs=n, device=dev)
    wc.set_posture_bias(op3_walking_posture())
    sched = WalkSchedule.on_device(wc, seed=3, K=40, scale_range=(0.02, 0.08))   # 2-8 steps per episode: 1000-2500 ticks
    resets = torch.zeros((), dtype=torch.int64, device=dev)
    failed = torch.zeros((), dtype=torch.int64, device=dev)
    flagged = torch.zeros((), dtype=torch.int64, device=dev)

    def loop(k):
        for i in range(k):
            wc.step_pipelined(walk=(sched, wc.t))
            if i % 50 == 49:
                over = (wc.t - sched.t_offset.double()) > (sched.t_start + sched.nsteps.double() * conf.step_duration + 1.0)
                wc.rows[:, wc.NOBS + 1] = torch.maximum(wc.rows[:, wc.NOBS + 1], over.to(wc.dtype))
                resets.add_((wc.rows[:, wc.NOBS + 1] != 0).sum())
                failed.add_((wc.status != 0).sum())
                wc.sync_sim()
                flagged.add_(((wc.info[:, 3] & (1 | 2 | 4 | 32)) != 0).sum())
                wc.reset_done(sched, t=wc.t)
        wc.sync_sim()

    with torch.cuda.stream(wc.tick_stream):
        loop(700)
        torch.cuda.synchronize()
        resets.zero_(); failed.zero_(); flagged.zero_()
        t0 = time.perf_counter()
        loop(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    res = {"workload": f"cfg3: {n} OP3 walkers, plans built on the device, episodes restarted on the device (reset_done + new plan every 50 steps)",
           "value": n * steps / el, "unit": "env-steps/s", "steps": steps, "ms_per_step": 1e3 * el / steps, "k_tick_ms": None, "k_sim_ms": None,
           "roofline": None, "episode_resets": int(resets), "failed_qp_samples": int(failed), "flagged_sim_samples": int(flagged),
           "states_finite": bool(torch.isfinite(wc.q).all() and torch.isfinite(wc.qpos).all())}
    del wc
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    from tsid_control_amd.sharding import init_distributed, shard_range
    import torch.distributed as dist

    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # TSIDB_BENCH_ONE_DEVICE=1 (+ TSIDB_DIST_BACKEND=gloo) rehearses the N > 1 plumbing on a 1-GPU box
    one_dev = os.environ.get("TSIDB_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda", local if (world > 1 and not one_dev) else 0)
    torch.cuda.set_device(dev)
    if args.weak:
        n = args.envs
    else:
        lo, hi = shard_range(args.envs, rank, world)
        if (hi - lo) * world != args.envs:
            raise SystemExit("the env count must be divisible by the number of GPUs")
        n = hi - lo
    res, wc, sched, t_next = run_workload(args, dev, rank, world, n)
    el, tick_ms, sim_ms = res["el"], res["tick_ms"], res["sim_ms"]

    wsz = 8 if args.dtype == "f64" else 4
    if res["graph_steps"]:
        # no HIP events inside a captured graph: take the per-kernel times from a short eager run of the same workload
        import copy
        ea = copy.copy(args)
        ea.graph, ea.steps = 0, 64
        er, ewc, _, _ = run_workload(ea, dev, rank, world, n, with_gather=False)
        tick_ms, sim_ms = er["tick_ms"], er["sim_ms"]
        del ewc
    tick_words, sim_words = kernel_words(wc)
    assert args.robot != "v1" or (tick_words, sim_words) == (TICK_WORDS, SIM_WORDS)
    dom, dom_ms, dom_words = ("k_tick", tick_ms, tick_words) if tick_ms >= sim_ms else ("k_sim", sim_ms, sim_words)
    if args.randomize and dom == "k_sim":
        dom_words += 8 + 20  # env_params row + terrain table row
    alg_bytes = n * dom_words * wsz
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9

    # HBM traffic is NOT measured by this run: it comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    # passes (tools/pmc_profile.sh) and is reported only for the configuration those passes were taken on
    traffic = traffic_src = pmc = None
    tf = ROOT / "profiles" / "pmc_traffic.json"
    if tf.exists():
        rec_all = json.loads(tf.read_text())
        cond = rec_all.get("measured_on", {})
        same = (args.dtype == cond.get("dtype", "f64") and n == cond.get("envs", 4096) and args.workload == cond.get("workload", "walk")
                and not args.randomize and args.dephase == 0 and args.tau_max_scaling is None
                and int(args.self_collision) == int(cond.get("self_collision", 0)))
        if same:
            rec = rec_all.get(dom, {})
            traffic = rec.get("bytes_per_launch")
            traffic_src = f"profiles/pmc_traffic.json ({rec_all.get('source', 'rocprofv3 --pmc passes')}; not measured by this run)"
            pmc = rec_all.get("valu", {}).get(dom)

    # what binds the dominant kernel, from the committed instruction-class counters (tools/pmc_profile.sh pass p5) and THIS run's
    # launch time: float64 lane-flops as issued against the 78.6 TFLOP/s vector peak, and the VALU issue floor (4 cycles per
    # double-precision instruction, 2 per other VALU instruction on a SIMD-32 shared by two wavefronts) against the launch time
    # at the 2.4 GHz maximum clock - both lower bounds of the utilisation (the clock under load is lower, idle lanes count)
    vec = None
    if pmc and "f64_lane_flops_issued_per_env" in pmc:
        t_s = dom_ms * 1e-3
        floor_s = (n / 1024.0) * pmc["issue_floor_cycles_per_env"] / 2.4e9
        vec = {"f64_issued_tflops": pmc["f64_lane_flops_issued_per_env"] * n / t_s / 1e12, "f64_vector_peak_tflops": VALU_PEAK_TFLOPS["f64"],
               "f64_frac_of_vector_peak": pmc["f64_lane_flops_issued_per_env"] * n / t_s / 1e12 / VALU_PEAK_TFLOPS["f64"],
               "f64_share_of_valu_instructions": pmc["f64_share_of_valu"],
               "valu_issue_floor_ms_at_2p4GHz": 1e3 * floor_s, "valu_issue_floor_frac_of_launch": floor_s / t_s}
    flat = {}
    if pmc and "issue_floor_cycles_per_env" in pmc and pmc.get("wave_cycles_per_env"):
        # two wavefronts share a SIMD; SQ_WAVE_CYCLES counts quad-cycles
        flat["valu_issue_frac"] = 2.0 * pmc["issue_floor_cycles_per_env"] / (4.0 * pmc["wave_cycles_per_env"])
        if pmc.get("launch_us_back_to_back"):   # (from the back-to-back kernel trace without counters: the PMC passes slow the kernels down)
            fr = pmc["f64_lane_flops_issued_per_env"] * 4096 / (pmc["launch_us_back_to_back"] * 1e-6) / 1e12 / VALU_PEAK_TFLOPS["f64"]
            flat["f64_issued_frac_of_vector_peak"] = fr
            flat["f64_useful_frac"] = fr * USEFUL_LANES / 64.0
    if rank == 0:
        value = world * n * args.steps / el
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": workload_name(args, n),
                       "envs_per_gpu": n, "global_envs": n * world, "preroll_steps": res["pre"],
                       "plan": None if args.workload != "walk" else ("device" if args.device_plan else "host"),
                       "parallelism": f"env-sharded x{world}, all-gather of obs + reward + done"
                                      + (" on a side stream" if res["side_gather"] else ""),
                       "streams": ("sim(t) overlapped with tick(t+1) on a second HIP stream" if res["overlap"] else "single stream")
                                  + (f"; {res['graph_steps']} steps per HIP graph launch" if res["graph_steps"] else ""),
                       "qp_failed_envs_last_step": res["stats"]["qp_failed_envs_last_step"], "last_step_stats": res["stats"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                         "k_tick_ms": tick_ms, "k_sim_ms": sim_ms,
                         # what actually binds these kernels: VALU issue + dependency latency of
                         # one wavefront per env; the HBM fraction above is reported because the north star asks for it
                         "binding_resource": "valu-issue + dependency latency (one wavefront per env)",
                         # the same as flat scalars (the driver's record keeps scalars only): share of the SIMDs' VALU issue
                         # slots used (two wavefronts per SIMD; 4 cycles per float64 instruction, 2 per other), float64
                         # lane-flops as issued / on the 26 of 64 lanes that carry a dof, against the 78.6 TFLOP/s vector
                         # peak, share of the wavefronts' cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), residency
                         # (all four from the committed back-to-back PMC passes, profiles/pmc_traffic.json - the counters
                         #  cannot be collected inside this run; the nested dicts below price the same counters with THIS
                         #  run's launch time, which under overlap includes the other kernel's share of the SIMDs)
                         "valu_issue_frac": flat.get("valu_issue_frac"),
                         "f64_issued_frac_of_vector_peak": flat.get("f64_issued_frac_of_vector_peak"),
                         "f64_useful_frac": flat.get("f64_useful_frac"),
                         "wait_any_frac": (pmc["wait_any_cycles_per_env"] / pmc["wave_cycles_per_env"]) if pmc and "wait_any_cycles_per_env" in pmc else None,
                         "waves_per_simd": WAVES_PER_SIMD[args.dtype],
                         "valu_issue_from_profiles": pmc, "vector_roofline_from_profiles": vec,
                         "valu_frac_nominal": (n * NOMINAL_FLOP_PER_ENV_STEP / ((tick_ms + sim_ms) * 1e-3)) / (VALU_PEAK_TFLOPS[args.dtype] * 1e12)},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(wc, sched, t_next, args.cpu_seconds, args.cpu_sample)
        if world == 1 and not args.no_secondary:
            del wc
            torch.cuda.empty_cache()
            out["secondary"] = secondary_runs(args, dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
