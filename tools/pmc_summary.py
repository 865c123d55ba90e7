"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.

    python tools/pmc_summary.py <dir> [--last N] [--traffic-json out.json]
--last N averages only the last N dispatches of each kernel (the timed window of bench.py; the earlier
ones are the workload's double-support start).  --traffic-json writes the per-launch HBM bytes bench.py
reports as roofline.traffic: k_tick = the three per-contact-configuration kernels of one tick together."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else None
tj = sys.argv[sys.argv.index("--traffic-json") + 1] if "--traffic-json" in sys.argv else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        # one row per (dispatch, counter[, dimension]); sum the dimensions of a dispatch
        d = rows[k][r["Counter_Name"]]
        did = int(r["Dispatch_Id"])
        d[did] = d.get(did, 0.0) + float(r["Counter_Value"])
    for k, cs in rows.items():
        for c, d in cs.items():
            vals = [d[i] for i in sorted(d)]
            acc[k][c].extend(vals[-last:] if last else vals)
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} mean/dispatch {sum(v)/len(v):16.1f}   (n={len(v)})")
if tj:
    mean = lambda k, c: sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else 0.0
    out = {}
    for name, kernels in (("k_tick", [k for k in acc if k.startswith("k_tick<double")]), ("k_sim", [k for k in acc if k.startswith("k_sim<double")])):
        fe = sum(mean(k, "FETCH_SIZE") for k in kernels)
        wr = sum(mean(k, "WRITE_SIZE") for k in kernels)
        out[name] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "bytes_per_launch": 1024.0 * (fe + wr),
                     "bytes_per_launch_fetch_x2": 1024.0 * (2 * fe + wr), "kernels": sorted(kernels),
                     "note": "f64, 4096 walking envs, last %s dispatches of each kernel; raw FETCH_SIZE + WRITE_SIZE (KiB). "
                             "FETCH_SIZE matches the known read byte count 1:1 on this access pattern (8 B per lane, "
                             "env-major rows), so the gfx950 x2 correction for 16 B/lane streams is not applied in "
                             "bytes_per_launch (bytes_per_launch_fetch_x2 applies it)." % (last or "all")}
    # what the passes were taken on (bench.py reports the traffic only for this configuration) and the issue
    # counters behind bench.py's "binding_resource": VALU / SALU / LDS instructions per env, wave cycles per env
    out["measured_on"] = {"dtype": "f64", "envs": 4096, "workload": "walk", "self_collision": 1}
    out["source"] = ("rocprofv3 --kernel-trace --pmc passes of tools/pmc_profile.sh: bench.py --steps 100 --warmup 5 "
                     "--no-overlap, last %s dispatches of each kernel" % (last or "all"))
    # launch durations of the same back-to-back runs (every PMC pass also carries the kernel trace)
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_"):
                per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            d = [x[1] for x in sorted(v)]
            dur[k].extend(d[-last:] if last else d)
    out["valu"] = {}
    for name, kernels in (("k_tick", [k for k in acc if k.startswith("k_tick<double")]), ("k_sim", [k for k in acc if k.startswith("k_sim<double")])):
        n_env = 4096.0
        g = lambda c: sum(mean(k, c) for k in kernels)
        out["valu"][name] = {"valu_inst_per_env": g("SQ_INSTS_VALU") / n_env, "salu_inst_per_env": g("SQ_INSTS_SALU") / n_env,
                             "lds_inst_per_env": g("SQ_INSTS_LDS") / n_env, "wave_cycles_per_env": g("SQ_WAVE_CYCLES") / n_env,
                             "wait_any_cycles_per_env": g("SQ_WAIT_ANY") / n_env,
                             "launch_us_back_to_back_under_pmc": sum(sum(dur[k]) / len(dur[k]) for k in kernels if dur[k]) / 1e3}
        # instruction classes (pass p5).  float64 flops per env as ISSUED (x 64 lanes per wavefront instruction, whatever the
        # execution mask: rows of a 26-wide problem leave most lanes idle, so the useful share is lower); the issue floor
        # counts 4 cycles per double-precision instruction and 2 per other VALU instruction (two wavefronts sharing a
        # SIMD-32: MI355X_MICROARCH.md "vector-instruction ISSUE cost"), per wavefront = per env
        f64 = {c: g("SQ_INSTS_VALU_%s_F64" % c) / n_env for c in ("ADD", "MUL", "FMA", "TRANS")}
        if sum(f64.values()) > 0:
            nf = sum(f64.values())
            valu = g("SQ_INSTS_VALU") / n_env
            out["valu"][name].update({"f64_add_per_env": f64["ADD"], "f64_mul_per_env": f64["MUL"], "f64_fma_per_env": f64["FMA"],
                                      "f64_trans_per_env": f64["TRANS"], "int32_per_env": g("SQ_INSTS_VALU_INT32") / n_env,
                                      "int64_per_env": g("SQ_INSTS_VALU_INT64") / n_env, "cvt_per_env": g("SQ_INSTS_VALU_CVT") / n_env,
                                      "f64_share_of_valu": nf / valu,
                                      "f64_lane_flops_issued_per_env": 64.0 * (f64["ADD"] + f64["MUL"] + 2.0 * f64["FMA"]),
                                      "issue_floor_cycles_per_env": 4.0 * nf + 2.0 * (valu - nf)})
    json.dump(out, open(tj, "w"), indent=1)
