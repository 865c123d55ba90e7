#!/bin/bash
# copy the judged summaries of a tools/final_profile.sh run from gpurun_out/<tag>_* into profiles/<round>_final_*
# usage: bash tools/collect_profiles.sh <tag> <round, e.g. r02>
set -e
TAG=${1:-r02}
RND=${2:-r02}
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/${TAG}_bench.json profiles/${RND}_final_bench.json
cp $G/${TAG}_bench_driver_window.json profiles/${RND}_final_bench_driver_window.json
cp $G/${TAG}_bench_no_overlap.json profiles/${RND}_final_bench_no_overlap.json
cp "$(ls -S $G/${TAG}_trace/*/*_kernel_stats.csv | head -1)" profiles/${RND}_final_kernel_stats_overlap.csv   # (largest: the bench process, not a helper child)
cp "$(ls -S $G/${TAG}_trace_serial/*/*_kernel_stats.csv | head -1)" profiles/${RND}_final_kernel_stats_serial.csv
cp $G/${TAG}_bench_under_rocprof.json profiles/${RND}_final_bench_under_rocprof_overlap.json
cp $G/${TAG}_bench_under_rocprof_serial.json profiles/${RND}_final_bench_under_rocprof_serial.json
python3 tools/pmc_summary.py $G/${TAG}_pmc --last 100 --traffic-json profiles/pmc_traffic.json > profiles/${RND}_final_pmc_walk_f64_last100.txt
python3 tools/pmc_summary.py $G/${TAG}_pmc > profiles/${RND}_final_pmc_walk_f64_all.txt
python3 tools/trace_window.py $G/${TAG}_trace_serial 1000 > profiles/${RND}_final_kernel_trace_timed_window_serial.txt
python3 tools/trace_window.py $G/${TAG}_trace 1000 > profiles/${RND}_final_kernel_trace_timed_window_overlap.txt
# the launch durations of the back-to-back kernel trace (no counters collected: PMC passes slow the kernels down) go
# beside the counters: bench.py prices the float64 lane-flops with them (roofline.f64_issued_frac_of_vector_peak)
python3 - "$G/${TAG}_trace_serial" <<'PY'
import collections, csv, glob, json, sys
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
t = json.load(open("profiles/pmc_traffic.json"))
for name, pre in (("k_tick", "k_tick<double"), ("k_sim", "k_sim<double")):
    d = [x[1] for k, v in rows.items() if k.startswith(pre) for x in sorted(v)[-1000:]]
    if d:
        t["valu"][name]["launch_us_back_to_back"] = sum(d) / len(d) / 1e3
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
