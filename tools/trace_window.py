"""Per-kernel duration over the LAST N dispatches of a rocprofv3 kernel trace (the timed window of bench.py;
the --stats averages also cover the pre-roll, where another k_tick variant does the work).

    python tools/trace_window.py <dir with *_kernel_trace.csv> [N]"""
import collections
import csv
import glob
import sys

root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows = collections.defaultdict(list)
meta = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
print(f"{'kernel':22s} {'launches':>8s} {'mean us (last %d)' % n:>20s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS B':>6s} {'scratch B':>9s}")
tick = 0.0
for k in sorted(rows):
    d = [x[1] for x in sorted(rows[k])][-n:]
    m = sum(d) / len(d) / 1e3
    if k.startswith("k_tick"):
        tick += m
    print(f"{k:22s} {len(rows[k]):8d} {m:20.1f} {meta[k][0]:>5s} {meta[k][1]:>5s} {meta[k][2]:>5s} {meta[k][3]:>6s} {meta[k][4]:>9s}")
print(f"k_tick<*> together: {tick:.1f} us per tick")
