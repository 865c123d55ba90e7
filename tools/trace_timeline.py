"""Timeline of the last launches of a rocprofv3 kernel trace: start / end (us, relative), queue, and for every k_tick launch the
gap to the previous k_tick's end - what the tick stream waited for.
    python tools/trace_timeline.py <dir> [N launches to print]"""
import csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_tick") or k.startswith("k_sim"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k[:28], r.get("Queue_Id", "?"), r.get("Workgroup_Size", "?"), r.get("Grid_Size", "?")))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
last_tick_end = None
gaps = []
for s, e, k, q, wg, grid in rows:
    extra = ""
    if k.startswith("k_tick"):
        if last_tick_end is not None:
            extra = f"  gap since the previous tick {(s - last_tick_end) / 1e3:7.1f} us"
            gaps.append((s - last_tick_end) / 1e3)
        last_tick_end = e
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  queue {q:>3s} wg {wg:>4s} grid {grid:>7s}  {k}{extra}")
if gaps:
    gaps.sort()
    print(f"tick-to-tick gaps: median {gaps[len(gaps) // 2]:.1f} us, max {gaps[-1]:.1f} us, mean {sum(gaps) / len(gaps):.1f} us over {len(gaps)}")
