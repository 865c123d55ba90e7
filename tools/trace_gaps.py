"""Gaps between consecutive launches of each kernel in a rocprofv3 kernel trace (last N launches)."""
import collections, csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 500
rows = collections.defaultdict(list)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in sorted(rows.items()):
    v = sorted(v)[-n:]
    if len(v) < 2:
        continue
    dur = sum(e - s for s, e in v) / len(v) / 1e3
    gap = sum(v[i + 1][0] - v[i][1] for i in range(len(v) - 1)) / (len(v) - 1) / 1e3
    per = (v[-1][0] - v[0][0]) / (len(v) - 1) / 1e3
    print(f"{k:18s} mean duration {dur:8.1f} us   mean gap to the next launch {gap:8.1f} us   period {per:8.1f} us")
