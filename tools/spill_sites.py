"""Where a kernel spills: scratch_ (VGPR spills) and v_writelane (SGPR spills) counts per source line.
    python tools/spill_sites.py k_simId [extra hipcc flags]      (mangled-name prefix after _Z<n>)"""
import re, subprocess, sys, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
pat = sys.argv[1] if len(sys.argv) > 1 else "k_simId"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-S", "--cuda-device-only",
                *sys.argv[2:], "-o", "/tmp/k.s", str(ROOT / "tsid_control_amd/csrc/tsidb_api.hip")], check=True)
files, cur, inside = {}, None, False
cnt = {k: collections.Counter() for k in ("scratch_store", "scratch_load", "v_writelane", "v_readlane")}
total = collections.Counter()
for line in open("/tmp/k.s"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[m.group(1)] = Path(m.group(3) or m.group(2)).name
        continue
    if re.match(r"_Z\d+" + pat, line):
        inside = True
        continue
    if not inside:
        continue
    if "s_endpgm" in line:
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    ins = line.split()[0] if line.split() else ""
    if ins and not ins.startswith((".", ";")):
        total[cur] += 1
    for k in cnt:
        if ins.startswith(k):
            cnt[k][cur] += 1
for k in ("scratch_store", "scratch_load", "v_writelane"):
    print(f"== {k}: {sum(cnt[k].values())}")
    for loc, n in cnt[k].most_common(14):
        print(f"   {loc[0]}:{loc[1]}  {n}")
print("instructions in the kernel:", sum(total.values()))
if "--regions" in sys.argv or True:
    import bisect
    # instruction counts per source function of tsidb_sim.hpp / tsidb_tick.hpp (by the line the instruction is attributed to)
    for fn in ("tsidb_sim.hpp", "tsidb_tick.hpp", "tsidb_common.hpp"):
        src = (ROOT / "tsid_control_amd/csrc" / fn).read_text().splitlines()
        starts = [(i + 1, re.search(r"(\w+)\s*\(", l).group(1)) for i, l in enumerate(src)
                  if re.match(r"^(template.*>\s*)?(__device__|static|inline).*\(", l) and re.search(r"(\w+)\s*\(", l)]
        if not starts:
            continue
        agg = collections.Counter()
        for (f, ln), n in total.items():
            if f != fn:
                continue
            j = bisect.bisect_right([s for s, _ in starts], ln) - 1
            agg[starts[j][1] if j >= 0 else "?"] += n
        for name, n in agg.most_common(12):
            print(f"   {fn}:{name:24s} {n}")
