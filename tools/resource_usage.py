"""Compact per-kernel register / scratch / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage.

    python tools/resource_usage.py [extra hipcc flags ...] > profiles/rNN_kernel_resource_usage.txt
Compiles csrc/tsidb_api.hip to a throw-away object (the product .so is not touched)."""
import re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-machine-licm", "-ffp-contract=on", *sys.argv[1:]]
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", "-Rpass-analysis=kernel-resource-usage", "-o", f"{td}/x.o",
                        str(ROOT / "tsid_control_amd/csrc/tsidb_api.hip")], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print("== hipcc", " ".join(flags), "tsidb_api.hip -Rpass-analysis=kernel-resource-usage")
cur, rows = None, {}
keys = ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        nm = m.group(1)
        d = re.match(r"_Z\d+(k_[a-z0-9]+)I([df])((?:L[bi]\dE?)*)", nm)
        cur = f"{d.group(1)}<{d.group(2)} {d.group(3).replace('E', ' ').strip()}>" if d else nm
        rows[cur] = {}
        continue
    for k in keys:
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
        if m and cur:
            rows[cur][k] = m.group(1)
for nm, d in rows.items():
    print(nm + "\t" + "\t".join(f"{k}: {d.get(k, '?')}" for k in keys))
