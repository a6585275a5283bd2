#!/bin/bash
# One line per kernel of nh_kernels.hip and nh_deflate.hip: registers, spills, scratch, LDS, occupancy (hipcc remarks).
cd "$(dirname "$0")/../nohuman_amd/csrc" || exit 1
make -s resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("nh::dfl::", "").replace("nh::", "").replace("(nh::KArgs)", "").replace("(DeflateArgs)", "").replace("void ", "")
    print("%-46s VGPR %3s  SGPR %3s  spill S/V %3s/%-3s scratch %3s B  LDS %6s B  occ %s" % (
        name[:46], r.get("VGPRs"), r.get("TotalSGPRs", r.get("SGPRs")), r.get("SGPRs Spill", r.get("SGPR Spill", "?")),
        r.get("VGPRs Spill", r.get("VGPR Spill", "?")), r.get("ScratchSize [bytes/lane]"),
        r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
