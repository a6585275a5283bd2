#!/bin/bash
# One line per kernel of nh_kernels.hip, nh_deflate.hip and nh_gunzip.hip: registers, spills, scratch, LDS, occupancy (hipcc
# remarks), headed by the hash of the sources it was taken from (bench.py / make_profile_summary.py use the same one for
# nh_kernels.hip + nh_device.h).   bash scripts/resource_usage.sh > profiles/rNN_resource_usage.txt
cd "$(dirname "$0")/../nohuman_amd/csrc" || exit 1
echo "# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, --offload-arch=gfx950), $(date -u +%Y-%m-%d)"
echo "# sha256(nh_kernels.hip + nh_device.h)[:16] = $(cat nh_kernels.hip nh_device.h | sha256sum | cut -c1-16)   (the key of profiles/traffic.json)"
for f in nh_kernels.hip nh_device.h nh_deflate.hip nh_deflate_core.h nh_gunzip.hip; do echo "# sha256($f)[:16] = $(sha256sum $f | cut -c1-16)"; done
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
    name = name.replace("nh::dfl::", "").replace("nh::gz::", "").replace("nh::fq::", "").replace("nh::", "").replace("(nh::KArgs)", "").replace("(DeflateArgs)", "").replace("void ", "")
    print("%-46s VGPR %3s  SGPR %3s  spill S/V %3s/%-3s scratch %3s B  LDS %6s B  occ %s" % (
        name[:46], r.get("VGPRs"), r.get("TotalSGPRs", r.get("SGPRs")), r.get("SGPRs Spill", r.get("SGPR Spill", "?")),
        r.get("VGPRs Spill", r.get("VGPR Spill", "?")), r.get("ScratchSize [bytes/lane]"),
        r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
