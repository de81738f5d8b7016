#!/bin/bash
# per-kernel resource usage of the gfx950 code object (VGPRs, spills, scratch, LDS, occupancy) from the compiler's remarks
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -c trust4_amd/csrc/t4_api.hip -o /tmp/t4_dev.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1); cur[k.strip()] = v.strip()
print("%-58s %5s %6s %7s %4s %7s" % ("kernel", "VGPR", "vspill", "scratch", "occ", "LDS"))
for r in rows:
    n = r["name"].replace("_ZN3t4k11queryKernelI", "queryKernel<").replace("EEv11T4IndexView11T4BatchView6T4Work11T4QueryArgs", ">").replace("ELi", ",").replace("Li", "")
    print("%-58s %5s %6s %7s %4s %7s" % (n[:58], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
