#!/bin/bash
# dev: registers, spills, scratch, occupancy and LDS of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
# Spilled scalar registers in a latency-bound kernel are instructions on its critical path: round 3 found 280 in the sparse evaluation.
cd "$(dirname "$0")/../cogaps_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result \
  -mllvm -amdgpu-kernarg-preload-count=16 -DGEN_WIN=${WIN:-256} --cuda-device-only -c -x hip -Rpass-analysis=kernel-resource-usage -o /dev/null cogaps_hip.cpp 2>&1 \
  | python3 -c '
import re, sys
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"): cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t: k, v = t.rsplit(":", 1); cur[k.strip()] = v.strip()
print("%-86s %5s %5s %7s %6s %6s %6s %7s" % ("kernel", "VGPRs", "SGPRs", "scratch", "waves", "sSpill", "vSpill", "LDS"))
for r in rows:
    print("%-86s %5s %5s %7s %6s %6s %6s %7s" % (r["name"][:86], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
'
