#!/bin/bash
# Per-kernel register / LDS / scratch / occupancy of the HIP module as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh [extra -D flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
for f in jade_hip jade_bvh; do
/opt/rocm/bin/hipcc "$@" -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -mfma -I"$R/include" -I"$R/jaderaytracerendering_amd/csrc" \
  -c "$R/jaderaytracerendering_amd/csrc/$f.hip" -o "$T/x.o" -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): ([0-9]+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    m = re.match(r"^_Z\d+(k_[a-z_0-9]+?)(?=\d|P|$)", k)
    if not m:
        continue
    name = m.group(1)
    print("%-16s VGPR %3d AGPR %3d SGPR %3d scratch %4d B  LDS %6d B  occupancy %d" % (name, v.get("VGPRs", -1), v.get("AGPRs", 0), v.get("TotalSGPRs", v.get("SGPRs", -1)), v.get("ScratchSize [bytes/lane]", v.get("ScratchSize", 0)), v.get("LDS Size [bytes/block]", v.get("LDS Size", 0)), v.get("Occupancy [waves/SIMD]", v.get("Occupancy", -1))))
'
done
