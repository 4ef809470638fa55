#!/bin/bash
# compile one .hip unit for gfx950 and print the per-kernel register / scratch usage (no GPU needed)
f=${1:-ddh_fftwave.hip}
cd /root/repo/dedalus_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-result -c $f -o /tmp/cc_check.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    if " error" in line or "error:" in line:
        print(line.rstrip())
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = dict(name=m.group(1)); continue
    if cur is None: continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m: cur[key.split(" ")[0]] = int(m.group(1))
    if "LDS Size" in line:
        print("%-110s VGPR %3d AGPR %3d scratch %5d occ %d" % (cur["name"][:110], cur.get("VGPRs", -1), cur.get("AGPRs", 0), cur.get("ScratchSize", -1), cur.get("Occupancy", -1)))
        cur = None
'
