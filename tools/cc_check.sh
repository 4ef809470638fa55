#!/bin/bash
# compile one .hip unit for gfx950 and print the per-kernel register / scratch usage (no GPU needed)
f=${1:-ddh_fftwave.hip}
cd /root/repo/dedalus_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-result -c $f -o /tmp/cc_check.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|Function Name|VGPRs:|ScratchSize|SGPRs:" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/^[^ ]*: remark://g' | paste - - - - | sed 's/  */ /g'
