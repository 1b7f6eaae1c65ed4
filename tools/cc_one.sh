#!/bin/bash
# compile one csrc/*.hip for gfx950 and print the per-kernel register / scratch report: tools/cc_one.sh meshdec
cd /root/repo/pdfnet_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wno-unused-result -c $1.hip -o $1.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|VGPRs Spill|SGPRs Spill" | paste - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[a-z_0-9]*.hip:[0-9]*:[0-9]*: remark: //g'
