#!/bin/bash
# Compact per-kernel resource table of the product build (VGPRs, scratch, LDS) from hipcc's kernel-resource-usage remarks.
# usage: tools/kernel_resources.sh [extra hipcc flags]   (does not touch the in-tree .so)
cd "$(dirname "$0")/../hrl_pybullet_envs_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/hrl_res_probe.so hrl_hip.hip 2>&1 |
  awk '/Function Name/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /LDS Size/ {print n, "vgpr", v, "scratch", sc, "lds", $(NF-1)}' |
  sed -E -e 's/_ZN12_GLOBAL__N_1[0-9]+//' -e 's/EvN3hrl[^ ]*//' -e 's/EN3hrl[^ ]*//' -e 's/EPKf[^ ]*//' -e 's/EPf[^ ]*//' | sort
