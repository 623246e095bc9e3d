#!/bin/bash
# Device assembly of the product build of a source tree, and per-kernel instruction counts.
#   tools/kernel_isa.sh <tree root, e.g. . or build/r3_tree> <out.s> [extra hipcc flags]
# Prints, for every k_step kernel, the number of instructions in its text (static count, not executed).
set -e
src="$1/hrl_pybullet_envs_amd/csrc/hrl_hip.hip"; out="$2"; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S "$@" -o "$out" "$src"
awk '/^_ZN.*k_step.*:$/ {name=$1; n=0; on=1; next} on && /^\s+[a-z_0-9]+ / {n++} on && /s_endpgm/ {print name, n; on=0}' "$out" | sed -E 's/_ZN12_GLOBAL__N_1[0-9]+//; s/EvN3hrl.*//'
