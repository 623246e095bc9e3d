#!/bin/bash
# GPU box: the rocprofv3 passes whose summaries are committed under profiles/ (tools/summarize_profile.py condenses them).
# Kernel trace + stats and every counter group run as separate passes of the same bench command.
#   usage: tools/profile_round.sh gpurun_out/prof_<tag> [kind [envs]]
O=${1:?out dir}; mkdir -p $O
# every kind is SETTLED before the window the counters are taken from (300 untimed steps: a PointBot has reached the walls, ants have
# tumbled; round 3 profiled the first 60 steps after a reset, a different regime than the bench line's): tools/summarize_profile.py
# condenses the LAST 50 launches of every pass.
B="python3 bench.py --steps 50 --warmup 300 --settle 0 --no-cpu-baseline --kind ${2:-gather} --envs ${3:-4096}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/w.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/pmc_sq1 -- $B > $O/s1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $O/pmc_sq2 -- $B > $O/s2.log 2>&1 || exit 1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $O/pmc_sq3 -- $B > $O/s3.log 2>&1 || exit 1
echo profiled into $O
