#!/bin/bash
# GPU box: everything the committed evidence under profiles/ is made from, in one call (about five minutes):
# the -m gpu tests, the rocprofv3 passes of all six kinds, and the bench lines of every BASELINE configuration.
#   usage: tools/round_end.sh <tag> [round prefix, default r6]     outputs under gpurun_out/<tag>_*; then on the build machine:
#          cp gpurun_out/<tag>_profiles/* profiles/ and the bench lines you want to keep
set -e
T=${1:?tag}; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 || { tail -30 gpurun_out/${T}_tests.log; exit 1; }
tail -2 gpurun_out/${T}_tests.log
for k in gather flat point maze maze_mj flagrun; do
  n=4096; [ $k = maze ] && n=8192
  rm -rf gpurun_out/prof_${T}_$k
  bash tools/profile_round.sh gpurun_out/prof_${T}_$k $k $n > gpurun_out/prof_${T}_$k.log 2>&1 && echo profiled $k
done
# condense them where the bench runs, so that its line carries the counters of exactly these kernels; the files travel back under gpurun_out/
R=${2:-r6}
for k in gather flat point maze maze_mj flagrun; do
  n=4096; [ $k = maze ] && n=8192
  name=${R}_$k; [ $k = gather ] && name=${R}_final
  python tools/summarize_profile.py gpurun_out/prof_${T}_$k $name $k $n > /dev/null
done
mkdir -p gpurun_out/${T}_profiles && cp profiles/pmc_summary.json profiles/${R}_*_pmc.json profiles/${R}_*_kernel_stats.csv gpurun_out/${T}_profiles/
: > gpurun_out/${T}_other_kinds.txt
for k in flat maze point maze_mj flagrun mixed; do
  n=4096; [ $k = maze ] && n=8192
  python bench.py --kind $k --envs $n --steps 1000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_other_kinds.txt
done
python bench.py --envs 32768 --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_other_kinds.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driverlike.json 2>/dev/null
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.json
