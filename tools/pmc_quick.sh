#!/bin/bash
# GPU box: per-wave instruction counters of k_step for the default bench workload.  usage: tools/pmc_quick.sh <outdir>
O=${1:-gpurun_out/pmc_quick}; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/sq.log 2>&1 || exit 1
python3 - $O <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/sq/*/*counter_collection.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_step' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f"{k:24s} {sum(v) / len(v) / 4096:10.1f} per env")
PY
