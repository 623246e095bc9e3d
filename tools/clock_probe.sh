#!/bin/bash
# GPU box: effective shader clock of k_step at different batch sizes = SQ_WAVE_CYCLES (quad-cycles, summed over waves) * 4
# / waves / kernel duration; waves live for nearly the whole kernel when they all fit at once (<= 4096 envs).
O=${1:-gpurun_out/clock}; mkdir -p $O
for N in 1024 2048 4096; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/n$N -- python3 bench.py --steps 200 --warmup 50 --envs $N --no-cpu-baseline > $O/n$N.log 2>&1 || exit 1
  python3 - $O/n$N $N <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + '/*/*counter_collection.csv')[0])):
    if 'k_step' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(glob.glob(d + '/*/*kernel_trace.csv')[0])) if 'k_step' in r['Kernel_Name']]
dur = sum(dur[50:]) / len(dur[50:])
wc = sum(acc['SQ_WAVE_CYCLES'][50:]) / len(acc['SQ_WAVE_CYCLES'][50:])
ga = sum(acc['GRBM_GUI_ACTIVE'][50:]) / max(1, len(acc['GRBM_GUI_ACTIVE'][50:])) if acc['GRBM_GUI_ACTIVE'] else 0
print(f'N {n}: kernel {dur/1e3:.1f} us, wave cycles/wave {wc*4/n:.0f}, => >= {wc*4/n/dur:.2f} GHz if waves live the whole kernel; GRBM_GUI_ACTIVE {ga:.0f} cycles => {ga/dur:.2f} GHz')
PY
done
