#!/bin/bash
# GPU box: calibrate FETCH_SIZE / WRITE_SIZE on the step kernels' own access pattern (tools/micro/traffic_calib.hip: known bytes per env).
#   usage: tools/traffic_calib.sh gpurun_out/calib [envs]      prints measured / known bytes per env for each kernel
O=${1:?out dir}; N=${2:-4096}; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o $O/traffic_calib tools/micro/traffic_calib.hip || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- $O/traffic_calib $N > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- $O/traffic_calib $N > $O/w.log 2>&1 || exit 1
python3 - $O $N <<'PY'
import collections, csv, glob, sys
o, n = sys.argv[1], int(sys.argv[2])
known = {'k_read': (304, 0), 'k_write': (0, 349), 'k_both': (304, 349)}
got = collections.defaultdict(dict)
for d, name in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'{o}/{d}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                for k in known:
                    if k in r['Kernel_Name']:
                        acc[k].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k, v in acc.items():
        v = [x for _, x in sorted(v)][-40:]          # the first launches of a kernel see cold lines
        got[k][name] = sum(v) / len(v) * 1024 / n
print(f'{n} envs; bytes per env MEASURED (rocprofv3 counter x 1024 / envs) against KNOWN:')
for k, (r, w) in known.items():
    f_, w_ = got[k].get('FETCH_SIZE', float('nan')), got[k].get('WRITE_SIZE', float('nan'))
    print(f'  {k:8s} FETCH_SIZE {f_:7.1f} B/env (known {r:3d}{"" if not r else f": x {f_ / r:.3f}"})   WRITE_SIZE {w_:7.1f} B/env (known {w:3d}{"" if not w else f": x {w_ / w:.3f}"})')
PY
