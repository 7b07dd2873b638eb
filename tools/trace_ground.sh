#!/bin/bash
# on the GPU box: timeline of one ground pass (kernel start / duration / gap to the previous kernel) from a rocprofv3 kernel trace of tools/time_ground.py
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/gr_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o gr -- python3 $GRAFT_REPO_ROOT/tools/time_ground.py > $OUT.log 2>&1
tail -2 $OUT.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/gr_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the 30th pass of the synthetic set: find k_pw_classify starts
idx = [i for i, r in enumerate(rows) if 'k_pw_classify' in r['Kernel_Name']]
a, b = idx[30], idx[31]
t0 = int(rows[a]['Start_Timestamp']); prev_end = None
for r in rows[a - 3:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = '' if prev_end is None else f'gap {(s - prev_end) / 1e3:6.1f}'
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  {gap}  {r['Kernel_Name'].split('(')[0][:60]}")
    prev_end = e
print('pass period', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'us')
PY
