#!/bin/bash
# on the GPU box: rocprofv3 kernel-trace stats of tools/seq_only.py; prints kernel time per frame (development aid)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/tseq
N=${1:-48}; NW=${2:-6}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o seq -- python3 $GRAFT_REPO_ROOT/tools/seq_only.py $N $NW > $OUT.log 2>&1
grep process_sequence $OUT.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/seq_kernel_stats.csv')))
nf = 2 * $N + 6
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('frames', nf, 'kernel ms per frame', round(tot / nf / 1e6, 3))
for r in rows[:${3:-22}]:
    print(f"{r['Name'].split('(')[0][:64]:64s} {int(r['Calls'])/nf:6.1f}/frame {float(r['TotalDurationNs'])/nf/1e6:7.3f} ms/frame avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $OUT
