#!/bin/bash
# on the GPU box: rocprofv3 kernel-trace stats of a short sequential bench run; prints the top kernels (development aid)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/tstats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras > $OUT.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/bench_kernel_stats.csv')))
nf = [int(r['Calls']) for r in rows if 'k_head' in r['Name']][0]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('frames', nf, 'kernel ms per frame', round(tot / nf / 1e6, 3))
for r in rows[:${1:-16}]:
    print(f"{r['Name'].split('(')[0][:64]:64s} {int(r['Calls'])/nf:6.1f}/frame {float(r['TotalDurationNs'])/nf/1e6:7.3f} ms/frame avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $OUT
