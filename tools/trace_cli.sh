#!/bin/bash
# on the GPU box: rocprofv3 kernel-trace stats of the entry point on one short synthetic sequence; prints the top kernels per frame
# (development aid: which kernels the entropy / ground / tracking stages of the CLI spend GPU time in)
set -u
FR=${1:-60}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcli
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cli -- python3 $GRAFT_REPO_ROOT/tools/time_cli.py $FR 150000 > $OUT.log 2>&1
tail -3 $OUT.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/cli_kernel_stats.csv')))
nf = $FR
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('frames', nf, 'kernel ms per frame', round(tot / nf / 1e6, 3))
for r in rows[:${2:-40}]:
    print(f"{r['Name'].split('(')[0][:70]:70s} {int(r['Calls'])/nf:7.1f}/frame {float(r['TotalDurationNs'])/nf/1e6:7.3f} ms/frame avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $OUT
