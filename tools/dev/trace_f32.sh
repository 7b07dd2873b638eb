set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/f32_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o f32 -- python3 $GRAFT_REPO_ROOT/tools/dev/time_f32.py > $OUT.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/f32_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:10]:
    print(f"{r['Name'].split('(')[0][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:8.1f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
rm -rf $OUT
