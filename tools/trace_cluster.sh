#!/bin/bash
# on the GPU box: rocprofv3 kernel-trace stats of tools/bench_cluster.py (one 150k-point frame, 11 MST runs); prints the clustering kernels
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/cl_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cl -- python3 $GRAFT_REPO_ROOT/tools/bench_cluster.py "$@" > $OUT.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/cl_kernel_stats.csv')))
runs = 11
tot = 0
for r in rows:
    if 'k_cl_' in r['Name'] or 'rocprim' in r['Name'] or 'hdbscan' in r['Name']:
        tot += float(r['TotalDurationNs'])
        print(f"{r['Name'].split('(')[0][:70]:70s} {int(r['Calls'])/runs:6.1f}/run {float(r['TotalDurationNs'])/runs/1e3:8.1f} us/run avg {float(r['AverageNs'])/1e3:8.1f} us")
print('clustering kernels per run: %.3f ms' % (tot / runs / 1e6))
PY
