#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06h7}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hierarchy.py -x -q > $OUT/pytest_hier.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_hier.txt
tail -3 $OUT/pytest_hier.txt
timeout 600 python tools/bench_hierarchy.py > $OUT/bench_hierarchy.txt 2>&1; grep frame $OUT/bench_hierarchy.txt
timeout 300 python tools/dev/hier_stamps.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o hier -- python3 $GRAFT_REPO_ROOT/tools/bench_hierarchy.py --frames 1 --reps 5 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/prof/**/hier_kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:60]:
        if 'k_hd' in r['Name']:
            print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
