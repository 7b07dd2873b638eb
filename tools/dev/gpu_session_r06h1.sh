#!/bin/bash
# round 6, hierarchy on the device: tests, stage timing, kernel trace of one stage, pipeline A/B host vs device, 768-thread walk A/B
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06h1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hierarchy.py -x -q -m gpu > $OUT/pytest_hier.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_hier.txt
tail -5 $OUT/pytest_hier.txt
timeout 600 python tools/bench_hierarchy.py > $OUT/bench_hierarchy.txt 2>&1; tail -6 $OUT/bench_hierarchy.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o hier -- python3 $GRAFT_REPO_ROOT/tools/bench_hierarchy.py --frames 1 --reps 5 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/prof/**/hier_kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        if 'k_hd' in r['Name'] or 'rocprim' in r['Name'] or 'k_cl_b_search' in r['Name']:
            print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for mode in host device; do
    VILGOD_HIERARCHY=$mode timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-sequence-pass --emulate-world > $OUT/bench_$mode.$rep.json 2> $OUT/bench_$mode.$rep.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$mode.$rep.json').read().strip().splitlines()[-1])
print('hierarchy $mode rep $rep:', d['value'], d.get('block_values'))"
  done
done
export VILGOD_HIP_LIB=$GRAFT_REPO_ROOT/vilgod_amd/libvilgod_hip_dev.so
for rep in 1 2; do
  for nt in 512 768; do
    VG_CLUSTER_SEARCH_NT=$nt timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-sequence-pass --emulate-world > $OUT/bench_nt$nt.$rep.json 2> $OUT/bench_nt$nt.$rep.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_nt$nt.$rep.json').read().strip().splitlines()[-1])
print('search NT $nt rep $rep:', d['value'], d.get('block_values'))"
  done
done
