#!/bin/bash
# round 6, hierarchy on the device: kernel trace of the stage; 768-thread walk A/B (development library)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06h2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o hier -- python3 $GRAFT_REPO_ROOT/tools/bench_hierarchy.py --frames 1 --reps 5 > $OUT/prof.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/prof/**/hier_kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:60]:
        if 'k_hd' in r['Name'] or 'rocprim' in r['Name']:
            print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
cd $GRAFT_REPO_ROOT
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
