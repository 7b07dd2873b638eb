#!/bin/bash
# on the GPU box: SQ counters of the clustering kernels (tools/bench_cluster.py under rocprofv3 --pmc, own pass, no --stats)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/cl_pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -o cl -- python3 $GRAFT_REPO_ROOT/tools/bench_cluster.py > $OUT.log 2>&1
tail -2 $OUT.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$OUT/cl_counter_collection.csv')))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not name.startswith('k_cl_'): continue
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    key = (name, r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[name] += 1
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:6]:
    n = cnt[name]
    wc = c['SQ_WAVE_CYCLES'] or 1
    print(f"{name:28s} launches {n:4d}  waves/launch {c['SQ_WAVES']/n:8.0f}  wave-cycles/launch {wc/n/1e6:8.2f} M  busy {c['SQ_BUSY_CYCLES']/n/1e6:6.2f} M | of wave cycles: waiting {c['SQ_WAIT_ANY']/wc:5.2f}  wait-inst {c['SQ_WAIT_INST_ANY']/wc:5.2f}  active-inst {c['SQ_ACTIVE_INST_ANY']/wc:5.2f}  valu-active {c['SQ_ACTIVE_INST_VALU']/wc:5.2f} | VALU insts/launch {c['SQ_INSTS_VALU']/n/1e6:7.2f} M")
PY
rm -rf $OUT
