#!/bin/bash
# on the GPU box: average duration of the projection GEMM launches (per instantiation) with ONE frame in flight and with SIX (the
# timed region's setting), from two rocprofv3 kernel traces of the same bench command: what co-scheduling with the other frames'
# kernels costs a persistent-workgroup GEMM launch.
set -u
cd /tmp && export TMPDIR=/tmp
for inf in 1 6; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/tgi_$inf
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --blocks 1 --inflight $inf --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras > $OUT.log 2>&1
  python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/bench_kernel_stats.csv')))
nf = [int(r['Calls']) for r in rows if 'k_head' in r['Name']][0]
tot = sum(float(r['TotalDurationNs']) for r in rows)
g = [r for r in rows if 'k_gemm_f16' in r['Name']]
print('inflight $inf: frames', nf, 'kernel ms per frame', round(tot / nf / 1e6, 3), ' projection GEMMs ms per frame', round(sum(float(r['TotalDurationNs']) for r in g) / nf / 1e6, 3))
for r in g:
    print(f"   {r['Name'].split('(')[0][:60]:60s} {int(r['Calls'])/nf:6.1f}/frame avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf $OUT
done
