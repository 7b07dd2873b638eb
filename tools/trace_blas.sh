#!/bin/bash
# on the GPU box: which kernels torch.matmul fp16 (hipBLASLt) runs for the four ViT projection shapes, with their durations
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/blas_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o blas -- python3 $GRAFT_REPO_ROOT/tools/bench_blas_ceiling.py --rounds 1 --iters 5 > $OUT.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/blas_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:400]}")
PY
