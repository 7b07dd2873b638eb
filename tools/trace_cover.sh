#!/bin/bash
# on the GPU box: kernel trace (timestamps) of a 48-step bench run with six frames in flight, reduced by tools/analyze_vit_cover.py
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/cover_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${1:-48} --warmup 6 --no-cpu-baseline --no-extras --no-sequence-pass --no-roofline-pass ${2:-} > $OUT.log 2>&1
tail -n 1 $OUT.log | cut -c 1-160
python3 $GRAFT_REPO_ROOT/tools/analyze_vit_cover.py $(find $OUT -name 'bench_kernel_trace.csv') ${1:-48}
rm -rf $OUT
