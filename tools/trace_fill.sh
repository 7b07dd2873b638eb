#!/bin/bash
# on the GPU box: kernel trace of a driver-style bench run, reduced to the busy / idle timeline of its timed region
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/fill_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline --no-extras --no-sequence-pass --no-roofline-pass ${2:-} > $OUT.log 2>&1
tail -n 1 $OUT.log | cut -c 1-200
python3 $GRAFT_REPO_ROOT/tools/analyze_fill.py $(find $OUT -name 'bench_kernel_trace.csv') ${1:-20}
rm -rf $OUT
