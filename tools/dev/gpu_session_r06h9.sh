#!/bin/bash
# round 6, hierarchy on the device in the pipeline: shader clock of the one-workgroup kernels, frame latency and throughput, host vs device stage
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06h9}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dev/hier_stamps.py 2>&1 | tail -2 | tee $OUT/stamps.txt
for mode in host device; do
  VILGOD_HIERARCHY=$mode timeout 600 python tools/frame_latency.py 8 > $OUT/latency_$mode.txt 2>&1
  echo "== frame latency, hierarchy $mode"; grep -E "mst|hierarchy|labels_d2h|pack|total" $OUT/latency_$mode.txt
done
for rep in 1 2 3; do
  for mode in host device; do
    VILGOD_HIERARCHY=$mode timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-sequence-pass --no-roofline-pass --emulate-world > $OUT/bench_$mode.$rep.json 2> $OUT/bench_$mode.$rep.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$mode.$rep.json').read().strip().splitlines()[-1])
print('hierarchy $mode rep $rep:', d['value'], d.get('block_values'))"
  done
done
timeout 900 python -m pytest tests/test_pipeline.py -x -q -m gpu -k "hierarchy" 2>&1 | tail -3
