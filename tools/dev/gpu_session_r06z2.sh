#!/bin/bash
# round 6, final validation with the device hierarchy as the default: GPU suite, smoke, the driver's bench command, frame latency
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r06z2}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.txt
tail -4 $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
python3 -c "
import json
d=json.loads(open('$OUT/bench_k20.json').read().strip().splitlines()[-1])
print('bench:', d['value'], d.get('block_values'), 'frac', d['roofline']['frac'], 'front latency', d['roofline'].get('front_stage_latency_ms'))"
for mode in host device; do
  VILGOD_HIERARCHY=$mode timeout 600 python tools/frame_latency.py 8 > $OUT/latency_$mode.txt 2>&1
  echo "== frame latency, hierarchy $mode"; grep -E "mst|hierarchy|labels_d2h|pack|total" $OUT/latency_$mode.txt
done
timeout 600 python tools/bench_hierarchy.py > $OUT/bench_hierarchy.txt 2>&1; grep frame $OUT/bench_hierarchy.txt
