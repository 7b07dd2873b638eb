#!/bin/bash
# default-configuration pass (entropy scores + two-frame clustering), hierarchy stage on the host against on the device, one box, alternating
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06z4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for mode in host device; do
    VILGOD_HIERARCHY=$mode timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline-pass --emulate-world > $OUT/bench_$mode.$rep.json 2> $OUT/bench_$mode.$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/bench_$mode.$rep.json').read().strip().splitlines()[-1])
print('hierarchy $mode rep $rep: metric', d['value'], ' default configuration', d.get('default_config_mode',{}).get('value'))"
  done
done
