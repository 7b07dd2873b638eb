#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_k20_$i.json 2> $O/err_$i.txt
  python3 -c "
import json,sys
d=[json.loads(l) for l in open('$O/bench_k20_$i.json') if l.startswith('{')][-1]
print('run $i', d['value'], d['block_values'], d['config']['setup_frames'])"
done
