#!/bin/bash
# the residual stream as an fp16 pair (VG_VIT_RESID_HL=1): tower tests, pipeline A/B
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vit.py -x -q -m gpu -k "pair_residual or families or fp16 or f16" -s 2>&1 | grep -E "pair stream|passed|failed|Error|error" | tail -6
for rep in 1 2 3; do
  for hl in 0 1; do
    VG_VIT_RESID_HL=$hl timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-sequence-pass --emulate-world > $OUT/bench_$hl.$rep.json 2> $OUT/bench_$hl.$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/bench_$hl.$rep.json').read().strip().splitlines()[-1])
r=d['roofline']
print('pair stream $hl rep $rep:', d['value'], d.get('block_values'), 'frac', r['frac'], 'gemm ms/frame', r.get('gemm_ms_per_frame'), 'avg launch us', r.get('avg_launch_us'))"
  done
done
