#!/bin/bash
# on the GPU box: the metric's run (48-step blocks, no extras) for a few (frames in flight, ViT passes at a time) settings, one process each, two rounds
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for cfg in "6 2" "6 3" "8 2" "8 3" "5 2" "6 1"; do
    set -- $cfg
    v=$(VILGOD_VIT_CONCURRENCY=$2 timeout 400 python bench.py --steps 48 --warmup 8 --inflight $1 --no-extras --no-cpu-baseline --no-roofline-pass --no-sequence-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['block_values'])")
    echo "inflight $1 vit_concurrency $2: $v"
  done
done
