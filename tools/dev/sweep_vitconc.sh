#!/bin/bash
# on the GPU box: ViT passes at a time (VILGOD_VIT_CONCURRENCY) in the driver's block length and in 48-step blocks, one process each, alternating
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for c in 2 3 4 0; do
    for steps in 20 48; do
      v=$(VILGOD_VIT_CONCURRENCY=$c timeout 400 python bench.py --steps $steps --warmup 5 --no-extras --no-cpu-baseline --no-roofline-pass --no-sequence-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['block_values'])")
      echo "vit_concurrency $c steps $steps: $v"
    done
  done
done
