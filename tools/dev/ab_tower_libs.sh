#!/bin/bash
# on the GPU box: the ViT tower (two encodes in flight, tools/exp_tile_tail.py) with two builds of the library, processes alternating on one box
# usage: tools/dev/ab_tower_libs.sh libA.so libB.so [rounds]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-3}); do
  for lib in $1 $2; do
    echo "== $lib"
    VILGOD_HIP_LIB=$GRAFT_REPO_ROOT/$lib COUNTS=328,337 REP=10 timeout 300 python tools/exp_tile_tail.py 2>&1 | grep "round 1"
  done
done
