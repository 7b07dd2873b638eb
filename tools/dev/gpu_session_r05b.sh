#!/bin/bash
# round 5, GPU session B: whole GPU suite on the new host tree / pack / masks code, frame latency, pipeline A/B under CU masks (one variant per process)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python tools/frame_latency.py 8 > $O/frame_latency.txt 2>&1; cat $O/frame_latency.txt
for rep in 1 2; do
  for v in base: r4:VILGOD_CU_RESERVE=4 r6:VILGOD_CU_RESERVE=6 r8:VILGOD_CU_RESERVE=8 r4all:VILGOD_CU_RESERVE=4,VILGOD_CU_TOWER=all; do
    timeout 400 python tools/ab_pipeline.py 48 3 $v 2>&1 | grep -E "median|round" >> $O/ab_masks.txt
  done
done
cat $O/ab_masks.txt | grep median
