#!/bin/bash
# round 5, GPU session A: CU-mask probe + interference sweep + frame latency + pipeline A/B under masks + per-kind GEMM PMC
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 120 tools/micro/cu_mask_probe > $O/cu_mask_probe.txt 2>&1
timeout 600 python tools/exp_interference.py 1 2 3 4 > $O/interference.txt 2>&1
timeout 300 python tools/frame_latency.py 8 > $O/frame_latency.txt 2>&1
timeout 600 python -m pytest tests/test_pipeline.py -x -q -m gpu -k "cu_masked" > $O/pytest_masks.txt 2>&1
timeout 900 python tools/ab_pipeline.py 48 3 base: r1:VILGOD_CU_RESERVE=1 r2:VILGOD_CU_RESERVE=2 r3:VILGOD_CU_RESERVE=3 r4:VILGOD_CU_RESERVE=4 r2all:VILGOD_CU_RESERVE=2,VILGOD_CU_TOWER=all r4all:VILGOD_CU_RESERVE=4,VILGOD_CU_TOWER=all > $O/ab_masks.txt 2>&1
timeout 1500 bash tools/collect_profiles.sh r05a > $O/collect.txt 2>&1
tail -5 $O/*.txt
