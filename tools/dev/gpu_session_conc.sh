#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/conc; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 900 python tools/ab_pipeline.py 48 3 c2w6: c3w6:VILGOD_VIT_CONCURRENCY=3 c2w8:AB_WORKERS=8 c3w8:VILGOD_VIT_CONCURRENCY=3,AB_WORKERS=8 c4w8:VILGOD_VIT_CONCURRENCY=4,AB_WORKERS=8 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
