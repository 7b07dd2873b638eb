#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/splitk; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout 600 python tools/ab_pipeline.py 48 3 split8:VG_GEMM_SPLITK=8 plain:VG_GEMM_SPLITK=0 split4:VG_GEMM_SPLITK=4 > $O/ab2.txt 2>&1; tail -12 $O/ab2.txt
for m in 0 8 4; do VG_GEMM_SPLITK=$m timeout 300 python tools/exp_tile_tail.py > $O/tail_$m.txt 2>&1; echo "== SPLITK=$m"; tail -12 $O/tail_$m.txt; done
