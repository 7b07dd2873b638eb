#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/cldbg; mkdir -p $O
VG_CLUSTER_DEBUG=1 timeout 300 python tools/bench_cluster.py > $O/dbg.txt 2>&1; grep "cluster dbg" $O/dbg.txt | head -12; tail -3 $O/dbg.txt
timeout 300 python tools/exp_nocluster.py > $O/nocluster.txt 2>&1; tail -8 $O/nocluster.txt
