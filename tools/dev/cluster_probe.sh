#!/bin/bash
# on the GPU box: clustering parity tests, the kernel table of tools/trace_cluster.sh (first lines) and the dev build's per-wave walk statistics
set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cluster.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t5.log
bash tools/trace_cluster.sh 2>&1 | head -${1:-6} > gpurun_out/cl5.log
grep -E "clustering kernels" gpurun_out/cl_trace.log >> gpurun_out/cl5.log
bash tools/trace_cluster.sh 2>&1 | grep -E "clustering kernels" >> gpurun_out/cl5.log
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so VG_CLUSTER_DEBUG=1 python tools/bench_cluster.py 2>&1 | grep "cluster dbg" | head -38 | tail -35 >> gpurun_out/cl5.log
