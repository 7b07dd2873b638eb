#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster.py tests/test_entropy.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
MODES="pipe_on pipe_off pipe_on pipe_off" timeout 900 bash tools/dev/seedsim_trace.sh 2>&1 | grep -E "^pipe"
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so VG_CLUSTER_PIPE=1 timeout 300 python tools/bench_cluster5d.py 2>&1 | tail -n 1
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so VG_CLUSTER_PIPE=0 timeout 300 python tools/bench_cluster5d.py 2>&1 | tail -n 1
