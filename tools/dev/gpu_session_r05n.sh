#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster.py tests/test_entropy.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
MODES="xcd_on xcd_on" timeout 900 bash tools/dev/seedsim_trace.sh 2>&1 | grep -E "^xcd"
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 300 python tools/bench_cluster5d.py 2>&1 | tail -n 1
