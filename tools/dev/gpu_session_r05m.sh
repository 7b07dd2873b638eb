#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster.py tests/test_entropy.py tests/test_pipeline.py tests/test_integration.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
timeout 600 bash tools/trace_cluster.sh > $O/cluster_trace.txt 2>&1; grep -E "k_cl_b_search|k_cl_b_purity|k_cl_b_emit|clustering kernels|mst \(GPU" $O/cluster_trace.txt gpurun_out/cl_trace.log | cut -c1-150
timeout 300 python tools/frame_latency.py 8 > $O/frame_latency.txt 2>&1; grep -E "mst_kernels|total" $O/frame_latency.txt
