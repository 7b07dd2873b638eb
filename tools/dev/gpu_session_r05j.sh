#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_render.py tests/test_vit.py tests/test_pipeline.py tests/test_integration.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python tools/time_render.py > $O/time_render.txt 2>&1; cat $O/time_render.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o r -- python3 $GRAFT_REPO_ROOT/tools/time_render.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
grep -E "k_render|k_to_origin|k_cluster_median|k_gather_ego" $GRAFT_REPO_ROOT/$O/trace/r_kernel_stats.csv | cut -c1-120
