#!/bin/bash
# round 5, GPU session G: attention with the deferred output of waves 4-6, O(1) level-0 purity, first batch of Boruvka rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_vit.py tests/test_cluster.py tests/test_entropy.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
CROPS=337 timeout 300 python tools/time_attention.py > $O/time_attention.txt 2>&1; cat $O/time_attention.txt
timeout 900 bash tools/dev/seedsim_trace.sh > $O/mst_trace.txt 2>&1; grep -E "sitout_on|sitout_off" $O/mst_trace.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 600 python tools/ab_pipeline.py 48 3 fb6: fb8:VG_CLUSTER_FIRST_BATCH=8 2>&1 | grep -E "median" > $O/ab_fb.txt; cat $O/ab_fb.txt
timeout 600 python tools/ab_pipeline.py 48 3 stagger: nostagger:VG_ATT_STAGGER=0 2>&1 | grep -E "median" > $O/ab_stagger.txt; cat $O/ab_stagger.txt
