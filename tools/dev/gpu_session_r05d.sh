#!/bin/bash
# round 5, GPU session D: skip-largest-component rule (whole GPU suite for bit-exactness), seed simulation + sit-out traces, pipeline A/B, fp16 ablation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 900 bash tools/dev/seedsim_trace.sh > $O/seedsim.txt 2>&1; cat $O/seedsim.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 600 python tools/ab_pipeline.py 48 3 sitout: nositout:VG_CLUSTER_SITOUT=0 2>&1 | grep -E "median|round" > $O/ab_sitout.txt; grep median $O/ab_sitout.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 300 python tools/bench_cluster5d.py > $O/cluster5d_on.txt 2>&1; tail -n 8 $O/cluster5d_on.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so VG_CLUSTER_SITOUT=0 timeout 300 python tools/bench_cluster5d.py > $O/cluster5d_off.txt 2>&1; tail -n 8 $O/cluster5d_off.txt
timeout 900 python tools/fp16_ablation.py 3 400 > $O/fp16_ablation.txt 2>&1; tail -n 22 $O/fp16_ablation.txt
