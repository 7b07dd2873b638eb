#!/bin/bash
# round 5, GPU session H: attention stagger variants in one binary; sit-out rule A/B in the pipeline and in the default-configuration sequence mode
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests/test_vit.py -x -q -m gpu -k "attention" > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
CROPS=337 timeout 300 python tools/time_attention.py > $O/time_attention.txt 2>&1; cat $O/time_attention.txt
CROPS=331 timeout 300 python tools/time_attention.py > $O/time_attention331.txt 2>&1; cat $O/time_attention331.txt
timeout 900 python tools/ab_pipeline.py 48 3 stag1:VG_ATT_STAGGER=1 stag2:VG_ATT_STAGGER=2 stag0:VG_ATT_STAGGER=0 2>&1 | grep -E "median" > $O/ab_stagger.txt; cat $O/ab_stagger.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 600 python tools/ab_pipeline.py 48 3 sitout: nositout:VG_CLUSTER_SITOUT=0 2>&1 | grep -E "median" > $O/ab_sitout.txt; cat $O/ab_sitout.txt
VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so timeout 900 python tools/ab_sequence.py 48 4 sitout: nositout:VG_CLUSTER_SITOUT=0 > $O/ab_sequence.txt 2>&1; tail -n 3 $O/ab_sequence.txt
