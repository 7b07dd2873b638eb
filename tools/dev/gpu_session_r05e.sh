#!/bin/bash
# round 5, GPU session E: attention stagger (bit-identity test, kernel timing interleaved, pipeline A/B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_vit.py -x -q -m gpu -k "attention or last_block or rendered" > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
CROPS=337 timeout 300 python tools/time_attention.py > $O/time_attention.txt 2>&1; cat $O/time_attention.txt
timeout 600 python tools/ab_pipeline.py 48 3 stagger: nostagger:VG_ATT_STAGGER=0 2>&1 | grep -E "median|round" > $O/ab_stagger.txt; grep median $O/ab_stagger.txt
