#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 300 python tools/exp_launch_noise.py > $O/launch_noise.txt 2>&1; cat $O/launch_noise.txt
timeout 300 python -m pytest tests/test_edge.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
