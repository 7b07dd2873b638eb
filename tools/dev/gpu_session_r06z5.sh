#!/bin/bash
# round 6, the rocprofv3 passes of the round's LAST state (hierarchy stage on the device): profiles/r06d_*
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z8; mkdir -p $O
timeout 1500 bash tools/collect_profiles.sh r06e > $O/collect.txt 2>&1; tail -n 4 $O/collect.txt
ls gpurun_out/profiles_r06e 2>/dev/null | head
