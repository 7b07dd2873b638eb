#!/bin/bash
# on the GPU box: clustering parity tests + kernel table for each variant library given (tags of tools/dev/build_variant.sh)
cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  lib=$PWD/vilgod_amd/libvilgod_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/vilgod_amd/libvilgod_hip.so
  echo "== $tag"
  VILGOD_HIP_LIB=$lib python -m pytest tests/test_cluster.py -x -q -m gpu 2>&1 | tail -1
  VILGOD_HIP_LIB=$lib bash tools/trace_cluster.sh 2>&1 | grep -E "b_search|clustering kernels"
done
