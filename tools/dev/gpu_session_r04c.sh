set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 900 python tools/ab_pipeline.py 48 4 nt256:VG_CLUSTER_SEARCH_NT=256 nt512:VG_CLUSTER_SEARCH_NT=512 > gpurun_out/r04c/ab.txt 2>&1
tail -12 gpurun_out/r04c/ab.txt
