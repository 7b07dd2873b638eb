set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_cluster.py tests/test_entropy.py tests/test_edge.py -x -q -m gpu > gpurun_out/r04h/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04h/pytest.txt
tail -4 gpurun_out/r04h/pytest.txt
timeout 900 python tools/ab_pipeline.py 48 4 walk_all:VG_CLUSTER_COMPACT=0 compact:VG_CLUSTER_COMPACT=1 > gpurun_out/r04h/ab.txt 2>&1
tail -12 gpurun_out/r04h/ab.txt
