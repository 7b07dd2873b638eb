cd $GRAFT_REPO_ROOT
echo "=== compact (default)"; bash tools/trace_cluster.sh 2>&1 | grep "k_cl_b_search\|k_cl_b_active\|clustering kernels\|k_cl_core"
echo "=== walk all"; VG_CLUSTER_COMPACT=0 bash tools/trace_cluster.sh 2>&1 | grep "k_cl_b_search\|k_cl_b_active\|clustering kernels\|k_cl_core"
