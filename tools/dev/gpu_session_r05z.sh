#!/bin/bash
# round 5, last GPU session: the committed tree once more -- whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; head -c 600 $O/bench_k20.json; echo
timeout 600 bash tools/trace_cluster.sh > $O/cluster_trace.txt 2>&1; grep -E "k_cl_b_search|clustering kernels" $O/cluster_trace.txt | cut -c1-130
