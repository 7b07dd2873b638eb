#!/bin/bash
# round 5, GPU session F: validation of the round's state -- whole GPU suite, smoke, A/B of two scheduling knobs, both bench commands, profile passes
cd $GRAFT_REPO_ROOT
TAG=${1:-r05b}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; head -c 700 $O/bench_k20.json; echo
timeout 1500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; head -c 700 $O/bench.json; echo
timeout 600 bash tools/trace_cluster.sh > $O/cluster_trace.txt 2>&1; tail -n 30 $O/cluster_trace.txt
timeout 1500 bash tools/collect_profiles.sh $TAG > $O/collect.txt 2>&1; tail -n 12 $O/collect.txt
