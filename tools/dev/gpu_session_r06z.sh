#!/bin/bash
# round 6, validation session on one box: the whole GPU suite, smoke, the driver's bench command (both GEMM families), the default bench
# command, the rocprofv3 passes behind profiles/r06b_*, the GEMM micro-benchmark against hipBLASLt, the cycle stamps of the w4 K loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; head -c 400 $O/bench_k20.json; echo
VG_GEMM_W4=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sequence-pass > $O/bench_k20_pp64.json 2> $O/bench_k20_pp64.err; head -c 300 $O/bench_k20_pp64.json; echo
timeout 1500 python bench.py --no-extras --no-cpu-baseline --no-sequence-pass > $O/bench_k96.json 2> $O/bench_k96.err; head -c 300 $O/bench_k96.json; echo
timeout 1200 bash tools/collect_profiles.sh r06b > $O/collect.txt 2>&1; tail -n 4 $O/collect.txt
timeout 400 python tools/bench_gemm_w4.py --json $O/w4.json > $O/w4.txt 2>&1; tail -n 8 $O/w4.txt
timeout 300 python tools/dev/w4_trace.py > $O/w4_trace.txt 2>&1; grep -v wave $O/w4_trace.txt | tail -9
timeout 600 bash tools/trace_cluster.sh > $O/cluster_trace.txt 2>&1; grep -E "k_cl_b_search|clustering kernels" $O/cluster_trace.txt | cut -c1-130
timeout 600 bash tools/dev/trace_gemm_inflight.sh > $O/gemm_inflight.txt 2>&1; grep inflight $O/gemm_inflight.txt
