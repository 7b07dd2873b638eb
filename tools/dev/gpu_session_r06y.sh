#!/bin/bash
# round 6, last GPU session: the committed tree once more -- whole GPU suite, smoke, the driver's bench command, the rocprofv3 passes behind profiles/r06c_*
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; head -c 300 $O/bench_k20.json; echo
timeout 1200 bash tools/collect_profiles.sh r06c > $O/collect.txt 2>&1; tail -n 3 $O/collect.txt
timeout 300 python tools/dev/w4_trace.py > $O/w4_trace.txt 2>&1; grep -v wave $O/w4_trace.txt | tail -9
timeout 400 python tools/bench_blas_ceiling.py > $O/blas_ceiling.txt 2>&1; tail -n 1 $O/blas_ceiling.txt > $O/blas_ceiling.json; head -n 4 $O/blas_ceiling.txt | cut -c1-200
timeout 400 python tools/bench_gemm_w4.py --json $O/w4.json > $O/w4.txt 2>&1; tail -n 7 $O/w4.txt | cut -c1-220
