#!/bin/bash
# round 5, GPU session C: tests of the last-block Q change, frame latency, fill-ramp A/B on 20-frame blocks, bench runs (20-step x3 blocks, 96-step)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_vit.py tests/test_gemm.py tests/test_pipeline.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
timeout 300 python tools/frame_latency.py 8 > $O/frame_latency.txt 2>&1; cat $O/frame_latency.txt
timeout 600 python tools/ab_pipeline.py 20 6 ramp1:VILGOD_FILL_RAMP=1 ramp2:VILGOD_FILL_RAMP=2 ramp3:VILGOD_FILL_RAMP=3 2>&1 | grep -E "median|round" > $O/ab_ramp.txt; grep median $O/ab_ramp.txt
VILGOD_FRAME_LOG=1 timeout 300 python tools/ab_pipeline.py 20 1 base: > $O/frame_log.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 1500 $O/bench_k20.json
timeout 900 python bench.py --blocks 1 --no-extras --no-cpu-baseline > $O/bench_k96.json 2> $O/bench_k96.err; head -c 600 $O/bench_k96.json
