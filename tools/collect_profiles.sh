#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summary + separate PMC passes for the dominant kernel
# (ViT projection GEMM) on the SAME command the bench uses, sequential mode (1 frame in flight) so that kernel
# durations are not inflated by other streams.  PMC passes are separate from --stats as the guide prescribes
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: one pass each).
# usage: tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
BENCH="$GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --blocks 1 --inflight 1 --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $BENCH > $OUT.trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python3 $BENCH > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python3 $BENCH > $OUT.write.log 2>&1
# MFMA pipe utilisation of the GEMM: busy cycles of the matrix pipe vs the time the GPU was active (SQ and GRBM slots, own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o bench -- python3 $BENCH > $OUT.sq.log 2>&1
# the raw per-dispatch CSVs are tens of MB each: reduce them here and take home only the summaries (gpurun_out/ comes back <= 64 MiB)
for f in fetch write sq; do tail -n 2 $OUT.$f.log; done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
VG_PROFILE_OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG python3 $GRAFT_REPO_ROOT/tools/summarize_profiles.py $TAG | tail -20
cp $OUT/trace/bench_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG/ 2>/dev/null
rm -rf $OUT
ls $GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
