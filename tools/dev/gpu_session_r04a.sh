set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
./tools/micro/valu_rate > gpurun_out/r04a/valu_rate.txt 2>&1
timeout 900 python -m pytest tests/test_gemm.py tests/test_vit.py tests/test_render.py -x -q -m gpu > gpurun_out/r04a/pytest_a.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/pytest_a.txt
tail -5 gpurun_out/r04a/pytest_a.txt
timeout 300 python tools/bench_gemm_ri.py > gpurun_out/r04a/gemm_ri.txt 2>&1
COLD=0 timeout 300 python tools/bench_gemm_ri.py >> gpurun_out/r04a/gemm_ri.txt 2>&1
cat gpurun_out/r04a/gemm_ri.txt
timeout 900 python tools/ab_pipeline.py 48 3 base:VG_GEMM_RI=0,VILGOD_PATCH_1CH=0 ri:VG_GEMM_RI=1,VILGOD_PATCH_1CH=0 ri1ch:VG_GEMM_RI=1,VILGOD_PATCH_1CH=1 > gpurun_out/r04a/ab.txt 2>&1
tail -12 gpurun_out/r04a/ab.txt
