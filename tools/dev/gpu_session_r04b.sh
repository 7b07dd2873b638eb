set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gemm.py tests/test_vit.py tests/test_render.py tests/test_ground.py -q -m gpu > gpurun_out/r04b/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04b/pytest.txt
tail -5 gpurun_out/r04b/pytest.txt
timeout 900 python tools/exp_tile_tail.py > gpurun_out/r04b/tile_tail.txt 2>&1
cat gpurun_out/r04b/tile_tail.txt
