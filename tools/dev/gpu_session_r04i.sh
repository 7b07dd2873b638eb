set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
timeout 900 python -m pytest tests/test_gemm.py tests/test_vit.py -x -q -m gpu > gpurun_out/r04i/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04i/pytest.txt
tail -4 gpurun_out/r04i/pytest.txt
CROPS=337 python tools/bench_gemm_epi.py 2>&1 | grep -v amdgpu
python tools/ab_pipeline.py 48 3 new: 2>&1 | grep -v amdgpu | tail -4
