set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 1500 python -m pytest tests/test_cli.py tests/test_integration.py tests/test_tracking.py tests/test_pipeline.py -x -q -m gpu > gpurun_out/r04e/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04e/pytest.txt
tail -5 gpurun_out/r04e/pytest.txt
SEQUENCES=2 VILGOD_STAGE_DETAIL=1 timeout 600 python tools/time_cli.py 199 150000 > gpurun_out/r04e/time_cli.txt 2>&1
tail -6 gpurun_out/r04e/time_cli.txt
