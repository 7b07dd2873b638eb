set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout 1500 python -m pytest tests/test_cli.py tests/test_integration.py tests/test_tracking.py -x -q -m gpu > gpurun_out/r04f/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r04f/pytest.txt
tail -5 gpurun_out/r04f/pytest.txt
SEQUENCES=3 VILGOD_STAGE_DETAIL=1 timeout 600 python tools/time_cli.py 199 150000 > gpurun_out/r04f/time_cli.txt 2>&1
tail -8 gpurun_out/r04f/time_cli.txt
SEQUENCES=3 timeout 600 python tools/time_cli.py 199 150000 device.overlap_sequences=false > gpurun_out/r04f/time_cli_noov.txt 2>&1
tail -5 gpurun_out/r04f/time_cli_noov.txt
