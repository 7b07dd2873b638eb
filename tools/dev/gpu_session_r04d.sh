set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
VILGOD_STAGE_DETAIL=1 timeout 600 python tools/time_cli.py 199 150000 > gpurun_out/r04d/time_cli.txt 2>&1
cat gpurun_out/r04d/time_cli.txt | tail -4
VILGOD_STAGE_DETAIL=1 timeout 600 python tools/time_cli.py 199 150000 >> gpurun_out/r04d/time_cli.txt 2>&1
tail -3 gpurun_out/r04d/time_cli.txt
timeout 900 python tools/profile_cli.py 199 150000 > gpurun_out/r04d/profile_cli.txt 2>&1
head -60 gpurun_out/r04d/profile_cli.txt
