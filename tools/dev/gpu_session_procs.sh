#!/bin/bash
# entry point with 1 / 2 / 3 processes on one GPU (device.processes_per_gpu), four / six sequences
cd $GRAFT_REPO_ROOT
O=gpurun_out/procs; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_cli.py -q -x -m gpu -k "two_processes_per_gpu or sequence_sharding" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
SEQUENCES=4 python tools/time_cli.py 199 150000 device.box_mode=reference > $O/p1.txt 2>&1; head -1 $O/p1.txt
SEQUENCES=4 PROCS=2 python tools/time_cli.py 199 150000 device.box_mode=reference > $O/p2.txt 2>&1; head -1 $O/p2.txt
SEQUENCES=4 PROCS=2 python tools/time_cli.py 199 150000 device.box_mode=reference device.frames_in_flight=4 > $O/p2_f4.txt 2>&1; head -1 $O/p2_f4.txt
SEQUENCES=4 PROCS=2 python tools/time_cli.py 199 150000 device.box_mode=reference device.frames_in_flight=3 > $O/p2_f3.txt 2>&1; head -1 $O/p2_f3.txt
SEQUENCES=6 PROCS=3 python tools/time_cli.py 199 150000 device.box_mode=reference device.frames_in_flight=3 > $O/p3_f3.txt 2>&1; head -1 $O/p3_f3.txt
SEQUENCES=8 PROCS=2 python tools/time_cli.py 199 150000 device.box_mode=reference device.frames_in_flight=4 > $O/p2_s8.txt 2>&1; head -1 $O/p2_s8.txt
