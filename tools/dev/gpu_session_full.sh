#!/bin/bash
# full GPU validation: the whole -m gpu suite, then the final session (bench K20, default bench, profiles)
cd $GRAFT_REPO_ROOT
TAG=${1:-r04d}
mkdir -p gpurun_out/$TAG
export GPU_MAX_HW_QUEUES=16
timeout 3000 python -m pytest tests -q -x -m gpu > gpurun_out/$TAG/pytest.txt 2>&1; tail -5 gpurun_out/$TAG/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.txt 2>&1; tail -2 gpurun_out/$TAG/smoke.txt
bash tools/dev/gpu_session_r04final.sh $TAG
