#!/bin/bash
# on the GPU box: per-launch durations of k_cl_b_search for ONE MST of a 150k-point frame, development library, in dispatch order:
#   VG_CLUSTER_SEEDSIM=1 -> round 1 is searched twice, the second launch walks only the points a k-NN seed could not serve
#   VG_CLUSTER_SITOUT=0 / default -> the rounds with and without the largest component searching
#   VG_CLUSTER_BOUNDSIM=1 (round 6) -> every round is searched twice, the second launch with every component's TRUE minimum published from the
#     start: the floor of any search that shares a component bound better (printed in pairs: real launch, perfect-bound launch)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/seedsim
cd /tmp && export TMPDIR=/tmp
export VILGOD_HIP_LIB=$GRAFT_REPO_ROOT/vilgod_amd/libvilgod_hip_dev.so
for mode in ${MODES:-seedsim sitout_on sitout_off}; do
  case $mode in
    seedsim) export VG_CLUSTER_SEEDSIM=1; unset VG_CLUSTER_SITOUT; unset VG_CLUSTER_BOUNDSIM;;
    boundsim) unset VG_CLUSTER_SEEDSIM; unset VG_CLUSTER_SITOUT; export VG_CLUSTER_BOUNDSIM=1;;
    sitout_on) unset VG_CLUSTER_SEEDSIM; unset VG_CLUSTER_SITOUT; unset VG_CLUSTER_BOUNDSIM;;
    sitout_off) unset VG_CLUSTER_SEEDSIM; export VG_CLUSTER_SITOUT=0;;
    xcd_on) unset VG_CLUSTER_SEEDSIM; unset VG_CLUSTER_SITOUT; export VG_CLUSTER_XCD_ORDER=1;;
    xcd_off) unset VG_CLUSTER_SEEDSIM; unset VG_CLUSTER_SITOUT; export VG_CLUSTER_XCD_ORDER=0;;
  esac
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -o cl -- python3 $GRAFT_REPO_ROOT/tools/bench_cluster.py > $OUT.$mode.log 2>&1
  grep -E "seedsim|mst \(GPU" $OUT.$mode.log | tail -3
  python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open('$OUT/$mode/cl_kernel_trace.csv')) if 'k_cl_' in r['Kernel_Name'] or 'rocprim' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the LAST complete MST of the run: from the last k_cl_b_init on
last = max(i for i, r in enumerate(rows) if 'k_cl_b_init' in r['Kernel_Name'])
first = max(i for i, r in enumerate(rows[:last]) if 'k_cl_bbox' in r['Kernel_Name'])
one = rows[first:]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('$mode: search launches (us):', ' '.join(f'{dur(r):.0f}' for r in one if 'k_cl_b_search' in r['Kernel_Name']))
print('$mode: MST kernels sum %.0f us, wall from first to last kernel %.0f us, %d launches' % (sum(dur(r) for r in one), (int(one[-1]['End_Timestamp']) - int(one[0]['Start_Timestamp'])) / 1e3, len(one)))
PY
done
