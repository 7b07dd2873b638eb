"""Per-launch durations of the clustering search kernels from a rocprofv3 kernel trace (gpurun_out/cl_trace/cl_kernel_trace.csv):
the last MST run's Boruvka rounds in order."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/cl_trace/cl_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
# last run = after the last k_cl_bbox
last = max(i for i, (n, _) in enumerate(seq) if n.startswith('k_cl_bbox'))
out = [(n, us) for n, us in seq[last:] if 'search' in n or 'core' in n]
print('  '.join(f"{n.replace('k_cl_', '')}:{us:.0f}" for n, us in out))
