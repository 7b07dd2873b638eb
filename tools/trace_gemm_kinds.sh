#!/bin/bash
# on the GPU box: kernel trace of a sequential bench run; average duration of every ViT GEMM by its position in the block
# (in_proj, out_proj, c_fc, c_proj) and by layer (development aid)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/tgk
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras ${1:-} > $OUT.log 2>&1
python3 - <<PY
import csv, collections
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open('$(find $OUT -name "bench_kernel_trace.csv")'))]
rows.sort()
seq = [(d, k) for s, d, k in rows if 'k_gemm_f16_pp64' in k or 'k_head' in k]
# split into frames at k_head
frames, cur = [], []
for d, k in seq:
    if 'k_head' in k:
        if len(cur) == 49: frames.append(cur)
        cur = []
    else:
        cur.append(d)
frames = frames[2:]          # skip the first (plain / capture) frames
print('frames used', len(frames))
names = ['in_proj', 'out_proj', 'c_fc', 'c_proj']
for j, nm in enumerate(names):
    per_layer = [sum(f[1 + 4 * l + j] for f in frames) / len(frames) / 1e3 for l in range(12)]
    print(f'{nm:9s} avg {sum(per_layer) / 12:7.1f} us   per layer: ' + ' '.join(f'{v:5.0f}' for v in per_layer))
print('patch embedding', sum(f[0] for f in frames) / len(frames) / 1e3)
PY
rm -rf $OUT
