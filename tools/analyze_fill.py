"""GPU busy / idle timeline of the timed region of a `bench.py --steps K` run from a rocprofv3 kernel trace (development aid):
    rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python3 bench.py --steps 20 --warmup 5 --no-roofline-pass ...
    python tools/analyze_fill.py DIR/bench_kernel_trace.csv 20
The timed region is found as the span from the first kernel of the K-th last ground pass (k_pw_classify) to the end of the trace."""
import csv, sys
path, K = sys.argv[1], int(sys.argv[2])
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [s for s, e, k in rows if k.startswith('k_pw_classify')]
t0 = starts[-K]
reg = [(s - t0, e - t0, k) for s, e, k in rows if s >= t0]
t_end = max(e for s, e, k in reg)
print(f'timed region: {t_end / 1e6:.2f} ms, {len(reg)} kernels, {t_end / 1e6 / K:.2f} ms per frame')
# union of busy intervals
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, k in reg:
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'busy {busy / 1e6:.2f} ms, idle {(t_end - busy) / 1e6:.2f} ms in {len(gaps)} gaps')
# idle time per 10 ms slice, and which kernels dominate each slice
import collections
slices = collections.defaultdict(float)
for a, b in gaps:
    slices[int(a // 10e6)] += (b - a) / 1e6
kt = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, k in reg:
    kt[int(s // 10e6)][k] += (e - s) / 1e6
for i in range(int(t_end // 10e6) + 1):
    top = sorted(kt[i].items(), key=lambda kv: -kv[1])[:3]
    print(f'  {10 * i:4d}-{10 * i + 10:4d} ms: idle {slices[i]:5.2f} ms   ' + ', '.join(f'{k} {v:.1f}' for k, v in top))
# when each frame's classification ends (k_head: once per frame)
# the ViT passes: from a frame's k_embed_lnpre to its k_head (passes may overlap, two at a time)
emb = sorted(s for s, e, k in reg if 'k_embed_lnpre' in k)
heads = sorted(e for s, e, k in reg if 'k_head' in k)
print('ViT pass starts (ms):', ' '.join(f'{t / 1e6:.1f}' for t in emb))
print('ViT pass ends   (ms):', ' '.join(f'{t / 1e6:.1f}' for t in heads))
last_gemm = max(e for s, e, k in reg if 'k_gemm' in k)
print(f'last GEMM ends {last_gemm / 1e6:.1f} ms, last kernel of any kind {t_end / 1e6:.1f} ms; kernels after the last GEMM:',
      ', '.join(sorted({k[:40] for s, e, k in reg if s > last_gemm})))
