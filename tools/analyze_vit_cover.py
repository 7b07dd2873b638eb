"""Where does the step time go beyond the ViT?  From a rocprofv3 kernel trace of `bench.py --steps K` (development aid):
    python tools/analyze_vit_cover.py DIR/bench_kernel_trace.csv K
Timed region = from the first kernel of the K-th last ground pass to the end.  Reports how long at least one / at least two ViT kernels
(projection GEMMs, attention, embedding, head, scores) were running, and what ran while none was."""
import csv, sys, collections
path, K = sys.argv[1], int(sys.argv[2])
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [s for s, e, k in rows if k.startswith('k_pw_classify')]
t0 = starts[-K]
reg = [(s - t0, e - t0, k) for s, e, k in rows if s >= t0]
T = max(e for s, e, k in reg)
def is_vit(k):
    return any(x in k for x in ('k_gemm', 'k_attention', 'k_embed', 'k_head', 'k_clip_scores', 'k_gather_cls', 'k_layernorm'))
ev = []
for s, e, k in reg:
    v = is_vit(k)
    ev.append((s, 1, v, k)); ev.append((e, -1, v, k))
ev.sort(key=lambda x: (x[0], x[1]))
nv = no = 0
last = 0
t_v1 = t_v2 = t_none_busy = t_idle = 0
other_when_no_vit = collections.defaultdict(float)
running = collections.Counter()
for t, d, v, k in ev:
    dt = t - last
    if dt > 0:
        if nv >= 1: t_v1 += dt
        if nv >= 2: t_v2 += dt
        if nv == 0 and no > 0:
            t_none_busy += dt
            for kk, c in running.items():
                if c > 0: other_when_no_vit[kk] += dt
        if nv == 0 and no == 0: t_idle += dt
    last = t
    if v: nv += d
    else:
        no += d
        running[k] += d
print(f'timed region {T/1e6:.2f} ms = {T/1e6/K:.3f} ms per frame over {K} frames')
print(f'  >= 1 ViT kernel running: {t_v1/1e6:.2f} ms ({100*t_v1/T:.1f} %), >= 2: {t_v2/1e6:.2f} ms ({100*t_v2/T:.1f} %)')
print(f'  no ViT kernel, other kernels running: {t_none_busy/1e6:.2f} ms ({100*t_none_busy/T:.1f} %); GPU idle: {t_idle/1e6:.2f} ms ({100*t_idle/T:.1f} %)')
print('  kernels running while no ViT kernel was (ms, may overlap):')
for k, v in sorted(other_when_no_vit.items(), key=lambda kv: -kv[1])[:14]:
    print(f'    {k[:60]:60s} {v/1e6:8.2f}')
vit_sum = sum(e - s for s, e, k in reg if is_vit(k))
print(f'  sum of ViT kernel durations {vit_sum/1e6:.2f} ms = {vit_sum/1e6/K:.3f} ms per frame (kernels of two passes overlap and stretch each other)')
# per kernel class: summed duration per frame in this (concurrent) run
cls = collections.defaultdict(float)
for s, e, k in reg:
    cls[k[:50]] += e - s
print('  top kernels by summed duration per frame in this run:')
for k, v in sorted(cls.items(), key=lambda kv: -kv[1])[:12]:
    print(f'    {k:50s} {v/1e6/K:8.3f} ms/frame')
