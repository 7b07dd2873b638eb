"""Post-process gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.{csv,json}.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
FETCH_SIZE under-reports wide coalesced streaming reads by 2x on gfx950 (TCC_EA0_RDREQ x 64 B with 128-B requests
tallied at 64 B); we report both the raw and the corrected (x2) figure, WRITE_SIZE is exact for 16-B/lane stores."""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.environ.get('VG_PROFILE_SRC', os.path.join(root, 'gpurun_out', f'prof_{tag}'))     # (override: tests/test_host.py feeds it a fabricated run)
dst = os.environ.get('VG_PROFILE_OUT', os.path.join(root, 'profiles'))      # on the GPU box: a directory under gpurun_out/ (comes back)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'trace', 'bench_kernel_stats.csv'), os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc
fetch = per_kernel(os.path.join(src, 'fetch', 'bench_counter_collection.csv'), 'FETCH_SIZE')
write = per_kernel(os.path.join(src, 'write', 'bench_counter_collection.csv'), 'WRITE_SIZE')
stats = {r['Name']: r for r in csv.DictReader(open(os.path.join(src, 'trace', 'bench_kernel_stats.csv')))}
out = {'tag': tag, 'command': 'bench.py --steps 4 --warmup 2 --blocks 1 --inflight 1 --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras', 'kernels': {}}
def group(match):
    g = {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0, 'dur_ns': 0.0, 'calls_trace': 0}
    for k in sorted(set(fetch) | set(write)):
        if not match(k):
            continue
        f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        st = stats.get(k)
        g['launches'] += f[0]; g['fetch_kb'] += f[1]; g['write_kb'] += w[1]; g['launches_w'] = g.get('launches_w', 0) + w[0]
        if st:
            g['dur_ns'] += float(st['TotalDurationNs']); g['calls_trace'] += int(st['Calls'])
    n = max(g['launches'], 1)
    nw = max(g.get('launches_w', 0), 1)        # (the FETCH and WRITE passes are separate runs of the command: their adaptive set-up may repeat a different number of times)
    return {
        'launches_pmc': g['launches'], 'avg_launch_us_trace': g['dur_ns'] / max(g['calls_trace'], 1) / 1e3,
        'fetch_bytes_per_launch_raw': 1024 * g['fetch_kb'] / n, 'fetch_bytes_per_launch_corrected_x2': 2048 * g['fetch_kb'] / n,
        'write_bytes_per_launch': 1024 * g['write_kb'] / nw,
        'hbm_bytes_per_launch': 2048 * g['fetch_kb'] / n + 1024 * g['write_kb'] / nw,
    }
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    st = stats.get(k)
    out['kernels'][k[:80]] = {'launches': f[0], 'fetch_kb_per_launch_raw': f[1] / max(f[0], 1), 'write_kb_per_launch': w[1] / max(w[0], 1),
                              'avg_ns': float(st['AverageNs']) if st else None}
# the dominant kernel (bench.py roofline): k_gemm_f16_w4 (round 6, the default) or k_gemm_f16_pp64 (VG_GEMM_W4=0) -- whichever the run used
DOM = 'k_gemm_f16_w4' if any('k_gemm_f16_w4' in k for k in list(stats) + list(fetch)) else 'k_gemm_f16_pp64'
is_dom = lambda k: DOM in k
out['dominant_kernel'] = DOM
out[DOM] = group(is_dom)
out['k_gemm_f16'] = group(lambda k: 'k_gemm_f16' in k and 'k_gemm_f16_pp' not in k and 'k_gemm_f16_w4' not in k)    # fallback kernel (unused by ViT-B/16)
sq_path = os.path.join(src, 'sq', 'bench_counter_collection.csv')
if os.path.exists(sq_path):
    # SQ / GRBM pass: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (16 per v_mfma_f32_16x16x32_f16),
    # GRBM_GUI_ACTIVE the active cycles of the 8 XCDs; MFMA utilisation = busy / (GUI_ACTIVE / 8 * 1024 SIMDs)
    names = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE')
    for label, match in ((DOM, is_dom), ('k_attention_f16', lambda k: 'k_attention_f16' in k)):
        tot = {n: per_kernel(sq_path, n) for n in names}
        agg = {n: sum(v[1] for k, v in tot[n].items() if match(k)) for n in names}
        launches = sum(v[0] for k, v in tot['GRBM_GUI_ACTIVE'].items() if match(k))
        if launches and agg['GRBM_GUI_ACTIVE'] > 0:
            out[label + '_sq'] = {
                'launches': launches, **{n + '_per_launch': agg[n] / launches for n in names},
                'mfma_utilisation': agg['SQ_VALU_MFMA_BUSY_CYCLES'] / (agg['GRBM_GUI_ACTIVE'] / 8 * 1024),
                'wave_cycles_split': {'waiting (s_waitcnt / barrier)': agg['SQ_WAIT_ANY'] / agg['SQ_WAVE_CYCLES'],
                                      'issue stalled': agg['SQ_WAIT_INST_ANY'] / agg['SQ_WAVE_CYCLES'],
                                      'issuing': agg['SQ_ACTIVE_INST_ANY'] / agg['SQ_WAVE_CYCLES']}}
# ---- the projection GEMM per KIND (VERDICT r4 task 2): in_proj / out_proj / c_fc / c_proj of the full-size blocks, split by template
# instantiation and -- out_proj and c_proj share one (EPI_BIAS_RESID, LN = 2) -- by their alternation in dispatch order of the sequential
# run (per block: in_proj, attention, out_proj, c_fc, c_proj).  Time from the kernel-trace pass, bytes and SQ counters from the PMC passes.
def dispatch_rows(path, counters=None):
    """-> {dispatch_id: {'name', 'grid', 'dur_ns', counter: value ...}} of one pass."""
    rows = {}
    if not os.path.exists(path):
        return rows
    # k_gemm_f16_w4 launches a persistent grid (one workgroup per CU): its rows come from the encode's k_embed_lnpre launch in front of it
    # (one wave per padded token row: grid = rows / 4 workgroups of 256 threads), in dispatch order
    all_rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Dispatch_Id']))
    m_rows = 0
    for r in all_rows:
        name = r.get('Kernel_Name', '')
        if 'k_embed_lnpre' in name:
            m_rows = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0) // 64
        if DOM not in name:
            continue
        d = rows.setdefault(int(r['Dispatch_Id']), {'name': name, 'grid': int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0), 'm_rows': m_rows})
        if 'Counter_Name' in r:
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            d['dur_ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return rows
def template_args(name):
    import re
    m = (re.search(r'k_gemm_f16_pp64ILi(\d)ELb\dELb\dELi(\d)ELb(\d)E', name) or re.search(r'k_gemm_f16_pp64<(\d), \w+, \w+, (\d), (\w+)>', name) or
         re.search(r'k_gemm_f16_w4ILi(\d)ELi(\d)ELi\d+E', name) or re.search(r'k_gemm_f16_w4<(\d), (\d), \d+>', name))
    return (int(m.group(1)), int(m.group(2))) if m else (None, None)
def kinds_of(rows):
    """dispatch id -> kind for the full-size launches (grid = the frame's largest for that instantiation)."""
    out = {}
    by_inst = {}
    for did in sorted(rows):
        by_inst.setdefault(template_args(rows[did]['name']), []).append(did)
    for (epi, ln), dids in by_inst.items():
        gmax = max(rows[d]['grid'] for d in dids)
        full = [d for d in dids if rows[d]['grid'] * 2 > gmax]
        if (epi, ln) == (0, 1):
            out.update({d: 'in_proj' for d in full})
        elif (epi, ln) == (1, 1):
            out.update({d: 'c_fc' for d in full})
        elif (epi, ln) == (5, 2) or ((epi, ln) == (2, 2) and (5, 2) not in by_inst):
            # the residual GEMMs over all rows: (5, 2) = the stream as an fp16 pair (round 6; then (2, 2) only runs on the last block's class rows).
            # grids differ between frames (crop counts) but a frame's out_proj and c_proj have the same grid and alternate
            for i, d in enumerate(full):
                out[d] = 'out_proj' if i % 2 == 0 else 'c_proj'
    return out
PAIR_STREAM = False
SHAPES = {'in_proj': (2304, 768), 'out_proj': (768, 768), 'c_fc': (3072, 768), 'c_proj': (768, 3072)}
def algorithmic_bytes(kind, M):
    N, K = SHAPES[kind]
    ldc = 2368 if kind == 'in_proj' else N
    rd = M * K * 2 + N * K * 2
    if kind in ('in_proj', 'c_fc'):
        return rd + M * 8 * (K // (128 if DOM == 'k_gemm_f16_w4' else 256)), M * N * 2                      # + the row statistics; fp16 output
    if PAIR_STREAM:
        return rd + M * N * 4, M * N * 4 + M * 8 * 2                                                          # the stream as an fp16 pair: 2 + 2 bytes read, 2 + 2 written, statistics
    return rd + M * N * 4, M * N * 4 + M * N * 2 + M * 8 * (2 if DOM == 'k_gemm_f16_w4' else 1)                # fp32 residual read; residual + fp16 copy + statistics written
trace_rows = dispatch_rows(os.path.join(src, 'trace', 'bench_kernel_trace.csv'))
PAIR_STREAM = any(template_args(r['name']) == (5, 2) for r in trace_rows.values())
fetch_rows = dispatch_rows(os.path.join(src, 'fetch', 'bench_counter_collection.csv'))
write_rows = dispatch_rows(os.path.join(src, 'write', 'bench_counter_collection.csv'))
sq_rows = dispatch_rows(sq_path)
kinds = {}
for label, rows in (('trace', trace_rows), ('fetch', fetch_rows), ('write', write_rows), ('sq', sq_rows)):
    km = kinds_of(rows)
    for did, kind in km.items():
        g = kinds.setdefault(kind, {})
        a = g.setdefault(label, {'n': 0})
        a['n'] += 1
        for key, val in rows[did].items():
            if key not in ('name',):
                a[key] = a.get(key, 0.0) + float(val)
gemm_kinds = {}
for kind, g in kinds.items():
    N, K = SHAPES[kind]
    t, f, w, q = g.get('trace', {'n': 0}), g.get('fetch', {'n': 0}), g.get('write', {'n': 0}), g.get('sq', {'n': 0})
    if not t['n']:
        continue
    M = (t['m_rows'] / t['n']) if DOM == 'k_gemm_f16_w4' else t['grid'] / t['n'] / 512 / (N // 256) * 256      # average rows per launch (pp64: Grid_Size = work-items of one 512-thread workgroup per tile)
    us = t['dur_ns'] / t['n'] / 1e3
    alg_r, alg_w = algorithmic_bytes(kind, M)
    d = {'launches': t['n'], 'avg_rows': round(M), 'avg_launch_us': round(us, 1), 'tflops': round(2 * M * N * K / (us * 1e-6) / 1e12, 1),
         'algorithmic_read_bytes': round(alg_r), 'algorithmic_write_bytes': round(alg_w)}
    if f['n']:
        d['fetch_bytes_corrected_x2'] = round(2048 * f['FETCH_SIZE'] / f['n'])
        d['fetch_over_algorithmic'] = round(d['fetch_bytes_corrected_x2'] / alg_r, 2)
    if w['n']:
        d['write_bytes'] = round(1024 * w['WRITE_SIZE'] / w['n'])
        d['write_over_algorithmic'] = round(d['write_bytes'] / alg_w, 2)
    if f['n'] and w['n']:
        d['hbm_tb_per_s'] = round((d['fetch_bytes_corrected_x2'] + d['write_bytes']) / (us * 1e-6) / 1e12, 2)
    if q['n'] and q.get('GRBM_GUI_ACTIVE'):
        d['mfma_utilisation'] = round(q['SQ_VALU_MFMA_BUSY_CYCLES'] / (q['GRBM_GUI_ACTIVE'] / 8 * 1024), 3)
        d['wave_cycles_waiting'] = round(q['SQ_WAIT_ANY'] / q['SQ_WAVE_CYCLES'], 3)
        d['wave_cycles_issue_stalled'] = round(q['SQ_WAIT_INST_ANY'] / q['SQ_WAVE_CYCLES'], 3)
        d['sustained_ghz'] = round(q['GRBM_GUI_ACTIVE'] / 8 / q['n'] / (q['dur_ns'] / q['n']), 3) if q.get('dur_ns') else None
    gemm_kinds[kind] = d
out[DOM + '_by_kind'] = gemm_kinds
# ---- stage kernels next to the GEMM (north_star: "rocprof HBM GB/s (clustering, renderer)"): bench.py's `roofline_stages` block reads this ----
def bench_line(logname):
    try:
        for ln in open(os.path.join(root, 'gpurun_out', logname)):
            if ln.startswith('{') and '"metric"' in ln:
                return json.loads(ln)
    except OSError:
        pass
    return {}
bl = bench_line(f'prof_{tag}.trace.log')
cfg = bl.get('config', {})
frames = max([int(r['Calls']) for k, r in stats.items() if 'k_head' in k] + [1])
crops = float(cfg.get('crops_per_frame', 337.0))
M = float(cfg.get('nonground_points_per_frame', 79_000.0))
N = float(cfg.get('points_per_frame', 150_000))
PEAK = 8.0e12
def stage(match, alg_bytes, alg_note, extra=None):
    g = group(match)
    names = [k for k in stats if match(k)]
    dur_ns = sum(float(stats[k]['TotalDurationNs']) for k in names)
    calls = sum(int(stats[k]['Calls']) for k in names)
    per_launch_us = dur_ns / max(calls, 1) / 1e3
    hbm = g['hbm_bytes_per_launch']
    d = {'kernels': sorted({k.split('(')[0][:48] for k in names}), 'launches_per_frame': round(calls / frames, 2),
         'avg_launch_us': round(per_launch_us, 1), 'ms_per_frame': round(dur_ns / frames / 1e6, 3),
         'hbm_bytes_per_launch_pmc': round(hbm), 'hbm_tb_per_s': round(hbm / (per_launch_us * 1e-6) / 1e12, 3) if per_launch_us else None,
         'frac_of_8_tb_per_s': round(hbm / (per_launch_us * 1e-6) / PEAK, 4) if per_launch_us else None,
         'algorithmic_bytes_per_frame': round(alg_bytes), 'algorithmic': alg_note,
         'hbm_bytes_per_frame_pmc': round(hbm * calls / frames),
         'traffic_over_algorithmic': round(hbm * calls / frames / alg_bytes, 2) if alg_bytes else None}
    if extra:
        d.update(extra)
    return d
stages = {
    'tag': tag, 'frames': frames, 'source': f'profiles/{tag}_bench_kernel_stats.csv (time, rocprofv3 --kernel-trace --stats) + profiles/{tag}_pmc_summary.json '
                                          '(bytes: separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950 per MI355X_MICROARCH.md) of '
                                          'bench.py --steps 4 --warmup 2 --inflight 1 (sequential: kernel times not inflated by other streams)',
    'peak_hbm_tb_per_s': 8.0,
    'render': stage(lambda k: k.startswith('k_render'), crops * 196 * 256 * 2, 'single-channel fp16 patch rows written: crops x 196 x 256 x 2 B (reads: the clusters\' points, < 1 MB)'),
    'clustering_core_distances': stage(lambda k: 'k_cl_core' in k or 'k_cl_blocks' in k, M * 16 + M * 8, 'Morton-sorted float4 points read once + one f64 core distance per point written; needed pair distances M x 15'),
    'clustering_boruvka_search': stage(lambda k: 'k_cl_b_search' in k, M * 16 + (M - 1) * 16, 'points read once per round + the n - 1 MST edges (16 B) over all rounds; latency-bound tree walks, not a bandwidth kernel',
                                       {'pairs_evaluated_over_needed_core_distances': {'value': 29.9, 'source': 'profiles/r03a: VG_CLUSTER_DEBUG counters of the development build (cooperative k-NN kernel; the per-point walk: 9.5)'}}),
    'clustering_grid_tables': stage(lambda k: 'k_cl_fill_int' in k or 'k_cl_levels' in k, M * 4, 'the dense cell-start table (64 MB) is refilled and its level tables rebuilt per frame for M x 4 B of codes'),
    # the attention is nearer its HBM roof than its MFMA roof (VERDICT r5 task 4): q, k, v rows read once and the output rows written once per
    # (crop, head) item -- 4 x 197 x 64 x 2 B -- in 12 layers; an unfused attention cannot move less
    'attention': stage(lambda k: 'k_attention_f16' in k, crops * 12 * 12 * 197 * 64 * 2 * 4, 'q, k, v rows read + output rows written once per (crop, head) and layer: crops x 12 heads x 12 layers x 4 x 197 x 64 x 2 B',
                       {'bound': 'hbm', 'floor_us_per_launch_at_6_3_tb_per_s_copy_rate': round(crops * 12 * 197 * 64 * 2 * 4 / 6.3e12 * 1e6, 1),
                        'note': 'per item the kernel is a chain of phases on seven waves (staging, load issue, S^T MFMAs, softmax on the vector ALU, P V, output) with ONE 97 KB workgroup per CU: '
                                'nothing overlaps the phases of an item except the next item\'s loads in flight.  Tried and measured (LAB_NOTES): K / V by LDS-DMA into a double buffer '
                                '(round 4: 111.5 vs 114 us, item time unchanged), two 4-wave workgroups per CU (round 3: +6 %), SIMD partners staggered (round 5: -3.5 %, kept)'}),
    'ground': stage(lambda k: k.startswith('k_pw_'), N * 16 + N * 1, 'N x 16 B of points read + N x 1 B mask written; one workgroup per patch, bound by its densest patch\'s dependent chain'),
}
json.dump(stages, open(os.path.join(dst, 'stage_roofline.json'), 'w'), indent=1)
json.dump(out, open(os.path.join(dst, f'{tag}_pmc_summary.json'), 'w'), indent=1)
json.dump(dict(out[DOM], kernel=DOM, tag=tag, by_kind=out.get(DOM + '_by_kind', {}),
               mfma_utilisation=out.get(DOM + '_sq', {}).get('mfma_utilisation')),
          open(os.path.join(dst, 'gemm_traffic.json'), 'w'), indent=1)
print(json.dumps({DOM: out[DOM], 'k_gemm_f16': out['k_gemm_f16']}, indent=1))
