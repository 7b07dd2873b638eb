"""Post-process gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.{csv,json}.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
FETCH_SIZE under-reports wide coalesced streaming reads by 2x on gfx950 (TCC_EA0_RDREQ x 64 B with 128-B requests
tallied at 64 B); we report both the raw and the corrected (x2) figure, WRITE_SIZE is exact for 16-B/lane stores."""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
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
out = {'tag': tag, 'command': 'bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline-pass --no-sequence-pass --no-extras', 'kernels': {}}
def group(match):
    g = {'launches': 0, 'fetch_kb': 0.0, 'write_kb': 0.0, 'dur_ns': 0.0, 'calls_trace': 0}
    for k in sorted(set(fetch) | set(write)):
        if not match(k):
            continue
        f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        st = stats.get(k)
        g['launches'] += f[0]; g['fetch_kb'] += f[1]; g['write_kb'] += w[1]
        if st:
            g['dur_ns'] += float(st['TotalDurationNs']); g['calls_trace'] += int(st['Calls'])
    n = max(g['launches'], 1)
    return {
        'launches_pmc': g['launches'], 'avg_launch_us_trace': g['dur_ns'] / max(g['calls_trace'], 1) / 1e3,
        'fetch_bytes_per_launch_raw': 1024 * g['fetch_kb'] / n, 'fetch_bytes_per_launch_corrected_x2': 2048 * g['fetch_kb'] / n,
        'write_bytes_per_launch': 1024 * g['write_kb'] / n,
        'hbm_bytes_per_launch': (2048 * g['fetch_kb'] + 1024 * g['write_kb']) / n,
    }
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
    st = stats.get(k)
    out['kernels'][k[:80]] = {'launches': f[0], 'fetch_kb_per_launch_raw': f[1] / max(f[0], 1), 'write_kb_per_launch': w[1] / max(w[0], 1),
                              'avg_ns': float(st['AverageNs']) if st else None}
out['k_gemm_f16_pp64'] = group(lambda k: 'k_gemm_f16_pp64' in k)                       # the dominant kernel (bench.py roofline)
out['k_gemm_f16'] = group(lambda k: 'k_gemm_f16' in k and 'k_gemm_f16_pp' not in k)    # fallback kernel (unused by ViT-B/16)
sq_path = os.path.join(src, 'sq', 'bench_counter_collection.csv')
if os.path.exists(sq_path):
    # SQ / GRBM pass: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (16 per v_mfma_f32_16x16x32_f16),
    # GRBM_GUI_ACTIVE the active cycles of the 8 XCDs; MFMA utilisation = busy / (GUI_ACTIVE / 8 * 1024 SIMDs)
    names = ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE')
    for label, match in (('k_gemm_f16_pp64', lambda k: 'k_gemm_f16_pp64' in k), ('k_attention_f16', lambda k: 'k_attention_f16' in k)):
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
json.dump(out, open(os.path.join(dst, f'{tag}_pmc_summary.json'), 'w'), indent=1)
json.dump(dict(out['k_gemm_f16_pp64'], kernel='k_gemm_f16_pp64', tag=tag), open(os.path.join(dst, 'gemm_traffic.json'), 'w'), indent=1)
print(json.dumps({'k_gemm_f16_pp64': out['k_gemm_f16_pp64'], 'k_gemm_f16': out['k_gemm_f16']}, indent=1))
