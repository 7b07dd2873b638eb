set -x
cd $GRAFT_REPO_ROOT
TAG=${1:-r04b}
mkdir -p gpurun_out/$TAG
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_k20.json 2> gpurun_out/$TAG/bench_k20.err
tail -c 1500 gpurun_out/$TAG/bench_k20.err
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 1500 gpurun_out/$TAG/bench.err
python - <<PY
import json
for f in ('bench_k20.json','bench.json'):
    try:
        d=json.loads([l for l in open('gpurun_out/$TAG/'+f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    print(f, 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gemm ms/frame', d['roofline']['gemm_ms_per_frame'], 'tower2', d['roofline'].get('tower_two_in_flight'))
    for k in ('multi_gpu_model','box_modes','hipgraph_loop','resident_input','angle_modes','resid16','f32_parity_mode','views6','dense200k','default_config_mode','cli_mode','cpu_baseline'):
        v=d.get(k,{})
        print('  ',k, {kk:(vv if not isinstance(vv,str) or len(vv)<50 else vv[:50]) for kk,vv in v.items() if kk not in ('note','workload','sample','config0_20k','roofline_stages')})
PY
bash tools/collect_profiles.sh $TAG > gpurun_out/$TAG/collect.log 2>&1
tail -12 gpurun_out/$TAG/collect.log
