set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04g/bench_k20.json 2> gpurun_out/r04g/bench_k20.err
tail -c 3000 gpurun_out/r04g/bench_k20.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04g/bench_k20.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'gemm ms/frame', d['roofline']['gemm_ms_per_frame'])
for k in ('box_modes','hipgraph_loop','resident_input','angle_modes','resid16','f32_parity_mode','views6','dense200k','default_config_mode','cli_mode','cpu_baseline'):
    v=d.get(k,{})
    print(k, {kk:(vv if not isinstance(vv,str) or len(vv)<60 else vv[:60]) for kk,vv in v.items() if kk not in ('note','workload','sample')})
PY
bash tools/collect_profiles.sh r04a > gpurun_out/r04g/collect.log 2>&1
tail -30 gpurun_out/r04g/collect.log
