mkdir -p gpurun_out/r5f
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5f/bench_default.txt 2> gpurun_out/r5f/bench_default.err
tail -c 1500 gpurun_out/r5f/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5f/bench_default.txt') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
for r in d.get('replicas_sweep', []): print(r)
print(d.get('single_chain'))
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('configs',{}).items()})
print(d.get('exchange'))
PY
