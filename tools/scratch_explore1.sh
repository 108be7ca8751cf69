mkdir -p gpurun_out/r5h
wl=spce_triclinic
for nf in 0 1; do
if [ $nf = 1 ]; then export MGPU_TRI_NO_FUSE=1; else unset MGPU_TRI_NO_FUSE; fi
timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 3 --sustained-steps 0 --no-cpu-baseline > gpurun_out/r5h/bench_${wl}_nf$nf.txt 2> gpurun_out/r5h/bench_${wl}_nf$nf.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r5h/bench_${wl}_nf$nf.txt') if l.startswith('{')][-1])
r=d['roofline']
print('$wl no_fuse=$nf', 'value', d['value'], 'ms/step', d['ms_per_step'], 'ns/eval', d.get('ns_per_dE_eval'), 'isolated us', r['isolated']['avg_launch_us'], 'frac', r['isolated']['frac'])
PY
done
