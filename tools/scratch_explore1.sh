mkdir -p gpurun_out/r5j
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gcmc.py tests/test_gpu_farm.py tests/test_gpu_farm_window.py tests/test_gpu_chain.py -q -m gpu -x > gpurun_out/r5j/tests.log 2>&1; tail -5 gpurun_out/r5j/tests.log
for round in 1 2; do
for pi in 0 1; do
echo "== pair_items $pi" >> gpurun_out/r5j/k2.txt
MGPU_RECIP_PAIR_ITEMS=$pi timeout -k 10 200 python tools/bench_kernels.py --replicas 2048 --reps 5 --workload framework_water 2>&1 | tail -1 >> gpurun_out/r5j/k2.txt
MGPU_RECIP_PAIR_ITEMS=$pi timeout -k 10 200 python tools/bench_kernels.py --replicas 4096 --reps 5 --workload framework_water 2>&1 | tail -1 >> gpurun_out/r5j/k2.txt
done
done
python - <<'PY'
import json
for l in open('gpurun_out/r5j/k2.txt'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print(d['candidates_per_launch'], {k: round(v,2) for k,v in d['avg_us'].items()})
PY
