#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3i
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for wl in spce co2_gcmc; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_R2048.json; done
python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_R4096.json
python tools/bench_kernels.py --replicas 1024 --reps 5 --workload framework_water > $out/k_framework_water_R1024.json
python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --workload co2_isotherm --no-cpu-baseline --steps 300 > $out/bench_co2_isotherm.json 2> $out/bench_co2_isotherm.err
python bench.py --no-cpu-baseline --steps 500 > $out/bench_spce.json 2> $out/bench_spce.err
echo done
