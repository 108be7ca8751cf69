#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3n
mkdir -p $out
for cfg in "2 4 8" "4 8 8" "4 8 12" "4 8 16" "3 6 12" "2 8 8"; do
  set -- $cfg
  python bench.py --workload co2_gcmc --drivers $1 --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_D$1_L$2_T$3.json 2> $out/bench_co2_gcmc_D$1_L$2_T$3.err
done
for cfg in "2 4 8" "4 8 12" "4 8 16"; do
  set -- $cfg
  python bench.py --workload framework_water --drivers $1 --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_framework_water_D$1_L$2_T$3.json 2> $out/bench_framework_water_D$1_L$2_T$3.err
  python bench.py --workload co2_isotherm --drivers $1 --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_co2_isotherm_D$1_L$2_T$3.json 2> $out/bench_co2_isotherm_D$1_L$2_T$3.err
done
python bench.py --lanes 8 --replicas 16384 --no-cpu-baseline --steps 300 > $out/bench_spce_R16384_L8.json 2> $out/bench_spce_R16384_L8.err
python bench.py --lanes 8 --drivers 2 --host-threads 8 --no-cpu-baseline --steps 500 > $out/bench_spce_L8_D2_T8.json 2> $out/bench_spce_L8_D2_T8.err
python bench.py --no-cpu-baseline --steps 500 > $out/bench_spce_default.json 2> $out/bench_spce_default.err
echo done
