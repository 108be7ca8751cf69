#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3h
mkdir -p $out
for v in st16 st40 st100; do
  for wl in spce co2_gcmc; do
    MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_$v.json
  done
  MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_R4096_$v.json
done
for wl in spce co2_gcmc; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_default.json; done
python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_R4096_default.json
MFARM_LANE_THREADS=2 python bench.py --workload co2_gcmc --lanes 4 --host-threads 8 --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_D2_L4_T8.json 2> $out/bench_co2_gcmc_D2_L4_T8.err
MGPU_DEFER_COMMIT=1 MFARM_LANE_THREADS=2 python bench.py --workload co2_gcmc --lanes 4 --host-threads 8 --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_D2_L4_T8_defer.json 2> $out/bench_co2_gcmc_D2_L4_T8_defer.err
MGPU_DEFER_COMMIT=1 python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_defer.json 2> $out/bench_co2_gcmc_defer.err
MFARM_LANE_THREADS=2 python bench.py --workload co2_gcmc --replicas 32768 --lanes 4 --host-threads 8 --no-cpu-baseline --steps 200 > $out/bench_co2_gcmc_R32768_D2_L4_T8.json 2> $out/bench_co2_gcmc_R32768_D2_L4_T8.err
MFARM_LANE_THREADS=2 python bench.py --workload co2_isotherm --lanes 4 --host-threads 8 --no-cpu-baseline --steps 300 > $out/bench_co2_isotherm_D2_L4_T8.json 2> $out/bench_co2_isotherm_D2_L4_T8.err
MFARM_LANE_THREADS=2 python bench.py --workload framework_water --replicas 16384 --lanes 4 --host-threads 8 --no-cpu-baseline --steps 200 > $out/bench_framework_water_R16384_D2_L4_T8.json 2> $out/bench_framework_water_R16384_D2_L4_T8.err
echo done
