#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3k
mkdir -p $out
for v in e1 e3 m5 m5e3; do
  MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload spce > $out/k_spce_$v.json
  MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_$v.json
  MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload framework_water > $out/k_framework_water_$v.json
done
python tools/bench_kernels.py --replicas 2048 --reps 5 --workload spce > $out/k_spce_default.json
python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_default.json
python tools/bench_kernels.py --replicas 2048 --reps 5 --workload framework_water > $out/k_framework_water_default.json
echo done
