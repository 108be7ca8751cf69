#!/bin/bash
# round-3 GPU session B: chain-count / lane scaling of the grand-canonical workloads (host-bound at 2048 x 2?)
set -e -o pipefail
out=gpurun_out/r3b
mkdir -p $out
for wl in co2_gcmc framework_water; do
  for cfg in "4096 2" "4096 4" "8192 4" "16384 4"; do
    set -- $cfg
    python bench.py --workload $wl --replicas $1 --lanes $2 --no-cpu-baseline --steps 400 > $out/bench_${wl}_R$1_L$2.json 2> $out/bench_${wl}_R$1_L$2.err
  done
done
for ht in 4 12 16; do
  python bench.py --workload co2_gcmc --replicas 8192 --lanes 4 --host-threads $ht --no-cpu-baseline --steps 400 > $out/bench_co2_gcmc_R8192_L4_T$ht.json 2> $out/bench_co2_gcmc_R8192_L4_T$ht.err
done
echo done
