#!/bin/bash
# round-3 GPU session E: why is the framework pair sweep slow? (kinds, PMC), prefetch variants, chain speed, benches
set -e -o pipefail
out=gpurun_out/r3e
mkdir -p $out
bk="python tools/bench_kernels.py --replicas 1024 --reps 5 --workload framework_water"
for k in moves insertions deletions; do $bk --kinds $k > $out/k_fw_flat_$k.json; done
MGPU_PAIR_FLAT=0 MGPU_NO_FROZEN=1 $bk --kinds moves > $out/k_fw_planes_nofrozen_moves.json
MGPU_PAIR_FLAT=0 MGPU_NO_FROZEN=1 $bk --kinds deletions > $out/k_fw_planes_nofrozen_deletions.json
for v in pf1 pf2 pf3; do
  for wl in spce co2_gcmc; do
    MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_$v.json
  done
done
for wl in spce co2_gcmc; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_default.json; done
bash tools/pmc_passes.sh $out/pmc r03a 1024 "" framework_water > $out/pmc_fw.log 2>&1
bash tools/pmc_passes.sh $out/pmc r03a 2048 "" co2_gcmc > $out/pmc_co2.log 2>&1
python tools/chain_speed.py --blocks 2 --steps 1500 > $out/chain_speed.txt 2>&1
python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --no-cpu-baseline --steps 500 > $out/bench_spce.json 2> $out/bench_spce.err
for t in 1 2 3; do python bench.py --no-cpu-baseline --steps 500 --host-threads $t > $out/bench_spce_T$t.json 2> $out/bench_spce_T$t.err; done
echo done
