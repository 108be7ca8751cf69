#!/bin/bash
# round-3 GPU session D: parity with the flat pair kernel + speculative chain windows, kernel A/B, chain speed
set -e -o pipefail
out=gpurun_out/r3d
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bk="python tools/bench_kernels.py --replicas 1024 --reps 5"
for wl in framework_water co2_gcmc; do
  $bk --workload $wl > $out/k_${wl}_flat.json
  MGPU_PAIR_FLAT=0 $bk --workload $wl > $out/k_${wl}_planes.json
  MGPU_PAIR_FLAT=0 MGPU_NO_FROZEN=1 $bk --workload $wl > $out/k_${wl}_planes_nofrozen.json
  for ns in 1 2 8; do MGPU_PAIR_NSPLIT=$ns $bk --workload $wl > $out/k_${wl}_flat_nsplit$ns.json; done
done
MGPU_PAIR_FUSE_MAX=4 $bk --workload framework_water > $out/k_framework_water_flat_fuse4.json
MGPU_PAIR_FUSE_MAX=4 MGPU_PAIR_NSPLIT=2 $bk --workload framework_water > $out/k_framework_water_flat_fuse4_nsplit2.json
MGPU_PAIR_FLAT=1 python tools/bench_kernels.py --replicas 2048 --reps 5 --workload spce > $out/k_spce_flat.json
for v in pf1 pf2 pf3; do
  for wl in spce co2_gcmc; do
    MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_$v.json
  done
done
for wl in spce co2_gcmc; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_default.json; done
python tools/chain_speed.py --blocks 2 --steps 1500 > $out/chain_speed.txt 2>&1
python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --no-cpu-baseline --steps 500 > $out/bench_spce.json 2> $out/bench_spce.err
echo done
