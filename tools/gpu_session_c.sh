#!/bin/bash
# round-3 GPU session C: parity of the new kernels, then A/B of kernel variants and host-thread / lane sweeps
set -e -o pipefail
out=gpurun_out/r3c
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bk="python tools/bench_kernels.py --replicas 1024 --reps 5"
# framework: frozen (type-sorted, per-lane charge) path against the site-major one; fused 4-site wide kernels; nsplit
$bk --workload framework_water > $out/k_fw_default.json
MGPU_NO_FROZEN=1 $bk --workload framework_water > $out/k_fw_nofrozen.json
MGPU_PAIR_FUSE_MAX=4 $bk --workload framework_water > $out/k_fw_fuse4.json
for ns in 1 2 8; do MGPU_PAIR_NSPLIT=$ns $bk --workload framework_water > $out/k_fw_nsplit$ns.json; done
for ns in 1 2 8; do MGPU_PAIR_FUSE_MAX=4 MGPU_PAIR_NSPLIT=$ns $bk --workload framework_water > $out/k_fw_fuse4_nsplit$ns.json; done
# k sweep / commit variants
for v in sincos early1 early2 early3; do
  for wl in spce co2_gcmc; do
    MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_$v.json
  done
done
for wl in spce co2_gcmc; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_default.json; done
# host sweeps
for cfg in "8192 2 4" "16384 2 4" "16384 4 4" "8192 4 2" "8192 4 6" "16384 2 6"; do
  set -- $cfg
  python bench.py --workload co2_gcmc --replicas $1 --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_R$1_L$2_T$3.json 2> $out/bench_co2_gcmc_R$1_L$2_T$3.err
done
for cfg in "4096 2 4" "8192 2 4" "8192 4 4" "8192 2 6"; do
  set -- $cfg
  python bench.py --workload framework_water --replicas $1 --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_framework_water_R$1_L$2_T$3.json 2> $out/bench_framework_water_R$1_L$2_T$3.err
done
for t in 4 6; do
  python bench.py --host-threads $t --no-cpu-baseline --steps 500 > $out/bench_spce_T$t.json 2> $out/bench_spce_T$t.err
done
python bench.py --no-cpu-baseline --steps 500 > $out/bench_spce_T8.json 2> $out/bench_spce_T8.err
echo done
