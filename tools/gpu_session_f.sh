#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3f
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gcmc.py tests/test_gpu_farm.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bk="python tools/bench_kernels.py --replicas 1024 --reps 5"
for wl in framework_water co2_gcmc; do
  $bk --workload $wl > $out/k_${wl}_flat.json
  MGPU_PAIR_FLAT=0 MGPU_NO_FROZEN=1 $bk --workload $wl > $out/k_${wl}_planes_nofrozen.json
  for ns in 1 2 8; do MGPU_PAIR_NSPLIT=$ns $bk --workload $wl > $out/k_${wl}_flat_nsplit$ns.json; done
done
MGPU_PAIR_FUSE_MAX=4 $bk --workload framework_water > $out/k_framework_water_flat_fuse4.json
MGPU_PAIR_FUSE_MAX=4 MGPU_PAIR_NSPLIT=2 $bk --workload framework_water > $out/k_framework_water_flat_fuse4_nsplit2.json
MGPU_PAIR_FLAT=1 python tools/bench_kernels.py --replicas 2048 --reps 5 --workload spce > $out/k_spce_flat.json
python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
echo done
MFARM_LANE_THREADS=1 python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_lane_threads.json 2> $out/bench_co2_gcmc_lane_threads.err
MFARM_LANE_THREADS=1 python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 --lanes 4 --host-threads 8 > $out/bench_co2_gcmc_lane_threads_L4_T8.json 2> $out/bench_co2_gcmc_lane_threads_L4_T8.err
MFARM_LANE_THREADS=1 python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water_lane_threads.json 2> $out/bench_framework_water_lane_threads.err
echo done2
