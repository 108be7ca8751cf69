#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3g
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_gcmc.py tests/test_gpu_farm.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bk="python tools/bench_kernels.py --replicas 1024 --reps 5"
for wl in framework_water co2_gcmc; do $bk --workload $wl > $out/k_${wl}_default.json; done
for wl in co2_gcmc framework_water; do
  for cfg in "2 4 6" "2 4 8" "2 4 12" "3 3 6"; do
    set -- $cfg
    R=16384; if [ $wl = framework_water ]; then R=8192; fi
    MFARM_LANE_THREADS=$1 python bench.py --workload $wl --replicas $R --lanes $2 --host-threads $3 --no-cpu-baseline --steps 300 > $out/bench_${wl}_D$1_L$2_T$3.json 2> $out/bench_${wl}_D$1_L$2_T$3.err
  done
done
for cfg in "2 4 6" "2 4 8"; do
  set -- $cfg
  MFARM_LANE_THREADS=$1 python bench.py --lanes $2 --host-threads $3 --no-cpu-baseline --steps 500 > $out/bench_spce_D$1_L$2_T$3.json 2> $out/bench_spce_D$1_L$2_T$3.err
done
python bench.py --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water.json 2> $out/bench_framework_water.err
echo done
