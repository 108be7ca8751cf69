#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3l
mkdir -p $out
python -m pytest tests/test_gpu_gcmc.py tests/test_gpu_farm.py tests/test_gpu_parity.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for db in 0 1; do
  python bench.py --device-build $db --no-cpu-baseline --steps 500 > $out/bench_spce_db$db.json 2> $out/bench_spce_db$db.err
  python bench.py --device-build $db --host-threads 2 --no-cpu-baseline --steps 500 > $out/bench_spce_T2_db$db.json 2> $out/bench_spce_T2_db$db.err
  python bench.py --device-build $db --host-threads 1 --no-cpu-baseline --steps 500 > $out/bench_spce_T1_db$db.json 2> $out/bench_spce_T1_db$db.err
  python bench.py --device-build $db --workload co2_gcmc --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_db$db.json 2> $out/bench_co2_gcmc_db$db.err
  python bench.py --device-build $db --workload framework_water --no-cpu-baseline --steps 300 > $out/bench_framework_water_db$db.json 2> $out/bench_framework_water_db$db.err
  python bench.py --device-build $db --workload co2_gcmc --drivers 1 --lanes 2 --host-threads 6 --no-cpu-baseline --steps 300 > $out/bench_co2_gcmc_D1_db$db.json 2> $out/bench_co2_gcmc_D1_db$db.err
done
echo done
