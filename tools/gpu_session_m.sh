#!/bin/bash
set -e -o pipefail
out=gpurun_out/r3m
mkdir -p $out
python -m pytest tests/test_gpu_farm.py tests/test_bench_cli.py -m gpu -x -q -k "long_run or isotherm" > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python bench.py --workload co2_gcmc > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --workload co2_isotherm > $out/bench_co2_isotherm.json 2> $out/bench_co2_isotherm.err
echo done
