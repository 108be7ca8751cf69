#!/bin/bash
# round-3 GPU session A: tests, baseline numbers of the grand-canonical workloads before any kernel work
set -e -o pipefail
out=gpurun_out/r3a
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python bench.py --steps 20 --warmup 5 > $out/bench_driver_format.json 2> $out/bench_driver_format.err
python bench.py --workload co2_gcmc > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water > $out/bench_framework_water.json 2> $out/bench_framework_water.err
for wl in spce co2_gcmc framework_water; do
  python tools/bench_kernels.py --workload $wl --replicas 1024 --reps 5 > $out/kernels_$wl.json 2> $out/kernels_$wl.err
done
for wl in co2_gcmc framework_water; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$wl -o p -- python3 /root/repo/bench.py --workload $wl --no-cpu-baseline --steps 300 > /root/repo/$out/bench_${wl}_under_rocprof.json 2> /root/repo/$out/bench_${wl}_under_rocprof.err)
done
echo done
