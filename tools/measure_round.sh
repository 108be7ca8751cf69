#!/bin/bash
# One measurement session on an MI355X box (run as: gpurun -- bash tools/measure_round.sh): GPU tests, PMC passes (tied to the
# library hash), rocprofv3 kernel stats of the three workloads, the bench lines and the kernel-only timings that
# profiles/rNN/ keeps.  Output under gpurun_out/r3final (traces are deleted at the end: 64 MiB merge limit).
set -e -o pipefail
out=gpurun_out/r3final
mkdir -p $out
export TMPDIR=/tmp
sha256sum maniac_mc_amd/libmaniac_hip.so > $out/lib_sha256.txt
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bash tools/pmc_passes.sh $out/pmc r03 4096 1 spce > $out/pmc_spce.log 2>&1   # the default bench: 16384 chains on 4 lanes, engine nsplit 1
bash tools/pmc_passes.sh $out/pmc r03 4096 2 co2_gcmc > $out/pmc_co2.log 2>&1
bash tools/pmc_passes.sh $out/pmc r03 2048 4 framework_water > $out/pmc_fw.log 2>&1
mkdir -p profiles/r03
for wl in spce co2_gcmc framework_water; do cp $out/pmc/pmc_${wl}_r03.json profiles/r03/pmc_${wl}.json; done
for wl in spce co2_gcmc framework_water; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$wl -o p -- python3 /root/repo/bench.py --workload $wl --no-cpu-baseline --steps 200 --settle-s 0 --sustained-steps 0 > /root/repo/$out/bench_${wl}_under_rocprof.json 2> /root/repo/$out/bench_${wl}_under_rocprof.err)
  rm -f $out/prof_$wl/*kernel_trace.csv
done
python bench.py --steps 20 --warmup 5 > $out/bench_driver_format.json 2> $out/bench_driver_format.err
python bench.py > $out/bench_spce.json 2> $out/bench_spce.err
python bench.py --workload co2_gcmc > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
python bench.py --workload framework_water > $out/bench_framework_water.json 2> $out/bench_framework_water.err
python bench.py --workload co2_isotherm > $out/bench_co2_isotherm.json 2> $out/bench_co2_isotherm.err
python bench.py --host-threads 2 --no-cpu-baseline > $out/bench_spce_T2.json 2> $out/bench_spce_T2.err
# opt-in: the acceptance rule and the commit on the device (the k sweep decides)
for wl in spce co2_gcmc framework_water; do
  python bench.py --workload $wl --device-accept 1 --no-cpu-baseline > $out/bench_${wl}_device_accept.json 2> $out/bench_${wl}_device_accept.err
done
PMC_EXTRA=--decide bash tools/pmc_passes.sh $out/pmc_decide r03 4096 2 co2_gcmc > $out/pmc_co2_decide.log 2>&1
python tools/chain_speed.py --blocks 2 --steps 1500 > $out/chain_speed.txt 2>&1
for wl in spce co2_gcmc framework_water; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_R2048.json; done
python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_R4096.json
MGPU_PAIR_NSPLIT=1 python tools/bench_kernels.py --replicas 4096 --reps 5 --workload spce > $out/k_spce_R4096_nsplit1.json   # the default bench's launch shape
# the same launch groups with the acceptance on the device (the k sweep decides and commits: the farm's default)
for wl in spce framework_water; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl --decide > $out/k_${wl}_R2048_decide.json; done
python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc --decide > $out/k_co2_gcmc_R4096_decide.json
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -size +2M -delete
du -a $out | sort -n | tail -5
du -sh $out
echo done
