#!/bin/bash
# Copy what tools/measure_round.sh left under gpurun_out/r3final into profiles/r03 under the names profiles/r03/README.md uses.
set -e
cd "$(dirname "$0")/.."
src=${1:-gpurun_out/r3final}
dst=${2:-profiles/r03}
mkdir -p $dst
cp $src/lib_sha256.txt $dst/lib_sha256.txt
for wl in spce co2_gcmc framework_water; do
  cp $src/pmc/pmc_${wl}_r03.json $dst/pmc_${wl}.json
  cp $src/pmc/pmc_kernels_${wl}_r03.txt $dst/pmc_kernels_${wl}.txt
  cp $src/prof_$wl/p_kernel_stats.csv $dst/bench_${wl}_kernel_stats.csv
  cp $src/bench_${wl}_under_rocprof.json $dst/bench_${wl}_under_rocprof.json
  cp $src/bench_${wl}.json $dst/bench_${wl}.json
done
for wl in spce co2_gcmc framework_water; do cp $src/bench_${wl}_device_accept.json $dst/bench_${wl}_device_accept.json; done
cp $src/pmc_decide/pmc_kernels_co2_gcmc_r03.txt $dst/pmc_kernels_co2_gcmc_device_accept.txt
cp $src/bench_co2_isotherm.json $dst/bench_co2_isotherm.json
cp $src/bench_driver_format.json $dst/bench_driver_format_steps20_warmup5.json
cp $src/bench_spce_T2.json $dst/bench_spce_host_threads2.json
cp $src/chain_speed.txt $dst/chain_speed.txt
cat $src/k_*.json | sed 's/}{/}\n{/g' > $dst/kernel_only.jsonl
grep -h "passed\|failed" $src/pytest.log | tail -1 > $dst/gpu_tests.txt
echo collected into $dst
