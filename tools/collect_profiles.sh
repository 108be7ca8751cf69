#!/bin/bash
# Copy what tools/measure_round.sh left under gpurun_out/r5final into profiles/r05 under the names profiles/r05/README.md uses.
set -e
cd "$(dirname "$0")/.."
src=${1:-gpurun_out/r5final}
dst=${2:-profiles/r05}
mkdir -p $dst
cp $src/lib_sha256.txt $dst/lib_sha256.txt
for wl in spce co2_gcmc framework_water; do
  cp $src/pmc/pmc_${wl}_r05.json $dst/pmc_${wl}.json
  cp $src/pmc/pmc_kernels_${wl}_r05.txt $dst/pmc_kernels_${wl}.txt
  cp $src/prof_$wl/p_kernel_stats.csv $dst/bench_${wl}_kernel_stats.csv
  cp $src/bench_${wl}_under_rocprof.json $dst/bench_${wl}_under_rocprof.json
  cp $src/bench_${wl}.json $dst/bench_${wl}.json
done
cp $src/bench_co2_isotherm.json $dst/bench_co2_isotherm.json
cp $src/bench_co2_isotherm_exchange_torch.json $dst/bench_co2_isotherm_exchange_torch.json
for wl in spce_triclinic adsorbate24; do cp $src/bench_$wl.json $dst/bench_$wl.json; done
cp $src/farm_window_speed.txt $dst/farm_window_speed.txt
cp $src/recip_many_sites.txt $dst/recip_many_sites.txt
cp $src/farm_stages.md $dst/farm_stages.md 2>/dev/null || true
(head -3 $src/window_farm_stress_full.txt; echo "..."; tail -4 $src/window_farm_stress_full.txt) > $dst/window_farm_stress.txt 2>/dev/null || true
(head -4 $src/recip_forms_stress_full.txt; echo "..."; grep "kmax (1[6-9]\|kmax (2" $src/recip_forms_stress_full.txt | head -4; tail -1 $src/recip_forms_stress_full.txt) > $dst/recip_forms_stress.txt 2>/dev/null || true
cp $src/recip_many_sites_vector_form.txt $dst/recip_many_sites_vector_form.txt 2>/dev/null || true
cp $src/farm_window_speed_co2.txt $dst/farm_window_speed_co2.txt 2>/dev/null || true
cp $src/probe_mfma_f64.txt $dst/probe_mfma_f64.txt 2>/dev/null || true
cp $src/host_team_matrix.txt $dst/host_team_matrix.txt 2>/dev/null || true
cp $src/bench_driver_format.json $dst/bench_driver_format_steps20_warmup5.json
cp $src/bench_driver_format_wall.txt $dst/bench_driver_format_wall.txt 2>/dev/null || true
cp $src/bench_driver_format_with_pmc.json $dst/bench_driver_format_steps20_warmup5_with_pmc.json 2>/dev/null || true
cp $src/chain_speed.txt $dst/chain_speed.txt
cp $src/chain_speed_batched_calls.txt $dst/chain_speed_batched_calls.txt
cp $src/chain_stages.md $dst/chain_stages.md
cp $src/chain_latency_traces.md $dst/chain_latency_traces.md
cat $src/k_*.json | sed 's/}{/}\n{/g' > $dst/kernel_only.jsonl
grep -h "passed\|failed" $src/pytest.log | tail -1 > $dst/gpu_tests.txt
echo collected into $dst
