#!/bin/bash
# One measurement session on an MI355X box (run as: gpurun --timeout 1200 -- bash tools/measure_round.sh [part]): what
# profiles/r05/ keeps.  Output under gpurun_out/r5final; tools/collect_profiles.sh copies it into profiles/r05.
# part 1: GPU tests, single-chain speed / stage stamps / rocprofv3 trace, kernel-only timings, the bench lines
# part 2: PMC passes (tied to the library hash) and rocprofv3 kernel stats of the three workloads
set -e -o pipefail
part=${1:-1}
out=gpurun_out/r5final
mkdir -p $out
export TMPDIR=/tmp
sha256sum maniac_mc_amd/libmaniac_hip.so > $out/lib_sha256.txt
if [ "$part" = 1 ]; then
  python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -80 $out/pytest.log; exit 1; }
  tail -3 $out/pytest.log
  python tools/chain_speed.py --cases co2_gcmc,framework_water_gcmc,spce_10125_nvt,spce_10125_triclinic_nvt > $out/chain_speed.txt 2>&1
  python tools/chain_speed.py --chain-windows 0 --ks 1,8 --cases co2_gcmc,framework_water_gcmc,spce_10125_nvt,spce_10125_triclinic_nvt > $out/chain_speed_batched_calls.txt 2>&1
  python tools/chain_stages.py --ks 1,4 > $out/chain_stages.md 2>&1
  bash tools/trace_chain.sh $out/trace_window 1 4 > $out/trace.log 2>&1
  # the batched-call path (round 3's single-chain loop) under the same tracer, for the before / after table
  EXTRA="--chain-windows 0" bash tools/trace_chain.sh $out/trace_batched 1 8 >> $out/trace.log 2>&1
  python tools/chain_latency.py $out/trace_batched/k1 $out/trace_batched/k8 $out/trace_window/k1 $out/trace_window/k4 > $out/chain_latency_traces.md
  for wl in spce co2_gcmc framework_water; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl > $out/k_${wl}_R2048.json; done
  python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc > $out/k_co2_gcmc_R4096.json
  MGPU_PAIR_NSPLIT=1 python tools/bench_kernels.py --replicas 4096 --reps 5 --workload spce > $out/k_spce_R4096_nsplit1.json   # the default bench's launch shape
  for wl in spce framework_water; do python tools/bench_kernels.py --replicas 2048 --reps 5 --workload $wl --decide > $out/k_${wl}_R2048_decide.json; done
  python tools/bench_kernels.py --replicas 4096 --reps 5 --workload co2_gcmc --decide > $out/k_co2_gcmc_R4096_decide.json
  t0=$(date +%s); python bench.py --steps 20 --warmup 5 > $out/bench_driver_format.json 2> $out/bench_driver_format.err; echo "bench.py --steps 20 --warmup 5: $(( $(date +%s) - t0 )) s wall" > $out/bench_driver_format_wall.txt
  python bench.py --configs 0 --replicas-sweep "" > $out/bench_spce.json 2> $out/bench_spce.err
  python bench.py --workload co2_gcmc --replicas-sweep 1,8,64,512,1024,4096 > $out/bench_co2_gcmc.json 2> $out/bench_co2_gcmc.err
  python bench.py --workload framework_water > $out/bench_framework_water.json 2> $out/bench_framework_water.err
  python bench.py --workload co2_isotherm > $out/bench_co2_isotherm.json 2> $out/bench_co2_isotherm.err
  python bench.py --workload co2_isotherm --exchange torch --no-cpu-baseline > $out/bench_co2_isotherm_exchange_torch.json 2> $out/bench_co2_isotherm_exchange_torch.err
  python bench.py --workload spce_triclinic > $out/bench_spce_triclinic.json 2> $out/bench_spce_triclinic.err
  python bench.py --workload adsorbate24 > $out/bench_adsorbate24.json 2> $out/bench_adsorbate24.err
  # round 5: the few-chain regime (one launch per lane step) against the batched path, and the many-site reciprocal kernels
  python tools/farm_window_speed.py --replicas 1,8,64,512,1024 --modes batched,w1,w2,w3 --lanes 1,2 --seconds 0.5 --json $out/farm_window_speed.json > $out/farm_window_speed.txt 2>&1
  python tools/farm_window_speed.py --workload co2_gcmc --replicas 1,8,64,512,1024,4096 --modes batched,w1,w3 --lanes 1,2 --drivers 1,2 --threads 6 --seconds 0.4 > $out/farm_window_speed_co2.txt 2>&1
  python tools/farm_stages.py --chains 1,8,64,512 --windows 300 > $out/farm_stages.md 2>&1 || true
  python tools/recip_many_sites.py > $out/recip_many_sites.txt 2>&1
  python tools/window_farm_stress.py --cases 400 --seed 11 > $out/window_farm_stress_full.txt 2>&1 || true
  python tools/recip_forms_stress.py --cases 60 --seed 9 > $out/recip_forms_stress_full.txt 2>&1 || true
  MGPU_RECIP_NO_MFMA=1 python tools/recip_many_sites.py --sites 24 > $out/recip_many_sites_vector_form.txt 2>&1
  hipcc -O3 --offload-arch=gfx950 -o /tmp/probe_mfma_f64 tools/probe_mfma_f64.hip > /dev/null 2>&1 && /tmp/probe_mfma_f64 > $out/probe_mfma_f64.txt 2>&1 || true
  python tools/host_team_matrix.py > $out/host_team_matrix.txt 2>&1 || true
else
  bash tools/pmc_passes.sh $out/pmc r05 4096 1 spce > $out/pmc_spce.log 2>&1   # the default bench: 16384 chains on 4 lanes, engine nsplit 1
  bash tools/pmc_passes.sh $out/pmc r05 4096 2 co2_gcmc > $out/pmc_co2.log 2>&1
  bash tools/pmc_passes.sh $out/pmc r05 2048 4 framework_water > $out/pmc_fw.log 2>&1
  mkdir -p profiles/r05
  for wl in spce co2_gcmc framework_water; do cp $out/pmc/pmc_${wl}_r05.json profiles/r05/pmc_${wl}.json; done
  for wl in spce co2_gcmc framework_water; do
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$wl -o p -- python3 /root/repo/bench.py --workload $wl --no-cpu-baseline --steps 200 --settle-s 0 --sustained-steps 0 --configs 0 --replicas-sweep "" > /root/repo/$out/bench_${wl}_under_rocprof.json 2> /root/repo/$out/bench_${wl}_under_rocprof.err)
    rm -f $out/prof_$wl/*kernel_trace.csv
  done
  # the headline line again, now that profiles/r05/pmc_*.json are of this build (roofline.traffic / valu_issue filled in)
  python bench.py --steps 20 --warmup 5 > $out/bench_driver_format_with_pmc.json 2> $out/bench_driver_format_with_pmc.err
fi
find $out -name '*kernel_trace.csv' -path '*pmc*' -delete
find $out -name '*counter_collection.csv' -size +2M -delete
du -sh $out
echo done part $part
