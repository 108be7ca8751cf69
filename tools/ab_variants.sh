#!/bin/bash
# Same-box A/B of library variants (maniac_mc_amd/variants/libmaniac_hip_<name>.so, each an earlier build kept aside):
#   bash tools/ab_variants.sh <out.txt> <name> [<name> ...]     -- SPC/E 4096 fused moves at nsplit 1, two rounds
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for round in 1 2; do
  for v in "$@"; do
    MANIAC_HIP_LIB=$PWD/maniac_mc_amd/variants/libmaniac_hip_$v.so MGPU_PAIR_NSPLIT=1 \
      python tools/bench_kernels.py --replicas 4096 --reps 5 --workload ${WL:-spce} 2>&1 | tail -1 | sed "s/^/$v /" >> "$out" || exit 1
  done
done
cat "$out"
