#!/bin/bash
# rocprofv3 HIP-API + kernel + memory-copy trace of the single-chain drop-in (tools/chain_speed.py) on the 10 125-atom SPC/E box:
#   [CASES=...] [EXTRA='--chain-windows 0'] bash tools/trace_chain.sh <outdir> [K ...]      (run on the GPU box from the repo root)
# tools/chain_latency.py folds the CSVs into the per-stage table of profiles/rNN/chain_latency.md.
set -e -o pipefail
out=${1:-gpurun_out/chain_trace}
shift || true
ks=${@:-1 8}
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
for k in $ks; do
  (cd /tmp && rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/k$k -o t -- \
      python3 $root/tools/chain_speed.py --blocks 1 --steps 600 --ks $k --cases ${CASES:-spce_10125_nvt} ${EXTRA:-} > $root/$out/k$k.log 2>&1)
done
find $out -name '*.csv' | xargs ls -la
