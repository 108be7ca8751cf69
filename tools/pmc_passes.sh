#!/bin/bash
# rocprofv3 PMC passes of the hot-path kernels (run on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh <outdir> <tag> [candidates per launch] [nsplit] [workload]
# Separate passes, --kernel-trace only (no other trace domain), as the MI355X guide prescribes: SQ counters,
# FETCH_SIZE, WRITE_SIZE; then tools/pmc_summary.py folds the CSVs into <outdir>/pmc_kernels_<workload>_<tag>.txt and
# <outdir>/pmc_<workload>_<tag>.json (what bench.py quotes as roofline.traffic / roofline.valu_issue, together with the
# sha256 of the libmaniac_hip.so that was profiled).
set -e
out=${1:-gpurun_out/pmc}
tag=${2:-r03}
repl=${3:-2048}          # candidates per launch, as in the default bench (8192 chains on 4 lanes)
nsplit=${4:-}
wl=${5:-spce}
if [ -n "$nsplit" ]; then export MGPU_PAIR_NSPLIT=$nsplit; fi   # the engine constant the bench's engine uses at its replica count
mkdir -p "$out"
export PMC_EXTRA=${PMC_EXTRA-}    # PMC_EXTRA=--decide profiles the opt-in path (the k sweep decides and commits) instead of k sweep + commit launch
export TMPDIR=/tmp
run() {  # name, counters
    rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$out/${wl}_$1" -o "$1" -- python3 tools/bench_kernels.py --workload $wl --reps 3 --replicas $repl $PMC_EXTRA > "$out/${wl}_$1.log" 2>&1
}
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 tools/pmc_summary.py "$out" "$tag" "$repl" "$wl"
