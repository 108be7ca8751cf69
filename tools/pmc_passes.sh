#!/bin/bash
# rocprofv3 PMC passes of the hot-path kernels (run on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh <outdir> [tag]
# Separate passes, --kernel-trace only (no other trace domain), as the MI355X guide prescribes: SQ counters,
# FETCH_SIZE, WRITE_SIZE; then tools/pmc_summary.py folds the CSVs into <outdir>/pmc_kernels_<tag>.txt and
# <outdir>/pmc_<tag>.json (what bench.py quotes as roofline.traffic / roofline.valu).
set -e
out=${1:-gpurun_out/pmc}
tag=${2:-r02}
repl=${3:-2048}          # candidates per launch, as in the default bench (8192 chains on 4 lanes)
export MGPU_PAIR_NSPLIT=${4:-2}   # the engine constant the bench's 8192-replica engine uses
mkdir -p "$out"
export TMPDIR=/tmp
run() {  # name, counters
    rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$out/$1" -o "$1" -- python3 tools/bench_kernels.py --reps 3 --replicas $repl > "$out/$1.log" 2>&1
}
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 tools/pmc_summary.py "$out" "$tag" "$repl"
