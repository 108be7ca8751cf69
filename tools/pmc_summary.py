#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc CSVs written by tools/pmc_passes.sh into a text table and a JSON summary.

    python3 tools/pmc_summary.py <dir> <tag> <candidates per launch> <workload>
Per kernel: mean of every counter over its dispatches (the first dispatch of each kernel is dropped: cold caches).
FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md section HBM; calibrated with
tools/probe_fetch.hip); both are reported in KiB by rocprofv3.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def main():
    d, tag = sys.argv[1], sys.argv[2]
    repl = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    wl = sys.argv[4] if len(sys.argv) > 4 else "spce"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "maniac_mc_amd", "libmaniac_hip.so"), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()          # the build these counters belong to (bench.py checks it)
    meta = {}
    for path in glob.glob(os.path.join(d, f"{wl}_*.log")):
        for line in open(path):
            if line.startswith("{") and '"evaluations_per_launch_group"' in line:
                meta = json.loads(line)
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(d, f"{wl}_*", "**", "*counter_collection.csv"), recursive=True):
        seen = defaultdict(int)
        rows = list(csv.DictReader(open(path)))
        first = {}
        for r in rows:
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            key = (k, r["Dispatch_Id"])
            first.setdefault(k, r["Dispatch_Id"])
            if r["Dispatch_Id"] == first[k]:
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(os.path.join(d, f"{wl}_*", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    keep = [k for k in vals if "mgpu::" in k and ("pair_sweep" in k or "pair_flat" in k or "pair_frozen" in k or "recip" in k)]
    extra = (" " + os.environ.get("PMC_EXTRA", "")).rstrip()
    lines = [f"# rocprofv3 --pmc passes (separate runs, --kernel-trace only) of `python3 tools/bench_kernels.py --workload {wl} --reps 3 --replicas {repl}{extra}`, build {tag}, libmaniac_hip.so sha256 {sha[:16]}",
             "# per-dispatch means over the dispatches after each kernel's first; whole GPU", ""]
    sys.path.insert(0, root)
    from maniac_mc_amd import _lib
    out = {"build": tag, "workload": wl, "lib_sha256": sha, "source_digest": _lib.source_digest(), "candidates_per_launch": repl,
           "evaluations_per_launch_group": meta.get("evaluations_per_launch_group", 2 * repl),
           "evaluations_per_pair_launch": 2 * repl if wl == "spce" else None, "bench_kernels": meta, "source": "tools/pmc_passes.sh (rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per the gfx950 calibration)",
           "kernels": {}}
    for k in sorted(keep):
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        us = sum(dur[k]) / max(1, len(dur[k]))
        for n in sorted(c):
            lines.append(f"{k:52s} {n:24s} {c[n]:16.0f}   (n={len(vals[k][n])})")
        busy = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0            # per shader engine -> cycles the kernel was resident
        n_disp = max((len(v) for v in vals[k].values()), default=0)
        e = {"avg_us_under_pmc": us, "dispatches_counted": n_disp, "counters": c}
        if busy > 0 and "SQ_ACTIVE_INST_VALU" in c:
            e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * busy)
            e["lds_busy"] = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * busy)
            e["lds_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0))
            e["valu_instr_per_launch"] = c.get("SQ_INSTS_VALU", 0.0)
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            e["hbm_bytes_per_launch"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        out["kernels"][k] = e
        lines.append(f"#   {k}: " + ", ".join(f"{a}={b:.4g}" for a, b in e.items() if a != "counters"))
        lines.append("")
    with open(os.path.join(d, f"pmc_kernels_{wl}_{tag}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(d, f"pmc_{wl}_{tag}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
