#!/usr/bin/env python3
"""Kernel-only timings (HIP events inside the library) of the hot-path kernels on the benchmark box.

    python tools/bench_kernels.py [--replicas 1024] [--reps 10]
Environment knobs read by the library: MGPU_PAIR_NSPLIT, MGPU_PAIR_BLOCKS_PER_CU.
MANIAC_HIP_LIB selects an alternative build of libmaniac_hip.so (tuning variants).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import _lib  # noqa: E402

if os.environ.get("MANIAC_HIP_LIB"):
    _lib.LIB_PATH = os.environ["MANIAC_HIP_LIB"]
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n-side", type=int, default=15)
    args = ap.parse_args()
    s = synth.spce_box(args.n_side)
    R = args.replicas
    eng = Engine.from_system(s, n_replicas=R, extra_capacity=0)
    eng.init_structure_factor(0, True)
    for r in range(1, R):
        eng.replica_copy(r, 0)
    rng = np.random.default_rng(0)
    n = int(s.n_mol[0])
    rep = np.arange(R, dtype=np.int32)
    t = np.zeros(R, np.int32)
    m = rng.integers(0, n, R).astype(np.int32)
    sites = s.all_sites(0)[m] + rng.uniform(-0.15, 0.15, (R, 1, 3))
    eng.trial_energy_candidates(rep, t, m, sites)
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(args.reps):
        eng.trial_energy_candidates(rep, t, m, sites)
        eng.commit_candidates(rep, t, m, np.zeros(R, np.int32), sites, (rng.random(R) < 0.7).astype(np.int32))
    names = ["pair_sweep", "recip", "commit", "sfactor"]
    out = {}
    for k, nm in enumerate(names):
        cnt, ms = eng.profile_get(k)
        if cnt:
            out[nm] = ms / cnt * 1e3
    evals = 2 * R
    print(f"R={R} lib={os.path.basename(_lib.LIB_PATH)} nsplit={os.environ.get('MGPU_PAIR_NSPLIT','auto')} "
          f"blocks/CU={os.environ.get('MGPU_PAIR_BLOCKS_PER_CU','auto')}: " +
          "  ".join(f"{k} {v:.1f} us" for k, v in out.items()) +
          f"  | pair {out['pair_sweep'] * 1e3 / evals:.1f} ns/eval, recip {out['recip'] * 1e3 / evals:.1f} ns/eval")
    eng.close()


if __name__ == "__main__":
    main()
