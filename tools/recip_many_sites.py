#!/usr/bin/env python3
"""Time of the reciprocal update for molecules beyond the row form's LDS budget (the per-k kernel, sites in LDS tiles):
trial energies (old + new from one pass) and commits of N candidates, one per replica, kernel time from dispatch events.

    python tools/recip_many_sites.py [--sites 24,128,300] [--replicas 1024]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import _lib, synth  # noqa: E402
from maniac_mc_amd.engine import Engine  # noqa: E402
from maniac_mc_amd._lib import MGPU_MOVE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", default="24,128,300")
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    for n_sites in [int(x) for x in args.sites.split(",")]:
        if n_sites <= 24:
            s = synth.rigid_adsorbate_box(n_mol=64, n_sites=n_sites, L=60.0, seed=17)
        else:
            s = synth.large_adsorbate_box(n_sites=n_sites, n_mol=3, L=44.0 if n_sites > 128 else 36.0)
        R = args.replicas
        eng = Engine.from_system(s, n_replicas=R)
        eng.init_structure_factor(0, True)
        for r in range(1, R):
            eng.replica_copy(r, 0)
        rng = np.random.default_rng(1)
        n_mol = int(s.n_mol[0])
        m = rng.integers(0, n_mol, R).astype(np.int32)
        sites = s.all_sites(0)[m] + rng.uniform(-0.2, 0.2, (R, 1, 3))
        rep = np.arange(R, dtype=np.int32)
        tt = np.zeros(R, np.int32)
        eng.profile_enable(True)
        eng.trial_energy_candidates(rep, tt, m, sites)
        eng.profile_reset()
        for _ in range(args.reps):
            eng.trial_energy_candidates(rep, tt, m, sites)
        n_k, ms_k = eng.profile_get(_lib.KERNEL_RECIP)
        n_p, ms_p = eng.profile_get(_lib.KERNEL_PAIR)
        eng.profile_reset()
        eng.commit_candidates(rep, tt, m, np.full(R, MGPU_MOVE, np.int32), sites, np.ones(R, np.int32))
        n_c, ms_c = eng.profile_get(_lib.KERNEL_COMMIT)
        nk = eng.nk
        # row form (narrow / wide / matrix-unit kernels: every molecule unless MGPU_RECIP_PER_K is set): 4 FMAs per site-state and
        # TASK (a +-kz pair); per-k form: two site sets, two complex products of 8 flops per site and k -- the form's own count
        wide = os.environ.get("MGPU_RECIP_PER_K") is None
        flop = 2.0 * n_sites * (nk / 2.0) * 8 if wide else 2.0 * 2 * n_sites * nk * 16
        print(f"sites {n_sites:4d}  Nk {nk:5d}  candidates {R}: k sweep {ms_k / max(1, n_k) * 1e3:9.1f} us  ({R * flop / (ms_k / max(1, n_k) * 1e-3) / 1e12:6.2f} TFLOP/s, "
              f"{'row' if wide else 'per-k'} form count)"
              f"   commit {ms_c / max(1, n_c) * 1e3:9.1f} us   pair sweep {ms_p / max(1, n_p) * 1e3:9.1f} us", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
