#!/usr/bin/env python3
"""The reciprocal update's forms against each other on random molecules and boxes: whatever form the engine picks by itself (narrow
row form, matrix-unit wide row form with one or several tiles of site-states) against the per-k kernel (MGPU_RECIP_PER_K=1) and,
where it applies, the vector wide row form (MGPU_RECIP_NO_MFMA=1).  Rigid molecules of 6-200 sites, cubic boxes of 14-78 A and
Ewald tolerances 1e-4..1e-6 (kmax 3-20: row tiles of fewer than 16 rows, more than 16 kz per row), sheared boxes, moves,
insertions and deletions.  Energies must agree within 5.03e-8 K (relative to the reciprocal energy where that is larger) and
A(k) after the commits within 1e-10.

    python tools/recip_forms_stress.py [--cases 30] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.engine import Engine  # noqa: E402
from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE  # noqa: E402


def engine_with(env, s, cap):
    os.environ.update(env)
    try:
        return Engine.from_system(s, n_replicas=3, mol_capacity=[cap])
    finally:
        for k in env:
            os.environ.pop(k, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        n_sites = int(rng.choice([6, 7, 9, 12, 17, 24, 33, 48, 64, 100, 128, 200]))
        radius = float(np.sqrt(n_sites * 1.2 ** 2 * 1.6 / (4 * np.pi)))
        L = float(rng.uniform(max(14.0, 4 * radius + 10.0), 78.0))
        tol = float(rng.choice([1e-4, 1e-5, 1e-6]))
        rc = min(12.0, L / 2 - 0.5)
        n_mol = int(rng.integers(2, 4)) if L > 6 * radius + 12 else 2
        s = synth.large_adsorbate_box(n_sites=n_sites, n_mol=n_mol, L=L, seed=int(rng.integers(1, 10 ** 6)), rc=rc, tol=tol)
        tilt = None
        if rng.uniform() < 0.3:
            tilt = rng.uniform(-0.2, 0.2, 3) * L
            s.box_matrix = np.array([[L, 0.0, 0.0], [tilt[0], L, 0.0], [tilt[1], tilt[2], L]])
            frac = (s.com[0] - s.bounds_lo[None, :]) / L
            s.com[0] = s.bounds_lo[None, :] + frac @ s.box_matrix.T
        cap = n_mol + 2
        engines = {"default": engine_with({}, s, cap), "per_k": engine_with({"MGPU_RECIP_PER_K": "1"}, s, cap),
                   "vector": engine_with({"MGPU_RECIP_NO_MFMA": "1"}, s, cap)}
        for e in engines.values():
            for r in range(3):
                e.init_structure_factor(r, True)
        n = n_mol
        sites = s.all_sites(0)[:n] + rng.uniform(-0.4, 0.4, (n, 1, 3))
        rep = np.zeros(n, np.int32); tt = np.zeros(n, np.int32); mm = np.arange(n, dtype=np.int32)
        new_site = (s.bounds_lo + rng.uniform(0.2, 0.8, 3) * L)[None, None, :] + s.offsets[0][:1]
        res = {}
        for name, e in engines.items():
            old, new = e.trial_energy_candidates(rep, tt, mm, sites)
            u_c = e.recip_energy_candidates([1], [0], [-1], [MGPU_CREATION], new_site)
            u_d = e.recip_energy_candidates([2], [0], [0], [MGPU_DELETION], np.zeros((1, n_sites, 3)))
            e.commit_candidates([0], [0], [n - 1], [MGPU_MOVE], sites[n - 1:n], [1])
            e.commit_candidates([1], [0], [-1], [MGPU_CREATION], new_site, [1])
            e.commit_candidates([2], [0], [0], [MGPU_DELETION], None, [1])
            res[name] = (old[:, 2], new[:, 2], u_c, u_d, [e.structure_factor(r) for r in range(3)])
        ref = res["per_k"]
        scale = max(1.0, float(np.max(np.abs(ref[0]))))
        worst_e = worst_a = 0.0
        for name in ("default", "vector"):
            got = res[name]
            for i in range(4):
                worst_e = max(worst_e, float(np.max(np.abs(np.asarray(got[i]) - np.asarray(ref[i])))))
            for r in range(3):
                worst_a = max(worst_a, float(np.max(np.abs(got[4][r] - ref[4][r]))))
        ok = worst_e <= 5.03e-8 * max(1.0, scale * 1e-6) and worst_a <= 1e-10
        bad += 0 if ok else 1
        kmax = engines["default"].kmax
        print(f"case {case:3d} sites {n_sites:3d} molecules {n_mol} L {L:5.1f} tol {tol:g} kmax {tuple(int(k) for k in kmax)} Nk {engines['default'].nk:5d} "
              f"{'sheared' if tilt is not None else 'cubic  '}: max |dE| {worst_e:.2e} K (|E| up to {scale:.2e}), max |dA| {worst_a:.2e}  {'ok' if ok else 'DIFFERENT'}", flush=True)
        for e in engines.values():
            e.close()
    print(f"{a.cases} cases, {bad} different")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
