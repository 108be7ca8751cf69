#!/usr/bin/env python3
"""How many 64-molecule sweep units could a spatially sorted molecule order skip?  (CPU only, numpy.)

The reference sums erfc(alpha r)/r over EVERY minimum-image pair (src/energy_utils.f90:427-432: no Coulomb cutoff), and at
the 10 125-atom benchmark box 35 % of the pairs lie beyond 25 A, where a term is below 1e-12 K.  The pair sweep works in
units of 64 consecutive molecules of one site plane, so a term can only be saved if ALL 64 molecules of a unit are beyond
r_far from EVERY site of the candidate.  This script sorts the molecules by Morton order of a cell grid (the most compact
64-molecule groups one can form: ~12.4 A cubes) and counts, for random candidates, the units whose nearest atom is
farther than r_far + 0.6 A (0.6 A: room for the new state of a 0.3 A trial move).

Result (seeded, deterministic): 0.2 - 1.6 % of the units for r_far = 24 ... 26 A -- the region beyond 25 A is the eight
thin corner wedges of the cube, into which a 12 A group does not fit.  Unit skipping under an error budget therefore
cannot pay at this box size, whatever the permutation layer costs; recorded in DESIGN.md section 4.1.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402


def morton(ix, iy, iz, bits=4):
    k = np.zeros_like(ix)
    for b in range(bits):
        k |= ((ix >> b) & 1) << (3 * b) | ((iy >> b) & 1) << (3 * b + 1) | ((iz >> b) & 1) << (3 * b + 2)
    return k


def main():
    s = synth.spce_box(15)
    L = float(s.box_matrix[0, 0])
    com = s.com[0]
    n = com.shape[0]
    sites = s.all_sites(0)
    rng = np.random.default_rng(0)
    cand = rng.choice(n, 200, replace=False)
    print("cells/axis  r_far[A]  skippable units")
    for ng in (4, 8, 16):
        cell = np.floor((com - s.bounds_lo) / L * ng).astype(int).clip(0, ng - 1)
        order = np.argsort(morton(cell[:, 0], cell[:, 1], cell[:, 2]), kind="stable")
        so = sites[order]
        n_units = (n + 63) // 64
        for r_far in (24.0, 25.0, 26.0):
            skip = tot = 0
            for c in cand:
                cs = sites[c]
                for u in range(n_units):
                    blk = so[u * 64:(u + 1) * 64].reshape(-1, 3)
                    d = blk[:, None, :] - cs[None, :, :]
                    d -= L * np.rint(d / L)
                    tot += 1
                    skip += np.sqrt((d ** 2).sum(-1)).min() > r_far + 0.6
            print(f"{ng:10d}  {r_far:8.1f}  {skip / tot:.4f}")


if __name__ == "__main__":
    main()
