"""Shared helpers for the test-suite (fixtures -> System, tolerances)."""
import os

import numpy as np

from maniac_mc_amd import synth
from maniac_mc_amd.system import KB_KCALMOL, System, Topology

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: Delta E within 1e-10 kcal/mol of the reference.  Internal unit is Kelvin.
TOL_KCALMOL = 1.0e-10
TOL_K = TOL_KCALMOL / KB_KCALMOL          # 5.03e-8 K


def tol_for(*values):
    """Absolute tolerance in K: the stated 1e-10 kcal/mol, or 16 ulp of the largest magnitude
    involved when that is bigger (static totals reach 1e7..1e8 K, where 1 ulp ~ 1e-8 K)."""
    big = max(abs(float(v)) for v in values) if values else 0.0
    return max(TOL_K, 16 * np.finfo(np.float64).eps * big)


def farm_tol(ref, steps):
    """Absolute tolerance in K for a chain's RUNNING energy components against a from-scratch evaluation after
    `steps` Metropolis steps: the stated 1e-10 kcal/mol plus a random-walk allowance of 64 ulp of the largest component
    per accepted move (every accepted move adds new - old, each carrying a rounding of that size)."""
    big = float(np.max(np.abs(np.asarray(ref, dtype=np.float64)))) if np.size(ref) else 0.0
    return TOL_K + 64 * np.finfo(np.float64).eps * big * np.sqrt(max(1, steps))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def system_from_golden(g):
    topo = Topology(g["atoms_in_res"], g["atom_types"], g["charges"], g["is_active"], g["epsilon"], g["sigma"])
    com = [g[f"com{t}"] for t in range(topo.n_res)]
    off = [g[f"off{t}"] for t in range(topo.n_res)]
    return System(topo, g["box_matrix"], g["bounds_lo"], float(g["rc_in"]), float(g["tol_in"]),
                  float(g["temperature"]), com, off)


# fixtures with stored coordinates, and the seeded generators for the scalar-only ones
GOLDEN_FULL = ["spce216", "mixture", "argon256", "co2_20", "framework_small"]
GOLDEN_SCALARS = {"spce1000_scalars": lambda: synth.spce_box(10), "spce3375_scalars": lambda: synth.spce_box(15),
                  "framework2208_scalars": lambda: synth.framework_water_box()}


def golden_system(name):
    g = load_golden(name)
    if name in GOLDEN_SCALARS:
        return g, GOLDEN_SCALARS[name]()
    return g, system_from_golden(g)


def split_sites(system, t, sites_padded):
    """(com, offsets) for a candidate given absolute padded sites: com := first site, off := rest."""
    n1 = int(system.topo.atoms_in_res[t])
    s = np.asarray(sites_padded)[:n1]
    com = s[0].copy()
    return com, s - com[None, :]
