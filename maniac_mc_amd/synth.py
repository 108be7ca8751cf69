"""Synthetic benchmark / parity systems (recipes and seeds: SURVEY.md section 8(d)).

The reference's own example systems live in the ``mc-topology`` git submodule, which is empty
in the snapshot (/root/reference/.gitmodules:1-3), so every system here is synthetic and
labelled as such.  Force-field numbers for SPC/E, argon, TraPPE CO2 and TIP4P-like water are
textbook values and are NOT taken from the reference.

epsilon is converted from kcal/mol to Kelvin exactly as the reference does it
(``epsilon / KB_kcalmol``, parameters_parser.f90:83); unlike pairs are filled by
Lorentz-Berthelot where both like-pair entries are non-zero (parameters_parser.f90:141-176).
"""
from __future__ import annotations

import numpy as np

from .system import KB_KCALMOL, System, Topology


def lorentz_berthelot(eps_diag_kcal, sig_diag):
    """(nt,) like-pair eps [kcal/mol] / sigma [A] -> (nt, nt) tables in K / A.

    Mirrors ApplyLorentzBerthelot (parameters_parser.f90:116-182): arithmetic-mean sigma,
    geometric-mean epsilon, applied only when the mixed values exceed 1e-6.
    """
    eps_k = np.asarray(eps_diag_kcal, dtype=np.float64) / KB_KCALMOL
    sig = np.asarray(sig_diag, dtype=np.float64)
    nt = eps_k.shape[0]
    e = np.zeros((nt, nt))
    s = np.zeros((nt, nt))
    for i in range(nt):
        e[i, i] = eps_k[i]
        s[i, i] = sig[i]
    for i in range(nt):
        for j in range(nt):
            if i == j:
                continue
            sm = (sig[i] + sig[j]) / 2
            em = np.sqrt(eps_k[i] * eps_k[j])
            if sm > 1e-6 and em > 1e-6:
                s[i, j] = sm
                e[i, j] = em
    return e, s


def _random_rotations(rng, n):
    """Random orientations: Q of the QR of a normal 3x3."""
    out = np.empty((n, 3, 3))
    for i in range(n):
        q, r = np.linalg.qr(rng.normal(size=(3, 3)))
        out[i] = q * np.sign(np.diag(r))[None, :]
    return out


def _spce_template():
    """Rigid SPC/E: r_OH = 1 A, HOH = 109.47 deg; offsets relative to the centre of mass."""
    r_oh = 1.0
    half = np.deg2rad(109.47) / 2
    pos = np.array([[0.0, 0.0, 0.0],
                    [r_oh * np.sin(half), r_oh * np.cos(half), 0.0],
                    [-r_oh * np.sin(half), r_oh * np.cos(half), 0.0]])
    mass = np.array([15.9994, 1.008, 1.008])
    com = (pos * mass[:, None]).sum(0) / mass.sum()
    return pos - com


def spce_topology():
    eps, sig = lorentz_berthelot([0.1553, 0.0], [3.166, 0.0])
    return Topology(atoms_in_res=[3], atom_types=[[1, 2, 2]], charges=[[-0.8476, 0.4238, 0.4238]],
                    is_active=[1], epsilon=eps, sigma=sig, names=["SPCE"])


def spce_box(n_side=10, seed=12345, spacing=3.104, jitter=0.2, rc=12.0, tol=1e-5, temperature=300.0):
    """n_side^3 rigid SPC/E molecules on a jittered lattice.

    n_side = 10 -> 1000 molecules, L = 31.04 A, Nk = 783   (BASELINE.json configs[1])
    n_side = 15 -> 3375 molecules, N = 10125, L = 46.56 A, Nk = 2242   (10k-atom headline point)
    """
    rng = np.random.default_rng(seed)
    L = n_side * spacing
    g = (np.arange(n_side) + 0.5) * spacing - L / 2
    com = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    com = com + rng.uniform(-jitter, jitter, size=com.shape)
    rot = _random_rotations(rng, com.shape[0])
    off = np.einsum("mij,aj->mai", rot, _spce_template())
    return System(spce_topology(), np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature,
                  [com], [off], label=f"spce_{com.shape[0]}mol_{3 * com.shape[0]}atoms")


def argon_box(n_cell=4, rho_star=0.8, rc=10.0, tol=1e-5, temperature=120.0):
    """256 Ar on an fcc 4x4x4 lattice, rho* = 0.8 -> L = 23.2899 A (BASELINE.json configs[0])."""
    sigma = 3.405
    n = 4 * n_cell ** 3
    L = (n / rho_star) ** (1.0 / 3.0) * sigma
    a = L / n_cell
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    cells = np.stack(np.meshgrid(*[np.arange(n_cell)] * 3, indexing="ij"), -1).reshape(-1, 3)
    com = ((cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) + 0.25) * a - L / 2
    eps, sig = lorentz_berthelot([0.238], [sigma])
    topo = Topology(atoms_in_res=[1], atom_types=[[1]], charges=[[0.0]], is_active=[1],
                    epsilon=eps, sigma=sig, names=["Ar"])
    return System(topo, np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature,
                  [com], [np.zeros((n, 1, 3))], label=f"argon_{n}")


def co2_topology():
    # TraPPE CO2 (textbook values, not from the reference): C 27 K / 2.80 A / +0.70,
    # O 79 K / 3.05 A / -0.35, r_CO = 1.16 A
    eps, sig = lorentz_berthelot([27.0 * KB_KCALMOL, 79.0 * KB_KCALMOL], [2.80, 3.05])
    return Topology(atoms_in_res=[3], atom_types=[[1, 2, 2]], charges=[[0.70, -0.35, -0.35]],
                    is_active=[1], epsilon=eps, sigma=sig, names=["CO2"])


def co2_box(n_mol=64, L=50.0, seed=7, rc=12.0, tol=1e-5, temperature=300.0):
    """Rigid linear CO2 in a cubic 50 A box (BASELINE.json configs[2] stand-in)."""
    rng = np.random.default_rng(seed)
    tmpl = np.array([[0.0, 0.0, 0.0], [1.16, 0.0, 0.0], [-1.16, 0.0, 0.0]])
    com = _spread_points(rng, n_mol, L, min_sep=4.0)
    rot = _random_rotations(rng, n_mol)
    off = np.einsum("mij,aj->mai", rot, tmpl)
    return System(co2_topology(), np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature,
                  [com], [off], label=f"co2_{n_mol}mol_L{L:g}")


def five_site_water_box(n_mol=24, L=20.0, seed=13, rc=9.0, tol=1e-5, temperature=298.0):
    """Rigid five-site water (TIP5P-like textbook geometry, not from the reference): O carries the
    Lennard-Jones site and no charge, two H (+0.241 e) and two lone-pair sites (-0.241 e).  Five sites per
    molecule take the generic (site count not templated) pair sweep and the per-k reciprocal kernel."""
    rng = np.random.default_rng(seed)
    r_oh, r_ol = 0.9572, 0.70
    a_h, a_l = np.deg2rad(104.52) / 2, np.deg2rad(109.47) / 2
    tmpl = np.array([[0.0, 0.0, 0.0],
                     [r_oh * np.sin(a_h), r_oh * np.cos(a_h), 0.0], [-r_oh * np.sin(a_h), r_oh * np.cos(a_h), 0.0],
                     [0.0, -r_ol * np.cos(a_l), r_ol * np.sin(a_l)], [0.0, -r_ol * np.cos(a_l), -r_ol * np.sin(a_l)]])
    tmpl = tmpl - tmpl.mean(0)
    eps, sig = lorentz_berthelot([0.16, 0.0, 0.0], [3.12, 0.0, 0.0])
    topo = Topology(atoms_in_res=[5], atom_types=[[1, 2, 2, 3, 3]], charges=[[0.0, 0.241, 0.241, -0.241, -0.241]],
                    is_active=[1], epsilon=eps, sigma=sig, names=["W5"])
    com = _spread_points(rng, n_mol, L, min_sep=3.0)
    off = np.einsum("mij,aj->mai", _random_rotations(rng, n_mol), tmpl)
    return System(topo, np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature, [com], [off],
                  label=f"water5_{n_mol}mol")


def rigid_adsorbate_box(n_mol=6, n_sites=24, L=26.0, seed=17, rc=10.0, tol=1e-5, temperature=300.0):
    """A rigid adsorbate of ``n_sites`` (default 24) sites: two stacked 12-rings of alternating +-0.25 e sites of
    two atom types (entirely synthetic).  With 24 sites the row-form reciprocal kernel's XY table exceeds its
    LDS budget, so this molecule can only take the per-k kernel (and the generic-site-count pair sweep)."""
    rng = np.random.default_rng(seed)
    per_ring = n_sites // 2
    ang = 2 * np.pi * np.arange(per_ring) / per_ring
    ring = np.stack([2.4 * np.cos(ang), 2.4 * np.sin(ang), np.zeros(per_ring)], 1)
    tmpl = np.vstack([ring + [0, 0, 0.9], ring[:n_sites - per_ring] @ np.array([[np.cos(0.26), -np.sin(0.26), 0],
                                                                                [np.sin(0.26), np.cos(0.26), 0], [0, 0, 1]]).T - [0, 0, 0.9]])
    tmpl = tmpl - tmpl.mean(0)
    types = np.array([1 + (i % 2) for i in range(n_sites)], dtype=np.int32)
    q = np.array([0.25 if i % 2 == 0 else -0.25 for i in range(n_sites)])
    q -= q.mean()
    eps, sig = lorentz_berthelot([0.09, 0.06], [3.3, 3.0])
    topo = Topology(atoms_in_res=[n_sites], atom_types=[types], charges=[q], is_active=[1], epsilon=eps, sigma=sig,
                    names=["CAGE"])
    com = _spread_points(rng, n_mol, L, min_sep=8.5)
    off = np.einsum("mij,aj->mai", _random_rotations(rng, n_mol), tmpl)
    return System(topo, np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature, [com], [off],
                  label=f"cage{n_sites}_{n_mol}mol")


def large_adsorbate_box(n_sites=300, n_mol=3, L=44.0, seed=23, rc=12.0, tol=1e-5, temperature=300.0):
    """A rigid adsorbate of MANY sites (default 300: a fullerene-like shell, sites >= 1.2 A apart, three atom types,
    charges of both signs summing to zero; entirely synthetic): larger than any LDS phase table, so its reciprocal
    update runs tile by tile and its intra-molecular sum a wave per molecule.  The reference moves, inserts and deletes
    a residue of any max_atom_in_residue (src/ewald_phase.f90:383-420, src/prepare_utils.f90:233-235)."""
    rng = np.random.default_rng(seed)
    radius = float(np.sqrt(n_sites * 1.2 ** 2 * 1.6 / (4 * np.pi)))
    pts = np.empty((0, 3))
    while pts.shape[0] < n_sites:
        v = rng.normal(size=3)
        v *= radius * (1.0 + 0.08 * rng.uniform(-1, 1)) / np.linalg.norm(v)
        if pts.shape[0] and np.min(np.linalg.norm(pts - v, axis=1)) < 1.2:
            continue
        pts = np.vstack([pts, v])
    tmpl = pts - pts.mean(0)
    types = (1 + (np.arange(n_sites) % 3)).astype(np.int32)
    q = rng.choice([-0.3, -0.1, 0.0, 0.1, 0.3], n_sites)
    q[q != 0.0] -= q.sum() / np.count_nonzero(q)          # neutral; the uncharged sites stay exactly uncharged
    eps, sig = lorentz_berthelot([0.09, 0.06, 0.0], [3.3, 3.0, 0.0])
    topo = Topology(atoms_in_res=[n_sites], atom_types=[types], charges=[q], is_active=[1], epsilon=eps, sigma=sig,
                    names=["SHELL"])
    com = _spread_points(rng, n_mol, L, min_sep=2 * radius + 4.0)
    off = np.einsum("mij,aj->mai", _random_rotations(rng, n_mol), tmpl)
    return System(topo, np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature, [com], [off],
                  label=f"shell{n_sites}_{n_mol}mol")


def _spread_points(rng, n, L, min_sep):
    """n points in [-L/2, L/2)^3 with pairwise minimum-image separation >= min_sep."""
    pts = np.empty((0, 3))
    while pts.shape[0] < n:
        p = rng.uniform(-L / 2, L / 2, size=3)
        if pts.shape[0]:
            d = pts - p
            d -= L * np.rint(d / L)
            if np.min(np.einsum("ij,ij->i", d, d)) < min_sep ** 2:
                continue
        pts = np.vstack([pts, p])
    return pts


def framework_water_box(n_water=40, seed=11, L=34.0, rc=12.0, tol=1e-5, temperature=300.0, n_frame=2208):
    """Synthetic stand-in for the README's ZIF-8 + water case (BASELINE.json configs[3]).

    One inactive residue of ``n_frame`` atoms (7 atom types, net-neutral charges, jittered
    simple-cubic lattice so that the minimum separation stays >= 1.5 A) plus a 4-site
    TIP4P-like water as the active species.  Entirely synthetic.
    """
    rng = np.random.default_rng(seed)
    side = int(np.ceil(n_frame ** (1 / 3)))
    a = L / side
    grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pick = rng.permutation(grid.shape[0])[:n_frame]
    fpos = (grid[np.sort(pick)] + 0.5) * a - L / 2
    fpos = fpos + rng.uniform(-(a - 1.5) / 2 * 0.9, (a - 1.5) / 2 * 0.9, size=fpos.shape)
    ftypes = rng.integers(1, 8, size=n_frame).astype(np.int32)
    fq = rng.uniform(-0.6, 0.6, size=n_frame)
    fq -= fq.mean()
    fcom = fpos.mean(0)
    # water: types 8 (O), 9 (H), 10 (M)
    r_oh, ang, r_om = 0.9572, np.deg2rad(104.52), 0.15
    h = np.array([[r_oh * np.sin(ang / 2), r_oh * np.cos(ang / 2), 0.0],
                  [-r_oh * np.sin(ang / 2), r_oh * np.cos(ang / 2), 0.0]])
    wt = np.vstack([[0.0, 0.0, 0.0], h, [0.0, r_om, 0.0]])
    mass = np.array([15.9994, 1.008, 1.008, 0.0])
    wt = wt - (wt * mass[:, None]).sum(0) / mass.sum()
    wcom = np.empty((0, 3))
    while wcom.shape[0] < n_water:
        p = rng.uniform(-L / 2, L / 2, size=3)
        d = fpos - p
        d -= L * np.rint(d / L)
        if np.min(np.einsum("ij,ij->i", d, d)) < 2.6 ** 2:
            continue
        if wcom.shape[0]:
            d = wcom - p
            d -= L * np.rint(d / L)
            if np.min(np.einsum("ij,ij->i", d, d)) < 3.0 ** 2:
                continue
        wcom = np.vstack([wcom, p])
    woff = np.einsum("mij,aj->mai", _random_rotations(rng, n_water), wt)
    eps_d = [0.05 + 0.02 * i for i in range(7)] + [0.1550, 0.0, 0.0]
    sig_d = [2.6 + 0.15 * i for i in range(7)] + [3.1536, 0.0, 0.0]
    eps, sig = lorentz_berthelot(eps_d, sig_d)
    max_atom = max(n_frame, 4)
    atom_types = np.zeros((2, max_atom), dtype=np.int32)
    charges = np.zeros((2, max_atom))
    atom_types[0, :n_frame] = ftypes
    charges[0, :n_frame] = fq
    atom_types[1, :4] = [8, 9, 9, 10]
    charges[1, :4] = [0.0, 0.52, 0.52, -1.04]
    topo = Topology(atoms_in_res=[n_frame, 4], atom_types=atom_types, charges=charges, is_active=[0, 1],
                    epsilon=eps, sigma=sig, names=["FRAME", "H2O"])
    return System(topo, np.diag([L, L, L]), np.full(3, -L / 2), rc, tol, temperature,
                  [fcom[None, :], wcom], [(fpos - fcom)[None, :, :], woff],
                  label=f"framework{n_frame}_water{n_water}")


def mixture_box(n_a=12, n_b=9, box=(18.0, 21.0, 24.0), seed=3, rc=8.0, tol=1e-4, temperature=250.0,
                bounds_lo=None, tilt=None):
    """Small two-residue mixture in an orthorhombic (or, with ``tilt``, triclinic) box.

    Residue 1: 3-site molecule with one uncharged site (exercises the |q| < 1e-10 skip,
    energy_utils.f90:430); residue 2: diatomic.  Used for edge-case parity only.
    """
    rng = np.random.default_rng(seed)
    box = np.asarray(box, dtype=np.float64)
    eps, sig = lorentz_berthelot([0.12, 0.20, 0.0, 0.07], [3.0, 3.4, 0.0, 2.7])
    topo = Topology(atoms_in_res=[3, 2], atom_types=[[1, 2, 3], [4, 2, 0]],
                    charges=[[0.5, -0.5, 0.0], [0.3, -0.3, 0.0]], is_active=[1, 1],
                    epsilon=eps, sigma=sig, names=["A3", "B2"])
    lo = -box / 2 if bounds_lo is None else np.asarray(bounds_lo, dtype=np.float64)
    n = n_a + n_b
    frac = np.empty((0, 3))
    while frac.shape[0] < n:
        p = rng.uniform(0, 1, size=3)
        if frac.shape[0]:
            d = frac - p
            d -= np.rint(d)
            if np.min(np.einsum("ij,ij->i", d * box, d * box)) < 3.2 ** 2:
                continue
        frac = np.vstack([frac, p])
    mat = np.diag(box)
    if tilt is not None:
        # LAMMPS convention as stored by the reference reader (readers_utils.f90:242-245):
        # rows a=(lx,0,0), b=(xy,ly,0), c=(xz,yz,lz)
        xy, xz, yz = tilt
        mat = np.array([[box[0], 0.0, 0.0], [xy, box[1], 0.0], [xz, yz, box[2]]])
    com = lo[None, :] + frac * box[None, :]
    ta = np.array([[0.0, 0.0, 0.0], [1.1, 0.0, 0.0], [-0.4, 0.9, 0.0]])
    ta -= ta.mean(0)
    tb = np.array([[0.6, 0.0, 0.0], [-0.6, 0.0, 0.0]])
    off_a = np.einsum("mij,aj->mai", _random_rotations(rng, n_a), ta)
    off_b = np.einsum("mij,aj->mai", _random_rotations(rng, n_b), tb)
    return System(topo, mat, lo, rc, tol, temperature, [com[:n_a], com[n_a:]], [off_a, off_b],
                  label=f"mixture_{n_a}_{n_b}")
