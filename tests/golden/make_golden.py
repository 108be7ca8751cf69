"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the container that has /root/reference: it drives oracle/_ref/libmaniac_ref.so
(the unmodified reference Fortran compiled by oracle/Makefile with amdflang, behind
oracle/ref_shim.f90) on the synthetic systems of maniac_mc_amd/synth.py and stores inputs and
the reference's outputs as .npz.  Fixtures are data only.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from maniac_mc_amd import synth  # noqa: E402
from oracle import reflib  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rot(axis, theta):
    c, s = np.cos(theta), np.sin(theta)
    r = np.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    r[i, i] = c; r[j, j] = c
    r[i, j] = -s if axis != 1 else s
    r[j, i] = s if axis != 1 else -s
    return r


def system_arrays(s):
    d = dict(atoms_in_res=s.topo.atoms_in_res, atom_types=s.topo.atom_types, charges=s.topo.charges,
             is_active=s.topo.is_active, epsilon=s.topo.epsilon, sigma=s.topo.sigma, box_matrix=s.box_matrix,
             bounds_lo=s.bounds_lo, rc_in=s.real_space_cutoff, tol_in=s.ewald_tolerance, temperature=s.temperature)
    for t in range(s.topo.n_res):
        d[f"com{t}"] = s.com[t]
        d[f"off{t}"] = s.offsets[t]
    return d


def make(name, s, moves, store_coords=True, n_amp_store=None):
    """moves: list of (t, m, dcom(3), axis, theta)"""
    rng = np.random.default_rng(99)
    R = reflib.Reference(s)
    out = system_arrays(s) if store_coords else dict(rc_in=s.real_space_cutoff, tol_in=s.ewald_tolerance)
    bt, vol, rcp, met = R.box()
    out.update(box_type=bt, volume=vol, reciprocal=rcp, metrics=met, alpha=R.alpha, rc_eff=R.rc, tol_eff=R.tol,
               kmax=R.kmax, nk=R.nk)
    kv = R.kvectors()
    out.update({f"k_{k}": v for k, v in kv.items()})
    e = R.system_energy()
    out["system_energy"] = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self",
                                                    "intra_coulomb", "total")])
    R.init_amplitude(True)
    A0 = R.amplitude()
    out["A_full"] = A0 if n_amp_store is None else A0[:n_amp_store]
    mv_t, mv_m, mv_sites, mv_old, mv_new, mv_intra = [], [], [], [], [], []
    A_after = []
    for (t, m, dcom, axis, theta) in moves:
        R.set_amplitude(A0)
        com, off = R.get_molecule(t, m)
        R.save_fourier(t, m)
        old = R.old_energy(t, m, 0)
        ncom = R.apply_pbc(com + np.asarray(dcom))
        noff = off @ rot(axis, theta).T
        R.set_molecule(t, m, ncom, noff)
        new = R.new_energy(t, m, 0)
        mv_t.append(t); mv_m.append(m)
        mv_sites.append(np.pad(ncom[None, :] + noff, ((0, s.topo.max_atom - noff.shape[0]), (0, 0))))
        mv_old.append(old); mv_new.append(new)
        mv_intra.append(R.intra_singlemol(t, m))
        A_after.append(R.amplitude() if n_amp_store is None else R.amplitude()[:n_amp_store])
        # reject: restore coordinates and tables
        R.set_molecule(t, m, com, off)
        R.restore_fourier(t, m)
    out.update(mv_t=np.array(mv_t), mv_m=np.array(mv_m), mv_sites=np.array(mv_sites), mv_old=np.array(mv_old),
               mv_new=np.array(mv_new), mv_intra=np.array(mv_intra), mv_A_after=np.array(A_after))
    # creation of one molecule of the first active type at a random position (slot n_mol)
    t = int(np.flatnonzero(s.topo.is_active)[0])
    n = R.num_residues(t)
    R.set_amplitude(A0)
    R.set_energy_recip(e["recip_coulomb"])
    old_c = R.old_energy(t, n, 1)
    _, off1 = R.get_molecule(t, 0)
    L = np.diag(s.box_matrix)
    coff = off1 @ rot(2, 1.234).T @ rot(0, 0.5).T
    # insertion point: the roomiest of 400 random positions (an overlapping insertion gives
    # ~1e9 K of LJ repulsion, where fp64 cannot resolve the absolute 1e-10 kcal/mol bar)
    allsites = np.concatenate([s.all_sites(tt).reshape(-1, 3) for tt in range(s.topo.n_res)])
    best, ccom = -1.0, None
    for _ in range(400):
        c = s.bounds_lo + rng.uniform(0.0, 1.0, 3) * L
        d = (allsites[None, :, :] - (c[None, :] + coff)[:, None, :])
        d -= L * np.rint(d / L)
        dmin = np.sqrt(np.min(np.einsum("sai,sai->sa", d, d)))
        if dmin > best:
            best, ccom = dmin, c
    R.set_num_residues(t, n + 1)
    R.save_fourier(t, n)
    R.set_molecule(t, n, ccom, coff)
    new_c = R.new_energy(t, n, 1)
    out.update(cr_t=t, cr_sites=ccom[None, :] + coff, cr_old=old_c, cr_new=new_c,
               cr_A_after=R.amplitude() if n_amp_store is None else R.amplitude()[:n_amp_store])
    R.set_num_residues(t, n)
    # deletion of molecule m_del: old energy as the reference computes it; the recip energy of the
    # (N-1) system the INTENDED way (A - S_mol; mode 2), since the reference's own call is F3-defective
    m_del = min(2, n - 1)
    R.set_amplitude(A0)
    R.all_fourier_terms()
    old_d = R.old_energy(t, m_del, 2)
    R.save_fourier(t, m_del)
    u_del = R.recip_singlemol(t, m_del, 2)
    out.update(dl_t=t, dl_m=m_del, dl_old=old_d, dl_recip_new=u_del,
               dl_A_after=R.amplitude() if n_amp_store is None else R.amplitude()[:n_amp_store])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "nk", R.nk, "alpha", R.alpha, "total", e["total"])


def main():
    only = set(sys.argv[1:])

    def make_if(name, *a, **k):
        if not only or name in only:
            make(name, *a, **k)
    _main(make_if)


def _main(make):
    make("spce216", synth.spce_box(6),
         [(0, 5, [0.11, -0.07, 0.13], 0, 0.0), (0, 40, [0, 0, 0], 1, 0.21), (0, 215, [-0.14, 0.02, 0.1], 2, -0.13)])
    make("mixture", synth.mixture_box(),
         [(0, 3, [0.2, 0.1, -0.3], 0, 0.1), (1, 2, [-0.25, 0.3, 0.05], 2, 0.3), (1, 8, [0, 0, 0], 1, -0.2)])
    make("argon256", synth.argon_box(), [(0, 0, [0.1, 0.2, -0.1], 0, 0.0), (0, 255, [-0.3, 0.1, 0.2], 0, 0.0)])
    make("co2_20", synth.co2_box(20), [(0, 7, [0.3, -0.2, 0.4], 1, 0.25), (0, 19, [0, 0, 0], 0, -0.3)])
    make("framework_small", synth.framework_water_box(n_water=12, n_frame=300, L=24.0),
         [(1, 4, [0.2, -0.1, 0.1], 2, 0.2), (1, 11, [0.05, 0.1, -0.2], 0, -0.15)])
    # full-size points: coordinates come from the seeded generator, only scalars are stored
    make("spce1000_scalars", synth.spce_box(10), [(0, 17, [0.1, 0.05, -0.12], 1, 0.2), (0, 999, [-0.1, 0.1, 0.1], 2, -0.1)],
         store_coords=False, n_amp_store=64)
    make("spce3375_scalars", synth.spce_box(15), [(0, 100, [0.1, 0.05, -0.12], 1, 0.2), (0, 3374, [-0.1, 0.1, 0.1], 2, -0.1)],
         store_coords=False, n_amp_store=64)
    # BASELINE.json configs[3] at its stated size: the 2208-atom inactive framework + 40 four-site waters in a 34 A box
    make("framework2208_scalars", synth.framework_water_box(),
         [(1, 3, [0.2, -0.1, 0.1], 2, 0.2), (1, 39, [0.05, 0.1, -0.2], 0, -0.15), (1, 17, [-0.3, 0.25, 0.1], 1, 0.4)],
         store_coords=False, n_amp_store=64)


if __name__ == "__main__":
    main()
