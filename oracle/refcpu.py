"""TEST INFRASTRUCTURE ONLY (oracle) -- ctypes view of ``oracle/librefcpu.so`` (oracle/refcpu.c).

Same method names and 0-based conventions as ``oracle.reflib.Reference`` so that the C
restatement can be pinned against the compiled reference call by call.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librefcpu.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def build(force=False):
    """Compile the restatement with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "refcpu.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "port"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.refcpu_create.restype = C.c_void_p
        for name in ("refcpu_distance", "refcpu_lj", "refcpu_coulomb", "refcpu_recip_singlemol",
                     "refcpu_recip_total", "refcpu_self_singlemol", "refcpu_intra_singlemol",
                     "refcpu_acceptance", "refcpu_acceptance_swap", "refcpu_convert_fugacity", "refcpu_table_lookup"):
            getattr(L, name).restype = C.c_double
        L.refcpu_get_num_residues.restype = C.c_int
        _lib = L
    return _lib


class RefCPU:
    def __init__(self, system, mol_capacity=None):
        self.L = lib()
        self.sys = system
        topo = system.topo
        self.n_res, self.max_atom = topo.n_res, topo.max_atom
        cap = int(mol_capacity or max(8, int(system.n_mol.max()) + 64))
        self.h = C.c_void_p(self.L.refcpu_create(
            C.c_int(self.n_res), _i(topo.atoms_in_res), C.c_int(self.max_atom), _i(topo.is_active),
            _d(np.ascontiguousarray(system.box_matrix)), _d(system.bounds_lo),
            C.c_int(1 if system.is_triclinic() else 0), C.c_double(system.real_space_cutoff),
            C.c_double(system.ewald_tolerance), _d(topo.charges), _i(topo.atom_types),
            C.c_int(topo.n_atom_types), _d(topo.epsilon), _d(topo.sigma), C.c_int(cap)))
        for t in range(self.n_res):
            self.set_molecules(t, system.com[t], system.offsets[t])
        alpha = C.c_double(); rcut = C.c_double(); tol = C.c_double(); scr = C.c_double(); fp = C.c_double()
        kmax = np.zeros(3, dtype=np.int32); nk = C.c_int()
        self.L.refcpu_get_ewald(self.h, C.byref(alpha), C.byref(rcut), C.byref(tol), C.byref(scr),
                                C.byref(fp), _i(kmax), C.byref(nk))
        self.alpha, self.rc, self.tol = alpha.value, rcut.value, tol.value
        self.screening, self.fourier_precision = scr.value, fp.value
        self.kmax, self.nk = kmax, nk.value

    def close(self):
        if self.h:
            self.L.refcpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ---------------------------------------------------------------------------
    def set_molecules(self, t, com, off):
        com = np.ascontiguousarray(com, dtype=np.float64)
        off = np.ascontiguousarray(off, dtype=np.float64)
        self.L.refcpu_set_molecules(self.h, C.c_int(t), C.c_int(com.shape[0]), _d(com), _d(off))

    def set_molecule(self, t, m, com, off):
        com = np.ascontiguousarray(com, dtype=np.float64)
        off = np.ascontiguousarray(off, dtype=np.float64)
        self.L.refcpu_set_molecule(self.h, C.c_int(t), C.c_int(m), _d(com), _d(off))

    def get_molecule(self, t, m):
        n1 = int(self.sys.topo.atoms_in_res[t])
        com = np.zeros(3); off = np.zeros((n1, 3))
        self.L.refcpu_get_molecule(self.h, C.c_int(t), C.c_int(m), _d(com), _d(off))
        return com, off

    def set_num_residues(self, t, n):
        self.L.refcpu_set_num_residues(self.h, C.c_int(t), C.c_int(n))

    def num_residues(self, t):
        return int(self.L.refcpu_get_num_residues(self.h, C.c_int(t)))

    def box(self):
        bt = C.c_int(); vol = C.c_double(); rec = np.zeros(9); met = np.zeros(9)
        self.L.refcpu_get_box(self.h, C.byref(bt), C.byref(vol), _d(rec), _d(met))
        return bt.value, vol.value, rec.reshape(3, 3).copy(), met

    def kvectors(self):
        nk = self.nk
        kx = np.zeros(nk, np.int32); ky = np.zeros(nk, np.int32); kz = np.zeros(nk, np.int32)
        k2n = np.zeros(nk); k2m = np.zeros(nk); ff = np.zeros(nk); w = np.zeros(nk)
        self.L.refcpu_get_kvectors(self.h, _i(kx), _i(ky), _i(kz), _d(k2n), _d(k2m), _d(ff), _d(w))
        return dict(kx=kx, ky=ky, kz=kz, k2norm=k2n, k2mag=k2m, form_factor=ff, weights=w)

    # ---- energies ------------------------------------------------------------------------
    def system_energy(self):
        out = np.zeros(6)
        self.L.refcpu_system_energy(self.h, _d(out))
        return dict(non_coulomb=out[0], coulomb=out[1], recip_coulomb=out[2], ewald_self=out[3],
                    intra_coulomb=out[4], total=out[5])

    def all_fourier_terms(self):
        self.L.refcpu_all_fourier_terms(self.h)

    def init_amplitude(self, full=True):
        self.L.refcpu_init_amplitude(self.h, C.c_int(1 if full else 0))

    def amplitude(self):
        a = np.zeros((self.nk, 2))
        self.L.refcpu_get_amplitude(self.h, _d(a))
        return a[:, 0] + 1j * a[:, 1]

    def set_amplitude(self, z):
        a = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1))
        self.L.refcpu_set_amplitude(self.h, _d(a))

    def set_energy_recip(self, u):
        self.L.refcpu_set_energy_recip(self.h, C.c_double(u))

    def pair_singlemol(self, t, m):
        a = C.c_double(); b = C.c_double()
        self.L.refcpu_pair_singlemol(self.h, C.c_int(t), C.c_int(m), C.byref(a), C.byref(b))
        return a.value, b.value

    def pair_ordered_singlemol(self, t, m):
        a = C.c_double(); b = C.c_double()
        self.L.refcpu_pair_ordered_singlemol(self.h, C.c_int(t), C.c_int(m), C.byref(a), C.byref(b))
        return a.value, b.value

    def distance(self, t1, m1, a1, t2, m2, a2):
        return self.L.refcpu_distance(self.h, *[C.c_int(v) for v in (t1, m1, a1, t2, m2, a2)])

    def apply_pbc(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float64).copy()
        self.L.refcpu_apply_pbc(self.h, _d(p))
        return p

    def lj(self, r, sigma, eps):
        return self.L.refcpu_lj(self.h, C.c_double(r), C.c_double(sigma), C.c_double(eps))

    def coulomb(self, r, q1, q2):
        return self.L.refcpu_coulomb(self.h, C.c_double(r), C.c_double(q1), C.c_double(q2))

    def fourier_singlemol(self, t, m):
        self.L.refcpu_fourier_singlemol(self.h, C.c_int(t), C.c_int(m))

    def save_fourier(self, t, m):
        self.L.refcpu_save_fourier(self.h, C.c_int(t), C.c_int(m))

    def restore_fourier(self, t, m):
        self.L.refcpu_restore_fourier(self.h, C.c_int(t), C.c_int(m))

    def replace_fourier(self, t, i1, i2):
        self.L.refcpu_replace_fourier(self.h, C.c_int(t), C.c_int(i1), C.c_int(i2))

    def phase_tables(self, t, m, a):
        k = self.kmax
        px = np.zeros((2 * k[0] + 1, 2)); py = np.zeros((2 * k[1] + 1, 2)); pz = np.zeros((2 * k[2] + 1, 2))
        self.L.refcpu_get_phase_tables(self.h, C.c_int(t), C.c_int(m), C.c_int(a), _d(px), _d(py), _d(pz))
        return [p[:, 0] + 1j * p[:, 1] for p in (px, py, pz)]

    def recip_singlemol(self, t, m, mode=0):
        return self.L.refcpu_recip_singlemol(self.h, C.c_int(t), C.c_int(m), C.c_int(mode))

    def recip_total(self):
        return self.L.refcpu_recip_total(self.h)

    def self_singlemol(self, t):
        return self.L.refcpu_self_singlemol(self.h, C.c_int(t))

    def intra_singlemol(self, t, m):
        return self.L.refcpu_intra_singlemol(self.h, C.c_int(t), C.c_int(m))

    def old_energy(self, t, m, kind=0):
        out = np.zeros(6)
        self.L.refcpu_old_energy(self.h, C.c_int(t), C.c_int(m), C.c_int(kind), _d(out))
        return out

    def new_energy(self, t, m, kind=0):
        out = np.zeros(6)
        self.L.refcpu_new_energy(self.h, C.c_int(t), C.c_int(m), C.c_int(kind), _d(out))
        return out

    def acceptance(self, old_total, new_total, t, move_type, fugacity=1.0):
        n = float(self.num_residues(t))
        vol = self.box()[1]
        return self.L.refcpu_acceptance(C.c_double(old_total), C.c_double(new_total), C.c_double(n),
                                        C.c_double(vol), C.c_double(fugacity),
                                        C.c_double(self.sys.temperature), C.c_int(move_type))

    def table_lookup(self, which, r):
        """The reference's tabulated potentials (dead code there): 1 erfc(alpha r)/r, 2 r**6, 3 r**12."""
        return self.L.refcpu_table_lookup(self.h, C.c_int(which), C.c_double(r))

    def acceptance_swap(self, old_total, new_total, t_old, t_new, fug_old, fug_new):
        return self.L.refcpu_acceptance_swap(C.c_double(old_total), C.c_double(new_total), C.c_int(self.num_residues(t_old)),
                                             C.c_int(self.num_residues(t_new)), C.c_double(fug_old), C.c_double(fug_new),
                                             C.c_double(self.sys.temperature))

    def rotation_matrix(self, axis, theta):
        r = np.zeros(9)
        self.L.refcpu_rotation_matrix(C.c_int(axis), C.c_double(theta), _d(r))
        return r.reshape(3, 3).copy()

    def convert_fugacity(self, f_atm, temp_K):
        return self.L.refcpu_convert_fugacity(C.c_double(f_atm), C.c_double(temp_K))


def trial_farm(system, n_chains, n_threads, budget_s, translation_step, rotation_step, seed=11):
    """All-core CPU baseline: n_chains independent copies of the sequential translation / rotation trial (the
    reference's loop, restated), spread over n_threads OpenMP threads for budget_s seconds.
    Returns (elapsed_s, trials_per_chain, accepted_per_chain)."""
    L = lib()
    L.refcpu_trial_farm.restype = C.c_double
    chains = []
    for _ in range(n_chains):
        P = RefCPU(system)
        P.all_fourier_terms()
        P.init_amplitude(True)
        chains.append(P)
    handles = (C.c_void_p * n_chains)(*[p.h for p in chains])
    counts = np.zeros(2 * n_chains, dtype=np.int64)
    el = L.refcpu_trial_farm(handles, C.c_int(n_chains), C.c_int(n_threads), C.c_double(budget_s),
                             C.c_ulonglong(seed), C.c_double(translation_step), C.c_double(rotation_step),
                             C.c_double(system.temperature), counts.ctypes.data_as(C.POINTER(C.c_longlong)))
    for p in chains:
        p.close()
    return float(el), counts[0::2].copy(), counts[1::2].copy()
