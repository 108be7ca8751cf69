"""TEST INFRASTRUCTURE ONLY (oracle) -- ctypes view of ``oracle/_ref/libmaniac_ref.so``.

That library is the reference's own Fortran (/root/reference/src/*.f90, compiled unmodified by
``oracle/Makefile`` with amdflang) plus ``oracle/ref_shim.f90``.  It exists so tests can ask the
reference itself for answers on explicit inputs.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product path never does.

The reference keeps all state in module globals (simulation_state.f90), so there is exactly one
live system per process: ``Reference(system)`` replaces whatever was loaded before.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmaniac_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def available() -> bool:
    return os.path.exists(LIB_PATH)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"reference oracle not built: {LIB_PATH} (run `make -C oracle ref` "
                               "in the container that has /root/reference)")
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_distance.restype = C.c_double
        _lib.ref_lj.restype = C.c_double
        _lib.ref_coulomb.restype = C.c_double
        _lib.ref_acceptance.restype = C.c_double
        _lib.ref_acceptance_swap.restype = C.c_double
        _lib.ref_table_lookup.restype = C.c_double
        _lib.ref_convert_fugacity.restype = C.c_double
        _lib.ref_setup.restype = C.c_int
        _lib.ref_get_num_residues.restype = C.c_int
    return _lib


class Reference:
    """The reference program's state, loaded from a ``maniac_mc_amd.system.System``."""

    def __init__(self, system):
        self.L = lib()
        self.sys = system
        topo = system.topo
        self.n_res = topo.n_res
        self.max_atom = topo.max_atom
        charges_f = np.asfortranarray(topo.charges)
        types_f = np.asfortranarray(topo.atom_types)
        eps_f = np.asfortranarray(topo.epsilon)
        sig_f = np.asfortranarray(topo.sigma)
        box_f = np.asfortranarray(system.box_matrix)
        rc = self.L.ref_setup(C.c_int(self.n_res), _i(topo.atoms_in_res), C.c_int(self.max_atom),
                              _i(topo.is_active), _d(box_f), _d(system.bounds_lo),
                              C.c_int(1 if system.is_triclinic() else 0),
                              C.c_double(system.real_space_cutoff), C.c_double(system.ewald_tolerance),
                              C.c_double(system.temperature), _d(charges_f), _i(types_f),
                              C.c_int(topo.n_atom_types), _d(eps_f), _d(sig_f))
        assert rc == 0
        for t in range(self.n_res):
            self.set_molecules(t, system.com[t], system.offsets[t])
        alpha = C.c_double(); rcut = C.c_double(); tol = C.c_double(); scr = C.c_double(); fp = C.c_double()
        kmax = np.zeros(3, dtype=np.int32); nk = C.c_int()
        self.L.ref_get_ewald(C.byref(alpha), C.byref(rcut), C.byref(tol), C.byref(scr), C.byref(fp),
                             _i(kmax), C.byref(nk))
        self.alpha, self.rc, self.tol = alpha.value, rcut.value, tol.value
        self.screening, self.fourier_precision = scr.value, fp.value
        self.kmax, self.nk = kmax, nk.value

    # ---- state ---------------------------------------------------------------------------
    def _pad_offsets(self, off):
        n_mol, n1 = off.shape[0], off.shape[1]
        out = np.zeros((n_mol, self.max_atom, 3))
        out[:, :n1, :] = off
        return out

    def set_molecules(self, t, com, off):
        """0-based residue type; com (n,3), off (n,n1,3)."""
        com = np.ascontiguousarray(com, dtype=np.float64)
        offp = np.ascontiguousarray(self._pad_offsets(np.asarray(off, dtype=np.float64)))
        # C (n, max_atom, 3) row-major == Fortran (3, max_atom, n) column-major
        self.L.ref_set_molecules(C.c_int(t + 1), C.c_int(com.shape[0]), _d(com), _d(offp))

    def set_molecule(self, t, m, com, off):
        com = np.ascontiguousarray(com, dtype=np.float64)
        offp = np.ascontiguousarray(self._pad_offsets(np.asarray(off, dtype=np.float64)[None])[0])
        self.L.ref_set_molecule(C.c_int(t + 1), C.c_int(m + 1), _d(com), _d(offp))

    def get_molecule(self, t, m):
        com = np.zeros(3); off = np.zeros((self.max_atom, 3))
        self.L.ref_get_molecule(C.c_int(t + 1), C.c_int(m + 1), _d(com), _d(off))
        return com, off[: int(self.sys.topo.atoms_in_res[t])]

    def set_num_residues(self, t, n):
        self.L.ref_set_num_residues(C.c_int(t + 1), C.c_int(n))

    def num_residues(self, t):
        return int(self.L.ref_get_num_residues(C.c_int(t + 1)))

    def box(self):
        bt = C.c_int(); vol = C.c_double(); rec = np.zeros(9); met = np.zeros(9)
        self.L.ref_get_box(C.byref(bt), C.byref(vol), _d(rec), _d(met))
        return bt.value, vol.value, rec.reshape(3, 3).T.copy(), met

    def kvectors(self):
        nk = self.nk
        kx = np.zeros(nk, np.int32); ky = np.zeros(nk, np.int32); kz = np.zeros(nk, np.int32)
        k2n = np.zeros(nk); k2m = np.zeros(nk); ff = np.zeros(nk); w = np.zeros(nk)
        self.L.ref_get_kvectors(_i(kx), _i(ky), _i(kz), _d(k2n), _d(k2m), _d(ff), _d(w))
        return dict(kx=kx, ky=ky, kz=kz, k2norm=k2n, k2mag=k2m, form_factor=ff, weights=w)

    # ---- energies ------------------------------------------------------------------------
    def system_energy(self):
        out = np.zeros(6)
        self.L.ref_system_energy(_d(out))
        return dict(non_coulomb=out[0], coulomb=out[1], recip_coulomb=out[2], ewald_self=out[3],
                    intra_coulomb=out[4], total=out[5])

    def all_fourier_terms(self):
        self.L.ref_all_fourier_terms()

    def init_amplitude(self, full=True):
        self.L.ref_init_amplitude(C.c_int(1 if full else 0))

    def amplitude(self):
        a = np.zeros((self.nk, 2))
        self.L.ref_get_amplitude(_d(a))
        return a[:, 0] + 1j * a[:, 1]

    def set_amplitude(self, z):
        a = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1))
        self.L.ref_set_amplitude(_d(a))

    def set_energy_recip(self, u):
        self.L.ref_set_energy_recip(C.c_double(u))

    def pair_singlemol(self, t, m):
        a = C.c_double(); b = C.c_double()
        self.L.ref_pair_singlemol(C.c_int(t + 1), C.c_int(m + 1), C.byref(a), C.byref(b))
        return a.value, b.value

    def pair_ordered_singlemol(self, t, m):
        a = C.c_double(); b = C.c_double()
        self.L.ref_pair_ordered_singlemol(C.c_int(t + 1), C.c_int(m + 1), C.byref(a), C.byref(b))
        return a.value, b.value

    def distance(self, t1, m1, a1, t2, m2, a2):
        return self.L.ref_distance(*[C.c_int(v + 1) for v in (t1, m1, a1, t2, m2, a2)])

    def apply_pbc(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float64).copy()
        self.L.ref_apply_pbc(_d(p))
        return p

    def lj(self, r, sigma, eps):
        return self.L.ref_lj(C.c_double(r), C.c_double(sigma), C.c_double(eps))

    def coulomb(self, r, q1, q2):
        return self.L.ref_coulomb(C.c_double(r), C.c_double(q1), C.c_double(q2))

    def fourier_singlemol(self, t, m):
        self.L.ref_fourier_singlemol(C.c_int(t + 1), C.c_int(m + 1))

    def save_fourier(self, t, m):
        self.L.ref_save_fourier(C.c_int(t + 1), C.c_int(m + 1))

    def restore_fourier(self, t, m):
        self.L.ref_restore_fourier(C.c_int(t + 1), C.c_int(m + 1))

    def replace_fourier(self, t, i1, i2):
        self.L.ref_replace_fourier(C.c_int(t + 1), C.c_int(i1 + 1), C.c_int(i2 + 1))

    def phase_tables(self, t, m, a):
        k = self.kmax
        px = np.zeros((2 * k[0] + 1, 2)); py = np.zeros((2 * k[1] + 1, 2)); pz = np.zeros((2 * k[2] + 1, 2))
        self.L.ref_get_phase_tables(C.c_int(t + 1), C.c_int(m + 1), C.c_int(a + 1), _d(px), _d(py), _d(pz))
        return [p[:, 0] + 1j * p[:, 1] for p in (px, py, pz)]

    def recip_singlemol(self, t, m, mode=0):
        """mode 0 move, 1 creation, 2 deletion; mutates A(k) like the reference."""
        u = C.c_double()
        self.L.ref_recip_singlemol(C.c_int(t + 1), C.c_int(m + 1), C.c_int(mode), C.byref(u))
        return u.value

    def recip_total(self):
        u = C.c_double()
        self.L.ref_recip_total(C.byref(u))
        return u.value

    def self_singlemol(self, t):
        e = C.c_double()
        self.L.ref_self_singlemol(C.c_int(t + 1), C.byref(e))
        return e.value

    def intra_singlemol(self, t, m):
        e = C.c_double()
        self.L.ref_intra_singlemol(C.c_int(t + 1), C.c_int(m + 1), C.byref(e))
        return e.value

    def old_energy(self, t, m, kind=0):
        out = np.zeros(6)
        self.L.ref_old_energy(C.c_int(t + 1), C.c_int(m + 1), C.c_int(kind), _d(out))
        return out

    def new_energy(self, t, m, kind=0):
        out = np.zeros(6)
        self.L.ref_new_energy(C.c_int(t + 1), C.c_int(m + 1), C.c_int(kind), _d(out))
        return out

    def acceptance(self, old_total, new_total, t, move_type, fugacity=1.0):
        return self.L.ref_acceptance(C.c_double(old_total), C.c_double(new_total), C.c_int(t + 1),
                                     C.c_int(move_type), C.c_double(fugacity))

    def table_lookup(self, which, r):
        return self.L.ref_table_lookup(C.c_int(which), C.c_double(r))

    def acceptance_swap(self, old_total, new_total, t_old, t_new, fug_old, fug_new):
        return self.L.ref_acceptance_swap(C.c_double(old_total), C.c_double(new_total), C.c_int(t_old + 1), C.c_int(t_new + 1),
                                          C.c_double(fug_old), C.c_double(fug_new))

    def rotation_matrix(self, axis, theta):
        r = np.zeros(9)
        self.L.ref_rotation_matrix(C.c_int(axis), C.c_double(theta), _d(r))
        return r.reshape(3, 3).T.copy()

    def convert_fugacity(self, f_atm, temp_K):
        return self.L.ref_convert_fugacity(C.c_double(f_atm), C.c_double(temp_K))

    def constants(self):
        out = np.zeros(8)
        self.L.ref_constants(_d(out))
        return dict(PI=out[0], TWOPI=out[1], SQRTPI=out[2], EPS0_INV_eVA=out[3], KB_eVK=out[4],
                    KB_kcalmol=out[5], error=out[6], NB_MAX_MOLECULE=int(out[7]))

    def close(self):
        self.L.ref_teardown()
