"""Python mirror of the energy engine's C ABI (include/maniac_gpu.h) -- plumbing only.

Two layers:

* batched calls (``pair_energy_candidates`` ...): thin ctypes wrappers, one per C entry point;
* reference-named B = 1 seams (``ComputePairInteractionEnergy_singlemol`` ...): the same names,
  argument meaning (residue type, molecule index) and outputs as the Fortran procedures listed in
  SURVEY.md section 8(b), so parity tests read like calls into the reference.  Indices are 0-based.

Every call goes to the HIP library; nothing here computes energies on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE, MGPU_NONE, check
from .system import System, Topology

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _ints(x, n=None):
    a = np.ascontiguousarray(np.atleast_1d(x), dtype=np.int32)
    if n is not None and a.shape[0] == 1 and n != 1:
        a = np.full(n, int(a[0]), dtype=np.int32)
    return a


def box_prepare(box_matrix):
    """mgpu_box_prepare: box type, volume, ``box%reciprocal`` and ``box%metrics``."""
    L = _lib.lib()
    m = np.ascontiguousarray(box_matrix, dtype=np.float64).reshape(9)
    bt = C.c_int(); vol = C.c_double(); rcp = np.zeros(9); met = np.zeros(9)
    check(L.mgpu_box_prepare(_d(m), C.byref(bt), C.byref(vol), _d(rcp), _d(met)))
    return bt.value, vol.value, rcp.reshape(3, 3), met


def ewald_setup(metrics, rc, tol):
    """mgpu_ewald_setup (SetupEwald, prepare_utils.f90:103-214)."""
    L = _lib.lib()
    met = np.ascontiguousarray(metrics, dtype=np.float64)
    rc_ = C.c_double(rc); tol_ = C.c_double(tol); alpha = C.c_double(); scr = C.c_double(); fp = C.c_double()
    kmax = np.zeros(3, dtype=np.int32); nk = C.c_int()
    check(L.mgpu_ewald_setup(_d(met), C.byref(rc_), C.byref(tol_), C.byref(alpha), C.byref(scr), C.byref(fp),
                             _i(kmax), C.byref(nk)))
    return dict(rc=rc_.value, tol=tol_.value, alpha=alpha.value, screening=scr.value,
                fourier_precision=fp.value, kmax=kmax, nk=nk.value)


def ewald_kvectors(reciprocal, alpha, kmax, nk):
    """mgpu_ewald_kvectors (PrecomputeValidReciprocalVectors + ComputeReciprocalWeights)."""
    L = _lib.lib()
    rcp = np.ascontiguousarray(reciprocal, dtype=np.float64).reshape(9)
    km = np.ascontiguousarray(kmax, dtype=np.int32)
    kx = np.zeros(nk, np.int32); ky = np.zeros(nk, np.int32); kz = np.zeros(nk, np.int32)
    k2 = np.zeros(nk); ff = np.zeros(nk); w = np.zeros(nk)
    check(L.mgpu_ewald_kvectors(_d(rcp), C.c_double(alpha), _i(km), C.c_int(nk), _i(kx), _i(ky), _i(kz),
                                _d(k2), _d(ff), _d(w)))
    return dict(kx=kx, ky=ky, kz=kz, k2mag=k2, form_factor=ff, weights=w)


class Engine:
    """One HIP device, R replicas of one (box, force field, k table)."""

    def __init__(self, topo: Topology, box_matrix, bounds_lo, real_space_cutoff, ewald_tolerance,
                 n_replicas: int = 1, device: int = 0, mol_capacity: Optional[Sequence[int]] = None):
        self.L = _lib.lib()
        self.topo = topo
        self.n_replicas = int(n_replicas)
        if mol_capacity is None:
            mol_capacity = [64] * topo.n_res
        self.mol_capacity = np.ascontiguousarray(mol_capacity, dtype=np.int32)
        self.h = C.c_void_p()
        bm = np.ascontiguousarray(box_matrix, dtype=np.float64).reshape(9)
        lo = np.ascontiguousarray(bounds_lo, dtype=np.float64)
        check(self.L.mgpu_engine_create(
            C.byref(self.h), C.c_int(device), C.c_int(self.n_replicas), C.c_int(topo.n_res),
            _i(topo.atoms_in_res), _i(self.mol_capacity), C.c_int(topo.max_atom), _i(topo.atom_types),
            _d(topo.charges), _i(topo.is_active), C.c_int(topo.n_atom_types), _d(topo.epsilon), _d(topo.sigma),
            _d(bm), _d(lo), C.c_double(real_space_cutoff), C.c_double(ewald_tolerance)))
        alpha = C.c_double(); rc = C.c_double(); tol = C.c_double(); vol = C.c_double(); bt = C.c_int()
        kmax = np.zeros(3, dtype=np.int32); nk = C.c_int()
        check(self.L.mgpu_engine_get_ewald(self.h, C.byref(alpha), C.byref(rc), C.byref(tol), _i(kmax),
                                           C.byref(nk), C.byref(vol), C.byref(bt)))
        self.alpha, self.rc, self.tol, self.volume = alpha.value, rc.value, tol.value, vol.value
        self.kmax, self.nk, self.box_type = kmax, nk.value, bt.value
        self.max_n1 = int(topo.atoms_in_res.max())

    @classmethod
    def from_system(cls, system: System, n_replicas: int = 1, device: int = 0, mol_capacity=None,
                    extra_capacity: int = 8):
        """Engine sized for ``system`` with every replica loaded with that configuration."""
        if mol_capacity is None:
            mol_capacity = [int(n) + (extra_capacity if system.topo.is_active[t] else 0)
                            for t, n in enumerate(system.n_mol)]
            mol_capacity = [max(1, c) for c in mol_capacity]
        eng = cls(system.topo, system.box_matrix, system.bounds_lo, system.real_space_cutoff,
                  system.ewald_tolerance, n_replicas, device, mol_capacity)
        eng.load_system(system, 0)
        for r in range(1, n_replicas):
            eng.replica_copy(r, 0)
        return eng

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.mgpu_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- replica state -------------------------------------------------------------------
    def load_system(self, system: System, replica: int = 0):
        for t in range(self.topo.n_res):
            self.set_molecules(replica, t, system.all_sites(t))

    def set_molecules(self, replica, t, sites):
        sites = np.ascontiguousarray(sites, dtype=np.float64).reshape(-1, int(self.topo.atoms_in_res[t]), 3)
        check(self.L.mgpu_replica_set_molecules(self.h, C.c_int(replica), C.c_int(t), C.c_int(sites.shape[0]),
                                                _d(sites)))

    def get_molecules(self, replica, t):
        n = self.num_molecules(replica, t)
        sites = np.zeros((n, int(self.topo.atoms_in_res[t]), 3))
        nm = C.c_int()
        check(self.L.mgpu_replica_get_molecules(self.h, C.c_int(replica), C.c_int(t), C.byref(nm), _d(sites)))
        return sites

    def set_frames(self, replica, t, com, off):
        """mgpu_replica_set_frames: com (n, 3) and offsets (n, n1, 3) as the reference keeps them; the sites are com + off."""
        n1 = int(self.topo.atoms_in_res[t])
        com = np.ascontiguousarray(com, dtype=np.float64).reshape(-1, 3)
        off = np.ascontiguousarray(np.asarray(off, dtype=np.float64)[:, :n1, :]).reshape(-1, n1, 3)
        check(self.L.mgpu_replica_set_frames(self.h, C.c_int(replica), C.c_int(t), C.c_int(com.shape[0]), _d(com), _d(off)))

    def get_frames(self, replica, t):
        n = self.num_molecules(replica, t)
        n1 = int(self.topo.atoms_in_res[t])
        com = np.zeros((n, 3)); off = np.zeros((n, n1, 3)); nm = C.c_int()
        check(self.L.mgpu_replica_get_frames(self.h, C.c_int(replica), C.c_int(t), C.byref(nm), _d(com), _d(off)))
        return com, off

    def move_trial(self, replica, t, m, move, u, translation_step, rotation_step, lane=0):
        """Device-built trials (mgpu_move_trial_submit + wait): (old[n,5], new[n,5])."""
        m = _ints(m); n = m.shape[0]
        replica = _ints(replica, n); t = _ints(t, n); move = _ints(move, n)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(n, 5)
        old = np.zeros((n, 5)); new = np.zeros((n, 5))
        self._last_stride = 0
        check(self.L.mgpu_move_trial_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(move), _d(u),
                                            C.c_double(translation_step), C.c_double(rotation_step)))
        check(self.L.mgpu_gcmc_trial_wait(self.h, C.c_int(lane), _d(old), _d(new)))
        return old, new

    def move_trial_decide(self, replica, t, m, move, u, translation_step, rotation_step, accept_u, accept_pref, temperature,
                          lane=0):
        """Device-built trials decided and committed on the device (mgpu_move_trial_decide_submit + wait):
        (old[n,5], new[n,5], accepted[n])."""
        m = _ints(m); n = m.shape[0]
        replica = _ints(replica, n); t = _ints(t, n); move = _ints(move, n)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(n, 5)
        au = np.ascontiguousarray(accept_u, dtype=np.float64).reshape(n)
        ap = np.ascontiguousarray(accept_pref, dtype=np.float64).reshape(n)
        old = np.zeros((n, 5)); new = np.zeros((n, 5)); acc = np.zeros(n, np.int32)
        self._last_stride = 0
        check(self.L.mgpu_move_trial_decide_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(move), _d(u),
                                                   C.c_double(translation_step), C.c_double(rotation_step), _d(au), _d(ap),
                                                   C.c_double(temperature)))
        check(self.L.mgpu_trial_decide_wait(self.h, C.c_int(lane), _d(old), _d(new), _i(acc)))
        return old, new, acc

    def farm_window_capacity(self):
        n = C.c_int(0); d = C.c_int(0)
        check(self.L.mgpu_farm_window_capacity(self.h, C.byref(n), C.byref(d)))
        return n.value, d.value

    def farm_window_submit(self, replica, t, m, move, u, translation_step, rotation_step, accept_u, accept_pref, temperature,
                           forced=None, lane=0, slot_u=None):
        """mgpu_farm_window_submit: one launch evaluates, decides and commits one step of every chain given."""
        m = _ints(m); n = m.shape[0]
        replica = _ints(replica, n); t = _ints(t, n); move = _ints(move, n)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(n, 5)
        au = np.ascontiguousarray(accept_u, dtype=np.float64).reshape(n)
        ap = np.ascontiguousarray(accept_pref, dtype=np.float64).reshape(n)
        fo = _ints(0 if forced is None else forced, n)
        su = None if slot_u is None else np.ascontiguousarray(slot_u, dtype=np.float64).reshape(n)
        check(self.L.mgpu_farm_window_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(move), _i(fo), _d(u),
                                             _d(au), _d(ap), _d(su), C.c_double(translation_step), C.c_double(rotation_step),
                                             C.c_double(temperature)))
        return n

    def farm_window_wait(self, n, lane=0):
        """The lane's oldest window: (old[n,5], new[n,5], verdict[n])."""
        old = np.zeros((n, 5)); new = np.zeros((n, 5)); v = np.zeros(n, np.int32)
        check(self.L.mgpu_farm_window_wait(self.h, C.c_int(lane), _d(old), _d(new), _i(v)))
        return old, new, v

    def farm_window_flush(self):
        check(self.L.mgpu_farm_window_flush(self.h))

    def farm_window_stats(self):
        w = C.c_longlong(0); u = C.c_longlong(0)
        check(self.L.mgpu_farm_window_get_stats(self.h, C.byref(w), C.byref(u)))
        return w.value, u.value

    def gcmc_trial_decide(self, replica, t, m, kind, sites, accept_u, accept_pref, temperature, lane=0):
        """Host-built rows, decided and committed on the device: (old[n,5], new[n,5], accepted[n])."""
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        kind = _ints(kind, n)
        au = np.ascontiguousarray(accept_u, dtype=np.float64).reshape(n)
        ap = np.ascontiguousarray(accept_pref, dtype=np.float64).reshape(n)
        old = np.zeros((n, 5)); new = np.zeros((n, 5)); acc = np.zeros(n, np.int32)
        self._last_stride = sites.shape[1]
        check(self.L.mgpu_gcmc_trial_decide_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(kind),
                                                   _d(sites), C.c_int(sites.shape[1]), _d(au), _d(ap), C.c_double(temperature)))
        check(self.L.mgpu_trial_decide_wait(self.h, C.c_int(lane), _d(old), _d(new), _i(acc)))
        return old, new, acc

    def num_molecules(self, replica, t):
        n = C.c_int()
        check(self.L.mgpu_replica_num_molecules(self.h, C.c_int(replica), C.c_int(t), C.byref(n)))
        return n.value

    def set_num_molecules(self, replica, t, n):
        check(self.L.mgpu_replica_set_num_molecules(self.h, C.c_int(replica), C.c_int(t), C.c_int(n)))

    def replica_copy(self, dst, src):
        check(self.L.mgpu_replica_copy(self.h, C.c_int(dst), C.c_int(src)))

    def structure_factor_add(self, replica, t, sites):
        """A(k) += sum_a q_a exp(i k . sites_a) for one molecule of type t; nothing else changes."""
        a = np.ascontiguousarray(sites, dtype=np.float64)[: int(self.topo.atoms_in_res[t])]
        check(self.L.mgpu_structure_factor_add(self.h, C.c_int(replica), C.c_int(t), _d(np.ascontiguousarray(a))))

    def replace_molecule(self, replica, t, m_dst, m_src):
        check(self.L.mgpu_replica_replace_molecule(self.h, C.c_int(replica), C.c_int(t), C.c_int(m_dst),
                                                   C.c_int(m_src)))

    def kvectors(self):
        nk = self.nk
        kx = np.zeros(nk, np.int32); ky = np.zeros(nk, np.int32); kz = np.zeros(nk, np.int32)
        k2 = np.zeros(nk); ff = np.zeros(nk); w = np.zeros(nk)
        check(self.L.mgpu_engine_get_kvectors(self.h, _i(kx), _i(ky), _i(kz), _d(k2), _d(ff), _d(w)))
        return dict(kx=kx, ky=ky, kz=kz, k2mag=k2, form_factor=ff, weights=w)

    # ---- static energies -----------------------------------------------------------------
    def system_energy(self, replica=0):
        out = np.zeros(6)
        check(self.L.mgpu_system_energy(self.h, C.c_int(replica), _d(out)))
        return dict(non_coulomb=out[0], coulomb=out[1], recip_coulomb=out[2], ewald_self=out[3],
                    intra_coulomb=out[4], total=out[5])

    def init_structure_factor(self, replica=0, full=True):
        check(self.L.mgpu_init_structure_factor(self.h, C.c_int(replica), C.c_int(1 if full else 0)))

    def structure_factor(self, replica=0):
        a = np.zeros((self.nk, 2))
        check(self.L.mgpu_get_structure_factor(self.h, C.c_int(replica), _d(a)))
        return a[:, 0] + 1j * a[:, 1]

    def set_structure_factor(self, z, replica=0):
        a = np.ascontiguousarray(np.stack([np.real(z), np.imag(z)], axis=1), dtype=np.float64)
        check(self.L.mgpu_set_structure_factor(self.h, C.c_int(replica), _d(a)))

    # ---- batched candidates --------------------------------------------------------------
    def _cand(self, replica, t, m, sites):
        m = _ints(m)
        n = m.shape[0]
        replica = _ints(replica, n)
        t = _ints(t, n)
        if sites is not None:
            sites = np.ascontiguousarray(sites, dtype=np.float64)
            if sites.ndim == 2:
                sites = sites[None]
            assert sites.shape[0] == n and sites.shape[2] == 3
        return n, replica, t, m, sites

    def pair_energy_candidates(self, replica, t, m, sites=None, use_resident=None):
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        if use_resident is None:
            use_resident = np.full(n, 1 if sites is None else 0, dtype=np.int32)
        use_resident = _ints(use_resident, n)
        e_nc = np.zeros(n); e_c = np.zeros(n)
        stride = 1 if sites is None else sites.shape[1]
        check(self.L.mgpu_pair_energy_candidates(self.h, C.c_int(n), _i(replica), _i(t), _i(m), _i(use_resident),
                                                 _d(sites), C.c_int(stride), _d(e_nc), _d(e_c)))
        return e_nc, e_c

    def recip_energy_candidates(self, replica, t, m, kind, sites=None):
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        kind = _ints(kind, n)
        u = np.zeros(n)
        stride = 1 if sites is None else sites.shape[1]
        check(self.L.mgpu_recip_energy_candidates(self.h, C.c_int(n), _i(replica), _i(t), _i(m), _i(kind),
                                                  _d(sites), C.c_int(stride), _d(u)))
        return u

    def intra_energy_candidates(self, replica, t, m, sites=None, use_resident=None):
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        if use_resident is None:
            use_resident = np.full(n, 1 if sites is None else 0, dtype=np.int32)
        use_resident = _ints(use_resident, n)
        u = np.zeros(n)
        stride = 1 if sites is None else sites.shape[1]
        check(self.L.mgpu_intra_energy_candidates(self.h, C.c_int(n), _i(replica), _i(t), _i(m), _i(use_resident),
                                                  _d(sites), C.c_int(stride), _d(u)))
        return u

    def self_energy(self, t):
        e = C.c_double()
        check(self.L.mgpu_self_energy(self.h, C.c_int(t), C.byref(e)))
        return e.value

    def trial_energy_candidates(self, replica, t, m, sites):
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        old = np.zeros((n, 3)); new = np.zeros((n, 3))
        check(self.L.mgpu_trial_energy_candidates(self.h, C.c_int(n), _i(replica), _i(t), _i(m), _d(sites),
                                                  C.c_int(sites.shape[1]), _d(old), _d(new)))
        return old, new

    def gcmc_trial(self, replica, t, m, kind, sites, lane=0):
        """Mixed batch (moves, creations, deletions): (old[n,5], new[n,5]) as ComputeOld/NewEnergy fill them."""
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        kind = _ints(kind, n)
        old = np.zeros((n, 5)); new = np.zeros((n, 5))
        self._last_stride = sites.shape[1]
        check(self.L.mgpu_gcmc_trial_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(kind),
                                            _d(sites), C.c_int(sites.shape[1])))
        check(self.L.mgpu_gcmc_trial_wait(self.h, C.c_int(lane), _d(old), _d(new)))
        return old, new

    def chain_window_capacity(self):
        n = C.c_int(0)
        check(self.L.mgpu_chain_window_capacity(self.h, C.byref(n)))
        return n.value

    def chain_window(self, replica, t, m, kind, sites, accept_u, accept_pref, temperature, recip_energy, link=None):
        """mgpu_chain_window: one launch evaluates the window's steps (all trials of replica's current state), decides
        them in order and commits the first accepted one.  Returns (old[n,5], new[n,5], first_accepted, undecided)."""
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        n = kind.shape[0]
        t = _ints(t, n); m = _ints(m, n)
        link = _ints(-1 if link is None else link, n)
        sites = np.ascontiguousarray(sites, dtype=np.float64)
        assert sites.ndim == 3 and sites.shape[0] == n and sites.shape[2] == 3
        u = np.ascontiguousarray(accept_u, dtype=np.float64); pref = np.ascontiguousarray(accept_pref, dtype=np.float64)
        assert u.shape == (n,) and pref.shape == (n,)
        old = np.zeros((n, 5)); new = np.zeros((n, 5))
        first = C.c_int(-1); und = C.c_int(-1)
        check(self.L.mgpu_chain_window(self.h, C.c_int(int(replica)), C.c_int(n), _i(t), _i(m), _i(kind), _i(link), _d(sites),
                                       C.c_int(sites.shape[1]), _d(u), _d(pref), C.c_double(temperature), C.c_double(recip_energy),
                                       _d(old), _d(new), C.byref(first), C.byref(und)))
        return old, new, first.value, und.value

    def chain_set_margin(self, rel):
        check(self.L.mgpu_chain_set_margin(self.h, C.c_double(rel)))

    def chain_set_timing(self, on=True):
        check(self.L.mgpu_chain_set_timing(self.h, C.c_int(1 if on else 0)))

    def chain_timing(self):
        """Stage times (us) of the last window: see include/maniac_gpu.h."""
        us = np.zeros(15)
        check(self.L.mgpu_chain_get_timing(self.h, _d(us)))
        return us

    def chain_stats(self):
        w = C.c_longlong(0); u = C.c_longlong(0)
        check(self.L.mgpu_chain_get_stats(self.h, C.byref(w), C.byref(u)))
        return w.value, u.value

    def commit_lane(self, lane, replica, t, m, kind, accept, sites=None, sync=True):
        """mgpu_commit_submit (+ synchronize unless sync=False); sites=None reuses the rows of the lane's last trial."""
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        kind = _ints(kind, n)
        accept = _ints(accept, n)
        stride = self._last_stride if sites is None else sites.shape[1]
        check(self.L.mgpu_commit_submit(self.h, C.c_int(lane), C.c_int(n), _i(replica), _i(t), _i(m), _i(kind),
                                        _d(sites), C.c_int(stride), _i(accept)))
        if sync:
            self.synchronize()

    def commit_candidates(self, replica, t, m, kind, sites, accept):
        n, replica, t, m, sites = self._cand(replica, t, m, sites)
        kind = _ints(kind, n)
        accept = _ints(accept, n)
        stride = 1 if sites is None else sites.shape[1]
        check(self.L.mgpu_commit_candidates(self.h, C.c_int(n), _i(replica), _i(t), _i(m), _i(kind), _d(sites),
                                            C.c_int(stride), _i(accept)))

    # ---- reference-named B = 1 seams ------------------------------
    def ComputeSystemEnergy(self, replica=0):
        """energy_utils.f90:18-35"""
        return self.system_energy(replica)

    def ComputePairInteractionEnergy_singlemol(self, residue_type, molecule_index, sites=None, replica=0):
        """energy_utils.f90:374-442 -> (e_non_coulomb, e_coulomb)"""
        a, b = self.pair_energy_candidates(replica, residue_type, molecule_index, sites)
        return float(a[0]), float(b[0])

    def ComputeRecipEnergySingleMol(self, residue_type, molecule_index, sites=None, is_creation=False,
                                    is_deletion=False, replica=0):
        """ewald_energy.f90:191-274 (non-mutating: commit applies the update)"""
        kind = MGPU_CREATION if is_creation else (MGPU_DELETION if is_deletion else
                                                  (MGPU_NONE if sites is None else MGPU_MOVE))
        return float(self.recip_energy_candidates(replica, residue_type, molecule_index, kind, sites)[0])

    def ComputeEwaldSelfInteractionSingleMol(self, residue_type):
        """ewald_energy.f90:308-336"""
        return self.self_energy(residue_type)

    def ComputeIntraResidueRealCoulombEnergySingleMol(self, residue_type, molecule_index, sites=None, replica=0):
        """ewald_energy.f90:371-411"""
        return float(self.intra_energy_candidates(replica, residue_type, molecule_index, sites)[0])

    def set_host_team(self, n_threads):
        """Host threads the candidate loops inside submit / wait / commit may use (mgpu_set_host_team)."""
        check(self.L.mgpu_set_host_team(self.h, C.c_int(int(n_threads))))

    def phase_factors(self, theta, k):
        """(cos, sin)(k * theta) as the device's phase tables hold them (mgpu_phase_factors; ewald_phase.f90:100-109)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        k = np.ascontiguousarray(k, dtype=np.int32)
        n = theta.shape[0]
        c, s = np.empty(n), np.empty(n)
        check(self.L.mgpu_phase_factors(self.h, C.c_int(n), _d(theta), _i(k), _d(c), _d(s)))
        return c, s

    # ---- measurement ---------------------------------------------------------------------
    def synchronize(self):
        check(self.L.mgpu_synchronize(self.h))

    def profile_enable(self, on=True):
        check(self.L.mgpu_profile_enable(self.h, C.c_int(1 if on else 0)))

    def profile_reset(self):
        check(self.L.mgpu_profile_reset(self.h))

    def profile_get(self, kernel):
        n = C.c_longlong(); ms = C.c_double()
        check(self.L.mgpu_profile_get(self.h, C.c_int(kernel), C.byref(n), C.byref(ms)))
        return n.value, ms.value
