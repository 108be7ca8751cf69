"""Replica farm: R independent Markov chains per GPU, one trial move per chain per step.

Host-side mirror (numpy, vectorised over replicas) of the reference's move drivers for the NVT
move set, with the energies coming from the HIP engine:

  Translation           /root/reference/src/translation.f90:36-112
  Rotation              /root/reference/src/rotation.f90:34-75, monte_carlo_utils.f90:30-92
  move selection        /root/reference/src/monte_carlo.f90:50-75
  acceptance            /root/reference/src/monte_carlo_utils.f90:184-226  (min(1, exp(-dE/T)), K units)
  AcceptMove            /root/reference/src/monte_carlo_utils.f90:410-422

A single chain is sequential, so concurrency comes from running many chains: every step evaluates
one candidate per replica in ONE batched launch (old + new state of each candidate), applies the
Metropolis test on the host and commits the accepted candidates in one more launch.  Rejected
candidates cost nothing: evaluation never mutates engine state.  Random numbers come from numpy's
PCG64, not from the Fortran runtime, so trajectories are not comparable move by move with the
reference (they are not comparable between Fortran compilers either, random_utils.f90:13-18).
"""
from __future__ import annotations

import numpy as np

from ._lib import MGPU_MOVE
from .engine import Engine
from .system import System


class ReplicaFarm:
    def __init__(self, system: System, n_replicas: int, device: int = 0, seed: int = 1,
                 translation_step: float = 0.3, rotation_step: float = 0.3, p_translation: float = 0.5):
        self.sys = system
        self.R = int(n_replicas)
        self.T = float(system.temperature)
        self.translation_step = float(translation_step)
        self.rotation_step = float(rotation_step)
        self.p_translation = float(p_translation)
        self.rng = np.random.default_rng(seed)
        self.eng = Engine.from_system(system, n_replicas=self.R, device=device, extra_capacity=0)
        self.eng.init_structure_factor(0, True)
        for r in range(1, self.R):
            self.eng.replica_copy(r, 0)
        topo = system.topo
        self.active = np.flatnonzero(topo.is_active).astype(np.int32)
        # host mirrors of primary%mol_com / site_offset for the active residue types
        self.com = {int(t): np.repeat(system.com[t][None], self.R, axis=0) for t in self.active}
        self.off = {int(t): np.repeat(system.offsets[t][None], self.R, axis=0) for t in self.active}
        self.L = np.diag(system.box_matrix).copy()
        self.lo = system.bounds_lo.copy()
        self.max_n1 = int(max(topo.atoms_in_res[t] for t in self.active))
        e0 = self.eng.system_energy(0)
        self.energy = np.tile(np.array([e0["non_coulomb"], e0["coulomb"], e0["recip_coulomb"]]), (self.R, 1))
        self.trials = 0
        self.accepted = 0
        self.n_translation = np.zeros(2, dtype=np.int64)   # trials, accepted
        self.n_rotation = np.zeros(2, dtype=np.int64)

    def close(self):
        self.eng.close()

    def step(self):
        """One Metropolis trial per replica (R trials, 2R Delta-E evaluations on the GPU)."""
        R, rng = self.R, self.rng
        rep = np.arange(R, dtype=np.int32)
        t = self.active[(rng.random(R) * len(self.active)).astype(np.int64)]          # PickRandomResidueType
        m = np.zeros(R, dtype=np.int32)
        sites = np.zeros((R, self.max_n1, 3))
        is_trans = rng.random(R) <= self.p_translation                                # monte_carlo.f90:53
        new_com = {}
        new_off = {}
        for tt in self.active:
            tt = int(tt)
            sel = np.flatnonzero(t == tt)
            if sel.size == 0:
                continue
            n_mol, n1 = self.com[tt].shape[1], self.off[tt].shape[2]
            mm = np.minimum((rng.random(sel.size) * n_mol).astype(np.int64), n_mol - 1)  # PickRandomMoleculeIndex
            m[sel] = mm
            com = self.com[tt][sel, mm]
            off = self.off[tt][sel, mm]
            tr = is_trans[sel] | (n1 == 1)
            # RandomTranslation: rand_symmetric(3) * translation_step, then ApplyPBC (translation.f90:104-110)
            disp = (rng.random((sel.size, 3)) - 0.5) * self.translation_step
            ncom = np.where(tr[:, None], self.lo + np.mod(com + disp - self.lo, self.L), com)
            # ApplyRandomRotation: theta = (u - 1/2) * rotation_step_angle about a random Cartesian axis
            theta = (rng.random(sel.size) - 0.5) * self.rotation_step
            axis = (rng.random(sel.size) * 3.0).astype(np.int64)
            theta = np.where(tr, 0.0, theta)
            c, s = np.cos(theta), np.sin(theta)
            rot = np.zeros((sel.size, 3, 3))
            rot[:, 0, 0] = rot[:, 1, 1] = rot[:, 2, 2] = 1.0
            i = (axis + 1) % 3
            j = (axis + 2) % 3
            k = np.arange(sel.size)
            # RotationMatrix (helper_utils.f90:39-77): axis X -> (2,2)=c (2,3)=-s (3,2)=s (3,3)=c, etc.
            rot[k, i, i] = c; rot[k, j, j] = c
            rot[k, i, j] = -s; rot[k, j, i] = s
            noff = np.where(tr[:, None, None], off, np.einsum("bij,baj->bai", rot, off))
            sites[sel, :n1] = ncom[:, None, :] + noff
            new_com[tt] = (sel, mm, ncom)
            new_off[tt] = (sel, mm, noff)
        old, new = self.eng.trial_energy_candidates(rep, t, m, sites)
        d_e = new.sum(axis=1) - old.sum(axis=1)
        with np.errstate(over="ignore"):
            prob = np.minimum(1.0, np.exp(-d_e / self.T))                             # monte_carlo_utils.f90:218
        accept = rng.random(R) <= prob
        self.eng.commit_candidates(rep, t, m, np.full(R, MGPU_MOVE, np.int32), sites, accept.astype(np.int32))
        for tt, (sel, mm, ncom) in new_com.items():
            a = accept[sel]
            self.com[tt][sel[a], mm[a]] = ncom[a]
            self.off[tt][sel[a], mm[a]] = new_off[tt][2][a]
        self.energy[accept] += new[accept] - old[accept]                              # AcceptMove
        n_acc = int(accept.sum())
        self.trials += R
        self.accepted += n_acc
        self.n_translation += (int(is_trans.sum()), int((accept & is_trans).sum()))
        self.n_rotation += (int((~is_trans).sum()), int((accept & ~is_trans).sum()))
        return n_acc

    def run(self, n_steps: int):
        acc = 0
        for _ in range(n_steps):
            acc += self.step()
        return acc
