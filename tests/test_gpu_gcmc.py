"""Mixed grand-canonical batches on the GPU (mgpu_gcmc_trial_submit / _wait / mgpu_commit_submit)."""
import numpy as np
import pytest

from maniac_mc_amd import synth
from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE
from maniac_mc_amd.engine import Engine
from tests.test_gpu_parity import amp_close, close

pytestmark = pytest.mark.gpu


def test_mixed_gcmc_batch_matches_oracle(refcpu_mod):
    """Moves, insertions and deletions in ONE batch; 5-component energy states as ComputeOldEnergy /
    ComputeNewEnergy fill them (monte_carlo_utils.f90:275-395); then a resident-row commit on lane 1."""
    rng = np.random.default_rng(21)
    s = synth.co2_box(18, seed=9)
    eng = Engine.from_system(s, n_replicas=3, mol_capacity=[30])
    for r in range(3):
        eng.init_structure_factor(r, True)
    P = refcpu_mod.RefCPU(s, mol_capacity=30)
    e_sys = P.system_energy()
    P.init_amplitude(True)
    P.set_energy_recip(e_sys["recip_coulomb"])
    n = 18
    tmpl = s.offsets[0][0]
    kinds = np.array([MGPU_MOVE, MGPU_CREATION, MGPU_DELETION, MGPU_MOVE, MGPU_CREATION, MGPU_DELETION], dtype=np.int32)
    rep = np.array([0, 1, 2, 1, 2, 0], dtype=np.int32)      # two candidates per replica, same state
    m = np.array([4, -1, 7, 11, -1, 2], dtype=np.int32)
    sites = np.zeros((6, 3, 3))
    exp_old, exp_new = np.zeros((6, 5)), np.zeros((6, 5))
    for c in range(6):
        A0 = P.amplitude()
        if kinds[c] == MGPU_MOVE:
            com, off = P.get_molecule(0, int(m[c]))
            sites[c] = P.apply_pbc(com + rng.uniform(-0.4, 0.4, 3))[None, :] + off @ P.rotation_matrix(1, 0.3).T
            P.save_fourier(0, int(m[c]))
            exp_old[c] = P.old_energy(0, int(m[c]), 0)[:5]
            P.set_molecule(0, int(m[c]), sites[c, 0], sites[c] - sites[c, 0][None, :])
            exp_new[c] = P.new_energy(0, int(m[c]), 0)[:5]
            P.set_molecule(0, int(m[c]), com, off)
            P.restore_fourier(0, int(m[c]))
        elif kinds[c] == MGPU_CREATION:
            sites[c] = (s.bounds_lo + rng.uniform(0.1, 0.9, 3) * 50.0)[None, :] + tmpl @ P.rotation_matrix(3, 1.1).T
            exp_old[c] = P.old_energy(0, n, 1)[:5]
            P.set_num_residues(0, n + 1)
            P.save_fourier(0, n)
            P.set_molecule(0, n, sites[c, 0], sites[c] - sites[c, 0][None, :])
            exp_new[c] = P.new_energy(0, n, 1)[:5]
            P.set_num_residues(0, n)
            P.set_amplitude(A0)
        else:
            P.all_fourier_terms()
            exp_old[c] = P.old_energy(0, int(m[c]), 2)[:5]
            P.save_fourier(0, int(m[c]))
            exp_new[c, 2] = P.recip_singlemol(0, int(m[c]), 2)      # intended physics: A - S_mol
            P.set_amplitude(A0)
        assert np.array_equal(P.amplitude(), A0)
    old, new = eng.gcmc_trial(rep, np.zeros(6, np.int32), m, kinds, sites, lane=1)
    close(old, exp_old, "mixed batch old")
    close(new, exp_new, "mixed batch new")
    accept = np.array([1, 1, 1, 0, 0, 0], dtype=np.int32)     # at most one commit per replica per call
    eng.commit_lane(1, rep, np.zeros(6, np.int32), m, kinds, accept)
    assert [eng.num_molecules(r, 0) for r in range(3)] == [18, 19, 17]
    assert np.array_equal(eng.get_molecules(0, 0)[4], sites[0])
    assert np.array_equal(eng.get_molecules(1, 0)[18], sites[1])
    for r in range(3):
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        amp_close(A, eng.structure_factor(r), f"replica {r} A after mixed commit")
    eng.close()
