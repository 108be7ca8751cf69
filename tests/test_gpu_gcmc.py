"""Mixed grand-canonical batches on the GPU (mgpu_gcmc_trial_submit / _wait / mgpu_commit_submit)."""
import ctypes as C

import numpy as np
import pytest

from maniac_mc_amd import synth
from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE
from maniac_mc_amd._lib import check
from maniac_mc_amd.engine import Engine
from tests.test_gpu_parity import amp_close, close

pytestmark = pytest.mark.gpu


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


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


def test_device_built_trials_match_host_built_rows():
    """mgpu_move_trial_submit builds the trial geometry on the device from the resident molecule frames and the host's
    uniform numbers (Translation / Rotation / CreateMolecule of the reference).  Here the same moves are constructed in
    numpy from the same numbers and handed over as explicit rows (mgpu_gcmc_trial_submit): energies agree to the parity
    bar (the rotated offsets may differ in the last bit: device sincos against numpy's); after committing a few of the
    device-built candidates from the lane's resident rows, sites == com + offsets of the frames read back, the counts
    follow and A(k) equals a fresh S(k)."""
    s = synth.co2_box(30, seed=3)
    L = np.diag(s.box_matrix)
    lo = s.bounds_lo
    eng = Engine(s.topo, s.box_matrix, s.bounds_lo, s.real_space_cutoff, s.ewald_tolerance, 2, 0, [48])
    eng.set_frames(0, 0, s.com[0], s.offsets[0])
    eng.init_structure_factor(0, True)
    eng.replica_copy(1, 0)
    com0, off0 = eng.get_frames(1, 0)
    assert np.array_equal(com0, s.com[0]) and np.array_equal(off0, s.offsets[0])
    assert np.array_equal(eng.get_molecules(1, 0), s.all_sites(0))
    rng = np.random.default_rng(12)
    n = 24
    rep = ((np.arange(n) // 4) % 2).astype(np.int32)        # every move code on both replicas
    t = np.zeros(n, np.int32)
    move = np.array([1, 2, 3, 4] * 6, dtype=np.int32)
    m = rng.integers(0, 30, n).astype(np.int32)
    u = rng.random((n, 5))
    t_step, r_step = 1.0, 0.6
    rows = np.zeros((n, 3, 3))
    kinds = np.zeros(n, np.int32)
    for c in range(n):
        com, off = s.com[0][m[c]].copy(), s.offsets[0][m[c]].copy()
        if move[c] == 1:
            x = (com + (u[c, :3] - 0.5) * t_step) - lo
            x = np.where((x < 0) | (x >= L), np.mod(x, L), x)
            com = lo + x
            kinds[c] = MGPU_MOVE
        elif move[c] in (2, 3):
            if move[c] == 3:
                com, off = lo + L * u[c, :3], s.offsets[0][0].copy()
            theta = (u[c, 3] - 0.5) * r_step if move[c] == 2 else u[c, 3] * 2 * np.pi
            axis = int(u[c, 4] * 3.0) + 1
            p, q = axis % 3, (axis + 1) % 3
            x, y = off[:, p].copy(), off[:, q].copy()
            off[:, p] = np.cos(theta) * x - np.sin(theta) * y
            off[:, q] = np.sin(theta) * x + np.cos(theta) * y
            kinds[c] = MGPU_MOVE if move[c] == 2 else MGPU_CREATION
        else:
            kinds[c] = MGPU_DELETION
        rows[c] = com[None, :] + off
    old_h, new_h = eng.gcmc_trial(rep, t, m, kinds, rows, lane=1)
    old_d, new_d = eng.move_trial(rep, t, m, move, u, t_step, r_step, lane=0)
    close(old_d, old_h, "device-built old")
    close(new_d, new_h, "device-built new")
    # commit: one move on replica 0, one insertion on replica 1 (resident rows of lane 0)
    acc = np.zeros(n, np.int32)
    c_mv = int(np.flatnonzero((move == 1) & (rep == 0))[0])
    c_cr = int(np.flatnonzero((move == 3) & (rep == 1))[0])
    acc[c_mv] = 1
    acc[c_cr] = 1
    eng.commit_lane(0, rep, t, m, kinds, acc)
    assert [eng.num_molecules(r, 0) for r in range(2)] == [30, 31]
    for r in range(2):
        com, off = eng.get_frames(r, 0)
        assert np.array_equal(eng.get_molecules(r, 0), com[:, None, :] + off)
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        amp_close(A, eng.structure_factor(r), f"replica {r} A after a device-built commit")
    assert np.max(np.abs(eng.get_molecules(0, 0)[m[c_mv]] - rows[c_mv])) < 1e-12
    assert np.max(np.abs(eng.get_molecules(1, 0)[30] - rows[c_cr])) < 1e-12
    # a deletion on replica 0 moves the last molecule's frame with its sites
    c_de = int(np.flatnonzero((move == 4) & (rep == 0))[0])
    old2, new2 = eng.move_trial([0], [0], [m[c_de]], [4], np.zeros((1, 5)), t_step, r_step)
    eng.commit_lane(0, [0], [0], [m[c_de]], [MGPU_DELETION], [1])
    com, off = eng.get_frames(0, 0)
    assert com.shape[0] == 29 and np.array_equal(eng.get_molecules(0, 0), com[:, None, :] + off)
    # bare-site commits would leave the frames behind: refused while the replica holds frames
    with pytest.raises(Exception):
        eng.commit_candidates([0], [0], [1], [MGPU_MOVE], rows[:1], [1])
    eng.close()


def test_device_built_coordinates_are_the_references_moves(refcpu_mod):
    """The COORDINATES trial_build_kernel writes (the bench's default move construction), read back after a commit, against
    the oracle's ApplyPBC and RotationMatrix -- both pinned bit for bit to the reference (tests/test_oracle_pin.py) --
    composed as Translation / Rotation / CreateMolecule compose them (src/translation.f90:93-112,
    src/monte_carlo_utils.f90:30-66, src/helper_utils.f90:39-77, src/create_molecule.f90:166-207): translations that leave
    the cell on either side, rotations about each Cartesian axis, insertions rotated about each axis.  <= 1e-12 A (the
    rotated offsets may differ in the last bit: device sincos against libm's)."""
    s = synth.co2_box(30, seed=3)
    P = refcpu_mod.RefCPU(s, mol_capacity=48)
    L = np.diag(s.box_matrix)
    lo = s.bounds_lo
    moves = [1] * 6 + [2] * 6 + [3] * 6
    n = len(moves)
    eng = Engine(s.topo, s.box_matrix, s.bounds_lo, s.real_space_cutoff, s.ewald_tolerance, n, 0, [48])
    eng.set_frames(0, 0, s.com[0], s.offsets[0])
    eng.init_structure_factor(0, True)
    for r in range(1, n):
        eng.replica_copy(r, 0)
    rng = np.random.default_rng(41)
    u = rng.random((n, 5))
    u[6:12, 4] = u[12:18, 4] = [0.05, 0.3, 0.4, 0.6, 0.7, 0.99]          # axes 1, 1, 2, 2, 3, 3
    # translations: molecules nearest to the cell faces, pushed outwards by a step larger than their distance to the face
    t_step, r_step = 6.0, 0.6
    d_lo = np.min(s.com[0] - lo, axis=1)
    d_hi = np.min(lo + L - s.com[0], axis=1)
    near = np.concatenate([np.argsort(d_lo)[:3], np.argsort(d_hi)[:3]])
    m = rng.integers(0, 30, n).astype(np.int32)
    m[:6] = near
    for c in range(3):
        u[c, :3] = 0.02              # towards -x, -y, -z by 2.9 A
        u[3 + c, :3] = 0.98
    rep = np.arange(n, dtype=np.int32)
    t = np.zeros(n, np.int32)
    move = np.array(moves, dtype=np.int32)
    kinds = np.where(move <= 2, MGPU_MOVE, MGPU_CREATION).astype(np.int32)
    eng.move_trial(rep, t, m, move, u, t_step, r_step, lane=0)
    eng.commit_lane(0, rep, t, m, kinds, np.ones(n, np.int32))
    crossed = 0
    for c in range(n):
        com0, off0 = s.com[0][m[c]].copy(), s.offsets[0][m[c]].copy()
        slot = int(m[c])
        if move[c] == 1:
            raw = com0 + (u[c, :3] - 0.5) * t_step
            crossed += int(np.any((raw < lo) | (raw >= lo + L)))
            com_e, off_e = P.apply_pbc(raw), off0
        elif move[c] == 2:
            com_e = com0
            off_e = off0 @ P.rotation_matrix(int(u[c, 4] * 3.0) + 1, (u[c, 3] - 0.5) * r_step).T
        else:
            slot = 30
            com_e = lo + L * u[c, :3]
            off_e = s.offsets[0][0] @ P.rotation_matrix(int(u[c, 4] * 3.0) + 1, u[c, 3] * 2 * np.pi).T
        com_d, off_d = eng.get_frames(c, 0)
        assert np.max(np.abs(com_d[slot] - com_e)) <= 1e-12, (c, move[c], com_d[slot], com_e)
        assert np.max(np.abs(off_d[slot] - off_e)) <= 1e-12, (c, move[c])
        assert np.array_equal(eng.get_molecules(c, 0)[slot], com_d[slot][None, :] + off_d[slot])       # sites = com + off, formed once
        if move[c] != 3:
            assert np.array_equal(com_d[np.arange(30) != slot], s.com[0][np.arange(30) != slot])
    assert crossed >= 4                       # the wrap of ApplyPBC was really exercised
    eng.close()


def _decide_host(old, new, u, pref, T):
    """mc_farm.f90 resolve_and_commit: totals in component order, min(1, pref exp(-dE / T)), accept iff u <= p"""
    e_old = np.zeros(len(u)); e_new = np.zeros(len(u))
    for k in range(5):
        e_old = e_old + old[:, k]
        e_new = e_new + new[:, k]
    with np.errstate(over="ignore", invalid="ignore"):
        x = pref * np.exp(-(e_new - e_old) / T)
    return ((x >= 1.0) | (u <= x)).astype(np.int32)


@pytest.mark.parametrize("system,built", [("co2", False), ("co2", True), ("framework_water", False), ("framework_water", True),
                                          ("mixture_triclinic", False)],
                         ids=["co2-host_rows", "co2-device_built", "framework_water-host_rows", "framework_water-device_built",
                              "mixture_triclinic-host_rows"])
def test_device_decided_trials_commit_the_same_state(system, built):
    """mgpu_*_trial_decide_submit: the k sweep's workgroup applies the acceptance rule and commits.  Two engines run the
    same scripted grand-canonical steps, one candidate per replica and step: engine A the classic way (trial, decision
    on the host, mgpu_commit_submit), engine B with the decision on the device.  Energies, flags, counts, coordinates,
    frames and A(k) must be bitwise equal after every step."""
    if system == "co2":
        s = synth.co2_box(20, seed=4)
        ta, caps = 0, [40]
    elif system == "mixture_triclinic":
        # two active species in a tilted cell: the 27-image pair sweep, single-state items, the general reciprocal matrix
        s = synth.mixture_box(seed=4, tilt=(1.5, -0.8, 0.6))
        ta, caps = 0, [30, 20]
    else:
        s = synth.framework_water_box(n_water=12, n_frame=300, L=24.0, seed=7)
        ta, caps = 1, [1, 30]
    R = 7
    T = 300.0
    engines = []
    for _ in range(2):
        e = Engine(s.topo, s.box_matrix, s.bounds_lo, s.real_space_cutoff, s.ewald_tolerance, R, 0, caps)
        for tt in range(s.topo.n_res):
            if built and tt == ta:
                e.set_frames(0, tt, s.com[tt], s.offsets[tt])
            else:
                e.set_molecules(0, tt, s.all_sites(tt))
        e.init_structure_factor(0, True)
        for r in range(1, R):
            e.replica_copy(r, 0)
        engines.append(e)
    A_, B_ = engines
    rng = np.random.default_rng(21)
    L = np.diag(s.box_matrix)
    V = float(abs(np.linalg.det(s.box_matrix)))
    n1 = int(s.topo.atoms_in_res[ta])
    tmpl = s.all_sites(ta)[0] - s.all_sites(ta)[0].mean(axis=0)
    rep = np.arange(R, dtype=np.int32)
    t = np.full(R, ta, np.int32)
    n_acc = 0
    for step in range(14):
        nm = np.array([A_.num_molecules(r, ta) for r in range(R)])
        assert np.array_equal(nm, [B_.num_molecules(r, ta) for r in range(R)])
        move = rng.integers(1, 5, R).astype(np.int32)
        move[nm <= 2] = 3
        move[nm >= caps[ta] - 1] = 4
        m = (rng.random(R) * nm).astype(np.int32)
        kinds = np.where(move <= 2, MGPU_MOVE, np.where(move == 3, MGPU_CREATION, MGPU_DELETION)).astype(np.int32)
        u5 = rng.random((R, 5))
        au = rng.random(R)
        phi = 30.0 / V
        pref = np.where(move <= 2, 1.0, np.where(move == 3, phi * V / (nm + 1.0), nm / (phi * V)))
        pref = pref * np.exp(rng.normal(0.0, 2.0, R))            # spread the outcomes: both answers occur
        if built:
            oa, na = A_.move_trial(rep, t, m, move, u5, 0.8, 0.7)
            acc_h = _decide_host(oa, na, au, pref, T)
            A_.commit_lane(0, rep, t, m, kinds, acc_h)
            ob, nb, acc_d = B_.move_trial_decide(rep, t, m, move, u5, 0.8, 0.7, au, pref, T)
        else:
            cur = [A_.get_molecules(r, ta) for r in range(R)]
            rows = np.zeros((R, n1, 3))
            for r in range(R):
                if move[r] == 3:
                    rows[r] = tmpl + (s.bounds_lo + u5[r, :3] @ s.box_matrix)      # create_molecule.f90:183-184 (rows = cell vectors)
                else:
                    rows[r] = cur[r][m[r]] + (u5[r, :3] - 0.5) * (0.8 if move[r] <= 2 else 0.0)
            oa, na = A_.gcmc_trial(rep, t, m, kinds, rows)
            acc_h = _decide_host(oa, na, au, pref, T)
            A_.commit_lane(0, rep, t, m, kinds, acc_h)
            ob, nb, acc_d = B_.gcmc_trial_decide(rep, t, m, kinds, rows, au, pref, T)
        assert np.array_equal(oa, ob) and np.array_equal(na, nb), step
        assert np.array_equal(acc_h, acc_d), (step, acc_h, acc_d)
        n_acc += int(acc_d.sum())
        for r in range(R):
            assert A_.num_molecules(r, ta) == B_.num_molecules(r, ta)
    assert 0.2 * 14 * R < n_acc < 0.8 * 14 * R
    for r in range(R):
        assert np.array_equal(A_.get_molecules(r, ta), B_.get_molecules(r, ta))
        assert np.array_equal(A_.structure_factor(r), B_.structure_factor(r))
        if built:
            ca, fa = A_.get_frames(r, ta)
            cb, fb = B_.get_frames(r, ta)
            assert np.array_equal(ca, cb) and np.array_equal(fa, fb)
    # a drain between submit and wait folds the outcomes into the engine's counts; the wait still returns the flags
    if built:
        nm0 = B_.num_molecules(0, ta)
        au1 = np.array([0.5]); m1 = np.array([0], np.int32)
        check(B_.L.mgpu_move_trial_decide_submit(B_.h, C.c_int(0), C.c_int(1), _ip(np.array([0], np.int32)), _ip(np.array([ta], np.int32)),
                                                 _ip(m1), _ip(np.array([3], np.int32)), _dp(np.array([[0.3, 0.6, 0.2, 0.1, 0.7]])),
                                                 C.c_double(0.8), C.c_double(0.7), _dp(au1), _dp(np.array([1e300])), C.c_double(T)))
        B_.synchronize()
        assert B_.num_molecules(0, ta) == nm0 + 1
        o1 = np.zeros((1, 5)); n1_ = np.zeros((1, 5)); a1 = np.zeros(1, np.int32)
        check(B_.L.mgpu_trial_decide_wait(B_.h, C.c_int(0), _dp(o1), _dp(n1_), _ip(a1)))
        assert a1[0] == 1
    # the device has committed: nothing is left to commit "from the lane's resident rows"
    with pytest.raises(Exception):
        B_.commit_lane(0, rep, t, m, kinds, np.ones(R, np.int32))
    # a second candidate on a replica is refused (the workgroups commit independently)
    with pytest.raises(Exception):
        if built:
            B_.move_trial_decide([0, 0], [ta, ta], [0, 1], [1, 1], np.zeros((2, 5)), 0.8, 0.7, [0.5, 0.5], [1.0, 1.0], T)
        else:
            B_.gcmc_trial_decide([0, 0], [ta, ta], [0, 1], [MGPU_MOVE] * 2, np.zeros((2, n1, 3)), [0.5, 0.5], [1.0, 1.0], T)
    for e in engines:
        e.close()
