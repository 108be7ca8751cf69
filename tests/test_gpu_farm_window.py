"""One launch per lane step of a farm of chains (mgpu_farm_window_submit / _wait, farm_window_kernel): energies, verdicts and
the committed state are held -- bit for bit -- to the batched device-built path (mgpu_move_trial_decide_submit: trial_build_kernel,
the pair and k sweeps, the rule behind the k sweep, the commit), itself held to the host-decided path and the oracle by
tests/test_gpu_farm.py, tests/test_gpu_gcmc.py and tests/test_gpu_parity.py.  Then the protocol around an UNDECIDED step
(margin, stall, forced resend), idle records, windows in flight, and the Fortran driver's window mode against its
batched mode.  Reference: src/monte_carlo.f90:40-86, src/monte_carlo_utils.f90:184-226, :275-395."""
import os

import numpy as np
import pytest

from maniac_mc_amd import synth
from maniac_mc_amd.engine import Engine

pytestmark = pytest.mark.gpu

V_REJ, V_ACC, V_UND, V_STALLED, V_IDLE = 0, 1, 2, 4, 5


def _twin(s, R, cap=None):
    """Two engines holding R copies of `s` with resident molecule frames.  MGPU_NO_FROZEN_BATCH: a framework box's batched
    path would otherwise sweep the framework with pair_frozen_kernel (the same sums in another order); the one-launch path
    uses the flat kernel's work units, and so does the batched path with the switch set."""
    out = []
    os.environ["MGPU_NO_FROZEN_BATCH"] = "1"
    try:
        for _ in range(2):
            e = Engine.from_system(s, n_replicas=R, mol_capacity=cap)
            e.load_system(s, 0)
            for t in range(s.topo.n_res):
                if s.topo.is_active[t]:
                    e.set_frames(0, t, s.com[t], s.offsets[t])
            e.init_structure_factor(0, True)
            for r in range(1, R):
                e.replica_copy(r, 0)
            out.append(e)
    finally:
        os.environ.pop("MGPU_NO_FROZEN_BATCH", None)
    return out


def _same_state(a, b, s, R):
    for r in range(R):
        for t in range(s.topo.n_res):
            assert a.num_molecules(r, t) == b.num_molecules(r, t), (r, t)
            assert np.array_equal(a.get_molecules(r, t), b.get_molecules(r, t)), (r, t)
            if s.topo.is_active[t]:
                ca, oa = a.get_frames(r, t)
                cb, ob = b.get_frames(r, t)
                assert np.array_equal(ca, cb) and np.array_equal(oa, ob), (r, t)
        assert np.array_equal(a.structure_factor(r), b.structure_factor(r)), r


def _nvt_records(rng, s, R, t_act):
    n_mol = int(s.n_mol[t_act])
    m = rng.integers(0, n_mol, R).astype(np.int32)
    move = rng.integers(1, 3, R).astype(np.int32)
    if int(s.topo.atoms_in_res[t_act]) == 1:
        move[:] = 1
    u = rng.uniform(0, 1, (R, 5))
    au = rng.uniform(0, 1, R)
    return m, move, u, au


@pytest.mark.parametrize("name,R", [("spce", 9), ("spce_many", 70), ("framework", 5), ("five_site", 4)])
def test_farm_window_is_the_batched_device_built_step(name, R):
    """NVT steps of R chains: the window's energies, verdicts and committed state (coordinates, frames, A(k)) are those of
    mgpu_move_trial_decide_submit, step after step; every second step rides in flight behind the one before it."""
    if name in ("spce", "spce_many"):
        s, t_act = synth.spce_box(6, seed=3), 0
    elif name == "framework":
        s, t_act = synth.framework_water_box(n_water=12, n_frame=300, L=24.0, seed=7), 1
    else:
        s, t_act = synth.five_site_water_box(), 0
    a, b = _twin(s, R)
    cap, depth = b.farm_window_capacity()
    assert cap >= R and depth >= 2
    rng = np.random.default_rng(5)
    rep = np.arange(R, dtype=np.int32)
    tt = np.full(R, t_act, np.int32)
    T = float(s.temperature)
    n_acc = 0
    for step in range(4):
        recs = [_nvt_records(rng, s, R, t_act) for _ in range(2)]
        # the window path: both steps queued before either is collected
        for m, move, u, au in recs:
            b.farm_window_submit(rep, tt, m, move, u, 0.4, 0.4, au, np.ones(R), T)
        for m, move, u, au in recs:
            o1, w1, acc = a.move_trial_decide(rep, tt, m, move, u, 0.4, 0.4, au, np.ones(R), T)
            a.synchronize()
            o2, w2, v = b.farm_window_wait(R)
            assert np.array_equal(o1, o2) and np.array_equal(w1, w2), (step, np.max(np.abs(o1 - o2)), np.max(np.abs(w1 - w2)))
            assert np.all((v == V_ACC) | (v == V_REJ)) and np.array_equal(v == V_ACC, acc != 0)
            n_acc += int(acc.sum())
        _same_state(a, b, s, R)
    assert 0 < n_acc <= 8 * R
    assert b.farm_window_stats() == (8, 0)
    a.close(); b.close()


def test_farm_window_gcmc_steps_are_the_batched_path():
    """Insertions, deletions and moves mixed in one window (CO2 box), with idle records: same energies (five components),
    verdicts, counts and state as the batched decide path -- which skips the idle chains."""
    s = synth.co2_box(20, seed=13)
    R = 12
    a, b = _twin(s, R, cap=[60])
    rng = np.random.default_rng(8)
    T = float(s.temperature)
    V = float(np.linalg.det(s.box_matrix))
    phi = 30.0 / V
    tt = np.zeros(R, np.int32)
    rep = np.arange(R, dtype=np.int32)
    seen = set()
    for step in range(14):
        n_now = np.array([b.num_molecules(r, 0) for r in range(R)])
        move = rng.integers(0, 5, R).astype(np.int32)
        move[(n_now <= 1) & (move == 4)] = 3
        m = np.array([rng.integers(0, n_now[r]) for r in range(R)], dtype=np.int32)
        u = rng.uniform(0, 1, (R, 5))
        au = rng.uniform(0, 1, R)
        pref = np.ones(R)
        pref[move == 3] = phi * V / (n_now[move == 3] + 1.0)
        pref[move == 4] = ((n_now[move == 4] - 1.0) + 1.0) / (phi * V)
        b.farm_window_submit(rep, tt, m, move, u, 1.0, 0.6, au, pref, T)
        o2, w2, v = b.farm_window_wait(R)
        live = move != 0
        assert np.all(v[~live] == V_IDLE) and not np.any(o2[~live]) and not np.any(w2[~live])
        o1, w1, acc = a.move_trial_decide(rep[live], tt[live], m[live], move[live], u[live], 1.0, 0.6, au[live], pref[live], T)
        a.synchronize()
        assert np.array_equal(o1, o2[live]) and np.array_equal(w1, w2[live]), step
        assert np.array_equal(v[live] == V_ACC, acc != 0) and np.all((v[live] == V_ACC) | (v[live] == V_REJ))
        seen.update((int(mv), int(vv)) for mv, vv in zip(move[live], v[live]))
        _same_state(a, b, s, R)
    assert {(3, V_ACC), (4, V_ACC)} <= seen and any(mv <= 2 and vv == V_ACC for mv, vv in seen)
    a.close(); b.close()


def test_by_count_records_follow_the_count_the_launch_sees():
    """Insertion / deletion windows IN FLIGHT: the driver hands over the residue type, the kind of move, the draw of
    PickRandomMoleculeIndex and phi V; the launch completes each record from the replica's molecule count as it is when it
    runs (`slot_u`).  Two windows are queued before either is collected -- the second one's slots and prefactors depend on
    what the first one accepted -- and held to the batched decide path run step by step with the slot and prefactor
    computed here from the twin's counts: energies, verdicts, counts, state.  Steps with nothing to do (a deletion from an
    empty type, an insertion into a full one) come back idle."""
    s = synth.co2_box(3, seed=13)
    R = 10
    cap = 5
    a, b = _twin(s, R, cap=[cap])
    rng = np.random.default_rng(21)
    T = float(s.temperature)
    V = float(np.linalg.det(s.box_matrix))
    phiV = 4.0
    tt = np.zeros(R, np.int32)
    rep = np.arange(R, dtype=np.int32)
    seen = set()
    for rnd in range(12):
        recs = []
        for _ in range(2):
            move = rng.integers(1, 5, R).astype(np.int32)
            move[rng.random(R) < 0.55] = rng.choice([3, 4])                 # mostly insertions / deletions: the counts hit 0 and cap
            u = rng.uniform(0, 1, (R, 5)); au = rng.uniform(0, 1, R); su = rng.uniform(0, 1, R)
            pv = np.where(move >= 3, phiV, 1.0)
            recs.append((move, u, au, su, pv))
            b.farm_window_submit(rep, tt, np.zeros(R, np.int32), move, u, 1.0, 0.6, au, pv, T, slot_u=su)
        for move, u, au, su, pv in recs:
            o2, w2, v = b.farm_window_wait(R)
            n_now = np.array([a.num_molecules(r, 0) for r in range(R)])
            # what the launch must have made of the records, from the twin's counts (mc_farm.f90 select_move)
            live = np.where(move == 3, n_now < cap, n_now > 0)
            m = np.minimum((su * n_now).astype(np.int32), np.maximum(n_now - 1, 0)).astype(np.int32)
            pref = np.ones(R)
            pref[move == 3] = phiV / (n_now[move == 3] + 1.0)
            pref[move == 4] = ((n_now[move == 4] - 1.0) + 1.0) / phiV
            assert np.all(v[~live] == V_IDLE) and not np.any(o2[~live]) and not np.any(w2[~live])
            if live.any():
                o1, w1, acc = a.move_trial_decide(rep[live], tt[live], m[live], move[live], u[live], 1.0, 0.6, au[live], pref[live], T)
                a.synchronize()
                assert np.array_equal(o1, o2[live]) and np.array_equal(w1, w2[live]), rnd
                assert np.array_equal(v[live] == V_ACC, acc != 0) and np.all((v[live] == V_ACC) | (v[live] == V_REJ))
            seen.update((int(mv), int(vv), int(nn)) for mv, vv, nn in zip(move, v, n_now))
        _same_state(a, b, s, R)                                # (both windows have run on the window engine)
    assert any(mv == 3 and vv == V_ACC for mv, vv, _ in seen) and any(mv == 4 and vv == V_ACC for mv, vv, _ in seen)
    assert any(mv == 4 and vv == V_IDLE and nn == 0 for mv, vv, nn in seen) or any(mv == 3 and vv == V_IDLE and nn == cap for mv, vv, nn in seen)
    a.close(); b.close()


def test_undecided_steps_stall_the_chain_until_the_host_decides():
    """With the margin wide open every step is UNDECIDED: nothing is committed, the replica is marked, a window already in
    flight for it does nothing (verdict 4), and the step sent again with the host's decision (`forced`) is obeyed --
    after which the state is the batched path's for the same decisions.  One chain is left alone (margin back to 16 ulp
    for it would need a second engine: instead it carries an idle record)."""
    s = synth.spce_box(5, seed=4)
    R = 6
    a, b = _twin(s, R)
    rng = np.random.default_rng(2)
    rep = np.arange(R, dtype=np.int32)
    tt = np.zeros(R, np.int32)
    T = float(s.temperature)
    b.chain_set_margin(1e9)
    m, move, u, au = _nvt_records(rng, s, R, 0)
    m2, move2, u2, au2 = _nvt_records(rng, s, R, 0)
    move[R - 1] = 0                                           # an idle chain is not stalled by anything
    move2[R - 1] = 0
    b.farm_window_submit(rep, tt, m, move, u, 0.4, 0.4, au, np.ones(R), T)
    b.farm_window_submit(rep, tt, m2, move2, u2, 0.4, 0.4, au2, np.ones(R), T)          # in flight behind the undecided one
    o, w, v = b.farm_window_wait(R)
    assert np.all(v[:R - 1] == V_UND) and v[R - 1] == V_IDLE
    o_skip, w_skip, v2 = b.farm_window_wait(R)
    assert np.all(v2[:R - 1] == V_STALLED) and v2[R - 1] == V_IDLE
    # the host's decision (its own exp) for step 1; the batched path gives the energies to decide from -- the same ones
    live = move != 0
    o1, w1 = a.move_trial(rep[live], tt[live], m[live], move[live], u[live], 0.4, 0.4)
    assert np.array_equal(o1, o[live]) and np.array_equal(w1, w[live])
    yes = au[live] <= np.minimum(1.0, np.exp(-(w1.sum(1) - o1.sum(1)) / T))
    a.commit_lane(0, rep[live], tt[live], m[live], np.zeros(int(live.sum()), np.int32), yes.astype(np.int32))
    forced = np.zeros(R, np.int32)
    forced[live] = np.where(yes, 1, 2)
    b.farm_window_submit(rep, tt, m, move, u, 0.4, 0.4, au, np.ones(R), T, forced=forced)
    o3, w3, v3 = b.farm_window_wait(R)
    assert np.array_equal(o3[live], o1) and np.array_equal(w3[live], w1)
    assert np.array_equal(v3[live] == V_ACC, yes) and np.all((v3[live] == V_ACC) | (v3[live] == V_REJ))
    _same_state(a, b, s, R)
    assert b.farm_window_stats()[1] == R - 1
    # and with the margin back the chains run on
    b.chain_set_margin(16 * np.finfo(float).eps)
    b.farm_window_submit(rep, tt, m2, move2, u2, 0.4, 0.4, au2, np.ones(R), T)
    _, _, v5 = b.farm_window_wait(R)
    assert np.all((v5[:R - 1] == V_ACC) | (v5[:R - 1] == V_REJ)) and v5[R - 1] == V_IDLE
    a.close(); b.close()


def test_farm_window_refusals():
    s = synth.spce_box(5, seed=4)
    from maniac_mc_amd import _lib
    e = Engine.from_system(s, n_replicas=3)
    e.init_structure_factor(0, True)
    ones = np.ones(2)
    with pytest.raises(_lib.MgpuError):                       # no molecule frames
        e.farm_window_submit([0, 1], [0, 0], [1, 2], [1, 1], np.zeros((2, 5)), 0.3, 0.3, ones, ones, 300.0)
    for r in range(3):
        if r:
            e.replica_copy(r, 0)
        e.set_frames(r, 0, s.com[0], s.offsets[0])
    with pytest.raises(_lib.MgpuError):                       # two records for one replica
        e.farm_window_submit([1, 1], [0, 0], [1, 2], [1, 1], np.zeros((2, 5)), 0.3, 0.3, ones, ones, 300.0)
    with pytest.raises(_lib.MgpuError):                       # nothing in flight
        e.farm_window_wait(2)
    e.farm_window_submit([0, 1], [0, 0], [1, 2], [1, 2], np.full((2, 5), 0.5), 0.3, 0.3, ones * 0.5, ones, 300.0)
    e.system_energy(0)                                        # a synchronous entry point in between: results stay collectable
    o, w, v = e.farm_window_wait(2)
    assert np.all((v == V_ACC) | (v == V_REJ)) and np.all(o[:, 2] != 0)
    e.close()


@pytest.mark.parametrize("case", ["spce_nvt", "spce_nvt_64", "mixture_nvt", "co2_gcmc", "framework_water_gcmc", "co2_gcmc_drivers",
                                  "mixture_gcmc", "spce_nvt_drivers"])
def test_window_farm_is_the_batched_farm(case):
    """mfarm_configure(3): mc_farm.f90 sends ONE launch per lane step (mgpu_farm_window_submit) and follows the outcomes,
    checking every device decision against its own rule.  Same seeds, same chains: counters, counts, running energies,
    coordinates and A(k) must be those of the farm that evaluates through mgpu_move_trial_submit, decides in Fortran and
    commits with mgpu_commit_submit -- bit for bit -- whatever runs beside a chain (64 chains on two lanes, windows in
    flight) and for insertions / deletions."""
    from maniac_mc_amd.fortran_host import FortranFarm
    from tests.util import farm_tol
    kw = dict(seed=23, n_threads=2, n_lanes=2, device_build=True)
    env = {}
    if case == "spce_nvt":
        s, R, steps = synth.spce_box(6, seed=3), 9, 60
        kw.update(translation_step=0.4, rotation_step=0.4, n_lanes=3)
    elif case == "spce_nvt_64":
        s, R, steps = synth.spce_box(6, seed=3), 64, 25
        kw.update(translation_step=0.4, rotation_step=0.4)
    elif case == "mixture_nvt":
        s, R, steps = synth.mixture_box(seed=4), 6, 60
        kw.update(translation_step=0.4, rotation_step=0.4)
    elif case == "mixture_gcmc":                  # two active residue types, each with its own count, capacity and fugacity
        s, R, steps = synth.mixture_box(seed=4), 10, 120
        kw.update(translation_step=0.4, rotation_step=0.4, mol_capacity=[16, 11],
                  gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=np.array([14.0, 8.0]) / (18.0 * 21.0 * 24.0)))
    elif case == "spce_nvt_drivers":              # four lanes, two driver threads
        s, R, steps = synth.spce_box(6, seed=3), 22, 40
        kw.update(translation_step=0.4, rotation_step=0.4, n_lanes=4, n_threads=2, n_drivers=2)
    elif case in ("co2_gcmc", "co2_gcmc_drivers"):
        s, R, steps = synth.co2_box(20, seed=13), 12, 150
        kw.update(translation_step=1.0, rotation_step=0.6, mol_capacity=[90],
                  gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=np.repeat([10.0, 30.0], 6) / 50.0 ** 3))
        if case == "co2_gcmc_drivers":            # three lanes, each driven by a host thread of its own
            kw.update(n_lanes=3, n_threads=3, n_drivers=3)
    else:
        s, R, steps = synth.framework_water_box(n_water=12, n_frame=300, L=24.0, seed=7), 8, 120
        kw.update(translation_step=0.5, rotation_step=0.5, mol_capacity=[1, 60],
                  gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=20.0 / 24.0 ** 3))
        env = {"MGPU_NO_FROZEN_BATCH": "1"}       # the batched path on the flat kernel's work units, as the window's
    os.environ.update(env)
    try:
        farms = [FortranFarm(s, R, window=w, window_depth=3, **kw) for w in (False, True)]
    finally:
        for k in env:
            os.environ.pop(k, None)
    assert farms[1].window and not farms[0].window
    for f in farms:
        f.run(steps)
    a, b = farms
    assert a.trials == b.trials and a.accepted == b.accepted and a.skipped == b.skipped and a.accepted > 0
    assert a.counters() == b.counters()
    assert np.array_equal(a.counts(), b.counts())
    for r in range(R):
        assert np.array_equal(a.energy(r), b.energy(r)), r
        assert np.array_equal(a.eng.structure_factor(r), b.eng.structure_factor(r)), r
        for t in a.active:
            assert a.eng.num_molecules(r, int(t)) == b.eng.num_molecules(r, int(t))
            assert np.array_equal(a.eng.get_molecules(r, int(t)), b.eng.get_molecules(r, int(t)))
    # the window farm is consistent with a from-scratch evaluation, and a second run continues
    for r in (0, R - 1):
        e = b.eng.system_energy(r)
        ref = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")])
        assert np.max(np.abs(b.energy(r) - ref)) < farm_tol(ref, steps), (r, b.energy(r) - ref)
    for f in farms:
        f.run(7)
    assert a.counters() == b.counters() and np.array_equal(a.energy(R - 1), b.energy(R - 1))
    assert b.window_mode()[2] == 0
    for f in farms:
        f.close()


@pytest.mark.parametrize("gcmc", [False, True], ids=["spce_nvt", "co2_gcmc"])
def test_window_farm_with_every_step_left_to_the_driver(gcmc):
    """The margin wide open: the device leaves EVERY step undecided, the Fortran driver decides each one with its own exp
    and sends it again, and the windows in flight behind an undecided step come back untouched and are sent again in
    order.  The trajectory is still the batched farm's, bit for bit -- also for an insertion / deletion farm, whose records
    the launches complete from the count they see."""
    from maniac_mc_amd.fortran_host import FortranFarm
    kw = dict(seed=5, n_threads=2, n_lanes=2, device_build=True, translation_step=0.4, rotation_step=0.4)
    if gcmc:
        s, R, steps = synth.co2_box(6, seed=13), 6, 60
        kw.update(translation_step=1.0, rotation_step=0.6, mol_capacity=[12],
                  gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=8.0 / 50.0 ** 3))
    else:
        s, R, steps = synth.spce_box(5, seed=9), 7, 30
    a = FortranFarm(s, R, **kw)
    b = FortranFarm(s, R, window=True, window_depth=3, **kw)
    b.eng.chain_set_margin(1e9)
    a.run(steps); b.run(steps)
    on, depth, undecided = b.window_mode()
    # (an insertion onto an atom: exp(-dE/T) underflows to 0 -- the one outcome no margin leaves open)
    assert (on, depth) == (True, 3) and b.trials - (3 if gcmc else 0) <= undecided <= b.trials
    assert b.trials == a.trials and b.skipped == a.skipped
    assert np.array_equal(a.counts(), b.counts())
    assert a.counters() == b.counters() and a.accepted == b.accepted > 0
    for r in range(R):
        assert np.array_equal(a.energy(r), b.energy(r)), r
        assert np.array_equal(a.eng.structure_factor(r), b.eng.structure_factor(r)), r
        assert np.array_equal(a.eng.get_molecules(r, 0), b.eng.get_molecules(r, 0))
    a.close(); b.close()


def test_randomised_window_farms_are_the_batched_farms():
    """tools/window_farm_stress.py: 30 random combinations of box (SPC/E, CO2 insertion / deletion, two active types with and without
    insertion / deletion, framework + water), chain count, lanes, driver threads, windows in flight, undecided margin and run length;
    every window farm must leave counters, counts, energies, coordinates and A(k) of the batched farm, bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "window_farm_stress.py"), "--cases", "30", "--seed", "3"],
                       capture_output=True, text=True, cwd=root, timeout=900)
    assert p.returncode == 0 and "30 cases, 0 different" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
