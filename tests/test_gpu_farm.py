"""The Fortran host Metropolis driver (maniac_mc_amd/fortran/mc_farm.f90) on the GPU: after a run
of overlapped batched steps every chain's running energy must equal a from-scratch evaluation of
its final configuration, A(k) must equal a fresh S(k), and the host mirrors must equal the device."""
import numpy as np
import pytest

from maniac_mc_amd import synth
from tests.util import TOL_K, farm_tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("maker,R,steps", [(lambda: synth.spce_box(6, seed=3), 7, 40), (lambda: synth.co2_box(24, seed=5), 4, 60),
                                           (lambda: synth.mixture_box(seed=4), 5, 50),
                                           (lambda: synth.mixture_box(seed=4, tilt=(1.5, -0.8, 0.6)), 4, 50),
                                           (lambda: synth.rigid_adsorbate_box(), 3, 30)])
@pytest.mark.parametrize("device_build", [False, True], ids=["host_built", "device_built"])
def test_fortran_farm_consistency(maker, R, steps, device_build):
    """device_built: the engine keeps the molecules' frames and builds the trial moves itself (mgpu_move_trial_submit);
    the Fortran driver then has no mirror, and farm.molecule() reads the frames back from the device (the triclinic
    case falls back to the host construction)."""
    from maniac_mc_amd.fortran_host import FortranFarm
    s = maker()
    farm = FortranFarm(s, R, seed=11, translation_step=0.4, rotation_step=0.4, device_build=device_build)
    assert farm.device_build == (device_build and not s.is_triclinic())
    acc = farm.run(steps)
    assert farm.trials + farm.skipped == R * steps and 0 < acc <= farm.trials
    assert acc == farm.accepted
    eng = farm.eng
    for r in range(R):
        e = eng.system_energy(r)
        run = farm.energy(r)
        # the running sums accumulate `steps` increments of O(1e5) K terms: allow their rounding
        tol = TOL_K + 64 * np.finfo(float).eps * max(abs(e["recip_coulomb"]), abs(e["coulomb"])) * np.sqrt(steps)
        assert abs(run[0] - e["non_coulomb"]) < tol and abs(run[1] - e["coulomb"]) < tol and abs(run[2] - e["recip_coulomb"]) < tol
        assert run[3] == e["ewald_self"] and abs(run[4] - e["intra_coulomb"]) <= 1e-6      # untouched by NVT moves
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        assert np.max(np.abs(A - eng.structure_factor(r))) < 1e-10
    # host mirrors == device coordinates
    for ia, t in enumerate(farm.active):
        dev = eng.get_molecules(R - 1, int(t))
        for slot in (0, dev.shape[0] - 1):
            com, off = farm.molecule(R - 1, ia, slot)
            n1 = dev.shape[1]
            assert np.array_equal(dev[slot], com[None, :] + off[:n1])
    # replicas diverged (independent chains) and a second run continues
    assert not np.array_equal(eng.get_molecules(0, int(farm.active[0])), eng.get_molecules(R - 1, int(farm.active[0])))
    acc2 = farm.run(5)
    assert farm.trials + farm.skipped == R * (steps + 5) and farm.accepted == acc + acc2
    farm.close()


def test_fortran_farm_matches_python_farm_statistically():
    """Different RNG streams, same physics: acceptance ratios of the two drivers agree."""
    from maniac_mc_amd.farm import ReplicaFarm
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.spce_box(6, seed=3)
    f1 = FortranFarm(s, 64, seed=5)
    a1 = f1.run(30) / (64 * 30)
    f1.close()
    f2 = ReplicaFarm(s, 64, seed=6)
    a2 = f2.run(30) / (64 * 30)
    f2.close()
    assert abs(a1 - a2) < 0.06, (a1, a2)


def test_fortran_farm_intrinsic_rng_and_single_replica():
    """rng_kind 0 = the reference's random_number; one replica = one lane idle."""
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.mixture_box(seed=4)
    farm = FortranFarm(s, 1, seed=3, rng_kind=0, n_threads=1)
    farm.run(25)
    assert farm.trials + farm.skipped == 25
    e = farm.eng.system_energy(0)
    run = farm.energy(0)
    assert abs(run[0] - e["non_coulomb"]) < 1e-6 and abs(run[1] - e["coulomb"]) < 1e-6
    farm.close()


@pytest.mark.parametrize("device_build", [False, True], ids=["host_built", "device_built"])
def test_gcmc_farm_consistency_and_ideal_gas_limit(device_build):
    """Grand-canonical chains (BASELINE.json configs[2]/[4] in miniature): insertion / deletion /
    translation / rotation of rigid CO2 in a 50 A box at several fugacities, one per replica group.
    (1) bookkeeping: counts, running 5-component energies, A(k) and host mirrors equal a from-scratch
    evaluation; (2) physics: at this density CO2 is nearly ideal, so <N> ~ fugacity * V."""
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.co2_box(20, seed=13)
    R = 24
    V = 50.0 ** 3
    targets = np.repeat([10.0, 20.0, 30.0], R // 3)               # phi V per replica
    farm = FortranFarm(s, R, seed=17, translation_step=1.0, rotation_step=0.6, n_threads=4, mol_capacity=[90],
                       gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=targets / V), device_build=device_build)
    farm.run(600)                                                   # equilibrate
    samples = []
    for _ in range(40):
        farm.run(40)
        samples.append(farm.counts()[:, 0].copy())
    samples = np.array(samples, dtype=np.float64)                  # (40, R)
    c = farm.counters()
    assert c["trial_creations"] > 0 and c["trial_deletions"] > 0 and c["creations"] > 0 and c["deletions"] > 0
    assert farm.trials == sum(c[k] for k in ("trial_translations", "trial_rotations", "trial_creations", "trial_deletions"))
    assert farm.trials + farm.skipped == R * (600 + 40 * 40)
    eng = farm.eng
    counts = farm.counts()[:, 0]
    for r in range(R):
        assert eng.num_molecules(r, 0) == counts[r]
        e = eng.system_energy(r)
        run = farm.energy(r)
        ref = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")])
        assert np.max(np.abs(run - ref)) < farm_tol(ref, 600 + 40 * 40), (r, run - ref)
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        assert np.max(np.abs(A - eng.structure_factor(r))) < 1e-9
        dev = eng.get_molecules(r, 0)
        for slot in range(counts[r]):
            com, off = farm.molecule(r, 0, slot)
            assert np.array_equal(dev[slot], com[None, :] + off[:3])
    # ideal-gas limit: mean N per fugacity group within 12 % of phi V (Poisson noise ~3 % here)
    for g, target in enumerate([10.0, 20.0, 30.0]):
        mean_n = samples[:, g * (R // 3):(g + 1) * (R // 3)].mean()
        assert abs(mean_n - target) < 0.12 * target, (target, mean_n)
    farm.close()


def test_gcmc_farm_in_a_triclinic_box():
    """Insertion / deletion / translation / rotation of both species of the small mixture in a tilted cell:
    translations wrap through fractional coordinates and insertions are placed with the cell matrix
    (ApplyPBC / InsertAndOrientMolecule for box%is_triclinic); bookkeeping as in the orthorhombic test."""
    from maniac_mc_amd.engine import box_prepare
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.mixture_box(seed=8, tilt=(1.5, -0.8, 0.6))
    R = 6
    _, volume, _, _ = box_prepare(s.box_matrix)
    farm = FortranFarm(s, R, seed=5, translation_step=0.8, rotation_step=0.5, n_threads=2, mol_capacity=[40, 40],
                       gcmc=dict(p_translation=0.3, p_rotation=0.3, fugacity=12.0 / volume))
    farm.run(300)
    c = farm.counters()
    assert c["creations"] > 0 and c["deletions"] > 0 and c["translations"] > 0
    eng = farm.eng
    counts = farm.counts()
    for r in range(R):
        e = eng.system_energy(r)
        run = farm.energy(r)
        ref = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")])
        assert np.max(np.abs(run - ref)) < farm_tol(ref, 300), (r, run - ref)
        for ia in range(2):
            assert eng.num_molecules(r, ia) == counts[r, ia]
            dev = eng.get_molecules(r, ia)
            for slot in range(counts[r, ia]):
                com, off = farm.molecule(r, ia, slot)
                assert np.array_equal(dev[slot], com[None, :] + off[: dev.shape[1]])
    farm.close()


@pytest.mark.parametrize("device_build", [False, True], ids=["host_built", "device_built"])
def test_farm_at_benchmark_size_three_lanes(device_build):
    """The bench workload itself (3375 SPC/E, N = 10 125, Nk = 2242) in miniature: 48 chains on three lanes,
    150 steps; every chain's running energy equals a from-scratch evaluation of its final configuration and
    A(k) equals a fresh S(k) -- the size-independent invariant of the whole submit / wait / commit pipeline."""
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.spce_box(15)
    R, steps = 48, 150
    farm = FortranFarm(s, R, seed=29, translation_step=0.3, rotation_step=0.3, n_threads=4, n_lanes=3, device_build=device_build)
    assert farm.n_lanes == 3
    acc = farm.run(steps)
    assert farm.trials == R * steps and 0.4 * farm.trials < acc < 0.95 * farm.trials
    eng = farm.eng
    for r in (0, 15, 16, 31, 32, R - 1):          # first / last chain of every lane
        e = eng.system_energy(r)
        run = farm.energy(r)
        tol = TOL_K + 64 * np.finfo(float).eps * max(abs(e["recip_coulomb"]), abs(e["coulomb"])) * np.sqrt(steps)
        assert abs(run[0] - e["non_coulomb"]) < tol and abs(run[1] - e["coulomb"]) < tol and abs(run[2] - e["recip_coulomb"]) < tol
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        assert np.max(np.abs(A - eng.structure_factor(r))) < 1e-9
    farm.close()


@pytest.mark.parametrize("device_build", [False, True], ids=["host_built", "device_built"])
def test_gcmc_farm_framework_water_at_stated_size(device_build):
    """BASELINE.json configs[3] at its stated size: the 2208-atom inactive framework (site-major sweep, 35
    chunks per molecule) + 4-site water as the adsorbate, full move set (translation / rotation / insertion /
    deletion) on 12 chains at two fugacities.  Bookkeeping invariants: counts, running 5-component energies
    and A(k) equal a from-scratch evaluation of every chain's final configuration; host mirrors equal the device."""
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.framework_water_box()
    assert s.topo.atoms_in_res[0] == 2208 and s.topo.atoms_in_res[1] == 4
    R = 12
    V = 34.0 ** 3
    fug = np.repeat([30.0, 80.0], R // 2) / V
    farm = FortranFarm(s, R, seed=23, translation_step=0.5, rotation_step=0.5, n_threads=4, mol_capacity=[1, 120],
                       gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=fug), device_build=device_build)
    farm.run(400)
    c = farm.counters()
    assert c["trial_creations"] > 0 and c["trial_deletions"] > 0 and c["creations"] > 0 and c["deletions"] > 0
    assert c["translations"] > 0 and c["rotations"] > 0
    eng = farm.eng
    counts = farm.counts()[:, 0]
    assert counts.min() >= 0 and len(set(counts.tolist())) > 1
    for r in range(R):
        assert eng.num_molecules(r, 1) == counts[r] and eng.num_molecules(r, 0) == 1
        e = eng.system_energy(r)
        run = farm.energy(r)
        ref = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")])
        assert np.max(np.abs(run - ref)) < farm_tol(ref, 400), (r, run - ref)
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        assert np.max(np.abs(A - eng.structure_factor(r))) < 1e-9
        dev = eng.get_molecules(r, 1)
        for slot in range(counts[r]):
            com, off = farm.molecule(r, 0, slot)
            assert np.array_equal(dev[slot], com[None, :] + off[:4])
    farm.close()


@pytest.mark.parametrize("env", [{"MFARM_LANE_THREADS": "1"}, {"MFARM_LANE_THREADS": "2"}], ids=["lane_threads", "two_drivers"])
def test_farm_options_keep_the_invariants(env):
    """The opt-in modes -- one host thread per lane (MFARM_LANE_THREADS=1), two driver threads sharing the lanes
    (MFARM_LANE_THREADS=2) -- are selected by environment variables read when the libraries start: re-run the
    NVT / GCMC consistency tests (running energies, A(k), host mirrors vs a from-scratch evaluation) in a child
    process with them set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_farm.py", "-q", "-x", "-m", "gpu", "-k",
                        "test_fortran_farm_consistency or test_gcmc_farm_consistency_and_ideal_gas_limit or "
                        "test_farm_at_benchmark_size_three_lanes"],
                       capture_output=True, text=True, env=dict(os.environ, **env), cwd=root, timeout=900)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.slow
@pytest.mark.parametrize("gcmc,host_build", [(False, False), (True, False), (True, True)],
                         ids=["spce_nvt_device_built", "co2_gcmc_device_built", "co2_gcmc_host_built"])
def test_long_run_drift(gcmc, host_build):
    """tools/long_run_check.py as a test: ~1 M trials at the benchmark size (SPC/E NVT: 256 chains x 4000 steps on four
    lanes; CO2 GCMC: 512 chains x 2000 steps), after which every sampled chain's running energies must equal a
    from-scratch evaluation to the random-walk tolerance of tests/util.py::farm_tol and A(k) a fresh S(k) to 1e-9
    (round 2 measured 9e-9 K / 5e-13 after 3 M trials)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps = 2000 if gcmc else 4000
    cmd = [sys.executable, os.path.join(root, "tools", "long_run_check.py"), "--steps", str(steps), "--lanes", "4",
           "--replicas", "512" if gcmc else "256"] + (["--gcmc"] if gcmc else []) + (["--host-build"] if host_build else [])
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=1200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    m = re.search(r"energy\| = ([0-9.e+-]+) K, max \|A - S\(k\)\| = ([0-9.e+-]+), largest \|E\| = ([0-9.e+-]+)", p.stdout)
    assert m, p.stdout
    worst_e, worst_a, big = (float(m.group(i)) for i in (1, 2, 3))
    assert worst_e < farm_tol([big], steps) and worst_a < 1e-9, p.stdout


def test_two_farms_in_one_process():
    """Farms are objects, not module state: two farms on two engines (different boxes) advance alternately in one
    process and each keeps its own invariants."""
    from maniac_mc_amd.fortran_host import FortranFarm
    a = FortranFarm(synth.spce_box(5, seed=2), 6, seed=3, n_threads=2)
    b = FortranFarm(synth.co2_box(20, seed=4), 5, seed=7, translation_step=1.0, rotation_step=0.6, n_threads=2, mol_capacity=[60],
                    gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=20.0 / 50.0 ** 3), device_build=True)
    assert a.slot != b.slot
    for _ in range(3):
        a.run(15)
        b.run(25)
    assert a.trials + a.skipped == 6 * 45 and b.trials + b.skipped == 5 * 75
    for farm, keys in ((a, ("non_coulomb", "coulomb", "recip_coulomb")),
                       (b, ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb"))):
        for r in range(farm.R):
            e = farm.eng.system_energy(r)
            run = farm.energy(r)
            ref = np.array([e[k] for k in keys])
            assert np.max(np.abs(run[: len(keys)] - ref)) < farm_tol(ref, 75), (r, run, ref)
    b.close()
    a.run(5)
    assert a.trials + a.skipped == 6 * 50
    a.close()


@pytest.mark.parametrize("case", ["spce_nvt", "co2_gcmc", "framework_water_gcmc", "mixture_nvt"])
def test_device_decided_farm_is_the_host_decided_farm(case):
    """mfarm_configure(2): the engine applies the acceptance rule behind the k sweep and commits accepted candidates
    itself (mgpu_move_trial_decide_submit); the Fortran driver only draws numbers, selects moves and follows the
    outcomes.  Same seeds, same chains: counters, counts, running energies, coordinates and A(k) must be those of the
    farm that decides on the host and commits with mgpu_commit_submit -- bit for bit."""
    from maniac_mc_amd.fortran_host import FortranFarm
    kw = dict(seed=23, n_threads=4, n_lanes=3, device_build=True)
    if case == "spce_nvt":
        s, R, steps = synth.spce_box(6, seed=3), 9, 60
        kw.update(translation_step=0.4, rotation_step=0.4)
    elif case == "mixture_nvt":
        s, R, steps = synth.mixture_box(seed=4), 6, 60
        kw.update(translation_step=0.4, rotation_step=0.4)
    elif case == "co2_gcmc":
        s, R, steps = synth.co2_box(20, seed=13), 12, 150
        kw.update(translation_step=1.0, rotation_step=0.6, mol_capacity=[90],
                  gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=np.repeat([10.0, 30.0], 6) / 50.0 ** 3))
    else:
        s, R, steps = synth.framework_water_box(n_water=12, n_frame=300, L=24.0, seed=7), 8, 120
        kw.update(translation_step=0.5, rotation_step=0.5, mol_capacity=[1, 60],
                  gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=20.0 / 24.0 ** 3))
    farms = [FortranFarm(s, R, device_accept=da, **kw) for da in (False, True)]
    assert farms[1].device_accept and not farms[0].device_accept
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
    # and the device-decided farm is consistent with a from-scratch evaluation
    for r in (0, R - 1):
        e = b.eng.system_energy(r)
        ref = np.array([e[k] for k in ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")])
        assert np.max(np.abs(b.energy(r) - ref)) < farm_tol(ref, steps), (r, b.energy(r) - ref)
    for f in farms:
        f.close()


def test_host_team_is_the_serial_loop():
    """mgpu_set_host_team: the per-candidate loops inside submit / wait / commit cut into ranges run by an OpenMP team
    (csrc/mgpu_engine.hip: for_parts).  Two engines on the same two-residue mixture, 1536 replicas, one with the
    calling thread alone and one with a team of three (the ranges end inside the batch: 512-candidate ranges against
    runs of kinds and types that do not): mixed batches of moves, insertions and deletions of both residue types give
    the same energies bit for bit, the same commits (counts, coordinates, A(k)), and a refused batch names the same --
    the lowest -- candidate."""
    from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE
    from maniac_mc_amd.engine import Engine
    s = synth.mixture_box(seed=4)
    R = 1536
    engines = []
    for team in (1, 3):
        e = Engine.from_system(s, n_replicas=R, mol_capacity=[20, 16])
        e.init_structure_factor(0, True)
        for r in range(1, R):
            e.replica_copy(r, 0)
        e.set_host_team(team)
        engines.append(e)
    rng = np.random.default_rng(5)
    n0 = [int(s.n_mol[0]), int(s.n_mol[1])]
    results = []
    for step in range(3):
        t = rng.integers(0, 2, R).astype(np.int32)
        kind = rng.choice([MGPU_MOVE, MGPU_MOVE, MGPU_CREATION, MGPU_DELETION], R).astype(np.int32)
        rep = rng.permutation(R).astype(np.int32)
        nm = np.array([engines[0].num_molecules(int(rep[c]), int(t[c])) for c in range(R)])
        m = (rng.random(R) * np.minimum(nm, [n0[int(x)] for x in t])).astype(np.int32)       # a slot this script knows
        sites = np.zeros((R, 3, 3))
        for c in range(R):
            base = s.all_sites(int(t[c]))[int(m[c])]
            n1 = base.shape[0]
            sites[c, :n1] = base + rng.uniform(-0.3, 0.3, 3)
            if kind[c] == MGPU_CREATION:
                sites[c, :n1] = base - base.mean(axis=0) + s.bounds_lo + rng.random(3) * np.diag(s.box_matrix)
        acc = (rng.random(R) < 0.6).astype(np.int32)
        out = []
        for e in engines:
            old, new = e.gcmc_trial(rep, t, m, kind, sites, lane=step % 2)
            e.commit_lane(step % 2, rep, t, m, kind, acc)
            out.append((old, new))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), step
        results.append(out[0])
    assert any(np.any(o != 0) for o, _ in results)
    for r in (0, 1, 511, 512, 1023, 1024, R - 1):
        for ty in (0, 1):
            assert engines[0].num_molecules(r, ty) == engines[1].num_molecules(r, ty)
            assert np.array_equal(engines[0].get_molecules(r, ty), engines[1].get_molecules(r, ty))
        assert np.array_equal(engines[0].structure_factor(r), engines[1].structure_factor(r))
    # a refused batch: candidates 700 and 1300 are both invalid; both engines name the lower one
    rep = np.arange(R, dtype=np.int32)
    t = np.zeros(R, np.int32)
    kind = np.full(R, MGPU_MOVE, np.int32)
    m = np.zeros(R, np.int32)
    m[700] = 999
    m[1300] = 998
    sites = np.tile(s.all_sites(0)[0], (R, 1, 1))
    msgs = []
    for e in engines:
        with pytest.raises(Exception) as ei:
            e.gcmc_trial(rep, t, m, kind, sites)
        msgs.append(str(ei.value))
    assert msgs[0] == msgs[1] and "candidate 700" in msgs[0], msgs
    # two accepted candidates for one replica (in different ranges of the team): refused by both, and a clean repeat passes
    m[:] = 0
    for e in engines:
        e.gcmc_trial(rep, t, m, kind, sites)
        bad = rep.copy()
        bad[1400] = 3
        with pytest.raises(Exception) as ei:
            e.commit_lane(0, bad, t, m, kind, np.ones(R, np.int32), sites=sites)
        assert "more than one accepted candidate" in str(ei.value)
        e.commit_lane(0, rep, t, m, kind, np.ones(R, np.int32), sites=sites)
    for e in engines:
        e.close()
