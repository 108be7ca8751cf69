"""One launch per window of one chain (mgpu_chain_window): the window's energies, the decisions taken on the device and the
state its commit leaves are held to the batched path (mgpu_gcmc_trial_submit / wait + the host's rule + mgpu_commit_submit)
-- itself held to the oracle by tests/test_gpu_gcmc.py and tests/test_gpu_parity.py -- on plane-major boxes, a framework
box (flat pair kernel), insertions / deletions, the reference's deletion as written, and at the margin where
the device must leave a step to the host.  Reference: src/monte_carlo.f90:40-86, src/monte_carlo_utils.f90:184-226,
:275-395, src/create_molecule.f90:100-112, src/delete_molecule.f90:100-142."""
import numpy as np
import pytest

from maniac_mc_amd import synth
from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE
from maniac_mc_amd.engine import Engine
from tests.test_gpu_parity import close
from tests.util import golden_system

pytestmark = pytest.mark.gpu


def _totals(kind, o, w, e_recip, w_link=None):
    """old%total / new%total as the move drivers form them (mc_chain.f90 resolve_step)."""
    if kind == MGPU_MOVE:
        return (o[0] + o[1]) + o[2], (w[0] + w[1]) + w[2]
    if kind == MGPU_CREATION:
        return e_recip, (((w[0] + w[1]) + w[2]) + w[3]) + w[4]
    return (((o[0] + o[1]) + e_recip) + o[3]) + o[4], (w[2] if w_link is None else w_link[2])


def _expected_first(kinds, old, new, u, pref, T, e_recip, link=None):
    for c in range(len(kinds)):
        if link is not None and link[c] == -2:
            continue
        wl = new[link[c]] if (link is not None and link[c] >= 0) else None
        eo, en = _totals(kinds[c], old[c], new[c], e_recip, wl)
        x = pref[c] * np.exp(-(en - eo) / T)
        assert abs(u[c] - x) > 1e-9 * max(x, 1e-300), "test data too close to the margin"
        if u[c] <= min(1.0, x):
            return c
    return -1


def _twin(s, cap=None):
    """Two engines on the same system.  MGPU_NO_FROZEN_BATCH: a framework box's batched path would otherwise sweep the
    framework with pair_frozen_kernel (candidates in the lanes, chunk partials): the same sums in another order; the
    one-launch path uses the flat kernel's work units, and so does the batched path with the switch set."""
    import os
    out = []
    os.environ["MGPU_NO_FROZEN_BATCH"] = "1"
    try:
        for _ in range(2):
            e = Engine.from_system(s, n_replicas=1, mol_capacity=cap)
            e.init_structure_factor(0, True)
            out.append(e)
    finally:
        os.environ.pop("MGPU_NO_FROZEN_BATCH", None)
    return out


def _same_state(a, b, n_res):
    for t in range(n_res):
        assert a.num_molecules(0, t) == b.num_molecules(0, t)
        assert np.array_equal(a.get_molecules(0, t), b.get_molecules(0, t))
    assert np.array_equal(a.structure_factor(0), b.structure_factor(0))


@pytest.mark.parametrize("name", ["spce216", "mixture", "framework_small", "mixture_triclinic"])
def test_window_of_moves_is_the_batched_path(name):
    """Several windows in a row: energies bitwise those of the batched path (same kernels' pieces, same nsplit), the
    device's verdict = the rule applied in order with numpy's exp, the committed state bitwise the batched commit's.
    Triclinic boxes too (round 5: the window's pair role runs the register-site sweeps with ComputeDistance's image search)."""
    if name == "mixture_triclinic":
        s = synth.mixture_box(tilt=(1.5, -0.8, 0.6))
    else:
        _, s = golden_system(name)
    A, B = _twin(s)
    assert A.chain_window_capacity() >= 8
    rng = np.random.default_rng(5)
    act = [t for t in range(s.topo.n_res) if s.topo.is_active[t] == 1]
    T = float(s.temperature)
    for win in range(6):
        n = 8
        t = rng.choice(act, n).astype(np.int32)
        m = np.array([rng.integers(0, B.num_molecules(0, int(tt))) for tt in t], dtype=np.int32)
        stride = max(int(s.topo.atoms_in_res[tt]) for tt in act)
        sites = np.zeros((n, stride, 3))
        for c in range(n):
            n1 = int(s.topo.atoms_in_res[t[c]])
            sites[c, :n1] = B.get_molecules(0, int(t[c]))[m[c]] + rng.uniform(-0.25, 0.25, 3)[None, :]
        kinds = np.full(n, MGPU_MOVE, dtype=np.int32)
        u = rng.random(n)
        pref = np.ones(n)
        old_b, new_b = B.gcmc_trial(np.zeros(n, np.int32), t, m, kinds, sites)
        old_a, new_a, first, und = A.chain_window(0, t, m, kinds, sites, u, pref, T, 0.0)
        assert und == -1
        assert np.array_equal(old_a, old_b) and np.array_equal(new_a, new_b), (np.abs(old_a - old_b).max(), np.abs(new_a - new_b).max())
        assert first == _expected_first(kinds, old_a, new_a, u, pref, T, 0.0)
        if first >= 0:
            acc = np.zeros(n, np.int32)
            acc[first] = 1
            B.commit_lane(0, np.zeros(n, np.int32), t, m, kinds, acc)
        _same_state(A, B, s.topo.n_res)
    assert A.chain_stats() == (6, 0)
    A.close(); B.close()


def test_grand_canonical_window_and_capacity():
    """Insertions, deletions and moves in one window (CO2 box of BASELINE configs[2]); prefactors and the running
    reciprocal energy enter the rule as create_molecule.f90:64 / delete_molecule.f90:73 / monte_carlo_utils.f90:366-372
    prescribe."""
    s = synth.co2_box(20, seed=4)
    A, B = _twin(s, cap=[40])
    rng = np.random.default_rng(8)
    e_recip = B.system_energy(0)["recip_coulomb"]
    T = float(s.temperature)
    tmpl = s.offsets[0][0]
    L = float(s.box_matrix[0, 0])
    for win in range(8):
        nm = B.num_molecules(0, 0)
        n = 9
        kinds = rng.choice([MGPU_MOVE, MGPU_CREATION, MGPU_DELETION], n).astype(np.int32)
        m = rng.integers(0, nm, n).astype(np.int32)
        m[kinds == MGPU_CREATION] = -1
        sites = np.zeros((n, 3, 3))
        pref = np.ones(n)
        for c in range(n):
            if kinds[c] == MGPU_MOVE:
                sites[c] = B.get_molecules(0, 0)[m[c]] + rng.uniform(-0.4, 0.4, 3)[None, :]
            elif kinds[c] == MGPU_CREATION:
                sites[c] = (s.bounds_lo + rng.random(3) * L)[None, :] + tmpl @ np.linalg.qr(rng.normal(size=(3, 3)))[0].T
                pref[c] = 30.0 / (nm + 1)
            else:
                pref[c] = nm / 30.0
        u = rng.random(n)
        t = np.zeros(n, np.int32)
        old_b, new_b = B.gcmc_trial(np.zeros(n, np.int32), t, m, kinds, sites)
        old_a, new_a, first, und = A.chain_window(0, t, m, kinds, sites, u, pref, T, e_recip)
        assert und == -1
        assert np.array_equal(old_a, old_b) and np.array_equal(new_a, new_b)
        assert first == _expected_first(kinds, old_a, new_a, u, pref, T, e_recip)
        if first >= 0:
            acc = np.zeros(n, np.int32)
            acc[first] = 1
            B.commit_lane(0, np.zeros(n, np.int32), t, m, kinds, acc)
            eo, en = _totals(kinds[first], old_a[first], new_a[first], e_recip)
            if kinds[first] == MGPU_MOVE:
                e_recip += new_a[first][2] - old_a[first][2]
            else:
                e_recip = new_a[first][2]
        _same_state(A, B, 1)
    # a full type refuses another insertion before anything is launched
    full = Engine.from_system(s, n_replicas=1, mol_capacity=[20])
    full.init_structure_factor(0, True)
    with pytest.raises(Exception, match="mol_capacity"):
        full.chain_window(0, [0], [-1], [MGPU_CREATION], np.zeros((1, 3, 3)) + 1.0, [0.5], [1.0], T, 0.0)
    full.close()
    A.close(); B.close()


def test_deletion_as_written_window():
    """link >= 0: the new reciprocal energy is the creation-kind energy of the molecule RemoveMolecule moves into the slot,
    and the commit adds THAT molecule's terms to A(k) while the coordinates lose slot m (monte_carlo_utils.f90:301-309).
    Held to the neutral primitives mc_chain.f90 composes the same update from."""
    s = synth.co2_box(16, seed=2)
    A, B = _twin(s, cap=[24])
    e_recip = B.system_energy(0)["recip_coulomb"]
    T = float(s.temperature)
    nm = 16
    last = B.get_molecules(0, 0)[nm - 1]
    kinds = np.array([MGPU_DELETION, MGPU_CREATION], dtype=np.int32)
    sites = np.stack([np.zeros((3, 3)), last])
    m = np.array([5, -1], dtype=np.int32)
    old_b, new_b = B.gcmc_trial(np.zeros(2, np.int32), [0, 0], m, kinds, sites)
    old_a, new_a, first, und = A.chain_window(0, [0, 0], m, kinds, sites, [1e-300, 0.0], [1.0, 0.0], T, e_recip, link=[1, -2])
    assert np.array_equal(old_a[0], old_b[0]) and new_a[1][2] == new_b[1][2]
    assert (first, und) == (0, -1)            # u = 1e-300: accepted whatever the energies
    B.replace_molecule(0, 0, 5, nm - 1)
    B.set_num_molecules(0, 0, nm - 1)
    B.structure_factor_add(0, 0, last)
    _same_state(A, B, 1)
    # ... and a rejected one (u = 1) changes nothing
    stateA = (A.get_molecules(0, 0).copy(), A.structure_factor(0).copy())
    last = A.get_molecules(0, 0)[nm - 2]
    sites = np.stack([np.zeros((3, 3)), last])
    _, _, first, und = A.chain_window(0, [0, 0], [3, -1], kinds, sites, [1.0, 0.0], [1e-6, 0.0], T, e_recip, link=[1, -2])
    assert (first, und) == (-1, -1)
    assert np.array_equal(A.get_molecules(0, 0), stateA[0]) and np.array_equal(A.structure_factor(0), stateA[1])
    A.close(); B.close()


def test_steps_too_close_to_call_are_left_to_the_host():
    """A draw within the engine's relative margin of the acceptance probability stops the device's walk: the step comes
    back undecided, nothing at or behind it is committed, earlier steps keep their verdicts.  (The default margin is 16
    ulp -- the band in which OCML's exp and glibc's could disagree; the test widens it.)"""
    _, s = golden_system("spce216")
    A, B = _twin(s)
    rng = np.random.default_rng(3)
    n0 = 24
    m = rng.choice(216, n0, replace=False).astype(np.int32)
    sites = B.get_molecules(0, 0)[m] + rng.uniform(-0.3, 0.3, (n0, 1, 3))
    T = float(s.temperature)

    def prob(old, new):
        return np.exp(-(((new[:, 0] + new[:, 1]) + new[:, 2]) - ((old[:, 0] + old[:, 1]) + old[:, 2])) / T)

    old, new = B.gcmc_trial(np.zeros(n0, np.int32), np.zeros(n0, np.int32), m, np.full(n0, MGPU_MOVE, np.int32), sites)
    uphill = np.flatnonzero(prob(old, new) < 0.5)[:6]        # steps that a draw can reject
    n = len(uphill)
    assert n == 6
    m, sites, old, new = m[uphill], sites[uphill], old[uphill], new[uphill]
    x = prob(old, new)
    t = np.zeros(n, np.int32)
    kinds = np.full(n, MGPU_MOVE, dtype=np.int32)
    u = np.minimum(x * 1.5, 0.99)               # everything rejected ...
    u[3] = x[3] * (1.0 + 1e-7)                  # ... step 3 too, but from inside a margin of 1e-6
    before = (A.get_molecules(0, 0).copy(), A.structure_factor(0).copy())
    A.chain_set_margin(1e-6)
    _, _, first, und = A.chain_window(0, t, m, kinds, sites, u, np.ones(n), T, 0.0)
    assert (first, und) == (-1, 3)
    assert np.array_equal(A.get_molecules(0, 0), before[0]) and np.array_equal(A.structure_factor(0), before[1])
    # behind an undecided step nothing is decided: step 4 would be accepted (u = 0) but the device has stopped at 3
    u2 = u.copy()
    u2[4] = 0.0
    _, _, first, und = A.chain_window(0, t, m, kinds, sites, u2, np.ones(n), T, 0.0)
    assert (first, und) == (-1, 3)
    assert np.array_equal(A.get_molecules(0, 0), before[0]) and np.array_equal(A.structure_factor(0), before[1])
    # ... while a step accepted BEFORE it is committed as usual
    u3 = u.copy()
    u3[1] = 0.0
    _, _, first, und = A.chain_window(0, t, m, kinds, sites, u3, np.ones(n), T, 0.0)
    assert (first, und) == (1, -1)
    assert np.array_equal(A.get_molecules(0, 0)[m[1]], sites[1])
    A.close(); B.close()
    # with the default margin (16 ulp) the same window is decided: step 3's draw lies 1e-7 above its probability
    A, B = _twin(s)
    _, _, first, und = A.chain_window(0, t, m, kinds, sites, u, np.ones(n), T, 0.0)
    assert (first, und) == (-1, -1)
    # an everything-undecided margin: the device never commits
    A.chain_set_margin(1e300)
    _, _, first, und = A.chain_window(0, t, m, kinds, sites, np.full(n, 0.5), np.ones(n), T, 0.0)
    assert (first, und) == (-1, 0)
    assert A.chain_stats() == (2, 1)
    A.close(); B.close()


def test_capacity_is_zero_where_the_path_does_not_apply():
    s = synth.mixture_box(tilt=(1.5, -0.8, 0.6))          # triclinic: single-chain windows yes, farm windows (device-built moves) no
    e = Engine.from_system(s, n_replicas=1)
    assert e.chain_window_capacity() > 0 and e.farm_window_capacity()[0] == 0
    e.close()
    s = synth.rigid_adsorbate_box()                        # a 24-site active molecule
    e = Engine.from_system(s, n_replicas=1)
    assert e.chain_window_capacity() == 0
    e.close()


def test_chain_loop_modes_write_the_same_files(tmp_path):
    """The single-chain driver with one launch per window against the batched calls, K = 1 and K = 8: same files."""
    import filecmp
    import json
    import os
    from maniac_mc_amd import run
    from tests.util import GOLDEN
    runs = os.path.join(GOLDEN, "runs")
    summary = json.load(open(os.path.join(runs, "summary.json")))
    for case in ("co2_gcmc", "framework_water_gcmc", "spce_nvt"):
        if case not in summary:
            continue
        inp = os.path.join(runs, case, "inputs")
        files = [os.path.join(inp, f) for f in ("system.maniac", "system.data", "system.inc")]
        outs = []
        for k, cw in ((1, False), (1, True), (8, True)):
            out = str(tmp_path / f"{case}_{k}_{int(cw)}") + "/"
            res = run.run_simulation(*files, out, seed=summary[case]["seed"], as_written=bool(summary[case].get("as_written")),
                                     speculate=k, chain_windows=cw, nb_block=3, nb_step=400)
            assert (res["chain_windows"][0] > 0) == cw
            outs.append(out)
        for o in outs[1:]:
            for f in sorted(os.listdir(outs[0])):
                if f != "log.maniac":
                    assert filecmp.cmp(os.path.join(o, f), os.path.join(outs[0], f), shallow=False), (case, o, f)


@pytest.mark.parametrize("margin", [0.2, 1e300], ids=["a_fifth_of_the_steps", "every_step"])
@pytest.mark.parametrize("case", ["co2_gcmc", "dumbbell_gcmc_reservoir", "spce_nvt"])
def test_steps_left_to_the_host_still_write_the_reference_files(case, margin, tmp_path):
    """The loop's side of the hand-over: with the engine's margin widened, a fifth of the steps -- or every one -- come back
    undecided; the loop then applies the rule itself, commits an accepted step with explicit sites (or, for the as-written
    deletion, through the neutral primitives), ends the window there and puts the generator back.  The files must still be
    the reference's, character for character."""
    import json
    import os
    from maniac_mc_amd import run
    from tests.util import GOLDEN
    runs = os.path.join(GOLDEN, "runs")
    summary = json.load(open(os.path.join(runs, "summary.json")))
    inputs = os.path.join(runs, case, "inputs")
    expected = os.path.join(runs, case, "expected")
    out = str(tmp_path / "out") + "/"
    cwd = os.getcwd()
    os.chdir(inputs)
    try:
        res = run.run_simulation("system.maniac", "system.data", "system.inc", out, seed=summary[case]["seed"],
                                 reservoir_path="reservoir.data" if summary[case]["reservoir"] else None,
                                 as_written=bool(summary[case].get("as_written")), chain_margin=margin)
    finally:
        os.chdir(cwd)
    windows, undecided = res["chain_windows"]
    assert windows > 0 and undecided > (0.05 * windows if margin < 1 else 0.99 * windows)
    for f in summary[case]["files"]:
        want = open(os.path.join(expected, f)).read().split("\n")
        got = open(os.path.join(out, f)).read().split("\n")
        if f == "log.maniac":
            got = ["<output path>" if out.rstrip("/") in ln else ln for ln in got]
        assert got == want, f
