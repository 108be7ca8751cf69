"""The two one-launch kernels held DIRECTLY to the oracle (oracle/refcpu.c, pinned bit for bit to the compiled reference):
chain_window_kernel (mgpu_chain_window) and farm_window_kernel (mgpu_farm_window_submit).  Elsewhere they are compared,
bitwise, with the batched path -- which is what is oracle-checked -- so this file is the short way round.
Reference: src/monte_carlo_utils.f90:275-395 (ComputeOldEnergy / ComputeNewEnergy), src/ewald_energy.f90:191-274."""
import numpy as np
import pytest

from maniac_mc_amd import synth
from maniac_mc_amd._lib import MGPU_CREATION, MGPU_DELETION, MGPU_MOVE
from maniac_mc_amd.engine import Engine
from tests.test_gpu_parity import amp_close, close

pytestmark = pytest.mark.gpu


def _oracle_move(P, t, m, sites):
    """(old[5], new[5], A after) of moving molecule (t, m) to `sites`, the oracle left as it was."""
    com, off = P.get_molecule(t, m)
    P.save_fourier(t, m)
    old = P.old_energy(t, m, 0)[:5]
    P.set_molecule(t, m, sites[0], sites - sites[0][None, :])
    new = P.new_energy(t, m, 0)[:5]
    A_after = P.amplitude()
    P.set_molecule(t, m, com, off)
    P.restore_fourier(t, m)
    return old, new, A_after


@pytest.mark.parametrize("maker,t_act", [(lambda: synth.spce_box(5, seed=4), 0), (lambda: synth.co2_box(16, seed=2), 0),
                                          (lambda: synth.framework_water_box(n_water=10, n_frame=260, L=23.0, seed=9), 1)])
def test_chain_window_energies_against_the_oracle(maker, t_act, refcpu_mod):
    """A window of translated + rotated candidates of one chain: every row's old / new components against the oracle's
    ComputeOldEnergy / ComputeNewEnergy, and A(k) + coordinates after the device's commit of the first accepted step."""
    s = maker()
    eng = Engine.from_system(s, n_replicas=1)
    eng.init_structure_factor(0, True)
    P = refcpu_mod.RefCPU(s)
    P.system_energy(); P.init_amplitude(True)
    rng = np.random.default_rng(11)
    n1 = int(s.topo.atoms_in_res[t_act])
    n = min(6, eng.chain_window_capacity())
    assert n >= 2
    m = rng.choice(int(s.n_mol[t_act]), n, replace=False).astype(np.int32)
    sites = np.zeros((n, n1, 3))
    exp = []
    for c in range(n):
        com, off = P.get_molecule(t_act, int(m[c]))
        sites[c] = P.apply_pbc(com + rng.uniform(-0.3, 0.3, 3))[None, :] + off @ P.rotation_matrix(1 + c % 3, 0.25).T
        exp.append(_oracle_move(P, t_act, int(m[c]), sites[c]))
    u = np.full(n, 0.999999)
    u[n // 2] = 1e-12                                   # the device accepts this step (and nothing before it, almost surely)
    old, new, first, und = eng.chain_window(0, np.full(n, t_act, np.int32), m, np.full(n, MGPU_MOVE, np.int32), sites, u, np.ones(n),
                                            float(s.temperature), 0.0)
    assert und == -1 and 0 <= first <= n // 2
    for c in range(n):
        close(old[c], exp[c][0], f"window row {c} old")
        close(new[c], exp[c][1], f"window row {c} new")
    amp_close(eng.structure_factor(0), exp[first][2], "A after the window's commit")
    assert np.array_equal(eng.get_molecules(0, t_act)[m[first]], sites[first])
    eng.close()


def test_farm_window_energies_against_the_oracle(refcpu_mod):
    """One farm window over four chains holding DIFFERENT configurations, every step sent with the driver's decision
    `accept` so that the candidate the device built is what the replica holds afterwards: old / new components of every
    chain against the oracle evaluated for exactly that move, and A(k) after the commit."""
    base = synth.spce_box(5, seed=4)
    R = 4
    rng = np.random.default_rng(3)
    eng = Engine.from_system(base, n_replicas=R)
    systems, oracles = [], []
    for r in range(R):
        s = base.copy()
        s.com[0] = s.com[0] + rng.uniform(-0.2, 0.2, s.com[0].shape) * (r > 0)
        eng.load_system(s, r)
        eng.set_frames(r, 0, s.com[0], s.offsets[0])
        eng.init_structure_factor(r, True)
        P = refcpu_mod.RefCPU(s)
        P.system_energy(); P.init_amplitude(True)
        systems.append(s); oracles.append(P)
    rep = np.arange(R, dtype=np.int32)
    m = rng.integers(0, int(base.n_mol[0]), R).astype(np.int32)
    move = np.array([1, 2, 2, 1], np.int32)
    u5 = rng.uniform(0, 1, (R, 5))
    eng.farm_window_submit(rep, np.zeros(R, np.int32), m, move, u5, 0.4, 0.4, np.full(R, 0.5), np.ones(R), float(base.temperature),
                           forced=np.ones(R, np.int32))
    old, new, v = eng.farm_window_wait(R)
    assert np.all(v == 1)
    for r in range(R):
        cand = eng.get_molecules(r, 0)[m[r]]                      # the committed candidate
        before = systems[r].sites(0, int(m[r]))
        assert np.max(np.abs(cand - before)) > 1e-3 and np.max(np.abs(cand - before)) < 1.5
        eo, en, A_after = _oracle_move(oracles[r], 0, int(m[r]), cand)
        close(old[r], eo, f"chain {r} old")
        close(new[r], en, f"chain {r} new")
        amp_close(eng.structure_factor(r), A_after, f"chain {r} A after the commit")
        # the construction itself: a translation keeps the offsets, a rotation the centre (src/translation.f90:93-112,
        # src/rotation.f90:34-75) -- site 0 is the centre's frame only up to the molecule's own offset, so compare shapes
        d_before = before - before[0][None, :]
        d_after = cand - cand[0][None, :]
        if move[r] == 1:
            assert np.max(np.abs(d_after - d_before)) < 1e-12
        else:
            assert np.max(np.abs(np.linalg.norm(d_after, axis=1) - np.linalg.norm(d_before, axis=1))) < 1e-12
    eng.close()
