"""The Fortran host Metropolis driver (maniac_mc_amd/fortran/mc_farm.f90) on the GPU: after a run
of overlapped batched steps every chain's running energy must equal a from-scratch evaluation of
its final configuration, A(k) must equal a fresh S(k), and the host mirrors must equal the device."""
import numpy as np
import pytest

from maniac_mc_amd import synth
from tests.util import TOL_K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("maker,R,steps", [(lambda: synth.spce_box(6, seed=3), 7, 40), (lambda: synth.co2_box(24, seed=5), 4, 60),
                                           (lambda: synth.mixture_box(seed=4), 5, 50)])
def test_fortran_farm_consistency(maker, R, steps):
    from maniac_mc_amd.fortran_host import FortranFarm
    s = maker()
    farm = FortranFarm(s, R, seed=11, translation_step=0.4, rotation_step=0.4)
    acc = farm.run(steps)
    assert farm.trials == R * steps and 0 < acc <= farm.trials
    assert acc == farm.accepted
    eng = farm.eng
    for r in range(R):
        e = eng.system_energy(r)
        run = farm.energy(r)
        # the running sums accumulate `steps` increments of O(1e5) K terms: allow their rounding
        tol = TOL_K + 64 * np.finfo(float).eps * max(abs(e["recip_coulomb"]), abs(e["coulomb"])) * np.sqrt(steps)
        assert abs(run[0] - e["non_coulomb"]) < tol and abs(run[1] - e["coulomb"]) < tol and abs(run[2] - e["recip_coulomb"]) < tol
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
    assert farm.trials == R * (steps + 5) and farm.accepted == acc + acc2
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
