"""MANIAC's input surface (maniac_mc_amd/io_maniac.py) on the REFERENCE'S OWN reader fixtures
(/root/reference/tests/readers/**, copied as data into tests/golden/ref_fixtures/).

Expected values come from the reference's own front end (tests/golden/make_ref_fixture_expectations.py
drives ReadInput / ReadSystemData / ReadParameters / PrepareSimulationParameters / ComputeSystemEnergy
of the compiled reference): good files must give the same state bit for bit, bad files must fail like
the reference's run-test.sh expects (readers/data/run-test.sh:28-40, readers/inputs/run-test.sh)."""
import json
import os

import numpy as np
import pytest

from maniac_mc_amd import io_maniac
from tests.util import GOLDEN, TOL_K

FIX = os.path.join(GOLDEN, "ref_fixtures")
OUTCOMES = json.load(open(os.path.join(FIX, "expected_outcomes.json")))
E_KEYS = ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb", "total")


def load(name):
    return io_maniac.load_system(os.path.join(FIX, "input.maniac"), os.path.join(FIX, name),
                                 os.path.join(FIX, "parameters.inc"))


@pytest.mark.parametrize("name", ["good-01", "good-02"])
def test_good_data_files_match_reference_front_end(name, refcpu_mod):
    assert OUTCOMES[name + ".data"]["ok"]
    g = np.load(os.path.join(FIX, name + ".expected.npz"))
    s, inp = load(name + ".data")
    topo = s.topo
    assert np.array_equal(topo.atoms_in_res, g["atoms_in_res"]) and np.array_equal(topo.is_active, g["is_active"])
    assert np.array_equal(s.n_mol, g["n_mol"])
    assert np.array_equal(topo.atom_types, g["atom_types"]) and np.array_equal(topo.charges, g["charges"])
    assert np.array_equal(s.box_matrix, g["box_matrix"]) and np.array_equal(s.bounds_lo, g["bounds_lo"])
    for t in range(topo.n_res):
        assert np.array_equal(s.com[t], g[f"com{t}"]), f"com of residue {t}"          # bit for bit
        assert np.array_equal(s.offsets[t], g[f"off{t}"]), f"offsets of residue {t}"
    seen = g["coeff_seen"]
    assert np.array_equal(topo.epsilon[seen], g["epsilon"][seen]) and np.array_equal(topo.sigma[seen], g["sigma"][seen])
    # input scalars, probabilities after rescaling, fugacity after ConvertFugacity
    exp = g["input"]
    got = [inp.nb_block, inp.nb_step, inp.temperature, inp.ewald_tolerance, inp.real_space_cutoff,
           inp.translation_step, inp.rotation_step_angle, float(inp.recalibrate_moves), inp.translation_proba,
           inp.rotation_proba, inp.insertion_deletion_proba, inp.swap_proba]
    assert got[:3] == list(exp[:3]) and got[5:] == list(exp[5:])
    assert np.array_equal(np.array(inp.fugacity_per_A3()), g["fugacity"])
    # the hot path on the parsed system: the C restatement reproduces the reference's energies exactly
    P = refcpu_mod.RefCPU(s)
    assert (P.alpha, P.nk, P.rc) == (float(g["alpha"]), int(g["nk"]), float(g["rc_eff"]))
    e = P.system_energy()
    assert np.array_equal(np.array([e[k] for k in E_KEYS]), g["system_energy"])
    # bonded tables per residue (DetectBondPerResidue & co.), which only the data-file writer uses
    _, _, dat = io_maniac.load_system(os.path.join(FIX, "input.maniac"), os.path.join(FIX, name + ".data"),
                                      os.path.join(FIX, "parameters.inc"), with_data=True)
    for key, ncol in (("bonds", 3), ("angles", 4), ("dihedrals", 5), ("impropers", 5)):
        assert dat["type_counts"][key] == int(g[key + "_types"])
        for t in range(topo.n_res):
            mine = np.array(dat["bonded_per_residue"][key][t], dtype=np.int32).reshape(-1, ncol)
            assert np.array_equal(mine, g[f"{key}_{t}"][:, :ncol]), (key, t)


def test_reference_program_output_on_its_fixture():
    """energy.dat of `maniac -i input.maniac -d good-01.data -p parameters.inc` (oracle/_ref/maniac,
    6 printed decimals, kcal/mol): total recip non-coulomb coulomb self intra."""
    g = np.load(os.path.join(FIX, "good-01.expected.npz"))
    kcal = g["system_energy"] * 0.0019872041
    printed = dict(total=0.019383, recip_coulomb=1.477379, non_coulomb=-0.000165, coulomb=0.000083,
                   ewald_self=-266.510672, intra_coulomb=265.052758)
    for i, k in enumerate(E_KEYS):
        assert abs(kcal[i] - printed[k]) <= 5.1e-7, (k, kcal[i], printed[k])


@pytest.mark.parametrize("name", ["bad-01", "bad-02", "bad-03", "bad-04"])
def test_bad_data_files_fail_like_the_reference(name):
    exp = OUTCOMES[name + ".data"]
    assert not exp["ok"]
    with pytest.raises(io_maniac.ManiacInputError) as ei:
        load(name + ".data")
    assert ei.value.code == exp["returncode"], (ei.value, exp)      # the reference's stop code


@pytest.mark.parametrize("name", sorted(k for k in OUTCOMES if k.startswith("inputs/")))
def test_maniac_input_files(name):
    exp = OUTCOMES[name]
    path = os.path.join(FIX, name)
    if exp["ok"]:
        inp = io_maniac.read_maniac_input(path)
        g = np.load(path.replace(".maniac", ".expected.npz"))
        assert [r.nb_atoms for r in inp.residues] == list(g["atoms_in_res"])
        assert [r.is_active for r in inp.residues] == list(g["is_active"])
        assert [len(r.types) for r in inp.residues] == list(g["types_per_res"])
        got = [inp.nb_block, inp.nb_step, inp.temperature, inp.ewald_tolerance, inp.real_space_cutoff,
               inp.translation_step, inp.rotation_step_angle, float(inp.recalibrate_moves), inp.translation_proba,
               inp.rotation_proba, inp.insertion_deletion_proba, inp.swap_proba]
        assert got == list(g["input"])
        # residues come out sorted by their smallest atom type (input_parser.f90:603-672)
        keys = [min(r.types) for r in inp.residues]
        assert keys == sorted(keys)
    else:
        with pytest.raises(io_maniac.ManiacInputError) as ei:
            io_maniac.read_maniac_input(path)
        # same first complaint as the reference where it prints one
        for phrase in ("Error reading translation_proba", "Invalid real_space_cutoff"):
            if phrase in exp["message"]:
                assert phrase in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["good-01", "good-02"])
def test_engine_on_reference_fixture(name):
    """File -> System -> HIP engine: the six energy components of the reference's front-end run."""
    from maniac_mc_amd.engine import Engine
    from tests.test_gpu_parity import close
    g = np.load(os.path.join(FIX, name + ".expected.npz"))
    s, _ = load(name + ".data")
    eng = Engine.from_system(s)
    assert eng.alpha == float(g["alpha"]) and eng.nk == int(g["nk"])
    e = eng.system_energy(0)
    for i, k in enumerate(E_KEYS):
        close(e[k], g["system_energy"][i], f"{name} {k}")
    eng.close()
