"""CPU checks on the whole-run fixtures (tests/golden/runs): the committed input files load through
io_maniac, write_input_files round-trips, and the C restatement's system energy equals the block-0
record of the energy.dat the reference wrote for them (6 decimals, kcal/mol)."""
import json
import os

import numpy as np
import pytest

from maniac_mc_amd import io_maniac
from maniac_mc_amd.system import KB_KCALMOL
from tests.util import GOLDEN

RUNS = os.path.join(GOLDEN, "runs")
SUMMARY = json.load(open(os.path.join(RUNS, "summary.json")))


def _load(case):
    d = os.path.join(RUNS, case, "inputs")
    return io_maniac.load_system(os.path.join(d, "system.maniac"), os.path.join(d, "system.data"),
                                 os.path.join(d, "system.inc"))


@pytest.mark.parametrize("case", sorted(SUMMARY))
def test_block0_energy_record(case, refcpu_mod):
    system, inp = _load(case)
    e = refcpu_mod.RefCPU(system).system_energy()
    rec = open(os.path.join(RUNS, case, "expected", "energy.dat")).read().split("\n")[1].split()
    assert int(rec[0]) == 0
    got = [e[k] * KB_KCALMOL for k in ("total", "recip_coulomb", "non_coulomb", "coulomb", "ewald_self", "intra_coulomb")]
    assert [f"{v:.6f}" for v in got] == rec[1:]
    # number_<res>.dat, block 0
    for t, r in enumerate(inp.residues):
        if r.is_active == 1 and system.n_mol[t] > 0:
            line = open(os.path.join(RUNS, case, "expected", f"number_{r.name}.dat")).read().split("\n")[1].split()
            assert [int(v) for v in line] == [0, int(system.n_mol[t])]


def test_written_input_files_round_trip(tmp_path):
    system, inp = _load("spce_nvt")
    files = io_maniac.write_input_files(system, str(tmp_path), nb_block=inp.nb_block, nb_step=inp.nb_step,
                                        translation_step=inp.translation_step, rotation_step_angle=inp.rotation_step_angle,
                                        translation_proba=inp.translation_proba, rotation_proba=inp.rotation_proba,
                                        masses=[15.9994, 1.008], atom_names=["OW", "HW"], seed=7)
    again, inp2 = io_maniac.load_system(*files)
    assert inp2.has_seed and inp2.seed == 7 and (inp2.nb_block, inp2.nb_step) == (inp.nb_block, inp.nb_step)
    assert np.array_equal(again.box_matrix, system.box_matrix) and np.array_equal(again.n_mol, system.n_mol)
    assert np.array_equal(again.topo.charges, system.topo.charges)
    # coordinates are printed with 17 significant digits: com + offset reproduces every site to the last bit or two
    for t in range(system.topo.n_res):
        assert np.max(np.abs(again.all_sites(t) - system.all_sites(t))) < 1e-13
    assert np.allclose(again.topo.epsilon, system.topo.epsilon, rtol=1e-15, atol=0)


@pytest.mark.parametrize("case", sorted(SUMMARY))
def test_log_header_is_the_references(case, tmp_path):
    """Everything the reference logs before "Started Monte Carlo Loop" -- banner, input echo, box and data-file
    summary (primary and reservoir), Lorentz-Berthelot listing, LogEwaldParameters block -- rebuilt by the front end
    (io_maniac.log_header_lines) and written through the Fortran driver's list-directed write (so long lines wrap as
    the runtime wraps them): character for character the head of the log the reference wrote for the fixture."""
    import ctypes as C
    from maniac_mc_amd import fortran_host, run
    from maniac_mc_amd.engine import box_prepare, ewald_setup
    if not os.path.exists(fortran_host.LIB_PATH):
        pytest.skip("Fortran host library not built")
    d = os.path.join(RUNS, case, "inputs")
    cwd = os.getcwd()
    os.chdir(d)                                   # the fixtures were generated with relative file names
    try:
        system, inp, dat = io_maniac.load_system("system.maniac", "system.data", "system.inc", with_data=True)
        res = "reservoir.data" if SUMMARY[case]["reservoir"] else None
        rdat = io_maniac.read_lammps_data(res, inp) if res else None
        _, _, _, metrics = box_prepare(dat["matrix"])
        ew = ewald_setup(metrics, inp.real_space_cutoff, inp.ewald_tolerance)
        text = run.header_text(inp, dat, "system.maniac", "system.data", "system.inc", ew, res, rdat)
    finally:
        os.chdir(cwd)
    H = fortran_host.lib()
    H.mchain_set_log_header(text, C.c_int(len(text)))
    out = tmp_path / "header.txt"
    H.mchain_write_log_header(str(out).encode())
    H.mchain_set_log_header(b"", C.c_int(0))
    got = open(out).read().split("\n")
    want = open(os.path.join(RUNS, case, "expected", "log.maniac")).read().split("\n")
    stop = next(i for i, ln in enumerate(want) if "Started Monte Carlo Loop" in ln) - 2
    assert got[-1] == "" and got[:-1] == want[:stop], next(
        (i, a, b) for i, (a, b) in enumerate(zip(got + [None] * 400, want[:stop] + [None])) if a != b)
