"""Whole-run parity: the Fortran chain driver on the engine (B = 1 seams of
maniac_gpu.f90) against the reference's own MonteCarloLoop on the same input files and seed.  The chain
draws its random numbers in the reference's order, so it must visit the same states: the output files --
energy.dat, moves.dat, number_<res>.dat, trajectory.lammpstrj, topology.data, reservoir.lammpstrj and the WHOLE
log.maniac (banner, input echo, data-file summary, Lorentz-Berthelot listing, Ewald parameters, status table, final
report; only the output directory printed in the closing box is blanked) -- are compared with the files the reference wrote
(tests/golden/runs/*/expected, made by tests/golden/make_run_fixtures.py), character for character.
The charged grand-canonical cases (summary "as_written": co2_gcmc = BASELINE.json configs[2] with Nk = 2975,
framework_water_gcmc = configs[3] in miniature) are the reference's files WITH its deletion defect;
the chain driver reproduces them in its as-written mode, which the host loop composes from neutral engine
primitives.  In the default (intended-physics) mode the same inputs must instead end with running energies that
equal a from-scratch evaluation -- the property the reference's own charged GCMC runs do not have.
"""
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, TOL_K

RUNS = os.path.join(GOLDEN, "runs")
SUMMARY = json.load(open(os.path.join(RUNS, "summary.json")))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["speculate8", "chain1", "batched8", "fused", "seams"])
@pytest.mark.parametrize("case", sorted(SUMMARY))
def test_run_writes_the_reference_files(case, mode, tmp_path):
    """mode: speculate8 = windows of 8 steps, ONE kernel launch per window that also decides and commits (the default of
    run.py; since round 5 for the triclinic case too); chain1 = the same with windows of one step;
    batched8 = windows of 8 through the batched submit / wait calls, the rule and the commit in the host loop; fused = one
    batched call per step; seams = one call per reference seam.  All five must write the reference's files."""
    seams = mode == "seams"
    from maniac_mc_amd import run
    inputs = os.path.join(RUNS, case, "inputs")
    expected = os.path.join(RUNS, case, "expected")
    out = str(tmp_path / "out") + "/"
    reservoir = "reservoir.data" if SUMMARY[case]["reservoir"] else None
    as_written = bool(SUMMARY[case].get("as_written"))
    cwd = os.getcwd()
    os.chdir(inputs)                    # the log echoes the file names as given: the fixtures used relative ones
    try:
        res = run.run_simulation("system.maniac", "system.data", "system.inc", out, seed=SUMMARY[case]["seed"],
                                 reservoir_path=reservoir, seams=seams, as_written=as_written,
                                 speculate=8 if mode in ("speculate8", "batched8") else 1,
                                 chain_windows=mode in ("speculate8", "chain1"))
        if mode in ("speculate8", "chain1"):
            assert res["chain_windows"][0] > 0
            assert res["chain_windows"][1] == 0          # no step fell inside the 16-ulp margin
    finally:
        os.chdir(cwd)
    # running energies of the chain == a full recomputation of the final configuration (as written, A(k) carries
    # the terms of deleted molecules, so the reciprocal energy and the total are exempt there)
    for k, v in res["energy"].items():
        if as_written and k in ("recip_coulomb", "total"):
            continue
        assert abs(v - res["recomputed_energy"][k]) <= 1e-9 * max(1.0, abs(v)) + 50 * TOL_K, k
    produced = sorted(os.listdir(out))
    assert produced == sorted(SUMMARY[case]["files"]) and "log.maniac" in produced
    for f in SUMMARY[case]["files"]:
        want = open(os.path.join(expected, f)).read().split("\n")
        got = open(os.path.join(out, f)).read().split("\n")
        if f == "log.maniac":
            got = ["<output path>" if out.rstrip("/") in ln else ln for ln in got]
        assert len(got) == len(want), f
        bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        assert not bad, f"{f}: first differing line {bad[0] + 1}: {got[bad[0]]!r} vs {want[bad[0]]!r} ({len(bad)} lines differ)"


@pytest.mark.parametrize("case", sorted(c for c in SUMMARY if SUMMARY[c].get("as_written")))
def test_intended_physics_on_the_charged_gcmc_inputs(case, tmp_path):
    """Default mode on the same charged grand-canonical inputs: deletions remove the molecule's terms from A(k), so
    the running energies (reciprocal part included) equal a from-scratch evaluation of the final configuration,
    and the trajectory leaves the reference's as-written one."""
    from maniac_mc_amd import run
    inputs = os.path.join(RUNS, case, "inputs")
    out = str(tmp_path / "out") + "/"
    res = run.run_simulation(os.path.join(inputs, "system.maniac"), os.path.join(inputs, "system.data"),
                             os.path.join(inputs, "system.inc"), out, seed=SUMMARY[case]["seed"])
    for k, v in res["energy"].items():
        assert abs(v - res["recomputed_energy"][k]) <= 1e-9 * max(1.0, abs(v)) + 50 * TOL_K, k
    assert res["counters"][5] > 0                                   # deletions were accepted
    want = open(os.path.join(RUNS, case, "expected", "energy.dat")).read()
    assert open(os.path.join(out, "energy.dat")).read() != want
