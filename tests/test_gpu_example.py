"""examples/seams_demo.f90: a stand-alone Fortran program binding the engine through module maniac_gpu (the
integration of INTEGRATION.md in miniature).  Build it with amdflang, run it on a seeded SPC/E box and compare
what it prints -- system energies, old / new energies of one translation trial through the reference-named
seams, system energies after the accepted move -- with the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from maniac_mc_amd import synth
from tests.util import TOL_K, tol_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fortran_seams_demo(tmp_path, refcpu_mod):
    if shutil.which("amdflang") is None:
        pytest.skip("amdflang not available")
    from maniac_mc_amd import fortran_host
    fortran_host.build()
    exe = str(tmp_path / "seams_demo")
    lib = os.path.join(ROOT, "maniac_mc_amd")
    subprocess.check_call(["amdflang", "-O2", "-fopenmp", os.path.join(ROOT, "examples", "seams_demo.f90"),
                           "-I" + os.path.join(ROOT, "build", "fmod"), "-L" + lib, "-lmaniac_host", "-lmaniac_hip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-o", exe], cwd=str(tmp_path))
    s = synth.spce_box(5, seed=8, rc=7.0)
    n = int(s.n_mol[0])
    m, disp = 17, np.array([0.21, -0.13, 0.08])
    cfg = tmp_path / "config.txt"
    with open(cfg, "w") as f:
        f.write(f"{n} {float(s.box_matrix[0, 0])!r} {s.real_space_cutoff!r} {s.ewald_tolerance!r}\n")
        for i in range(n):
            f.write(" ".join(repr(float(v)) for v in s.com[0][i]) + " " +
                    " ".join(repr(float(v)) for v in s.offsets[0][i].reshape(-1)) + "\n")      # off(:, a) = site a
        f.write(f"{m + 1} " + " ".join(repr(float(v)) for v in disp) + "\n")
    out = subprocess.run([exe, str(cfg)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {ln.split()[0]: np.array([float(v) for v in ln.split()[1:]]) for ln in out.stdout.splitlines() if ln.strip()}
    # the demo's force field is the same SPC/E as synth.spce_topology: check against the oracle
    P = refcpu_mod.RefCPU(s)
    e = P.system_energy()
    P.init_amplitude(True)
    keys = ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb", "total")

    def close(a, b, what):
        for x, y in zip(a, b):
            assert abs(x - y) <= tol_for(x, y), (what, x, y)
    close(got["system"], [e[k] for k in keys], "system")
    com, off = P.get_molecule(0, m)
    P.save_fourier(0, m)
    old = P.old_energy(0, m, 0)
    P.set_molecule(0, m, com + disp, off)
    new = P.new_energy(0, m, 0)
    close(got["old"], old[:3], "old")
    close(got["new"], new[:3], "new")
    e2 = P.system_energy()
    close(got["after"], [e2[k] for k in keys], "after")
    assert abs((new[5] - old[5]) - (got["after"][5] - got["system"][5])) < 50 * TOL_K
