"""CPU-only checks of the product library: it loads, exports every symbol include/maniac_gpu.h
declares, and its host-side setup arithmetic (box products, SetupEwald, k table) reproduces the
reference's values bit for bit.  No compute entry point is called here (there is no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from maniac_mc_amd import _lib
from tests.util import GOLDEN_FULL, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    _lib.build()
    return _lib.lib()


def test_exports_match_header(hiplib):
    header = open(os.path.join(ROOT, "include", "maniac_gpu.h")).read()
    declared = set(re.findall(r"\b(mgpu_[a-z_0-9]+)\s*\(", header))
    declared.discard("mgpu_engine")
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(hiplib, name), name
    m = re.search(r"#define MGPU_ABI_VERSION (\d+)", header)
    assert m and hiplib.mgpu_abi_version() == int(m.group(1)) == _lib.ABI_VERSION


@pytest.mark.parametrize("name", GOLDEN_FULL + ["spce1000_scalars", "spce3375_scalars", "framework2208_scalars"])
def test_host_setup_bitwise_vs_reference(name, hiplib):
    from maniac_mc_amd import engine
    g = load_golden(name)
    if "box_matrix" in g:
        bm = g["box_matrix"]
    else:
        L = float(g["metrics"][0])
        bm = np.diag([L, L, L])
    bt, vol, rcp, met = engine.box_prepare(bm)
    assert bt == int(g["box_type"]) and vol == float(g["volume"])
    assert np.array_equal(rcp, g["reciprocal"]) and np.array_equal(met, g["metrics"])
    ew = engine.ewald_setup(met, float(g["rc_in"]), float(g["tol_in"]))
    assert ew["alpha"] == float(g["alpha"]) and ew["rc"] == float(g["rc_eff"]) and ew["tol"] == float(g["tol_eff"])
    assert np.array_equal(ew["kmax"], g["kmax"]) and ew["nk"] == int(g["nk"])
    kv = engine.ewald_kvectors(rcp, ew["alpha"], ew["kmax"], ew["nk"])
    for k in ("kx", "ky", "kz", "k2mag", "form_factor", "weights"):
        assert np.array_equal(kv[k], g["k_" + k]), k


def test_setup_edge_cases(hiplib):
    from maniac_mc_amd import engine
    # cutoff larger than the box is halved to min(L)/2 (prepare_utils.f90:141-150)
    bt, vol, rcp, met = engine.box_prepare(np.diag([10.0, 12.0, 14.0]))
    assert bt == 2
    ew = engine.ewald_setup(met, 11.0, 1e-5)
    assert ew["rc"] == 5.0
    # tolerance is clamped to |tol| <= 0.5 (prepare_utils.f90:157-160)
    assert engine.ewald_setup(met, 4.0, -0.9)["tol"] == 0.5
    # triclinic boxes are recognised (type 3)
    assert engine.box_prepare(np.array([[10.0, 0, 0], [1.0, 10.0, 0], [0, 0, 10.0]]))[0] == 3
    # degenerate box: error code, not a crash
    with pytest.raises(_lib.MgpuError):
        engine.box_prepare(np.diag([0.5, 0.5, 0.5]))
    with pytest.raises(_lib.MgpuError):
        engine.ewald_kvectors(rcp, ew["alpha"], ew["kmax"], ew["nk"] - 1)


def test_engine_create_fails_loudly_without_gpu(hiplib):
    """On a box without a HIP device the product path must raise, never fall back."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from maniac_mc_amd import engine, synth
    with pytest.raises(_lib.MgpuError) as ei:
        engine.Engine.from_system(synth.argon_box())
    assert ei.value.code in (2, 4)


def test_product_path_does_not_import_oracle():
    """maniac_mc_amd/ must never import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "maniac_mc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".f90")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("# oracle-free", ""), os.path.join(dirpath, f)


def test_fortran_host_library_exports():
    """libmaniac_host.so (Fortran: farm driver, single-chain driver, writers) loads without a GPU and exports the
    bind(C) entry points the Python plumbing calls; no compute is attempted."""
    import shutil
    if shutil.which("amdflang") is None:
        pytest.skip("amdflang not available")
    from maniac_mc_amd import fortran_host
    H = fortran_host.lib()
    for name in ("mfarm_create", "mfarm_set_gcmc", "mfarm_set_triclinic", "mfarm_run", "mfarm_destroy", "mfarm_get_energy",
                 "mfarm_get_molecule", "mfarm_get_counts", "mfarm_get_counters", "mfarm_get_timers", "mfarm_recalibrate",
                 "mchain_reset", "mchain_set_box", "mchain_set_residue", "mchain_set_bonded", "mchain_set_tables",
                 "mchain_set_moves", "mchain_set_reservoir_box", "mchain_set_reservoir_residue", "mchain_run",
                 "mchain_set_mode", "mchain_get_energy", "mchain_get_counters", "mchain_get_counts", "mchain_get_steps", "mchain_get_molecule"):
        assert hasattr(H, name), name


def test_header_is_plain_c(tmp_path):
    """include/maniac_gpu.h is the C ABI: it must compile as C99 (no C++-isms), and a C caller must link."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "caller.c"
    src.write_text('#include "maniac_gpu.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    int n = -1;\n'
                   '    int rc = mgpu_device_count(&n);\n'
                   '    printf("abi %d rc %d devices %d lanes %d\\n", mgpu_abi_version(), rc, n, MGPU_LANES);\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "caller"
    lib = os.path.join(root, "maniac_mc_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                           "-L" + lib, "-lmaniac_hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("abi "), out.stdout + out.stderr


def test_chain_generators_are_seeded_independently():
    """Per-chain xoshiro256+ states come from splitmix64 streams (mgpu_rng_seed_streams); the first outputs of
    neighbouring chains must be uncorrelated, uniform, and reproducible for a given seed."""
    import ctypes as C
    from maniac_mc_amd import _lib, fortran_host
    L = _lib.lib()
    st = np.zeros(4 * 64, dtype=np.int64)
    assert L.mgpu_rng_seed_streams(C.c_longlong(7), C.c_int(64), st.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
    st = st.reshape(64, 4)
    assert len({tuple(r) for r in st.tolist()}) == 64 and not np.any(np.all(st == 0, axis=1))
    # the recipe itself: splitmix64 of the seed, then per stream base + odd * (r + 1), four outputs
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9e3779b97f4a7c15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
        return x, z ^ (z >> 31)
    _, base = sm(7)
    x = (base + 0xd1342543de82ef95 * 3) & M
    want = []
    for _ in range(4):
        x, z = sm(x)
        want.append(z - (1 << 64) if z >= (1 << 63) else z)
    assert st[2].tolist() == want
    if not os.path.exists(fortran_host.LIB_PATH):
        pytest.skip("Fortran host library not built")
    H = fortran_host.lib()
    n, per = 256, 1500
    u = np.zeros((n, per))
    H.mfarm_rng_sample(C.c_int(2024), C.c_int(n), C.c_int(per), u.ctypes.data_as(C.POINTER(C.c_double)))
    u2 = np.zeros((n, per))
    H.mfarm_rng_sample(C.c_int(2024), C.c_int(n), C.c_int(per), u2.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(u, u2) and 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.003 and abs(u.var() * 12 - 1.0) < 0.02
    c = np.corrcoef(u[:96])
    np.fill_diagonal(c, 0.0)
    assert np.abs(c).max() < 5.0 / np.sqrt(per)                     # ~4.3 sigma over 4560 pairs
    # the very first draws of neighbouring chains (what a linear seeding rule would line up)
    first = u[:, 0]
    assert abs(np.corrcoef(first[:-1], first[1:])[0, 1]) < 0.2
    # the overflow-free form of (s1 + s4) >> 11: cross-check the first number of chain 0 in exact integer arithmetic
    s = np.zeros(4, dtype=np.int64)
    L.mgpu_rng_seed_streams(C.c_longlong(2024), C.c_int(1), s.ctypes.data_as(C.POINTER(C.c_longlong)))
    s1, s4 = int(s[0]) & M, int(s[3]) & M
    assert u[0, 0] == (((s1 + s4) & M) >> 11) / 9007199254740992.0


def test_oracle_all_core_trial_farm_runs_the_same_trial(refcpu_mod):
    """bench.py's all-core CPU leg: independent chains of the restated sequential trial under OpenMP."""
    from maniac_mc_amd import synth
    el, tr, ac = refcpu_mod.trial_farm(synth.spce_box(5, seed=2), 2, 2, 0.6, 0.3, 0.3)     # one chain per thread
    assert 0.5 < el < 5.0 and tr.shape == (2,) and np.all(tr > 20) and np.all(ac <= tr)
    assert 0.4 < ac.sum() / tr.sum() < 0.95


def test_atom_record_formatter_is_the_fortran_runtimes(tmp_path):
    """mgpu_append_atom_records formats the atom records of trajectory.lammpstrj / topology.data outside the Fortran
    runtime (whose Fw.d conversion dominated a single chain's run time at 10 125 atoms).  Its conversion is exact --
    integer mantissa x 10^d in 128-bit arithmetic, round half to even -- and must be the runtime's bytes: random values
    over twenty decades, exact ties of the last printed digit, signed zeros, values that round up into the next width and
    values that overflow it (asterisks), in the three widths the writers use (src/write_utils.f90:86, :297-300)."""
    import ctypes as C
    import shutil
    if not shutil.which("amdflang"):
        pytest.skip("needs the Fortran runtime (amdflang) to compare with")
    from maniac_mc_amd import _lib, fortran_host
    L, H = _lib.lib(), fortran_host.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.uniform(-1, 1, 60000) * 10.0 ** rng.integers(-12, 8, 60000),
        rng.integers(-4000000, 4000000, 20000) / 256.0, rng.integers(-4000000, 4000000, 20000) / 512.0,      # ties of F12.7 / F12.8
        rng.integers(-400, 400, 2000) / 256.0 * 1e-3,
        [0.0, -0.0, 1e-9, -1e-9, 5e-8, -5e-8, 4.9999999e-8, 0.5, -0.5, 999.99999995, 9999.99999995, -999.99999995, 9999.9999999,
         -999.9999999, 99999.0, 123456.7891234, 1e15, -1e15, 1e300, 4.5e15, 2.0 ** 52, 2.0 ** 53, 1e-300, 5e-324],
    ])
    n = len(x)
    dp = C.POINTER(C.c_double)
    for w, d in ((12, 7), (12, 8), (15, 8)):
        mine = C.create_string_buffer(n * w)
        ref = C.create_string_buffer(n * w)
        _lib.check(L.mgpu_format_fixed(C.c_int(n), x.ctypes.data_as(dp), C.c_int(w), C.c_int(d), mine))
        H.mout_format_fixed(C.c_int(n), x.ctypes.data_as(dp), C.c_int(w), C.c_int(d), ref)
        a = np.frombuffer(mine.raw, dtype=f"S{w}")
        b = np.frombuffer(ref.raw, dtype=f"S{w}")
        bad = np.flatnonzero(a != b)
        assert len(bad) == 0, [(x[i], a[i], b[i]) for i in bad[:10]]
    # ... and whole records, both layouts, against formatted writes of the same numbers
    m = 500
    xyz = np.ascontiguousarray(rng.uniform(-40, 40, (m, 3)))
    ty = rng.integers(1, 12, m).astype(np.int32)
    mol = rng.integers(1, 4000, m).astype(np.int32)
    q = np.ascontiguousarray(rng.uniform(-1.2, 1.2, m))
    ip = C.POINTER(C.c_int)
    p1, p2 = str(tmp_path / "traj"), str(tmp_path / "topo")
    open(p1, "w").write("HEADER\n")
    _lib.check(L.mgpu_append_atom_records(p1.encode(), C.c_int(m), C.c_int(1), None, ty.ctypes.data_as(ip), None, xyz.ctypes.data_as(dp)))
    _lib.check(L.mgpu_append_atom_records(p2.encode(), C.c_int(m), C.c_int(7), mol.ctypes.data_as(ip), ty.ctypes.data_as(ip),
                                          q.ctypes.data_as(dp), xyz.ctypes.data_as(dp)))
    want1 = "HEADER\n" + "".join("%6d %4d %12.7f %12.7f %12.7f\n" % (i + 1, ty[i], *xyz[i]) for i in range(m))
    want2 = "".join("%6d %6d %4d %12.8f %12.7f %12.7f %12.7f\n" % (i + 7, mol[i], ty[i], q[i], *xyz[i]) for i in range(m))
    assert open(p1).read() == want1 and open(p2).read() == want2
    bad = xyz.copy(); bad[3, 1] = np.nan
    assert L.mgpu_append_atom_records(p1.encode(), C.c_int(m), C.c_int(1), None, ty.ctypes.data_as(ip), None, bad.ctypes.data_as(dp)) != 0
    assert open(p1).read() == want1                       # nothing was written by the failed call


def test_rng_fill_is_the_scalar_generator(hiplib):
    """mgpu_rng_fill (four xoshiro256+ streams abreast, AVX2 where the CPU has it) against the recurrence written out in
    Python integers -- the numbers of mc_farm.f90's chain_random: top 53 bits of s0 + s3 times 2^-53 -- for stream counts
    around the vector width, and the states it leaves behind."""
    import ctypes as C
    M = (1 << 64) - 1
    for n in (1, 3, 4, 7, 33):
        st = np.zeros((n, 4), dtype=np.int64)
        assert hiplib.mgpu_rng_seed_streams(C.c_longlong(1234), C.c_int(n), st.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
        ref_state = [[int(x) & M for x in row] for row in st]
        per = 11
        out = np.zeros((n, per))
        assert hiplib.mgpu_rng_fill(st.ctypes.data_as(C.POINTER(C.c_longlong)), C.c_int(n), C.c_int(per),
                                    out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        for r in range(n):
            s0, s1, s2, s3 = ref_state[r]
            for i in range(per):
                assert out[r, i] == float(((s0 + s3) & M) >> 11) * 2.0 ** -53, (n, r, i)
                t = (s1 << 17) & M
                s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3
                s2 ^= t
                s3 = ((s3 << 45) | (s3 >> 19)) & M
            assert [int(x) & M for x in st[r]] == [s0, s1, s2, s3], (n, r)
