"""Accuracy of the Coulomb table the pair sweep reads from LDS (CPU test: mgpu_coulomb_table_eval builds
the same rows and evaluates them with the same index / Horner / FMA arithmetic on the host), against
50-digit mpmath.

What the path needs: each site-atom term is q1 q2 * 167101 K A * G, G = erfc(alpha r)/r, |q1 q2| <~ 1.
A relative error of ~2 ulp where the terms are large (alpha r < 1) and an absolute error that decays
with G elsewhere keep every term within ~1e-11 K of the reference's libm erfc -- far below the
5e-8 K (1e-10 kcal/mol) budget, and comparable to libm's own last-bit error.
"""
import ctypes as C

import numpy as np
import pytest

from maniac_mc_amd import _lib

mp = pytest.importorskip("mpmath")


def table_G(alpha, r2_max, r2):
    L = _lib.lib()
    r2 = np.ascontiguousarray(r2, dtype=np.float64)
    out = np.zeros_like(r2)
    _lib.check(L.mgpu_coulomb_table_eval(C.c_double(alpha), C.c_double(r2_max), C.c_int(r2.size),
                                         r2.ctypes.data_as(C.POINTER(C.c_double)),
                                         out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


@pytest.mark.parametrize("alpha,half_diag", [(0.2346370178899813, 40.4), (0.30995665486983587, 18.3), (0.16, 87.0)])
def test_coulomb_table_accuracy(alpha, half_diag):
    _lib.build()
    mp.mp.dps = 50
    rng = np.random.default_rng(1)
    r2_max = half_diag ** 2
    # log-uniform in r^2 over the whole table, every octave edge (both sides), and a dense patch at contact
    r2 = np.concatenate([np.exp(rng.uniform(np.log(0.25), np.log(r2_max), 12000)),
                         rng.uniform(0.8, 16.0, 6000),
                         [2.0 ** e for e in range(-2, int(np.log2(r2_max)) + 1)],
                         [np.nextafter(2.0 ** e, 0) for e in range(-1, int(np.log2(r2_max)) + 1)]])
    r2 = r2[r2 < r2_max]
    got = table_G(alpha, r2_max, r2)
    a = mp.mpf(alpha)
    exact = [mp.erfc(a * mp.sqrt(mp.mpf(float(s)))) / mp.sqrt(mp.mpf(float(s))) for s in r2]
    ref = np.array([float(e) for e in exact])
    abs_err = np.array([abs(float(mp.mpf(float(g)) - e)) for g, e in zip(got, exact)])
    x = alpha * np.sqrt(r2)
    rel = abs_err / ref
    assert np.max(rel[x < 1.0]) <= 4.5e-16, np.max(rel[x < 1.0])           # <= 2 ulp where terms are large (r < 4 A)
    # further out the two fp32 coefficients (c5, c6) show: a few ulp of an ever smaller G
    assert np.max(rel[x < 2.0]) <= 1.5e-15, np.max(rel[x < 2.0])
    # beyond alpha r = 1 what matters is the absolute error in Kelvin per unit charge product:
    # 167101 K A * |dG| stays below 1e-11 K
    assert np.max(abs_err[x >= 1.0]) * 167101.0 <= 1e-11, np.max(abs_err[x >= 1.0]) * 167101.0
    assert np.all(got > 0) or alpha * half_diag > 26


def test_coulomb_table_below_range_uses_direct_evaluation():
    alpha = 0.2346370178899813
    r2 = np.array([1e-6, 1e-3, 0.01, 0.2, 0.2499999])
    got = table_G(alpha, 1600.0, r2)
    mp.mp.dps = 40
    for g, s in zip(got, r2):
        e = mp.erfc(mp.mpf(alpha) * mp.sqrt(mp.mpf(float(s)))) / mp.sqrt(mp.mpf(float(s)))
        assert abs(float(mp.mpf(float(g)) - e) / float(e)) < 5e-16


def test_coulomb_table_argument_errors():
    with pytest.raises(_lib.MgpuError):
        table_G(-1.0, 100.0, np.array([1.0]))
    with pytest.raises(_lib.MgpuError):
        table_G(0.2, float("inf"), np.array([1.0]))
