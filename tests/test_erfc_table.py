"""Accuracy of the erfc table the pair sweep reads from LDS (CPU test: mgpu_erfc_table_eval runs the
same rows with the same Horner/FMA order on the host), against 50-digit mpmath.

What the path needs: each site-atom term is q1 q2 erfc(alpha r)/r * 167101 K A, |q1 q2| <~ 1, r >~ 1 A.
A relative error of a few ulp where erfc ~ 1 and an ABSOLUTE error ~1e-17 further out keep every term
within ~1e-11 K of the reference's libm erfc, far below the 5e-8 K (1e-10 kcal/mol) budget.
"""
import ctypes as C

import numpy as np
import pytest

from maniac_mc_amd import _lib

mp = pytest.importorskip("mpmath")


def table_erfc(x):
    L = _lib.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    _lib.check(L.mgpu_erfc_table_eval(C.c_int(x.size), x.ctypes.data_as(C.POINTER(C.c_double)),
                                      out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def test_erfc_table_accuracy():
    _lib.build()
    mp.mp.dps = 50
    rng = np.random.default_rng(0)
    # dense random sample plus every interval boundary (both sides)
    edges = np.arange(0, 12 * 32 + 1) / 32.0
    x = np.concatenate([rng.uniform(0, 12.5, 20000), edges, np.nextafter(edges[1:], 0), [0.0, 5e-324, 1e-300, 1e-8]])
    got = table_erfc(x)
    ref = np.array([float(mp.erfc(mp.mpf(float(v)))) for v in x])
    exact = [mp.erfc(mp.mpf(float(v))) for v in x]
    abs_err = np.array([abs(float(mp.mpf(float(g)) - e)) for g, e in zip(got, exact)])
    rel_err = abs_err / np.maximum(ref, 1e-300)
    assert np.max(abs_err) <= 1.2e-16, np.max(abs_err)            # < 1 ulp of erfc ~ 1
    assert np.max(rel_err[x < 2.0]) <= 4.5e-16, np.max(rel_err[x < 2.0])    # 2 ulp where terms are large
    assert np.max(abs_err[x >= 2.0]) <= 2e-18, np.max(abs_err[x >= 2.0])
    # tail: erfc(x >= 12) < 1.4e-64 is returned as exactly 0
    assert np.all(table_erfc(np.array([12.0, 13.7, 40.0, 1e6, 1e300])) == 0.0)
    # monotone and within (0, 1]
    xs = np.linspace(0, 11.99, 4001)
    ys = table_erfc(xs)
    assert ys[0] == 1.0 and np.all(np.diff(ys) <= 0) and np.all(ys > 0)


def test_erfc_table_rejects_negative():
    with pytest.raises(_lib.MgpuError):
        table_erfc(np.array([-0.1]))
