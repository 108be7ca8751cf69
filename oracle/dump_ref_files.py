"""TEST INFRASTRUCTURE ONLY (oracle).  Run the reference's own front end on input files and dump what
it built.  One call per process: the reference allocates its state once (and `stop`s on bad input),
so tests invoke this script in a subprocess:

    python oracle/dump_ref_files.py <input.maniac> <topology.data> <parameters.inc> <outdir> <stage> <out.npz>

stage 1: ReadInput only (input-file fixtures); stage 2: full front end + ComputeSystemEnergy.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reflib  # noqa: E402

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def main():
    maniac, data, inc, outdir, stage, out = sys.argv[1:7]
    stage = int(stage)
    L = reflib.lib()
    L.ref_load_files.restype = C.c_int
    rc = L.ref_load_files(maniac.encode(), data.encode(), inc.encode(), outdir.encode(), C.c_int(stage))
    assert rc == 0
    n_res = C.c_int(); max_atom = C.c_int(); n_types = C.c_int()
    L.ref_get_sizes(C.byref(n_res), C.byref(max_atom), C.byref(n_types))
    n_res, max_atom = n_res.value, max_atom.value
    a = np.zeros(n_res, np.int32); tp = np.zeros(n_res, np.int32); act = np.zeros(n_res, np.int32)
    nm = np.zeros(n_res, np.int32); fug = np.zeros(n_res)
    L.ref_get_residues(a.ctypes.data_as(_ip), tp.ctypes.data_as(_ip), act.ctypes.data_as(_ip), fug.ctypes.data_as(_dp),
                       nm.ctypes.data_as(_ip))
    inp = np.zeros(12)
    L.ref_get_input(inp.ctypes.data_as(_dp))
    d = dict(atoms_in_res=a, types_per_res=tp, is_active=act, fugacity=fug, input=inp, max_atom=max_atom)
    if stage > 1:
        d["n_atom_types"] = n_types.value
        d["n_mol"] = nm
        types = np.zeros((n_res, max_atom), np.int32); charges = np.zeros((n_res, max_atom))
        for t in range(n_res):
            L.ref_get_template(C.c_int(t + 1), types[t].ctypes.data_as(_ip), charges[t].ctypes.data_as(_dp))
        d["atom_types"] = types
        d["charges"] = charges
        m = np.zeros(9); lo = np.zeros(3); hi = np.zeros(3)
        L.ref_get_box_matrix(m.ctypes.data_as(_dp), lo.ctypes.data_as(_dp), hi.ctypes.data_as(_dp))
        d["box_matrix"] = m.reshape(3, 3).T.copy(); d["bounds_lo"] = lo; d["bounds_hi"] = hi
        for t in range(n_res):
            com = np.zeros((nm[t], 3)); off = np.zeros((nm[t], a[t], 3))
            for k in range(nm[t]):
                c = np.zeros(3); o = np.zeros((max_atom, 3))
                L.ref_get_molecule(C.c_int(t + 1), C.c_int(k + 1), c.ctypes.data_as(_dp), o.ctypes.data_as(_dp))
                com[k] = c; off[k] = o[: a[t]]
            d[f"com{t}"] = com; d[f"off{t}"] = off
        # per-site-pair epsilon / sigma, flattened over (t1, a1, t2, a2)
        eps = {}; sig = {}
        tab_e = np.zeros((n_types.value, n_types.value)); tab_s = np.zeros((n_types.value, n_types.value))
        seen = np.zeros((n_types.value, n_types.value), bool)
        e_ = C.c_double(); s_ = C.c_double()
        for t1 in range(n_res):
            for a1 in range(a[t1]):
                for t2 in range(n_res):
                    for a2 in range(a[t2]):
                        L.ref_get_coeff(C.c_int(t1 + 1), C.c_int(a1 + 1), C.c_int(t2 + 1), C.c_int(a2 + 1), C.byref(e_), C.byref(s_))
                        i, j = types[t1, a1] - 1, types[t2, a2] - 1
                        if seen[i, j]:
                            assert tab_e[i, j] == e_.value and tab_s[i, j] == s_.value   # purely by atom type
                        tab_e[i, j] = e_.value; tab_s[i, j] = s_.value; seen[i, j] = True
        d["epsilon"] = tab_e; d["sigma"] = tab_s; d["coeff_seen"] = seen
        alpha = C.c_double(); rc_ = C.c_double(); tol = C.c_double(); scr = C.c_double(); fp = C.c_double()
        kmax = np.zeros(3, np.int32); nk = C.c_int()
        L.ref_get_ewald(C.byref(alpha), C.byref(rc_), C.byref(tol), C.byref(scr), C.byref(fp), kmax.ctypes.data_as(_ip), C.byref(nk))
        d.update(alpha=alpha.value, rc_eff=rc_.value, tol_eff=tol.value, kmax=kmax, nk=nk.value)
        for kind, key in ((1, "bonds"), (2, "angles"), (3, "dihedrals"), (4, "impropers")):
            for t in range(n_res):
                n = C.c_int(); ntyp = C.c_int(); tab = np.zeros((64, 5), np.int32)
                L.ref_get_bonded(C.c_int(kind), C.c_int(t + 1), C.byref(n), tab.ctypes.data_as(_ip), C.byref(ntyp))
                d[f"{key}_{t}"] = tab[: min(n.value, 64)].copy()
                d[f"{key}_types"] = ntyp.value
        e6 = np.zeros(6)
        L.ref_system_energy(e6.ctypes.data_as(_dp))
        d["system_energy"] = e6
    np.savez_compressed(out, **d)
    print("DUMP_OK")


if __name__ == "__main__":
    main()
