#!/usr/bin/env python3
"""Stage times inside one farm-window launch (mgpu_farm_window_submit), measured by the kernel itself in a DIAGNOSTIC build of
the library (-DMGPU_FARM_STAMPS, built here into maniac_mc_amd/variants/; the shipped library carries no stamps).

    python tools/farm_stages.py [--chains 1,8,64,512] [--windows 300]

The 10 125-atom SPC/E box, random translations / rotations of every chain per window, one window at a time; stamps of chain 0's
k role, of the launch's first pair workgroup and of chain 0's resolver; medians in microseconds since the earliest stamp."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maniac_mc_amd import _lib, synth  # noqa: E402

VARIANT = os.path.join(ROOT, "maniac_mc_amd", "variants", "libmaniac_hip_farm_stamps.so")
NAMES = {(1, 0): "pair role (first workgroup): start", (1, 1): "pair role: records, Coulomb table staged, by-count completion",
         (1, 2): "pair role: candidates built, work units swept", (1, 3): "pair role: partials stored and acknowledged, barrier",
         (1, 4): "pair role: ticket drawn",
         (0, 0): "k role (chain 0): start", (0, 1): "k role: record in, current A(k) buffer known", (0, 2): "k role: candidate built",
         (0, 3): "k role: phase tables built", (0, 4): "k role: k sweep passed (A + delta stored)", (0, 5): "k role: energies reduced and stored",
         (0, 6): "k role: stores acknowledged, barrier", (0, 7): "k role: ticket drawn",
         (2, 0): "resolver (chain 0): started by the last ticket", (2, 1): "resolver: partials and k-role results loaded",
         (2, 2): "resolver: totals, exp, verdict", (2, 3): "resolver: eleven words + tag in host memory (acknowledged)",
         (2, 4): "resolver: accepted step committed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="1,8,64,512")
    ap.add_argument("--windows", type=int, default=300)
    a = ap.parse_args()
    if not os.path.exists(VARIANT) or os.path.getmtime(VARIANT) < max(os.path.getmtime(os.path.join(_lib.CSRC, f)) for f in os.listdir(_lib.CSRC)):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "farm_stamps", "-DMGPU_FARM_STAMPS"],
                              stdout=subprocess.DEVNULL)
    _lib.LIB_PATH = VARIANT
    from maniac_mc_amd.engine import Engine
    s = synth.spce_box(15)
    for R in [int(x) for x in a.chains.split(",")]:
        eng = Engine.from_system(s, n_replicas=R)
        eng.set_frames(0, 0, s.com[0], s.offsets[0])
        eng.init_structure_factor(0, True)
        for r in range(1, R):
            eng.replica_copy(r, 0)
        L = _lib.lib()
        L.mgpu_farm_window_get_stamps.restype = C.c_int
        rng = np.random.default_rng(3)
        rep = np.arange(R, dtype=np.int32)
        tt = np.zeros(R, np.int32)
        rows = []
        accepted_rows = []
        for w in range(a.windows + 20):
            m = rng.integers(0, int(s.n_mol[0]), R).astype(np.int32)
            move = rng.integers(1, 3, R).astype(np.int32)
            u = rng.uniform(0, 1, (R, 5))
            au = rng.uniform(0, 1, R)
            eng.farm_window_submit(rep, tt, m, move, u, 0.3, 0.3, au, np.ones(R), float(s.temperature))
            old, new, v = eng.farm_window_wait(R)
            st = np.zeros(24, dtype=np.int64)
            _lib.check(L.mgpu_farm_window_get_stamps(eng.h, st.ctypes.data_as(C.POINTER(C.c_longlong))))
            if w >= 20:
                rows.append(st.reshape(3, 8).copy())
                accepted_rows.append(int(v[0]) == 1)
        rows = np.array(rows, dtype=np.float64)
        t0 = np.minimum(rows[:, 0, 0], rows[:, 1, 0])[:, None, None]
        us = (rows - t0) / 100.0
        acc = np.array(accepted_rows)
        print(f"## {R} chain{'s' if R > 1 else ''} per window (engine nsplit {os.environ.get('MGPU_PAIR_NSPLIT', 'default')}), {len(rows)} windows, chain 0 accepted in {int(acc.sum())}\n")
        print("| stage | us since the launch's first stamp (median) |\n|---|---|")
        for (role, i), name in NAMES.items():
            col = us[:, role, i]
            if role == 2 and i == 4:
                col = col[acc]                     # (only an accepted step is committed)
            if col.size:
                print(f"| {name} | {np.median(col):.2f} |")
        print()
        eng.close()


if __name__ == "__main__":
    main()
