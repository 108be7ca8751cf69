#!/usr/bin/env python3
"""Randomised comparison of the window farm with the batched farm (same seeds, same chains): chain counts, lanes, driver threads,
windows in flight, NVT / insertion-deletion (one or two active types) / framework boxes, and the device's undecided margin from its
default (nothing undecided) over 1e-3 (a few steps left to the driver at random places) to wide open (every step).  Counters, molecule
counts, running energies, coordinates and A(k) must agree bit for bit.  `tests/test_gpu_farm_window.py` holds the fixed cases; this
tool is the sweep behind them.

    python tools/window_farm_stress.py [--cases 40] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.fortran_host import FortranFarm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        kind = rng.choice(["spce", "co2", "mixture_gcmc", "mixture_nvt", "framework"])
        R = int(rng.integers(1, 41))
        lanes = int(rng.integers(1, 5))
        drivers = int(rng.integers(1, min(lanes, 3) + 1))
        depth = int(rng.integers(1, 5))
        margin = [None, None, 1e-3, 1e-2, 1e9][int(rng.integers(0, 5))]
        steps = int(rng.integers(5, 120))
        seed = int(rng.integers(1, 10 ** 6))
        kw = dict(seed=seed, n_threads=max(2, drivers), n_lanes=lanes, n_drivers=drivers, device_build=True)
        env = {}
        if kind == "spce":
            s = synth.spce_box(int(rng.integers(3, 7)), seed=seed)
            kw.update(translation_step=0.4, rotation_step=0.4)
        elif kind == "co2":
            s = synth.co2_box(int(rng.integers(2, 12)), seed=seed)
            kw.update(translation_step=1.0, rotation_step=0.6, mol_capacity=[int(rng.integers(12, 40))],
                      gcmc=dict(p_translation=float(rng.uniform(0, 0.4)), p_rotation=float(rng.uniform(0, 0.4)),
                                fugacity=float(rng.uniform(2.0, 30.0)) / 50.0 ** 3))
        elif kind == "mixture_gcmc":
            s = synth.mixture_box(seed=seed % 1000)
            kw.update(translation_step=0.4, rotation_step=0.4, mol_capacity=[int(rng.integers(13, 20)), int(rng.integers(10, 16))],
                      gcmc=dict(p_translation=0.2, p_rotation=0.2, fugacity=np.array([rng.uniform(5, 20), rng.uniform(3, 12)]) / (18.0 * 21.0 * 24.0)))
        elif kind == "mixture_nvt":
            s = synth.mixture_box(seed=seed % 1000)
            kw.update(translation_step=0.4, rotation_step=0.4)
        else:
            s = synth.framework_water_box(n_water=int(rng.integers(4, 16)), n_frame=int(rng.choice([72, 200, 300])), L=24.0, seed=seed % 1000)
            kw.update(translation_step=0.5, rotation_step=0.5, mol_capacity=[1, 40],
                      gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=float(rng.uniform(5, 30)) / 24.0 ** 3))
            env = {"MGPU_NO_FROZEN_BATCH": "1"}      # the batched path on the window's work units (same sums)
        os.environ.update(env)
        try:
            fa = FortranFarm(s, R, window=False, **kw)
            fb = FortranFarm(s, R, window=True, window_depth=depth, **kw)
        finally:
            for k in env:
                os.environ.pop(k, None)
        ok = fb.window
        what = ""
        if ok:
            if margin is not None:
                fb.eng.chain_set_margin(margin)
            for chunk in (steps, int(rng.integers(1, 9))):           # a run, then a short second run (state carried over)
                fa.run(chunk); fb.run(chunk)
            same = (fa.trials == fb.trials and fa.accepted == fb.accepted and fa.skipped == fb.skipped and fa.counters() == fb.counters()
                    and np.array_equal(fa.counts(), fb.counts()))
            for r in range(R):
                if not same:
                    break
                same = np.array_equal(fa.energy(r), fb.energy(r)) and np.array_equal(fa.eng.structure_factor(r), fb.eng.structure_factor(r))
                for t in fa.active:
                    same = same and np.array_equal(fa.eng.get_molecules(r, int(t)), fb.eng.get_molecules(r, int(t)))
            what = "same" if same else "DIFFERENT"
            bad += 0 if same else 1
        else:
            what = "window mode not available"
        print(f"case {case:3d} {kind:13s} chains {R:3d} lanes {lanes} drivers {drivers} in flight {depth} margin {margin} steps {steps:3d} "
              f"seed {seed:6d}: accepted {fa.accepted:6d} left to the driver {fb.window_mode()[2] if ok else 0:5d}  {what}", flush=True)
        fa.close(); fb.close()
    print(f"{a.cases} cases, {bad} different")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
