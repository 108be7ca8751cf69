#!/usr/bin/env python3
"""Long-run consistency check of the farm at the benchmark size (not part of the test-suite: minutes of GPU time).

    python tools/long_run_check.py [--replicas 512] [--lanes 4] [--steps 4000] [--gcmc]
After the run every sampled chain's running energies must equal a from-scratch evaluation of its final configuration
and its A(k) a fresh S(k); prints the worst deviations."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.fortran_host import FortranFarm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=512)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--gcmc", action="store_true", help="CO2 in the 50 A box with insertion / deletion instead of SPC/E NVT")
    ap.add_argument("--host-build", action="store_true", help="trial moves built by the Fortran driver from its mirror (default: on the device)")
    ap.add_argument("--device-accept", action="store_true", help="the k sweep applies the acceptance rule and commits (mfarm_configure(2))")
    ap.add_argument("--framework", action="store_true", help="4-site water in the 2208-atom framework, full move set (with --gcmc semantics)")
    ap.add_argument("--window", type=int, default=0, help="> 0: window mode (one launch per lane step) with this many windows of a lane in flight")
    ap.add_argument("--drivers", type=int, default=1, help="driver threads")
    a = ap.parse_args()
    if a.framework:
        s = synth.framework_water_box()
        V = float(abs(np.linalg.det(s.box_matrix)))
        farm = FortranFarm(s, a.replicas, seed=5, translation_step=0.5, rotation_step=0.5, n_threads=4, n_lanes=a.lanes,
                           mol_capacity=[1, 200], gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=40.0 / V),
                           device_build=not a.host_build, device_accept=a.device_accept, window=a.window > 0, window_depth=max(1, a.window), n_drivers=a.drivers)
        keys = ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")
    elif a.gcmc:
        s = synth.co2_box(64, seed=13)
        V = 50.0 ** 3
        fug = np.geomspace(20.0, 160.0, 8)[np.arange(a.replicas) % 8] / V
        farm = FortranFarm(s, a.replicas, seed=5, translation_step=1.0, rotation_step=0.6, n_threads=8, n_lanes=a.lanes,
                           mol_capacity=[400], gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=fug),
                           device_build=not a.host_build, device_accept=a.device_accept, window=a.window > 0, window_depth=max(1, a.window), n_drivers=a.drivers)
        keys = ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb")
    else:
        s = synth.spce_box(15)
        farm = FortranFarm(s, a.replicas, seed=5, translation_step=0.3, rotation_step=0.3, n_threads=8, n_lanes=a.lanes,
                           device_build=not a.host_build, device_accept=a.device_accept, window=a.window > 0, window_depth=max(1, a.window), n_drivers=a.drivers)
        keys = ("non_coulomb", "coulomb", "recip_coulomb")
    t0 = time.perf_counter()
    acc = farm.run(a.steps)
    el = time.perf_counter() - t0
    eng = farm.eng
    worst_e, worst_a, big = 0.0, 0.0, 0.0
    sample = sorted(set(np.linspace(0, a.replicas - 1, 12).astype(int).tolist()))
    for r in sample:
        e = eng.system_energy(r)
        run = farm.energy(r)
        worst_e = max(worst_e, max(abs(run[i] - e[k]) for i, k in enumerate(keys)))
        big = max(big, max(abs(e[k]) for k in keys))
        A = eng.structure_factor(r)
        eng.init_structure_factor(r, True)
        worst_a = max(worst_a, float(np.max(np.abs(A - eng.structure_factor(r)))))
    mode = f"windows, {farm.window_mode()[1]} in flight, {a.drivers} driver(s), {farm.window_mode()[2]} steps left to the driver" if farm.window else "batched"
    print(f"{'framework + water GCMC' if a.framework else ('GCMC CO2' if a.gcmc else 'SPC/E 10125 atoms')} ({'host' if a.host_build else 'device'}-built moves, "
          f"{'device' if a.device_accept else 'host'} rule; {mode}): {a.replicas} chains x {a.steps} steps on {a.lanes} lanes, "
          f"{farm.trials} trials, {acc} accepted in {el:.1f} s ({acc / el:.3e} accepted/s); over {len(sample)} sampled chains: "
          f"max |running - recomputed energy| = {worst_e:.3e} K, max |A - S(k)| = {worst_a:.3e}, largest |E| = {big:.3e} K")
    farm.close()
    ok = worst_e < 1e-6 and worst_a < 1e-9
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
