#!/usr/bin/env python3
"""Stage times inside one single-chain window launch (mgpu_chain_window), measured by the kernel itself.

    python tools/chain_stages.py [--windows 400] [--ks 1,4,8] [--cases spce,framework]

For each case (the 10 125-atom SPC/E box; the 2208-atom framework + 40 waters) and window size K: K random trial moves of
one chain per window, random acceptance draws, the kernel's stage stamps (mgpu_chain_set_timing / get_timing) collected
over many windows; medians in microseconds since the window's first workgroup started.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd._lib import MGPU_MOVE  # noqa: E402
from maniac_mc_amd.engine import Engine  # noqa: E402

NAMES = ["k role: start", "k role: phase tables built", "k role: k sweep summed", "k role: at the ticket",
         "pair role: start", "pair role: Coulomb table staged", "pair role: work units swept", "pair role: at the ticket",
         "resolver: last ticket drawn", "resolver: acquire fence passed", "resolver: partials reduced", "resolver: decided",
         "resolver: tag published (host sees the window)", "resolver: commit tables built", "resolver: commit done"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=400)
    ap.add_argument("--ks", default="1,4,8")
    ap.add_argument("--cases", default="spce,framework")
    a = ap.parse_args()
    print(f"HIP_FORCE_DEV_KERNARG={os.environ.get('HIP_FORCE_DEV_KERNARG', '(unset)')}\n")
    for case in a.cases.split(","):
        s = synth.spce_box(15) if case == "spce" else synth.framework_water_box()
        t_act = 0 if case == "spce" else 1
        eng = Engine.from_system(s, n_replicas=1)
        eng.init_structure_factor(0, True)
        eng.chain_set_timing(True)
        n1 = int(s.topo.atoms_in_res[t_act])
        rng = np.random.default_rng(1)
        T = float(s.temperature)
        for k in [int(x) for x in a.ks.split(",")]:
            rows, acc = [], 0
            for w in range(a.windows):
                nm = eng.num_molecules(0, t_act)
                m = rng.integers(0, nm, k).astype(np.int32)
                sites = eng.get_molecules(0, t_act)[m] + rng.uniform(-0.15, 0.15, (k, 1, 3)) if w % 50 == 0 else sites_cache[m] + rng.uniform(-0.15, 0.15, (k, 1, 3))
                if w % 50 == 0:
                    sites_cache = eng.get_molecules(0, t_act)
                _, _, first, und = eng.chain_window(0, np.full(k, t_act, np.int32), m, np.full(k, MGPU_MOVE, np.int32), sites[:, :n1],
                                                    rng.random(k), np.ones(k), T, 0.0)
                us = eng.chain_timing()
                if first >= 0:
                    acc += 1
                    sites_cache[m[first]] = sites[first]
                    rows.append(us)
            med = np.median(np.array(rows), axis=0)
            print(f"## {case}: N = {s.n_atoms}, K = {k} (engine nsplit {os.environ.get('MGPU_PAIR_NSPLIT', 'default')}), {len(rows)} windows with an accepted step of {a.windows}\n")
            print("| stage | us since the window's first workgroup started |\n|---|---|")
            for nme, v in zip(NAMES, med):
                print(f"| {nme} | {v:.2f} |")
            print()
        eng.close()


if __name__ == "__main__":
    main()
