#!/usr/bin/env python3
"""Kernel-only timings (HIP events inside the library) of the hot-path kernels on the benchmark box.

    python tools/bench_kernels.py [--workload spce|co2_gcmc|framework_water] [--replicas 1024] [--reps 10]

One launch group per repetition, shaped like one lane step of bench.py's farm for that workload:
  spce             R trial moves (translation-sized displacements) of the 10 125-atom SPC/E box: fused old + new pair
                   sweep, k sweep, commit of ~70 %
  co2_gcmc         R insertions / deletions (50 / 50) of rigid CO2 in the 50 A box (BASELINE.json configs[2])
  framework_water  R trials of the full move set (25 % translation, 25 % rotation, 25 % insertion, 25 % deletion) of
                   4-site water in the 2208-atom framework (configs[3])
This is also the command the rocprofv3 --pmc passes profile (tools/pmc_passes.sh).
Environment knob read by the library: MGPU_PAIR_NSPLIT.
MANIAC_HIP_LIB selects an alternative build of libmaniac_hip.so (tuning variants).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import _lib  # noqa: E402

if os.environ.get("MANIAC_HIP_LIB"):
    _lib.LIB_PATH = os.environ["MANIAC_HIP_LIB"]
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.engine import Engine  # noqa: E402


def rotate_about_com(sites, rng, angle):
    """rigid rotation of every row's molecule by a random angle in [-angle/2, angle/2] about a random Cartesian axis"""
    com = sites.mean(axis=1, keepdims=True)
    off = sites - com
    out = np.empty_like(sites)
    for i in range(sites.shape[0]):
        th = (rng.random() - 0.5) * angle
        ax = int(rng.integers(0, 3))
        p, q = (ax + 1) % 3, (ax + 2) % 3
        c, s = np.cos(th), np.sin(th)
        o = off[i].copy()
        o[:, p] = c * off[i, :, p] - s * off[i, :, q]
        o[:, q] = s * off[i, :, p] + c * off[i, :, q]
        out[i] = com[i] + o
    return out


def build(workload, R, n_side):
    """(system, engine, active type) with every replica loaded and A(k) initialised"""
    if workload == "spce":
        s = synth.spce_box(n_side)
        eng = Engine.from_system(s, n_replicas=R, extra_capacity=0)
        ta = 0
    elif workload == "co2_gcmc":
        s = synth.co2_box(64, seed=13)
        eng = Engine(s.topo, s.box_matrix, s.bounds_lo, s.real_space_cutoff, s.ewald_tolerance, R, 0, [400])
        eng.load_system(s, 0)
        ta = 0
    else:
        s = synth.framework_water_box()
        eng = Engine(s.topo, s.box_matrix, s.bounds_lo, s.real_space_cutoff, s.ewald_tolerance, R, 0, [1, 200])
        eng.load_system(s, 0)
        ta = 1
    eng.init_structure_factor(0, True)
    for r in range(1, R):
        eng.replica_copy(r, 0)
    return s, eng, ta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["spce", "co2_gcmc", "framework_water"], default="spce")
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n-side", type=int, default=15)
    ap.add_argument("--json", default=None, help="also write the figures to this file")
    ap.add_argument("--kinds", choices=["workload", "moves", "insertions", "deletions"], default="workload",
                    help="override the workload's mix of trial kinds (diagnostics)")
    ap.add_argument("--decide", action="store_true",
                    help="the acceptance on the device (mgpu_gcmc_trial_decide_submit): the k sweep decides and commits; the "
                         "outcomes are scripted through the prefactors (the same share of acceptances as the commit launch gets)")
    args = ap.parse_args()
    R = args.replicas
    s, eng, ta = build(args.workload, R, args.n_side)
    rng = np.random.default_rng(0)
    n0 = int(s.n_mol[ta])
    L = np.diag(s.box_matrix)
    rep = np.arange(R, dtype=np.int32)
    t = np.full(R, ta, np.int32)
    base = s.all_sites(ta)
    n1 = base.shape[1]

    def batch():
        """one lane step: kinds, slots, candidate sites; the live count of every replica stays n0 +- a few"""
        if args.workload == "spce":
            kind = np.zeros(R, np.int32)
        elif args.workload == "co2_gcmc":
            kind = np.where(rng.random(R) < 0.5, _lib.MGPU_CREATION, _lib.MGPU_DELETION).astype(np.int32)
        else:
            u = rng.random(R)
            kind = np.where(u < 0.5, _lib.MGPU_MOVE, np.where(u < 0.75, _lib.MGPU_CREATION, _lib.MGPU_DELETION)).astype(np.int32)
        if args.kinds != "workload":
            kind = np.full(R, {"moves": _lib.MGPU_MOVE, "insertions": _lib.MGPU_CREATION, "deletions": _lib.MGPU_DELETION}[args.kinds], np.int32)
        nm = np.array([eng.num_molecules(r, ta) for r in range(R)]) if args.workload != "spce" else np.full(R, n0)
        m = (rng.random(R) * np.maximum(nm, 1)).astype(np.int32)
        m = np.minimum(m, np.minimum(nm, n0) - 1)          # a slot whose coordinates this script knows (never a grown one)
        sites = base[np.maximum(m, 0)].copy()
        mv = kind == _lib.MGPU_MOVE
        half = rng.random(R) < 0.5
        sites[mv & half] += rng.uniform(-0.15, 0.15, (int((mv & half).sum()), 1, 3))
        if (mv & ~half).any():
            sites[mv & ~half] = rotate_about_com(sites[mv & ~half], rng, 0.3)
        cr = kind == _lib.MGPU_CREATION
        if cr.any():
            sites[cr] = rotate_about_com(base[0][None].repeat(int(cr.sum()), 0), rng, 2 * np.pi)
            sites[cr] += (s.bounds_lo + L * rng.random((int(cr.sum()), 3)))[:, None, :] - sites[cr].mean(axis=1, keepdims=True)
        return kind, m, sites

    def outcomes(kind):
        if args.workload == "spce":
            return (rng.random(R) < 0.7).astype(np.int32)
        # keep every replica's count near n0: accept an insertion only below n0 + 4, a deletion only above n0 - 4
        nm = np.array([eng.num_molecules(r, ta) for r in range(R)])
        acc = (rng.random(R) < 0.6).astype(np.int32)
        acc[(kind == _lib.MGPU_CREATION) & (nm >= n0 + 4)] = 0
        acc[(kind == _lib.MGPU_DELETION) & (nm <= max(1, n0 - 4))] = 0
        return acc

    kind, m, sites = batch()
    eng.gcmc_trial(rep, t, m, kind, sites)
    eng.profile_enable(True)
    eng.profile_reset()
    evals = 0
    for _ in range(args.reps):
        kind, m, sites = batch()
        evals += int(2 * (kind == _lib.MGPU_MOVE).sum() + (kind != _lib.MGPU_MOVE).sum())
        if args.decide:
            acc = outcomes(kind)
            # prefactor 1e300: exp(-dE / T) * 1e300 >= 1 for any dE a liquid produces -> accepted; 0 -> rejected
            _, _, got = eng.gcmc_trial_decide(rep, t, m, kind, sites, np.full(R, 0.5), np.where(acc == 1, 1e300, 0.0), 300.0)
            continue
        eng.gcmc_trial(rep, t, m, kind, sites)
        acc = outcomes(kind)
        eng.commit_lane(0, rep, t, m, kind, acc)
    names = ["pair_sweep", "recip", "commit", "sfactor"]
    out = {}
    tot = {}
    for k, nm_ in enumerate(names):
        cnt, ms = eng.profile_get(k)
        if cnt:
            out[nm_] = ms / cnt * 1e3
            tot[nm_] = ms * 1e3
    ev = evals / args.reps
    N = int(sum(int(s.n_mol[i]) * int(s.topo.atoms_in_res[i]) for i in range(s.topo.n_res)))
    line = {"workload": args.workload, "candidates_per_launch": R, "evaluations_per_launch_group": ev, "n_atoms": N, "nk": eng.nk,
            "sites_per_molecule": int(n1), "avg_us": out, "us_per_group": {k: v / args.reps for k, v in tot.items()},
            "pair_ns_per_eval": tot.get("pair_sweep", 0.0) / args.reps * 1e3 / ev,
            "recip_ns_per_eval": out.get("recip", 0.0) * 1e3 / ev,
            "acceptance": "device" if args.decide else "host + commit launch", "lib": os.path.basename(_lib.LIB_PATH), "nsplit": os.environ.get("MGPU_PAIR_NSPLIT", "auto"),
            "blocks_per_cu": "auto"}
    print(json.dumps(line))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(line, f)
    eng.close()


if __name__ == "__main__":
    main()
