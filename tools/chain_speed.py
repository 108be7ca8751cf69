#!/usr/bin/env python3
"""Speed of the single-chain drop-in (mc_chain.f90 through run.run_simulation) with and without speculative windows.

    python tools/chain_speed.py [--blocks 4] [--steps 2000]
Cases: the two charged grand-canonical whole-run fixtures (tests/golden/runs/co2_gcmc = BASELINE.json configs[2],
framework_water_gcmc = configs[3] in miniature; their own inputs and seeds, more steps) in the as-written mode the
fixtures were made in, and the 10 125-atom SPC/E box (NVT).  For every case the loop is run with K = 1 (one engine call
per step), 4, 8 and 16; the output files of all K must be identical (checked here), only the time differs.
"""
import argparse
import filecmp
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import io_maniac, run, synth  # noqa: E402

RUNS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "runs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--cases", default="co2_gcmc,framework_water_gcmc,spce_10125_nvt")
    ap.add_argument("--chain-windows", type=int, default=1, help="0: batched submit / wait calls instead of one launch per window")
    a = ap.parse_args()
    ks = [int(k) for k in a.ks.split(",")]
    summary = json.load(open(os.path.join(RUNS, "summary.json")))
    tmp = tempfile.mkdtemp()
    cases = []
    for name in ("co2_gcmc", "framework_water_gcmc"):
        inp = os.path.join(RUNS, name, "inputs")
        cases.append((name, [os.path.join(inp, f) for f in ("system.maniac", "system.data", "system.inc")],
                      dict(seed=summary[name]["seed"], as_written=bool(summary[name].get("as_written")))))
    s = synth.spce_box(15)
    files = io_maniac.write_input_files(s, tmp + "/spce_in", nb_block=2, nb_step=100, translation_step=0.3, rotation_step_angle=0.3,
                                        translation_proba=0.5, rotation_proba=0.5, masses=[15.9994, 1.008], atom_names=["OW", "HW"])
    cases.append(("spce_10125_nvt", list(files), dict(seed=5)))
    # the same box sheared (bench.py's spce_triclinic workload): single-chain windows in a triclinic cell (round 5)
    import numpy as np
    st = synth.spce_box(15)
    L = float(st.box_matrix[0, 0])
    st.box_matrix = np.array([[L, 0.0, 0.0], [3.0, L, 0.0], [-2.0, 1.5, L]])
    frac = (st.com[0] - st.bounds_lo[None, :]) / L
    st.com[0] = st.bounds_lo[None, :] + frac @ st.box_matrix.T
    files_t = io_maniac.write_input_files(st, tmp + "/spce_tri_in", nb_block=2, nb_step=100, translation_step=0.3, rotation_step_angle=0.3,
                                          translation_proba=0.5, rotation_proba=0.5, masses=[15.9994, 1.008], atom_names=["OW", "HW"])
    cases.append(("spce_10125_triclinic_nvt", list(files_t), dict(seed=5)))
    want = a.cases.split(",")
    for name, files, kw in cases:
        if name not in want:
            continue
        base = None
        for k in ks:
            out = os.path.join(tmp, f"{name}_k{k}") + "/"
            res = run.run_simulation(*files, out, nb_block=a.blocks, nb_step=a.steps, speculate=k, chain_windows=bool(a.chain_windows), **kw)
            n = a.blocks * a.steps
            c = res["counters"]
            acc = int(c[1] + c[3] + c[5] + c[7])
            same = ""
            if base is None:
                base = (out, res["loop_seconds"])
            else:
                diff = [f for f in sorted(os.listdir(out)) if f != "log.maniac" and not filecmp.cmp(os.path.join(out, f), os.path.join(base[0], f), shallow=False)]
                same = f"  files identical to K={ks[0]}: {not diff}{' ' + str(diff) if diff else ''}  speed-up {base[1] / res['loop_seconds']:.2f}x"
            print(f"{name:24s} K={k:2d}: {n} steps in {res['loop_seconds']:.3f} s -> {n / res['loop_seconds']:.0f} steps/s "
                  f"(Monte Carlo loop alone {res['mc_seconds']:.3f} s -> {n / res['mc_seconds']:.0f} steps/s, initial energy {res['init_seconds']:.3f} s, "
                  f"files {res['file_seconds']:.3f} s; windows {res['chain_windows'][0]}, "
                  f"left to the host {res['chain_windows'][1]}), acceptance {acc / n:.2f}{same}", flush=True)


if __name__ == "__main__":
    main()
