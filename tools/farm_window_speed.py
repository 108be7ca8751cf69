#!/usr/bin/env python3
"""Accepted moves/s of the Fortran farm (10 125-atom SPC/E box, or the CO2 insertion / deletion box) against the number of chains: batched path
(mgpu_move_trial_submit / wait + mgpu_commit_submit, five launches per lane step) against window mode (ONE launch per lane
step, mgpu_farm_window_submit), for several lane counts and windows in flight.

    python tools/farm_window_speed.py [--replicas 8,64,512] [--seconds 1.0] [--modes batched,w1,w2,w3] [--lanes 1,2,4]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", default="1,8,64,512")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--modes", default="batched,w1,w2,w3")
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--drivers", default="1", help="driver threads (each runs its own lanes' windows)")
    ap.add_argument("--side", type=int, default=15)
    ap.add_argument("--workload", default="spce", choices=("spce", "co2_gcmc"),
                    help="spce: the 10 125-atom box, translation / rotation; co2_gcmc: bench.py's 50 A CO2 box, insertion / deletion only")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    from maniac_mc_amd import synth
    from maniac_mc_amd.fortran_host import FortranFarm
    if args.workload == "spce":
        s = synth.spce_box(args.side, seed=12345)
        kw = dict(translation_step=0.3, rotation_step=0.3, p_translation=0.5)
    else:
        s = synth.co2_box(64, seed=13)
        kw = dict(translation_step=1.0, rotation_step=0.6, mol_capacity=[400],
                  gcmc=dict(p_translation=0.0, p_rotation=0.0, fugacity=100.0 / 50.0 ** 3))
    rows = []
    for R in [int(x) for x in args.replicas.split(",")]:
        for mode in args.modes.split(","):
            for lanes, drivers in [(int(x), int(y)) for x in args.lanes.split(",") for y in args.drivers.split(",")]:
                if lanes > R or drivers > lanes:
                    continue
                window = mode.startswith("w")
                depth = int(mode[1:]) if window else 1
                farm = FortranFarm(s, R, seed=77, n_threads=max(args.threads, drivers), n_lanes=lanes, n_drivers=drivers, device_build=True,
                                   window=window, window_depth=depth, **kw)
                try:
                    farm.run(20)
                    chunk = (400 if R <= 64 else 200) if window else 50       # (see bench.py replicas_sweep: a chunk ends with a synchronise)
                    farm.run(chunk)
                    farm.eng.synchronize()
                    steps = acc = 0
                    t0 = time.perf_counter()
                    while True:
                        acc += farm.run(chunk)
                        steps += chunk
                        farm.eng.synchronize()
                        el = time.perf_counter() - t0
                        if el >= args.seconds:
                            break
                    row = {"replicas": R, "mode": mode, "lanes": lanes, "drivers": drivers, "window": farm.window, "accepted_per_s": acc / el,
                           "us_per_step": el / steps * 1e6, "nsplit_note": "engine constant", "timers": farm.timers()}
                    rows.append(row)
                    print(f"R {R:5d}  {mode:8s} lanes {lanes} drivers {drivers}  {acc / el / 1e6:8.4f} M accepted/s   {el / steps * 1e6:8.1f} us/step", flush=True)
                finally:
                    farm.close()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
