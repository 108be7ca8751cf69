#!/usr/bin/env python3
"""Several INDEPENDENT single-chain runs on one GPU at the same time (one process and one engine per chain, as a user with a
handful of chains would run the drop-in): aggregate step rate against the number of processes.

    python tools/chains_in_parallel.py [--procs 1,2,4] [--steps 100000] [--case spce_10125_nvt]

Every process is `tools/chain_speed.py --cases <case> --ks 4 --blocks 2 --steps <steps>`; the rate of a process is its Monte
Carlo loop alone (the loops of the processes overlap for all but the start-up skew).
"""
import argparse
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="1,2,4")
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--case", default="spce_10125_nvt")
    a = ap.parse_args()
    for p in [int(x) for x in a.procs.split(",")]:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "chain_speed.py"), "--cases", a.case, "--ks", "4", "--blocks", "2", "--steps", str(a.steps)]
        t0 = time.perf_counter()
        children = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(p)]
        outs = [c.communicate()[0] for c in children]
        wall = time.perf_counter() - t0
        rates = []
        for o in outs:
            m = re.search(r"Monte Carlo loop alone ([0-9.]+) s -> (\d+) steps/s", o)
            if not m:
                print(o[-400:])
                continue
            rates.append(float(m.group(2)))
        print(f"{a.case}: {p} process(es) x {2 * a.steps} steps: per process {', '.join(f'{r:.0f}' for r in rates)} steps/s -> "
              f"aggregate {sum(rates):.0f} steps/s (wall {wall:.1f} s with start-up)", flush=True)


if __name__ == "__main__":
    main()
