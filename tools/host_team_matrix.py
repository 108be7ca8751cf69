#!/usr/bin/env python3
"""Host cores against throughput for the host-bound farm: bench.py --workload co2_gcmc with D driver threads x T host threads
(accepted moves/s and the host's seconds per phase over the timed region).  Round 5: the chains' uniform numbers come from
mgpu_rng_fill (four xoshiro256+ streams abreast) instead of one stream at a time.

    python tools/host_team_matrix.py [--steps 300]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--combos", default="1x1,1x2,1x4,2x4,3x6")
    a = ap.parse_args()
    print(f"# CO2 farm (bench.py --workload co2_gcmc --steps {a.steps} --drivers D --host-threads T), accepted moves/s and host ms per step")
    for combo in a.combos.split(","):
        d, t = (int(x) for x in combo.split("x"))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "co2_gcmc", "--steps", str(a.steps), "--warmup", "20",
               "--drivers", str(d), "--host-threads", str(t), "--no-cpu-baseline", "--configs", "0", "--sustained-steps", "0",
               "--replicas-sweep", ""]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            print(f"co2 drivers {d} threads {t} FAILED {p.stderr[-300:]}")
            continue
        j = json.loads(line[-1])
        hs = {k: round(v / a.steps * 1e3, 3) for k, v in (j.get("host_seconds") or {}).items()}
        print(f"co2 drivers {d} threads {t} {j['value'] / 1e6:.2f} M  ms/step {j['ms_per_step']:.3f} {hs}", flush=True)


if __name__ == "__main__":
    main()
