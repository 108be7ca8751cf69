"""Throughput of the grand-canonical farm on the CO2 stand-in of BASELINE.json configs[2] / [4]:
rigid 3-site CO2 in a cubic 50 A box (kmax = 11, Nk = 2975), insertion / deletion / translation / rotation,
one fugacity per chain group.  Prints trial and accepted moves per second."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maniac_mc_amd import synth  # noqa: E402
from maniac_mc_amd.fortran_host import FortranFarm  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
s = synth.co2_box(64, seed=13)
V = 50.0 ** 3
fug = np.repeat(np.geomspace(20.0, 160.0, 8), R // 8) / V          # 8 isotherm points, as configs[4]
farm = FortranFarm(s, R, seed=3, translation_step=1.0, rotation_step=0.6, n_threads=8, mol_capacity=[400],
                   gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=fug))
farm.run(200)
t0 = time.perf_counter()
acc = farm.run(steps)
el = time.perf_counter() - t0
c = farm.counters()
n = farm.counts()[:, 0]
print(f"R={R}: {R * steps / el:.3e} selections/s, {acc / el:.3e} accepted moves/s, {el / steps * 1e3:.3f} ms/step; "
      f"<N> per fugacity group {[round(float(n[g * (R // 8):(g + 1) * (R // 8)].mean()), 1) for g in range(8)]}; counters {c}")
farm.close()
