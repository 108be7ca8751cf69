"""TEST INFRASTRUCTURE ONLY (oracle).  Run the reference's own program flow (main.f90:16-33) on three
input files through oracle/_ref: readers, PrepareSimulationParameters, ComputeSystemEnergy,
MonteCarloLoop, FinalReport -- with A(k) initialised to S(k) first and the generator seeded
by the reference's seed_rng.  One run per process (the reference allocates its state once):

    python oracle/run_ref_mc.py <input.maniac> <topology.data> <parameters.inc> <outdir/> <seed> [reservoir.data]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import reflib  # noqa: E402


def main():
    maniac, data, inc, outdir, seed = sys.argv[1:6]
    reservoir = sys.argv[6] if len(sys.argv) > 6 else ""
    L = reflib.lib()
    L.ref_load_files.restype = C.c_int
    L.ref_run_mc.restype = C.c_int
    L.ref_set_reservoir_file(reservoir.encode())
    rc = L.ref_load_files(maniac.encode(), data.encode(), inc.encode(), outdir.encode(), C.c_int(2))
    assert rc == 0
    rc = L.ref_run_mc(C.c_int(int(seed)), C.c_int(1))
    assert rc == 0
    print("RUN_OK")


if __name__ == "__main__":
    main()
