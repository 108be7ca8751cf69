"""One MANIAC run on the GPU engine: the reference's program flow (main.f90:16-33) for ONE chain.

    run_simulation("input.maniac", "topology.data", "parameters.inc", "outputs/")

reads the reference's three input files (io_maniac), creates the engine for one replica and hands
the state to the Fortran chain driver (fortran/mc_chain.f90), which mirrors MonteCarloLoop move by
move -- same random-number order -- and writes the reference's output files (fortran/
maniac_output.f90): log.maniac (Monte Carlo part), energy.dat, number_<res>.dat, moves.dat,
trajectory.lammpstrj, topology.data, and reservoir.lammpstrj when a reservoir topology is given
(the reference's ``-r`` option, cli_utils.f90:60-63).  Python here is plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib, fortran_host, io_maniac
from .engine import Engine, box_prepare
from .system import NB_MAX_MOLECULE

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _f(a):
    """Fortran (column-major) view of a 2-D array as a flat double pointer."""
    arr = np.asfortranarray(np.asarray(a, dtype=np.float64))
    return arr, arr.ctypes.data_as(_dp)


def _box_args(dat):
    box_type, volume, reciprocal, _ = box_prepare(dat["matrix"])
    m, mp = _f(dat["matrix"])
    r, rp = _f(reciprocal)
    lo = np.ascontiguousarray(dat["lo"], dtype=np.float64)
    hi = np.ascontiguousarray(dat["hi"], dtype=np.float64)
    tilt = np.ascontiguousarray(dat["tilt"], dtype=np.float64)
    keep = (m, r, lo, hi, tilt)
    return keep, (mp, rp, lo.ctypes.data_as(_dp), hi.ctypes.data_as(_dp), tilt.ctypes.data_as(_dp),
                  C.c_int(1 if dat["triclinic"] else 0), C.c_int(box_type), C.c_double(volume))


def _mol_arrays(com, off, n1):
    """(n, 3) / (n, n1, 3) numpy -> the Fortran shapes com(3, n) / off(3, n1, n)."""
    com = np.ascontiguousarray(np.asarray(com, dtype=np.float64).reshape(-1, 3))
    off = np.ascontiguousarray(np.asarray(off, dtype=np.float64).reshape(-1, n1, 3))
    return com, off


def header_text(inp, dat, maniac_path, data_path, inc_path, eng_or_ewald, reservoir_path=None, rdat=None):
    """The messages above "Started Monte Carlo Loop" as one byte string (one message per line) for mchain_set_log_header.
    File names are echoed as given, like the reference does.  ``eng_or_ewald``: an Engine, or the dict ewald_setup returns."""
    present = sorted({int(t) for i, r in enumerate(inp.residues) for t in dat["atom_types"][i, : r.nb_atoms] if t > 0})
    _, _, eps_raw, sig_raw = io_maniac.read_parameters(inc_path, dat["n_atom_types"], present, with_raw=True)
    if isinstance(eng_or_ewald, dict):
        ew = eng_or_ewald
    else:
        from .engine import ewald_setup
        _, _, _, metrics = box_prepare(dat["matrix"])
        ew = ewald_setup(metrics, inp.real_space_cutoff, inp.ewald_tolerance)
        assert ew["nk"] == eng_or_ewald.nk and ew["alpha"] == eng_or_ewald.alpha      # what the engine itself set up
    lines = io_maniac.log_header_lines(inp, dat, maniac_path, data_path, inc_path, eps_raw, sig_raw, ew,
                                       reservoir=(reservoir_path, rdat) if reservoir_path else None)
    return "\n".join(lines).encode("utf-8")


def run_simulation(maniac_path, data_path, inc_path, outdir, seed=None, reservoir_path=None, device=0,
                   mol_capacity=None, nb_block=None, nb_step=None, seams=False, as_written=False, speculate=4,
                   chain_windows=True, chain_margin=None):
    """Run the chain; returns a dict with the final energies (K), counters, molecule counts, step sizes.

    ``seed``: None -> the input file's ``seed`` if present, else the generator is left unseeded
    (as the reference leaves it when the input names a seed, input_parser.f90:597).
    ``mol_capacity``: molecule slots per residue type (default: NB_MAX_MOLECULE for active types).
    ``seams``: True -> one engine call per reference seam (ComputePairInteractionEnergy_singlemol, ...), the literal
    integration of INTEGRATION.md; False (default) -> one batched call per move, same energies, ~3x fewer waits.
    ``as_written``: True -> the reference's deletion update exactly as written (A(k) gains the swapped-in
    molecule's terms, monte_carlo_utils.f90:308), composed in the host loop from neutral engine primitives; for
    charged grand-canonical runs this reproduces the reference's files but not the intended physics (default False).
    ``speculate``: K > 1 (default 4 -- measured best with one launch per window, where a longer window costs more than the
    steps it saves; batched mode only) -> the next K steps are drawn in the reference's random-number
    order assuming every one is rejected and evaluated in ONE engine call; the first accepted one is applied, the
    generator is put back to its state after that step and the rest is redrawn.  Same states, same files, up to
    1 / acceptance fewer round trips.  1 -> one step per engine call.
    ``chain_windows``: True (default; batched mode only) -> a window is ONE kernel launch (mgpu_chain_window): the engine
    evaluates its steps, applies the acceptance rule to them in order with the loop's own draws and commits the first
    accepted one, leaving to the loop only the steps too close to call; where the engine cannot (triclinic box, molecules
    of more than five sites) the loop falls back to the batched calls by itself.  False -> the batched calls always.
    ``chain_margin``: relative width of the band around an acceptance probability inside which the engine leaves the step to
    this loop's own exp (default: the engine's 16 ulp; tests widen it to drive the loop's side of that hand-over).
    """
    system, inp, dat = io_maniac.load_system(maniac_path, data_path, inc_path, with_data=True)
    rdat = io_maniac.read_lammps_data(reservoir_path, inp) if reservoir_path else None
    topo = system.topo
    n_res = topo.n_res
    if mol_capacity is None:
        mol_capacity = [NB_MAX_MOLECULE if topo.is_active[t] == 1 else max(1, int(system.n_mol[t])) for t in range(n_res)]
    eng = Engine.from_system(system, n_replicas=1, device=device, mol_capacity=mol_capacity)
    if chain_margin is not None:
        eng.chain_set_margin(float(chain_margin))
    H = fortran_host.lib()
    H.mchain_run.restype = C.c_int
    try:
        H.mchain_reset(eng.h, C.c_int(n_res), C.c_int(topo.n_atom_types), C.c_double(system.temperature))
        keep, args = _box_args(dat)
        H.mchain_set_box(*args)
        fug = inp.fugacity_per_A3()
        hold = []
        for t in range(n_res):
            n1 = int(topo.atoms_in_res[t])
            com, off = _mol_arrays(system.com[t], system.offsets[t], n1)
            types = np.ascontiguousarray(topo.atom_types[t, :n1], dtype=np.int32)
            q = np.ascontiguousarray(topo.charges[t, :n1], dtype=np.float64)
            hold += [com, off, types, q]
            H.mchain_set_residue(C.c_int(t + 1), inp.residues[t].name.encode(), C.c_int(n1), C.c_int(int(topo.is_active[t])),
                                 C.c_int(int(mol_capacity[t])), C.c_int(com.shape[0]), types.ctypes.data_as(_ip),
                                 q.ctypes.data_as(_dp), com.ctypes.data_as(_dp), off.ctypes.data_as(_dp),
                                 C.c_double(float(fug[t])))
            for kind, key in enumerate(("bonds", "angles", "dihedrals", "impropers"), start=1):
                rows = dat["bonded_per_residue"][key][t]
                tab = np.zeros((max(1, len(rows)), 5), dtype=np.int32)
                for k, row in enumerate(rows):
                    tab[k, : len(row)] = row
                hold.append(tab)
                H.mchain_set_bonded(C.c_int(t + 1), C.c_int(kind), C.c_int(len(rows)), tab.ctypes.data_as(_ip))
        masses = np.ascontiguousarray(dat["masses"], dtype=np.float64)
        ntypes = np.array([dat["type_counts"][k] for k in ("bonds", "angles", "dihedrals", "impropers")], dtype=np.int32)
        H.mchain_set_tables(masses.ctypes.data_as(_dp), ntypes.ctypes.data_as(_ip))
        H.mchain_set_mode(C.c_int(1 if seams else 0))
        H.mchain_set_as_written(C.c_int(1 if as_written else 0))
        H.mchain_set_speculation(C.c_int(1 if seams else max(1, int(speculate))))
        H.mchain_set_chain_windows(C.c_int(1 if chain_windows else 0))
        H.mchain_get_loop_seconds.restype = C.c_double
        header = header_text(inp, dat, maniac_path, data_path, inc_path, eng, reservoir_path, rdat)
        H.mchain_set_log_header(header, C.c_int(len(header)))
        H.mchain_set_moves(C.c_double(inp.translation_step), C.c_double(inp.rotation_step_angle),
                           C.c_double(inp.translation_proba), C.c_double(inp.rotation_proba),
                           C.c_int(1 if inp.recalibrate_moves else 0))
        if reservoir_path:
            rkeep, rargs = _box_args(rdat)
            any_bonded = np.array([1 if rdat["bonded_counts"][k] > 0 else 0
                                   for k in ("bonds", "angles", "dihedrals", "impropers")], dtype=np.int32)
            H.mchain_set_reservoir_box(*rargs, any_bonded.ctypes.data_as(_ip))
            for t in range(n_res):
                n1 = int(topo.atoms_in_res[t])
                com, off = _mol_arrays(rdat["com"][t], rdat["off"][t], n1)
                hold += [com, off]
                H.mchain_set_reservoir_residue(C.c_int(t + 1), C.c_int(n1), C.c_int(NB_MAX_MOLECULE), C.c_int(com.shape[0]),
                                               com.ctypes.data_as(_dp), off.ctypes.data_as(_dp))
        if seed is None:
            seed = inp.seed if inp.has_seed else 0
        outdir = os.path.join(outdir, "")
        os.makedirs(outdir, exist_ok=True)
        import time
        t_loop = time.perf_counter()
        rc = H.mchain_run(C.c_int(inp.nb_block if nb_block is None else nb_block),
                          C.c_int(inp.nb_step if nb_step is None else nb_step), C.c_int(int(seed)), outdir.encode())
        t_loop = time.perf_counter() - t_loop       # initial energy + Monte Carlo loop + files
        _lib.check(rc)
        e = np.zeros(6); cnt = np.zeros(8, dtype=np.int32); nm = np.zeros(n_res, dtype=np.int32); st = np.zeros(2)
        H.mchain_get_energy(e.ctypes.data_as(_dp))
        H.mchain_get_counters(cnt.ctypes.data_as(_ip))
        H.mchain_get_counts(nm.ctypes.data_as(_ip))
        H.mchain_get_steps(st.ctypes.data_as(_dp))
        mc_seconds = float(H.mchain_get_loop_seconds())
        times = np.zeros(3)
        H.mchain_get_times(times.ctypes.data_as(_dp))
        chain_stats = eng.chain_stats()
        e_final = eng.system_energy(0)
    finally:
        eng.close()
    keys = ("non_coulomb", "coulomb", "recip_coulomb", "ewald_self", "intra_coulomb", "total")
    # loop_seconds: mchain_run as a whole (initial energy, Monte Carlo loop, every output file); mc_seconds: the Monte
    # Carlo steps alone; chain_windows: (one-launch windows, windows with a step left to the host's exp)
    return dict(energy=dict(zip(keys, e)), recomputed_energy=e_final, counters=cnt, n_mol=nm, loop_seconds=t_loop,
                mc_seconds=mc_seconds, init_seconds=float(times[0]), file_seconds=float(times[2]), chain_windows=chain_stats, translation_step=st[0], rotation_step=st[1])


def main(argv=None):
    """`python -m maniac_mc_amd.run -i input.maniac -d topology.data -p parameters.inc [-r reservoir.data] [-o outputs/]`
    -- the reference's command line (cli_utils.f90:36-83: -i, -d, -p mandatory, -r optional, -o defaults to
    outputs/), plus --seed and --device."""
    import argparse
    import sys
    ap = argparse.ArgumentParser(prog="python -m maniac_mc_amd.run", description=main.__doc__)
    ap.add_argument("-i", dest="maniac", required=True, help="MANIAC input file")
    ap.add_argument("-d", dest="data", required=True, help="LAMMPS data file (atom_style full)")
    ap.add_argument("-p", dest="inc", required=True, help="pair_coeff include file")
    ap.add_argument("-r", dest="reservoir", default=None, help="reservoir data file")
    ap.add_argument("-o", dest="out", default="outputs/", help="output directory")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--speculate", type=int, default=4,
                    help="speculative window: steps evaluated per engine call (same states and files; 1: one step per call)")
    ap.add_argument("--no-chain-windows", action="store_true",
                    help="evaluate windows through the batched submit / wait calls instead of the one-launch path")
    ap.add_argument("--as-written", action="store_true",
                    help="the reference's deletion update exactly as written (SURVEY F3) instead of the intended physics")
    a = ap.parse_args(argv)
    for path, what in ((a.maniac, "Input"), (a.data, "Data"), (a.inc, "Parameter"), (a.reservoir, "Reservoir")):
        if path is not None and not os.path.isfile(path):
            print(f"{what} file not found: {path}", file=sys.stderr)
            return 1
    res = run_simulation(a.maniac, a.data, a.inc, a.out, seed=a.seed, reservoir_path=a.reservoir, device=a.device,
                         as_written=a.as_written, speculate=a.speculate, chain_windows=not a.no_chain_windows)
    e = res["energy"]
    print(f"final energy (K): total {e['total']:.6f}  non_coulomb {e['non_coulomb']:.6f}  coulomb {e['coulomb']:.6f}  "
          f"recip {e['recip_coulomb']:.6f};  molecules {res['n_mol'].tolist()};  output in {os.path.join(a.out, '')}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
