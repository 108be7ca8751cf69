#!/usr/bin/env python3
"""Benchmark of the per-move Delta-E hot path on MI355X (contract: see the task statement).

Workload (BASELINE.json metric: "MC moves/sec ... 10k-atom LJ+Ewald"): the 10 125-atom SPC/E box
of SURVEY.md section 8(d) -- 3375 rigid molecules, L = 46.56 A, lj/cut/coul/long, rc = 12 A,
ewald_tolerance 1e-5 -> alpha = 0.2346, kmax = 10, Nk = 2242 -- 50 % translation / 50 % rotation,
steps 0.3 A / 0.3 rad, T = 300 K.  Synthetic coordinates (seeded lattice + jitter), no files read.

A "step" = one Metropolis trial in each of the R replicas a GPU holds: R candidates x (old + new)
= 2R Delta-E evaluations in batched launches (two pair sweeps per candidate; one pass over k that
yields both reciprocal energies, sum ff W |A|^2 and sum ff W |A + delta|^2), the acceptance test on
the host (Fortran, mc_farm.f90), and one commit launch for the accepted candidates.  Nothing is
cached or skipped: both halves of ComputeOldEnergy / ComputeNewEnergy are computed for every trial.
value = accepted MC moves per second over all GPUs (replicas are independent chains; weak scaling).
"""
import argparse
import json
import os
import sys
import time

ORIG_AFFINITY = os.sched_getaffinity(0)      # before any OpenMP runtime binds this thread to its place

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 fp64 lanes x 2 flop x 2.4 GHz (spec)
FP64_SUSTAINED_TFLOPS = 61.0    # tools/probe_fma.hip on MI355X: what back-to-back v_fma_f64 sustains (clock under load)
PMC_SUMMARY = os.path.join("profiles", "r02", "pmc_latest.json")


def pmc_summary(kernel_substr):
    """Static figures from the committed rocprofv3 --pmc passes (tools/pmc_passes.sh; PMC cannot be collected
    from inside this process): HBM-side bytes (FETCH_SIZE doubled per the gfx950 calibration + WRITE_SIZE),
    VALU instructions and VALU-busy share per launch of tools/bench_kernels.py's batch (1024 trial moves =
    2048 evaluations per pair-sweep launch, as in the default bench).  None if the summary is absent."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            d = json.load(f)
        for name, e in d["kernels"].items():
            if kernel_substr in name:
                return dict(e, build=d.get("build"), kernel=name, evals=float(d.get("evaluations_per_pair_launch", 2048)))
    except Exception:
        pass
    return None


def cpu_baseline(system, translation_step, rotation_step, budget_s=15.0, seed=3, all_cores_budget_s=6.0):
    """The reference itself (oracle/_ref, amdflang -O2, 1 thread) -- or, if that build is not on
    this box, the C restatement -- running the same Metropolis trial sequentially for a bounded
    number of moves.  Reported beside the GPU number; it is a baseline, not the target."""
    from oracle import reflib, refcpu
    kind = "reference" if reflib.available() else "port"
    X = reflib.Reference(system) if kind == "reference" else refcpu.RefCPU(system)
    X.all_fourier_terms()
    X.init_amplitude(True)
    rng = np.random.default_rng(seed)
    n = int(system.n_mol[0])
    T = system.temperature
    trials = accepted = 0
    t0 = time.perf_counter()
    while True:
        m = int(rng.integers(0, n))
        com, off = X.get_molecule(0, m)
        X.save_fourier(0, m)
        old = X.old_energy(0, m, 0)
        if rng.random() <= 0.5:
            X.set_molecule(0, m, X.apply_pbc(com + (rng.random(3) - 0.5) * translation_step), off)
        else:
            rot = X.rotation_matrix(int(rng.random() * 3) + 1, (rng.random() - 0.5) * rotation_step)
            X.set_molecule(0, m, com, off @ rot.T)
        new = X.new_energy(0, m, 0)
        trials += 1
        if rng.random() <= min(1.0, np.exp(-(new[5] - old[5]) / T)):
            accepted += 1
        else:
            X.set_molecule(0, m, com, off)
            X.restore_fourier(0, m)
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    out = {"value": accepted / el, "unit": "accepted MC moves/s", "cores": 1, "kind": kind,
           "sample": f"{trials} sequential translation/rotation trials ({2 * trials} Delta-E evaluations) of the same "
                     f"{system.n_atoms}-atom box in {el:.1f} s",
           "trial_moves_per_s": trials / el, "ns_per_dE_eval": el / (2 * trials) * 1e9,
           "acceptance": accepted / max(1, trials)}
    if all_cores_budget_s > 0:
        # the Fortran driver's OpenMP runtime has bound this thread to one core and left OMP_PLACES in the
        # environment: undo both before the oracle's own OpenMP runtime (libgomp) starts
        os.sched_setaffinity(0, ORIG_AFFINITY)
        os.environ.pop("OMP_PLACES", None)
        os.environ.pop("OMP_PROC_BIND", None)
        # SURVEY 8(d)(ii), labelled extra: the C restatement (oracle/refcpu.c, kind "port") running one independent
        # chain per host core with OpenMP -- the same replica-level parallelism the GPU farm uses
        cores = usable_cores()
        try:
            el2, tr, ac = refcpu.trial_farm(system, cores, cores, all_cores_budget_s, translation_step, rotation_step)
            out["all_cores"] = {"value": float(ac.sum()) / el2, "unit": "accepted MC moves/s", "cores": cores, "kind": "port",
                                "sample": f"{cores} independent chains (one per core, OpenMP) x the same sequential trial for "
                                          f"{el2:.1f} s: {int(tr.sum())} trials ({2 * int(tr.sum())} Delta-E evaluations)",
                                "trial_moves_per_s": float(tr.sum()) / el2,
                                "ns_per_dE_eval_per_core": el2 / (2.0 * max(1.0, float(tr.mean()))) * 1e9}
        except Exception as exc:                       # the extra leg must never take the bench line down
            out["all_cores"] = {"error": str(exc)}
    return out


def usable_cores():
    cores = len(ORIG_AFFINITY)
    try:                                             # a cgroup CPU quota (the GPU boxes: 16 CPUs per GPU) caps it too
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(q) // int(per)))
    except Exception:
        pass
    return cores


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def pin_host_threads(torch, device, local_rank, local_world, n_threads):
    """Bind the OpenMP threads of the Fortran driver to physical cores of the NUMA node the GPU hangs off,
    split among the ranks whose GPUs share that node (unpinned, the step time varied by +-10 % run to run
    with where the threads happened to land).  Must run before the OpenMP runtime starts, i.e. before
    libmaniac_host.so is used; does nothing when the topology cannot be read or the user set OMP_PLACES."""
    if "OMP_PLACES" in os.environ or "OMP_PROC_BIND" in os.environ:
        return None
    try:
        def node_of(dev):
            p = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        n_dev = torch.cuda.device_count()
        node = node_of(device)
        if node < 0:
            return None
        peers = [j for j in range(local_world) if node_of(j % n_dev) == node] if local_world > 1 else [local_rank]
        cpus = _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        sib = set()
        cores = []
        for c in cpus:                       # one hardware thread per physical core
            if c in sib:
                continue
            cores.append(c)
            sib.update(_cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
        allowed = os.sched_getaffinity(0)
        cores = [c for c in cores if c in allowed]
        # keep clear of cpu 0 (it serves most interrupts and whatever else the box runs)
        if len(cores) > n_threads * len(peers) and cores and cores[0] == 0:
            cores = cores[1:]
        idx = peers.index(local_rank) if local_rank in peers else 0
        share = cores[idx * len(cores) // len(peers):(idx + 1) * len(cores) // len(peers)]
        if len(share) < n_threads:
            return None
        # one physical core per thread (places of two cores were measured: the threads then migrate and the
        # mirror gathers slow down by 45 %)
        places = [[c] for c in share[:n_threads]]
        os.environ["OMP_PLACES"] = ",".join("{" + ",".join(str(c) for c in pl) + "}" for pl in places)
        os.environ["OMP_PROC_BIND"] = "spread,close"     # lane threads spread over the places, each lane's team close to it
        return [c for pl in places for c in pl]
    except Exception:
        return None



ISOTHERM_POINTS = 8            # BASELINE.json configs[4]: isotherm sweep over 8 fugacities


def isotherm_fugacities(volume):
    """The 8 fugacity points of the CO2 isotherm (molecules per cubic Angstrom), log-spaced 20/V .. 160/V."""
    return np.geomspace(20.0, 160.0, ISOTHERM_POINTS) / volume


def isotherm_points_of_rank(rank, world):
    """Fugacity points a rank holds: dealt round-robin, so 8 GPUs hold one point each (configs[4]) and
    one GPU holds all eight; a rank beyond the eighth repeats point rank % 8 with its own seeds."""
    pts = [p for p in range(ISOTHERM_POINTS) if p % world == rank]
    return pts if pts else [rank % ISOTHERM_POINTS]


def spawn_ranks(n_gpus, argv):
    """`bench.py --gpus N` started without a launcher: start the N ranks (one process per GPU) as a child
    `python -m torch.distributed.run`, BEFORE anything in this process has touched the GPU (a process that has
    initialised HIP must never be replaced or forked into GPU work), pass its output through and return its
    exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def plan_host_threads(n_threads_req, local_world):
    cores = usable_cores()
    if n_threads_req > 0:
        return n_threads_req
    return max(1, min(8, cores // max(1, local_world)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--replicas", type=int, default=None,
                    help="independent chains per GPU (default: 8192 for the SPC/E workload, 2048 for the isotherm, whose small "
                         "boxes are host-bound)")
    ap.add_argument("--n-side", type=int, default=15, help="SPC/E lattice side (15 -> 10 125 atoms)")
    ap.add_argument("--host", choices=["fortran", "python"], default="fortran",
                    help="Metropolis driver: the Fortran farm (mc_farm.f90, two overlapped lanes) or the numpy one")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="OpenMP threads of the Fortran driver per GPU (0: min(8, cores available / ranks on the node))")
    ap.add_argument("--no-pin", action="store_true", help="do not bind the host threads to the GPU's NUMA node")
    ap.add_argument("--lanes", type=int, default=None,
                    help="submission lanes (chain groups in flight) of the Fortran driver: the host prepares / resolves one "
                         "group while the GPU evaluates the others (measured: 2048 chains x 2 lanes 5.6 M, 8192 x 4 lanes 6.9 M "
                         "accepted moves/s; kernels of different lanes overlap, which stretches their individual durations); "
                         "default 4 (SPC/E) or 2 (isotherm)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of the 1-core reference leg")
    ap.add_argument("--cpu-all-cores-budget", type=float, default=6.0,
                    help="seconds of the labelled all-core OpenMP leg of the C restatement (0: skip)")
    ap.add_argument("--settle-s", type=float, default=0.5,
                    help="untimed settle phase after the warm-up steps, seconds of the same step (0: none)")
    ap.add_argument("--workload", choices=["spce", "co2_isotherm"], default="spce",
                    help="spce: the 10 125-atom SPC/E box, translation / rotation (BASELINE metric, default); co2_isotherm: "
                         "configs[4], GCMC of CO2 in a 50 A box, 8 fugacity points dealt over the ranks, per-block gather "
                         "of the uptake (molecule-count) histogram")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank reports its placement (device, host threads, fugacity points) and "
                         "rank 0 prints the rank-ordered table gathered over the process group")
    ap.add_argument("--dump-counts", default=None, help="directory: every rank writes its chains' final molecule counts (tests)")
    args = ap.parse_args()

    if args.replicas is None:
        args.replicas = 8192 if args.workload == "spce" else 2048
    if args.lanes is None:
        args.lanes = 4 if args.workload == "spce" else 2
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    args.host_threads = plan_host_threads(args.host_threads, local_world)
    device = local_rank if args.device is None else args.device

    if args.dry_run:
        # placement rehearsal (CPU only): same process group, same exchange call, no engine
        from maniac_mc_amd import exchange
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo" if args.dist_backend != "gloo" else args.dist_backend)
        pts = isotherm_points_of_rank(rank, world) if args.workload == "co2_isotherm" else []
        row = [float(rank), float(local_rank), float(device), float(args.host_threads), float(args.replicas),
               float(len(pts)), float(pts[0] if pts else -1)]
        table, _ = exchange.gather_block_stats(row)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "workload": args.workload,
                              "ranks": [dict(rank=int(r[0]), local_rank=int(r[1]), device=int(r[2]), host_threads=int(r[3]),
                                             replicas=int(r[4]), fugacity_points=int(r[5]), first_point=int(r[6]))
                                        for r in table]}))
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.dist_backend)

    pinned = None
    if args.host == "fortran" and not args.no_pin:
        pinned = pin_host_threads(torch, device, local_rank, local_world, args.host_threads)

    from maniac_mc_amd import _lib, synth
    if args.host == "fortran":
        from maniac_mc_amd.fortran_host import FortranFarm as Farm
    else:
        from maniac_mc_amd.farm import ReplicaFarm as Farm
    kw = dict(n_threads=args.host_threads, n_lanes=args.lanes) if args.host == "fortran" else {}
    R = args.replicas
    iso_pts, fug_grid = None, None
    if args.workload == "spce":
        system = synth.spce_box(args.n_side)
        t_step, r_step = 0.3, 0.3
        farm = Farm(system, R, device=device, seed=1000 + rank,
                    translation_step=t_step, rotation_step=r_step, p_translation=0.5, **kw)
    else:
        if args.host != "fortran":
            sys.exit("bench.py: the isotherm workload runs on the Fortran farm")
        # configs[2] / [4]: rigid 3-site CO2, empty-ish cubic 50 A box (Nk = 2975), 25 % translation, 25 % rotation,
        # 50 % insertion / deletion; this rank's chains are split evenly over the fugacity points it holds
        system = synth.co2_box(64, seed=13)
        volume = float(np.prod(np.diag(system.box_matrix)))
        fug_grid = isotherm_fugacities(volume)
        iso_pts = isotherm_points_of_rank(rank, world)
        point_of_chain = np.array([iso_pts[(i * len(iso_pts)) // R] for i in range(R)])
        t_step, r_step = 1.0, 0.6
        farm = Farm(system, R, device=device, seed=1000 + rank, translation_step=t_step, rotation_step=r_step,
                    mol_capacity=[400], gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=fug_grid[point_of_chain]), **kw)
    eng = farm.eng
    N, Nk = system.n_atoms, eng.nk

    # profiling on BEFORE the warm-up: the first event-carrying dispatch of a stream costs ~7 ms once
    eng.profile_enable(True)
    farm.run(args.warmup)
    # Declared, untimed settle phase: the timed region may be as short as 20 steps (~5 ms), far below the time
    # the GPU clocks, the OpenMP team and the host caches need to reach their steady state; run the same step
    # until --settle-s seconds have passed (reported as settle_steps; `steps` and `warmup` stay as given).
    settle_steps = 0
    t_settle = time.perf_counter()
    while args.settle_s > 0 and time.perf_counter() - t_settle < args.settle_s:
        farm.run(100)
        settle_steps += 100
    eng.profile_reset()
    timers0 = farm.timers() if hasattr(farm, "timers") else None

    from maniac_mc_amd import exchange

    def fence():
        exchange.barrier()
        torch.cuda.synchronize()

    fence()
    trials0 = farm.trials if hasattr(farm, "trials") else 0
    t0 = time.perf_counter()
    accepted = farm.run(args.steps)
    # the path's one real exchange step: per-block all-gather of every rank's counters and
    # molecule-count histogram (NVT: a single bin per rank, the message size is the same; isotherm: one
    # histogram of the chains' current N per fugacity point, ISOTHERM_POINTS x 5001 bins)
    nbins = 5001
    trials_now = float(farm.trials - trials0) if hasattr(farm, "trials") else float(args.steps * R)
    if args.workload == "spce":
        hist = exchange.molecule_count_histogram([int(system.n_mol[0])] * R, nbins)
    else:
        counts = farm.counts()[:, 0]
        hist = np.concatenate([exchange.molecule_count_histogram(counts[point_of_chain == p], nbins)
                               for p in range(ISOTHERM_POINTS)])
    sums_by_rank, hist_by_rank = exchange.gather_block_stats([float(accepted), trials_now], hist)
    fence()
    elapsed = exchange.max_over_ranks(time.perf_counter() - t0)
    tot_acc, tot_trials = float(sums_by_rank[:, 0].sum()), float(sums_by_rank[:, 1].sum())
    assert int(hist_by_rank.sum()) == R * world
    if args.dump_counts and args.workload == "co2_isotherm":
        os.makedirs(args.dump_counts, exist_ok=True)
        np.savez(os.path.join(args.dump_counts, f"rank{rank}.npz"), counts=counts, point_of_chain=point_of_chain)

    n_pair, ms_pair = eng.profile_get(_lib.KERNEL_PAIR)
    n_rec, ms_rec = eng.profile_get(_lib.KERNEL_RECIP)
    n_com, ms_com = eng.profile_get(_lib.KERNEL_COMMIT)

    # The same pair-sweep batch launched alone (outside the timed region, rank 0): with several lanes in
    # flight the kernels of different lanes share the CUs, which raises throughput but stretches every
    # kernel's own duration; this gives the kernel's un-shared time for comparison.
    iso_us = None
    if rank == 0 and args.host == "fortran" and args.workload == "spce":
        rng = np.random.default_rng(5)
        n_l = max(1, R // farm.n_lanes)
        m_iso = rng.integers(0, int(system.n_mol[0]), n_l).astype(np.int32)
        sites_iso = system.all_sites(0)[m_iso] + rng.uniform(-0.15, 0.15, (n_l, 1, 3))
        eng.profile_reset()
        for _ in range(10):
            eng.trial_energy_candidates(np.arange(n_l, dtype=np.int32), np.zeros(n_l, np.int32), m_iso, sites_iso)
        n_iso, ms_iso = eng.profile_get(_lib.KERNEL_PAIR)
        iso_us = ms_iso / max(1, n_iso) * 1e3
    eng.profile_enable(False)

    if rank == 0:
        # algorithmic bytes per Delta-E evaluation: 36 N + 52 Nk; the pair sweep owns
        # the 36 N part (x, y, z, q fp64 + int32 type per atom), the k sweep the 52 Nk part.
        bytes_pair_eval = 36.0 * N
        bytes_eval = 36.0 * N + 52.0 * Nk
        # the Fortran driver splits the replicas over the engine's lanes: each launch carries one group
        n_lanes = farm.n_lanes if args.host == "fortran" else 1
        evals_per_launch = (2 * R) / n_lanes
        avg_pair_s = ms_pair / max(1, n_pair) * 1e-3
        alg_gbs = bytes_pair_eval * evals_per_launch / avg_pair_s / 1e9 if n_pair else 0.0
        evals_total = 2.0 * tot_trials
        # The pair sweep is bound by fp64 VALU issue, not by HBM (its measured memory-side traffic is a third of the
        # algorithmic bytes and it sits at ~80 % VALU-busy), so the headline fraction is VALU work against the
        # fp64 vector peak: every VALU instruction slot counted as one 64-lane FMA (2 flop), i.e. achieved =
        # VALU wave-instructions x 128 / launch time.  The instruction count per launch is the PMC figure
        # (SQ_INSTS_VALU of the same batch shape); the launch time is measured live by the dispatch events.
        pmc = pmc_summary("pair_sweep_kernel<3, false, false, true") or pmc_summary("pair_sweep_kernel")
        pmc_evals = pmc["evals"] if pmc else 2048.0
        scale = evals_per_launch / pmc_evals
        valu_instr = pmc["valu_instr_per_launch"] * scale if pmc and pmc.get("valu_instr_per_launch") else None
        # (a) per launch, in the pipeline: with several lanes in flight the kernels of different lanes share the
        #     device, so a launch's begin-to-end time includes waiting for CUs -- what rocprofv3 reports too;
        # (b) job level: the pair sweep's VALU work of ALL launches of the timed region over the region's wall time
        #     (every other kernel, every gap and the host's share included) -- the headline `achieved` / `frac`;
        # (c) isolated: the same batch launched alone after the timed region (the kernel's own efficiency).
        launch_tflops = valu_instr * 128.0 / avg_pair_s / 1e12 if (valu_instr and n_pair) else None
        valu_tflops = valu_instr * 128.0 * n_pair / elapsed / 1e12 if (valu_instr and n_pair) else None
        iso = None
        if iso_us:
            iso = {"avg_launch_us": iso_us,
                   "achieved": valu_instr * 128.0 / (iso_us * 1e-6) / 1e12 if valu_instr else None,
                   "frac": valu_instr * 128.0 / (iso_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS if valu_instr else None,
                   "algorithmic_hbm_GBs": bytes_pair_eval * evals_per_launch / (iso_us * 1e-6) / 1e9,
                   "note": "same batch launched alone after the timed region; in the timed region the kernels of "
                           "the lanes overlap on the device"}
        out = {
            "metric": "MC moves/sec", "value": tot_acc / elapsed, "unit": "accepted MC moves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "settle_steps": settle_steps, "timed_region_s": elapsed,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"spce_{system.n_mol[0]}mol_{N}atoms_lj_cut_coul_long_ewald_Nk{Nk}",
                       "replicas_per_gpu": R, "host_driver": args.host, "host_threads": args.host_threads if args.host == "fortran" else 1,
                       "lanes": farm.n_lanes if args.host == "fortran" else 1, "host_cores": pinned, "moves": "50% translation / 50% rotation, 0.3 A / 0.3 rad, 300 K",
                       "trials_per_step": R * world, "dE_evals_per_step": 2 * R * world, "parallelism": f"replicas x{world}"},
            "trial_moves_per_s": tot_trials / elapsed,
            "acceptance": tot_acc / max(1.0, tot_trials),
            "ns_per_dE_eval": elapsed / evals_total * 1e9 * world,
            "ns_per_dE_eval_note": "wall time per Delta-E evaluation per GPU (pair sweep + k sweep), host loop included",
            "roofline": {"bound": "valu", "bound_note": "fp64 vector issue (the contract's hbm / mfma do not apply: measured HBM-side "
                                                         "traffic is 0.34 x the algorithmic bytes, no MFMA-shaped work; SURVEY 8(d))",
                         "kernel": "pair_sweep_kernel<3,false,false,true,true> (old + new state of a trial move in one sweep, two-instruction fold)",
                         "achieved": valu_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": valu_tflops / FP64_VECTOR_PEAK_TFLOPS if valu_tflops else None,
                         "traffic": pmc["hbm_bytes_per_launch"] * scale if pmc and pmc.get("hbm_bytes_per_launch") else None,
                         "traffic_source": f"static: {PMC_SUMMARY} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch, build "
                                           f"{pmc.get('build') if pmc else None}; not a measurement of this run)",
                         "frac_basis": "job level: VALU instruction slots of all pair-sweep launches of the timed region (x 128 flop) / "
                                       "timed_region_s / fp64 vector peak; per_launch and isolated give the other two views",
                         "avg_launch_us": avg_pair_s * 1e6, "launches": n_pair, "evaluations_per_launch": evals_per_launch,
                         "per_launch": {"avg_launch_us": avg_pair_s * 1e6, "achieved": launch_tflops,
                                        "frac": launch_tflops / FP64_VECTOR_PEAK_TFLOPS if launch_tflops else None,
                                        "note": "dispatch events in the timed region (what rocprofv3 --kernel-trace shows): with "
                                                f"{n_lanes} lanes in flight a launch's begin-to-end time includes waiting for CUs "
                                                "held by the other lanes' kernels"},
                         "valu": {"instr_per_launch": valu_instr, "valu_busy": pmc.get("valu_busy") if pmc else None,
                                  "lds_busy": pmc.get("lds_busy") if pmc else None,
                                  "frac_of_sustained_fma_rate": valu_tflops / FP64_SUSTAINED_TFLOPS if valu_tflops else None,
                                  "sustained_peak": FP64_SUSTAINED_TFLOPS,
                                  "source": f"static: {PMC_SUMMARY} (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x busy cycles)); "
                                            "sustained peak from tools/probe_fma.hip"},
                         "hbm": {"algorithmic_bytes_per_launch": bytes_pair_eval * evals_per_launch,
                                 "algorithmic_equivalent_GBs": alg_gbs, "peak": HBM_PEAK_GBS,
                                 "algorithmic_frac": alg_gbs / HBM_PEAK_GBS,
                                 "note": "36 N bytes per evaluation (SURVEY 8(d)) / launch time: an algorithmic figure served mostly "
                                         "from L2 / Infinity Cache, NOT an HBM utilisation"},
                         "isolated": iso,
                         "job_frac": (evals_total / world / elapsed) * bytes_eval / 1e9 / HBM_PEAK_GBS,
                         "recip_avg_launch_us": ms_rec / max(1, n_rec) * 1e3, "commit_avg_launch_us": ms_com / max(1, n_com) * 1e3},
        }
        if args.workload == "co2_isotherm":
            # configs[4]: GCMC isotherm.  Trials are move SELECTIONS that reach the engine (no-op selections of the
            # reference -- empty type, full type -- are skipped by the host); an insertion / deletion costs ONE
            # evaluation, so the SPC/E roofline figures do not carry over: report the launch times only.
            hist_pts = hist_by_rank.reshape(world, ISOTHERM_POINTS, nbins).sum(axis=0)
            nn = np.arange(nbins)
            out["config"] = {"workload": f"co2_isotherm_{ISOTHERM_POINTS}fugacities_50A_box_Nk{Nk}", "replicas_per_gpu": R,
                             "host_driver": args.host, "host_threads": args.host_threads, "lanes": farm.n_lanes,
                             "host_cores": pinned, "moves": "25% translation / 25% rotation / 50% insertion-deletion, 300 K",
                             "parallelism": f"replicas x{world}; fugacity points dealt round-robin over the ranks"}
            out["isotherm"] = [{"fugacity_molecules_per_A3": float(fug_grid[p]), "chains": int(hist_pts[p].sum()),
                                "mean_N": float((hist_pts[p] * nn).sum() / max(1, hist_pts[p].sum())),
                                "N_min": int(nn[hist_pts[p] > 0].min()) if hist_pts[p].sum() else None,
                                "N_max": int(nn[hist_pts[p] > 0].max()) if hist_pts[p].sum() else None}
                               for p in range(ISOTHERM_POINTS)]
            out["exchange"] = {"collective": "all_gather", "backend": args.dist_backend if world > 1 else None,
                               "bytes_per_rank": int(hist.nbytes + 16), "per": "block (= the timed region)"}
            out["roofline"] = {"bound": "valu", "kernel": "pair_sweep_kernel (fused old + new for moves, single state for insertions / deletions)",
                               "achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "avg_launch_us": avg_pair_s * 1e6, "launches": n_pair,
                               "recip_avg_launch_us": ms_rec / max(1, n_rec) * 1e3, "commit_avg_launch_us": ms_com / max(1, n_com) * 1e3,
                               "note": "not the BASELINE metric's workload: launch times only"}
            for k in ("ns_per_dE_eval", "ns_per_dE_eval_note"):
                out.pop(k, None)
        if hasattr(farm, "timers"):
            out["host_seconds"] = {k: v - timers0[k] for k, v in farm.timers().items()}   # timed region only
        if world == 1 and not args.no_cpu_baseline and args.workload == "spce":
            out["cpu_baseline"] = cpu_baseline(system, t_step, r_step, budget_s=args.cpu_budget,
                                                all_cores_budget_s=args.cpu_all_cores_budget)
        print(json.dumps(out))
    farm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
