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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic_per_evaluation():
    """HBM bytes per Delta-E evaluation of the pair sweep from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled per the gfx950 calibration, + WRITE_SIZE); None if the summary is absent.
    PMC cannot be collected from inside this process, so the figure comes from profiles/."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["pair_sweep_kernel"]["hbm_bytes_per_evaluation"])
    except Exception:
        return None


def cpu_baseline(system, translation_step, rotation_step, budget_s=15.0, seed=3):
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
    return {"value": accepted / el, "unit": "accepted MC moves/s", "cores": 1, "kind": kind,
            "sample": f"{trials} sequential translation/rotation trials ({2 * trials} Delta-E evaluations) of the same "
                      f"{system.n_atoms}-atom box in {el:.1f} s",
            "trial_moves_per_s": trials / el, "ns_per_dE_eval": el / (2 * trials) * 1e9,
            "acceptance": accepted / max(1, trials)}


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
        os.environ["OMP_PROC_BIND"] = "true"
        return [c for pl in places for c in pl]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--replicas", type=int, default=2048, help="independent chains per GPU")
    ap.add_argument("--n-side", type=int, default=15, help="SPC/E lattice side (15 -> 10 125 atoms)")
    ap.add_argument("--host", choices=["fortran", "python"], default="fortran",
                    help="Metropolis driver: the Fortran farm (mc_farm.f90, two overlapped lanes) or the numpy one")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="OpenMP threads of the Fortran driver per GPU (0: min(8, cores available / ranks on the node))")
    ap.add_argument("--no-pin", action="store_true", help="do not bind the host threads to the GPU's NUMA node")
    ap.add_argument("--lanes", type=int, default=2,
                    help="submission lanes (chain groups in flight) of the Fortran driver; 3 lanes x 1024 chains give ~20 %% more "
                         "moves/s because kernels of different lanes overlap, at the price of stretched per-kernel durations")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--settle-s", type=float, default=0.5,
                    help="untimed settle phase after the warm-up steps, seconds of the same step (0: none)")
    args = ap.parse_args()

    if args.host_threads <= 0:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        args.host_threads = max(1, min(8, len(os.sched_getaffinity(0)) // max(1, local_world)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    device = local_rank if args.device is None else args.device
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
        local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        pinned = pin_host_threads(torch, device, local_rank, local_world_size, args.host_threads)

    from maniac_mc_amd import _lib, synth
    system = synth.spce_box(args.n_side)
    t_step, r_step = 0.3, 0.3
    if args.host == "fortran":
        from maniac_mc_amd.fortran_host import FortranFarm as Farm
    else:
        from maniac_mc_amd.farm import ReplicaFarm as Farm
    kw = dict(n_threads=args.host_threads, n_lanes=args.lanes) if args.host == "fortran" else {}
    farm = Farm(system, args.replicas, device=device, seed=1000 + rank,
                translation_step=t_step, rotation_step=r_step, p_translation=0.5, **kw)
    eng = farm.eng
    N, Nk, R = system.n_atoms, eng.nk, args.replicas

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
    t0 = time.perf_counter()
    accepted = farm.run(args.steps)
    # the path's one real exchange step: per-block all-gather of every rank's counters and
    # molecule-count histogram (NVT here, so the histogram is a single bin; the message size is the same)
    n_now = [int(system.n_mol[0])] * R
    sums_by_rank, hist_by_rank = exchange.gather_block_stats(
        [float(accepted), float(args.steps * R)], exchange.molecule_count_histogram(n_now, 5001))
    fence()
    elapsed = exchange.max_over_ranks(time.perf_counter() - t0)
    tot_acc, tot_trials = float(sums_by_rank[:, 0].sum()), float(sums_by_rank[:, 1].sum())
    assert int(hist_by_rank.sum()) == R * world

    n_pair, ms_pair = eng.profile_get(_lib.KERNEL_PAIR)
    n_rec, ms_rec = eng.profile_get(_lib.KERNEL_RECIP)
    n_com, ms_com = eng.profile_get(_lib.KERNEL_COMMIT)

    # The same pair-sweep batch launched alone (outside the timed region, rank 0): with several lanes in
    # flight the kernels of different lanes share the CUs, which raises throughput but stretches every
    # kernel's own duration; this gives the kernel's un-shared time for comparison.
    iso_us = None
    if rank == 0 and args.host == "fortran":
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
        achieved = bytes_pair_eval * evals_per_launch / avg_pair_s / 1e9 if n_pair else 0.0
        evals_total = 2.0 * tot_trials
        traffic_eval = pmc_traffic_per_evaluation()
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
            "roofline": {"bound": "hbm", "kernel": "pair_sweep_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic_eval * evals_per_launch) if traffic_eval else None,
                         "traffic_source": "profiles/r01/pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
                         "algorithmic_bytes_per_launch": bytes_pair_eval * evals_per_launch,
                         "avg_launch_us": avg_pair_s * 1e6, "launches": n_pair,
                         "isolated": None if not iso_us else {
                             "avg_launch_us": iso_us, "achieved": bytes_pair_eval * evals_per_launch / (iso_us * 1e-6) / 1e9,
                             "frac": bytes_pair_eval * evals_per_launch / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "note": "same batch launched alone after the timed region; in the timed region the "
                                     "kernels of the lanes overlap on the device"},
                         "job_frac": (evals_total / world / elapsed) * bytes_eval / 1e9 / HBM_PEAK_GBS,
                         "recip_avg_launch_us": ms_rec / max(1, n_rec) * 1e3, "commit_avg_launch_us": ms_com / max(1, n_com) * 1e3},
        }
        if hasattr(farm, "timers"):
            out["host_seconds"] = {k: v - timers0[k] for k, v in farm.timers().items()}   # timed region only
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system, t_step, r_step, budget_s=args.cpu_budget)
        print(json.dumps(out))
    farm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
