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
# nested teams of the Fortran driver's lane-thread mode: in the environment before ANY OpenMP runtime starts
# (numpy / torch may load one on import)
os.environ.setdefault("OMP_MAX_ACTIVE_LEVELS", "2")
os.environ.setdefault("KMP_HOT_TEAMS_MAX_LEVEL", "2")
os.environ.setdefault("KMP_HOT_TEAMS_MODE", "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 fp64 lanes x 2 flop x 2.4 GHz (spec)
FP64_SUSTAINED_TFLOPS = 61.0    # tools/probe_fma.hip on MI355X: what back-to-back v_fma_f64 sustains (clock under load)
FLOP_PER_PAIR = 64.0            # ~64 fp64 flops per site-atom pair term incl. erfc (the ALGORITHMIC flop count)
PMC_DIR = os.path.join("profiles", "r05")


def lib_sha256():
    import hashlib
    from maniac_mc_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_summary(workload, kernel_substr):
    """Static figures from the committed rocprofv3 --pmc passes (tools/pmc_passes.sh; PMC cannot be collected from
    inside this process): HBM-side bytes (FETCH_SIZE doubled per the gfx950 calibration + WRITE_SIZE), VALU
    instructions and VALU-busy share of tools/bench_kernels.py's launch group for this workload, summed over the
    kernels whose name contains `kernel_substr` (a GCMC step launches two pair-sweep instantiations).  The summary
    names the sha256 of the libmaniac_hip.so it was taken from: if that is not the library loaded now, the
    counters are STALE and only that fact is returned.  None if there is no summary."""
    path = os.path.join(PMC_DIR, f"pmc_{workload}.json")
    try:
        with open(os.path.join(ROOT, path)) as f:
            d = json.load(f)
    except Exception:
        return None
    out = {"path": path, "build": d.get("build"), "lib_sha256": d.get("lib_sha256"), "candidates": float(d.get("candidates_per_launch") or 0.0),
           "evals": float(d.get("evaluations_per_launch_group") or 0.0), "kernels": []}
    # the counters belong to this build if the BINARY is the one profiled, or -- a rebuild changes the binary's hash through
    # its build id and output path, not its code -- if the library's SOURCES are the ones it was built from
    from maniac_mc_amd import _lib
    out["stale"] = d.get("lib_sha256") != lib_sha256() and d.get("source_digest") != _lib.source_digest()
    if out["stale"] or not out["evals"]:
        return out
    hbm = valu = 0.0
    busy_w = lds_w = wsum = 0.0
    for name, e in d["kernels"].items():
        if kernel_substr not in name:
            continue
        out["kernels"].append(name)
        hbm += e.get("hbm_bytes_per_launch", 0.0)
        valu += e.get("valu_instr_per_launch", 0.0)
        w = e.get("avg_us_under_pmc", 0.0)
        busy_w += w * e.get("valu_busy", 0.0)
        lds_w += w * e.get("lds_busy", 0.0)
        wsum += w
    out.update(hbm_bytes_per_eval=hbm / out["evals"], valu_instr_per_eval=valu / out["evals"],
               valu_busy=busy_w / wsum if wsum else None, lds_busy=lds_w / wsum if wsum else None, us_under_pmc=wsum)
    return out


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


def plan_host_threads(n_threads_req, local_world, want=6):
    cores = usable_cores()
    if n_threads_req > 0:
        return n_threads_req
    if want != 6:
        return max(1, min(want, cores // max(1, local_world)))
    # six threads per GPU: measured round 3 on the 16-CPU boxes (SPC/E: 4 / 6 / 8 threads 7.40 / 7.40 / 7.35 M accepted moves/s,
    # GPU-bound; CO2 GCMC 8192 x 4 lanes: 2 / 4 / 6 / 8 / 12 / 16 threads 7.8 / 9.3 / 10.5 / 6.6 / 7.3 / 6.7 M -- beyond six the
    # regions' fork / join and the threads' spinning cost more than the extra threads bring)
    return max(1, min(6, cores // max(1, local_world)))


def cpu_baseline_gcmc(system, t_act, p_move, translation_step, rotation_step, fugacity, budget_s=12.0, seed=3):
    """Grand-canonical workloads: the reference itself (oracle/_ref, 1 thread; the C restatement if that build is
    absent) evaluating the same kinds of trial -- translation / rotation with probability p_move, else insertion or
    deletion 50 / 50 -- one after the other on the workload's INITIAL configuration for a bounded time.  Every trial is
    evaluated and then undone (the acceptance test is applied for the count only), so the sample stays at the initial
    molecule count; energies per trial are what ComputeOldEnergy / ComputeNewEnergy compute."""
    from oracle import reflib, refcpu
    kind = "reference" if reflib.available() else "port"
    n = int(system.n_mol[t_act])
    if kind == "reference":
        X = reflib.Reference(system)
    else:
        X = refcpu.RefCPU(system, mol_capacity=n + 2)
    e_sys = X.system_energy()
    X.all_fourier_terms()
    X.init_amplitude(True)
    X.set_energy_recip(e_sys["recip_coulomb"])
    rng = np.random.default_rng(seed)
    T = system.temperature
    volume = float(abs(np.linalg.det(system.box_matrix)))
    L = np.diag(system.box_matrix)
    tmpl = system.offsets[t_act][0]
    trials = accepted = evals = 0
    t0 = time.perf_counter()
    while True:
        u = rng.random()
        if u < p_move:
            m = int(rng.integers(0, n))
            com, off = X.get_molecule(t_act, m)
            X.save_fourier(t_act, m)
            old = X.old_energy(t_act, m, 0)
            if rng.random() <= 0.5:
                X.set_molecule(t_act, m, X.apply_pbc(com + (rng.random(3) - 0.5) * translation_step), off)
            else:
                rot = X.rotation_matrix(int(rng.random() * 3) + 1, (rng.random() - 0.5) * rotation_step)
                X.set_molecule(t_act, m, com, off @ rot.T)
            new = X.new_energy(t_act, m, 0)
            prob = min(1.0, np.exp(-(new[5] - old[5]) / T))
            X.set_molecule(t_act, m, com, off)
            X.restore_fourier(t_act, m)
            evals += 2
        elif rng.random() < 0.5:
            A0 = X.amplitude()
            old = X.old_energy(t_act, n, 1)
            X.set_num_residues(t_act, n + 1)
            X.save_fourier(t_act, n)
            rot = X.rotation_matrix(int(rng.random() * 3) + 1, rng.random() * 2 * np.pi)
            X.set_molecule(t_act, n, system.bounds_lo + L * rng.random(3), tmpl @ rot.T)
            new = X.new_energy(t_act, n, 1)
            prob = min(1.0, fugacity * volume / (n + 1) * np.exp(-(new[5] - old[5]) / T))
            X.set_num_residues(t_act, n)
            X.set_amplitude(A0)
            evals += 1
        else:
            m = int(rng.integers(0, n))
            A0 = X.amplitude()
            old = X.old_energy(t_act, m, 2)
            X.save_fourier(t_act, m)
            u_new = X.recip_singlemol(t_act, m, 2)
            # new%total of a deletion: only the reciprocal term survives (monte_carlo_utils.f90:301-309)
            prob = min(1.0, n / (fugacity * volume) * np.exp(-(u_new - old[5]) / T))
            X.set_amplitude(A0)
            evals += 1
        trials += 1
        accepted += rng.random() <= prob
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": accepted / el, "unit": "accepted MC moves/s", "cores": 1, "kind": kind,
            "sample": f"{trials} sequential trials ({evals} Delta-E evaluations; moves with probability {p_move:g}, else "
                      f"insertion / deletion) on the workload's initial {system.n_atoms}-atom configuration in {el:.1f} s, "
                      "each evaluated and undone",
            "trial_moves_per_s": trials / el, "ns_per_dE_eval": el / max(1, evals) * 1e9, "acceptance": accepted / max(1, trials)}



def replicas_sweep(system, counts, device, host_threads, seconds, t_step, r_step, farm_kw=None):
    """Throughput against the number of chains a GPU holds (same SPC/E step, same Fortran farm, device-built moves).  A farm
    of R chains advances in lock step.  Up to 1024 chains a lane step is ONE launch (window mode: mgpu_farm_window_submit --
    the engine builds, evaluates, decides with the driver's draws and commits; mc_farm.f90 checks every decision against its
    own rule and keeps two windows in flight); above, the batched path (evaluation launches, the rule in Fortran, a commit
    launch), which a launch of a thousand chains or more fills."""
    from maniac_mc_amd.fortran_host import FortranFarm
    out = []
    for R in counts:
        # batched path, lanes: measured round 4 at the 10 125-atom box, accepted moves/s with 1 / 2 / 4 lanes: 8 chains 0.115 / 0.098 /
        # 0.10 M, 64: 0.72 / 0.75 / 0.69, 512: 2.6 / 2.8 / 1.3, 1024: - / 4.9 / 3.4, 2048: - / 6.3 / 5.7, 4096: - / 7.2 / 6.2,
        # 8192: - / 6.9 / 7.4, 16384: - / 7.2 / 7.7 (a lane's launch should fill the GPU: at nsplit 4 that takes ~1000 chains).
        # window mode, round 5 (profiles/r05/farm_window_speed.txt): ONE lane -- a second lane doubles the driver's launches per
        # lock step, and two lanes' launches sharing the chip scatter from box to box (512 chains: 3.9-6.3 M on two lanes,
        # 5.1-5.5 M on one)
        # insertion / deletion farms (farm_kw: the CO2 box; its steps are host-bound): windows up to the engine's 4096 chains
        # per launch, and from 1024 chains two lanes, each with a driver thread of its own (round 5, accepted moves/s, batched /
        # windows: 512 chains 2.5 / 8.5 M on one lane, 1024: 4.5 / 13.0 M, 4096: 10.2 / 12.6 M on two lanes with two drivers)
        gc = farm_kw is not None
        window = R <= (4096 if gc else 1024)
        drivers = 1
        if window:
            lanes = 2 if gc and R >= 1024 else 1
            drivers = lanes
        else:
            lanes = 2 if R < 8192 else 4
            drivers = 2 if gc and lanes == 4 else 1
        kw = farm_kw if gc else dict(p_translation=0.5)
        farm = FortranFarm(system, R, device=device, seed=77, translation_step=t_step, rotation_step=r_step,
                           n_threads=max(drivers, min(host_threads, 2 if R < 512 else (6 if gc else 4))), n_lanes=lanes,
                           n_drivers=drivers, device_build=True, window=window, window_depth=3 if gc else 2, **kw)
        try:
            farm.run(20)
            # (a chunk ends with mgpu_synchronize, which drains the windows in flight and copies A(k) back to its primary buffer:
            #  ~40 us that a real run pays once per block, so a window farm's chunk is a few thousand lock steps' worth of time)
            chunk = (400 if R <= 64 else 200) if farm.window else 50
            farm.run(chunk)
            farm.eng.synchronize()
            steps = acc = 0
            t0 = time.perf_counter()
            while True:
                acc += farm.run(chunk)
                steps += chunk
                farm.eng.synchronize()
                el = time.perf_counter() - t0
                if el >= seconds:
                    break
            out.append({"replicas": R, "lanes": farm.n_lanes, "value": acc / el, "unit": "accepted MC moves/s",
                        "trial_moves_per_s": steps * R / el, "steps": steps, "us_per_step": el / steps * 1e6,
                        "per_chain_steps_per_s": steps / el,
                        "drivers": drivers,
                        "path": (f"window: one launch per lane step, {farm.window_mode()[1]} in flight (mgpu_farm_window_submit)" if farm.window
                                 else "batched: mgpu_move_trial_submit / wait, rule in Fortran, mgpu_commit_submit"),
                        "steps_left_to_the_driver": farm.window_mode()[2],
                        "engine_nsplit_note": "engine constant of this replica count"})
        finally:
            farm.close()
    return out


def single_chain_leg(system, device, t_step, r_step, steps=4000, k=4):
    """ONE chain of the same box through the single-chain drop-in (mc_chain.f90 -> mgpu_chain_window: one kernel launch per
    window of up to k speculative steps, the launch also decides and commits): what a user with one chain gets."""
    import shutil
    import tempfile
    from maniac_mc_amd import io_maniac, run
    tmp = tempfile.mkdtemp(prefix="bench_chain_")
    try:
        files = io_maniac.write_input_files(system, os.path.join(tmp, "in"), nb_block=1, nb_step=steps, translation_step=t_step,
                                            rotation_step_angle=r_step, translation_proba=0.5, rotation_proba=0.5,
                                            masses=[15.9994, 1.008], atom_names=["OW", "HW"])
        res = run.run_simulation(*files, os.path.join(tmp, "out") + "/", seed=5, device=device, speculate=k)
        c = res["counters"]
        acc = int(c[1] + c[3] + c[5] + c[7])
        return {"replicas": 1, "driver": f"mc_chain.f90, one launch per window of up to {k} steps (mgpu_chain_window)",
                "value": acc / res["mc_seconds"], "unit": "accepted MC moves/s", "steps": steps,
                "per_chain_steps_per_s": steps / res["mc_seconds"], "us_per_step": res["mc_seconds"] / steps * 1e6,
                "windows": int(res["chain_windows"][0]), "steps_left_to_the_host": int(res["chain_windows"][1]),
                "note": "Monte Carlo loop alone (initial energy and output files excluded); the chain writes the reference's files"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config_legs(args, device):
    """BASELINE.json configs[2]-[4] as short self-timed legs: child processes of this one, one after the other (this
    process has closed its farm; a child is an ordinary `bench.py --workload ...` run and prints its own JSON line)."""
    import subprocess
    out = {}
    for wl in ("co2_gcmc", "framework_water", "co2_isotherm", "spce_triclinic", "adsorbate24"):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(args.config_steps), "--warmup", "20",
               "--sustained-steps", "0", "--configs", "0", "--replicas-sweep", "", "--cpu-budget", "1.0", "--cpu-all-cores-budget", "0",
               "--device", str(device)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.perf_counter()
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_PLACES", "OMP_PROC_BIND")}
        try:
            # (this process's main thread has been bound to one core by the Fortran driver's OpenMP runtime, and a child
            #  inherits the mask: give the child the whole original set back)
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env,
                               preexec_fn=lambda: os.sched_setaffinity(0, ORIG_AFFINITY))
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[wl] = {"error": (p.stderr or p.stdout)[-400:], "returncode": p.returncode}
                continue
            d = json.loads(line[-1])
        except Exception as exc:
            out[wl] = {"error": str(exc)}
            continue
        r = d["roofline"]
        roof = {k: r.get(k) for k in ("bound", "basis", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us")}
        iso = r.get("isolated") or {}
        if r.get("measured") and iso.get("measured_traffic_frac") is not None:
            # The k sweep only reads A(k): the rate on the bytes the counters saw is the one that means something, and the
            # kernel's own efficiency is what it reaches ALONE (in the farm's pipeline four lanes' kernels share the CUs,
            # so a launch's begin-to-end time is mostly waiting for them: `in_pipeline`)
            roof.update({"basis": "measured_traffic, one lane's batch launched alone", "achieved": iso["measured_traffic_GBs"],
                         "frac": iso["measured_traffic_frac"], "avg_launch_us": iso["avg_launch_us"]})
            roof["in_pipeline"] = {k: r["measured"][k] for k in ("basis", "achieved", "frac", "avg_launch_us")}
            roof["algorithmic"] = {"achieved": r.get("achieved"), "frac": r.get("frac"), "basis": "algorithmic_bytes (52 Nk per evaluation), in the pipeline"}
        elif iso:
            roof["isolated_frac"] = iso.get("frac")
        out[wl] = {"baseline_config": d["config"]["baseline_config"], "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                   "steps": d["steps"], "ms_per_step": d["ms_per_step"], "timed_region_s": d["timed_region_s"],
                   "replicas_per_gpu": d["config"]["replicas_per_gpu"], "acceptance": d["acceptance"],
                   "trial_moves_per_s": d["trial_moves_per_s"], "ns_per_dE_eval": d["ns_per_dE_eval"], "roofline": roof,
                   "cpu_baseline": d.get("cpu_baseline"), "leg_seconds": time.perf_counter() - t0}
        if "isotherm" in d:
            out[wl]["isotherm_mean_N"] = [round(pt["mean_N"], 2) for pt in d["isotherm"]]
    return out


WORKLOADS = {
    # name: default chains per GPU, lanes, what BASELINE.json calls it
    # device_build: the engine keeps the molecules' frames and builds the trial geometry itself (no host mirror, no candidate
    # rows staged; measured round 3, SPC/E: 1 / 2 / 6 host threads 5.25 / 7.02 / 7.41 M host-built, 7.24 / 7.31 / 7.32 M
    # device-built -- one host thread per GPU then keeps the GPU busy)
    # device_accept (opt-in, --device-accept 1): the engine also applies the acceptance rule behind the k sweep and commits
    # accepted candidates in the same workgroup (mgpu_move_trial_decide_submit): no commit launch, no second upload, no host
    # commit phase.  The DEFAULT keeps the rule in the Fortran driver, as BASELINE.json's north star asks ("the sequential
    # Metropolis acceptance loop stays on the host in Fortran").  Host threads, measured round 3 on a 16-CPU box (drivers x
    # team; host rule / device rule, M accepted moves/s): SPC/E 1 x 2 7.39 / 7.59, 1 x 4 7.36 / 7.60, 2 x 2 7.35 / 7.34;
    # CO2 2 x 2 19.0-19.8 / 21.9, 3 x 2 19.6-20.0 / 21.5, 2 x 4 16.7 / 19.9; framework + water 2 x 2 5.3-6.0 / 6.27,
    # 3 x 2 5.8-6.05 / 6.12, 2 x 4 5.4 / 4.7 -- teams of two (larger teams lose to their fork / join); two drivers are
    # 82 % busy at the CO2 box and fall behind on a slow host (15.1 M on one box), three have margin: default 3 x 2 there.
    # 16384 chains on four lanes (engine nsplit 1: a lane step is 4096 fused moves = one work unit per resident wave, each a
    # whole sweep): 8192 / 16384 / 32768 chains 7.34 / 7.56 / 7.52 M on one box (round 3)
    "spce": dict(replicas=16384, lanes=4, drivers=1, threads=4, device_build=1, device_accept=0, config="metric workload: 10 125-atom SPC/E box (configs[1] recipe at 15^3)"),
    # the grand-canonical boxes are small (a few hundred atoms): per lane step the fixed host costs (OpenMP regions, HIP calls)
    # weigh as much as the kernels, so they run MANY chains (measured round 3, co2_gcmc, one driver thread: 2048 x 2 lanes
    # 5.4 M, 8192 x 4 6.6 M, 8192 x 2 11.2 M, 16384 x 2 13.7 M accepted moves/s) and TWO host driver threads sharing four
    # lanes (the SPC/E box is GPU-bound and gains nothing from a second driver)
    "co2_gcmc": dict(replicas=16384, lanes=4, drivers=3, threads=6, device_build=1, device_accept=0, config="configs[2]: GCMC of CO2 in a 50 A box, insertion / deletion at one fugacity"),
    "framework_water": dict(replicas=8192, lanes=4, drivers=3, threads=6, device_build=1, device_accept=0, config="configs[3] stand-in: 2208-atom framework + 4-site water, full move set"),
    "co2_isotherm": dict(replicas=16384, lanes=4, drivers=3, threads=6, device_build=1, device_accept=0, config="configs[4]: 8 fugacity points dealt over the ranks"),
    # the two regimes the parity tests exercise that had never been timed (round 5): a TRICLINIC box -- the metric's
    # 10 125-atom SPC/E box with the tilt of the spce_triclinic_nvt fixture; ComputeDistance's 27-image search,
    # src/geometry_utils.f90:397-411 -- and a molecule beyond the register-site sweeps and the row-form k sweep (a 24-site
    # rigid adsorbate: LDS-staged NS = 0 sweep, per-k reciprocal kernel)
    "spce_triclinic": dict(replicas=4096, lanes=4, drivers=1, threads=4, device_build=0, device_accept=0, config="the metric's SPC/E box, triclinic (tilt 3.0 / -2.0 / 1.5 A): parity-tested regime, timed"),
    "adsorbate24": dict(replicas=4096, lanes=4, drivers=1, threads=4, device_build=0, device_accept=0, config="64 rigid 24-site adsorbates (1536 atoms) in a 60 A box: parity-tested regime, timed"),
}
NVT_WORKLOADS = ("spce", "spce_triclinic", "adsorbate24")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--replicas", type=int, default=None,
                    help="independent chains per GPU (default: 8192 for the SPC/E and framework workloads, 16384 for the CO2 ones)")
    ap.add_argument("--n-side", type=int, default=15, help="SPC/E lattice side (15 -> 10 125 atoms)")
    ap.add_argument("--host", choices=["fortran", "python"], default="fortran",
                    help="Metropolis driver: the Fortran farm (mc_farm.f90, overlapped lanes) or the numpy one")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="OpenMP threads of the Fortran driver per GPU (0: min(6, cores available / ranks on the node))")
    ap.add_argument("--no-pin", action="store_true", help="do not bind the host threads to the GPU's NUMA node")
    ap.add_argument("--lanes", type=int, default=None,
                    help="submission lanes (chain groups in flight) of the Fortran driver: the host prepares / resolves one "
                         "group while the GPU evaluates the others; default 4")
    ap.add_argument("--drivers", type=int, default=None,
                    help="host driver threads of the Fortran farm that share the lanes, each with a team of host-threads / drivers "
                         "(default 1 for SPC/E, 2 for the grand-canonical workloads)")
    ap.add_argument("--device-accept", type=int, default=None, choices=[0, 1],
                    help="1: the engine applies the acceptance rule behind the k sweep and commits accepted candidates itself "
                         "(mgpu_move_trial_decide_submit; needs --device-build 1); 0: the Fortran driver decides and commits "
                         "(default: the workload's setting)")
    ap.add_argument("--device-build", type=int, default=None, choices=[0, 1],
                    help="1: the engine keeps the molecules' frames and builds the trial moves on the device (no host mirror, no "
                         "candidate rows staged); 0: the Fortran driver builds them from its mirror (default: see WORKLOADS)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--exchange", choices=["torch", "c-abi"], default=None,
                    help="the per-block all-gather of counters + uptake histogram: through the C ABI (mgpu_comm_create / "
                         "mgpu_allgather_block_stats: RCCL called from libmaniac_hip.so, what a Fortran host uses; the id travels "
                         "over the torch.distributed group) -- the default with --host fortran and the nccl backend -- or through "
                         "torch.distributed (the default otherwise).  If the C-ABI communicator cannot be created on every rank "
                         "within 90 s the ranks agree to fall back to torch.distributed and the line's `exchange` block says so.")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timing", type=int, default=1, choices=[0, 1],
                    help="0 (diagnostic): no HIP events on the launches of the timed region -- the per-launch figures of the roofline "
                         "block are then empty; shows what the event bookkeeping costs a host-bound workload")
    ap.add_argument("--cpu-budget", type=float, default=4.0, help="seconds of the 1-core reference leg")
    ap.add_argument("--cpu-all-cores-budget", type=float, default=2.0,
                    help="seconds of the labelled all-core OpenMP leg of the C restatement (0: skip; SPC/E workload only)")
    ap.add_argument("--settle-s", type=float, default=0.5,
                    help="untimed settle phase after the warm-up steps, seconds of the same step (0: none)")
    ap.add_argument("--sustained-steps", type=int, default=1000,
                    help="a second, self-timed window of this many steps after the timed region when --steps is smaller "
                         "(reported as `sustained`; 0: none)")
    ap.add_argument("--fugacity-n", type=float, default=None,
                    help="grand-canonical workloads: fugacity as the ideal-gas molecule count f V (default 100 for co2_gcmc, "
                         "40 for framework_water)")
    ap.add_argument("--workload", choices=list(WORKLOADS), default="spce",
                    help="spce: the 10 125-atom SPC/E box, translation / rotation (BASELINE metric, default); co2_gcmc: "
                         "configs[2]; framework_water: configs[3]; co2_isotherm: configs[4] (8 fugacity points dealt over "
                         "the ranks, per-block gather of the uptake histogram)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank reports its placement (device, host threads, fugacity points) and "
                         "rank 0 prints the rank-ordered table gathered over the process group")
    ap.add_argument("--dump-counts", default=None, help="directory: every rank writes its chains' final molecule counts (tests)")
    ap.add_argument("--configs", type=int, default=None, choices=[0, 1],
                    help="1: after the headline workload, time the other BASELINE.json configs (co2_gcmc, framework_water, "
                         "co2_isotherm) as short self-timed legs (child processes of this one, one after the other) and put them "
                         "into the same JSON line as `configs` (default: 1 for the default single-GPU SPC/E run)")
    ap.add_argument("--config-steps", type=int, default=300, help="steps of each `configs` leg")
    ap.add_argument("--replicas-sweep", default=None,
                    help="comma-separated chain counts: after the headline region, the same SPC/E step with that many chains per "
                         "GPU (`replicas_sweep` in the JSON line: throughput against chain count), plus the single-chain driver "
                         "(mc_chain.f90, one launch per window); default 1,8,64,512,4096 for the default single-GPU SPC/E run, "
                         "'' for none")
    ap.add_argument("--sweep-seconds", type=float, default=0.4, help="timed seconds per point of --replicas-sweep")
    args = ap.parse_args()

    wl = args.workload
    default_run = wl == "spce" and args.replicas is None and args.gpus <= 1 and args.host == "fortran" and not args.dry_run
    if args.configs is None:
        args.configs = 1 if default_run else 0
    if args.replicas_sweep is None:
        args.replicas_sweep = "1,8,64,512,1024,4096" if default_run else ""
    if args.replicas is None:
        args.replicas = WORKLOADS[wl]["replicas"]
    if args.lanes is None:
        args.lanes = WORKLOADS[wl]["lanes"]
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    args.host_threads = plan_host_threads(args.host_threads, local_world, WORKLOADS[wl]["threads"])
    if args.drivers is None:
        args.drivers = WORKLOADS[wl]["drivers"]
    if args.host_threads < 2 * args.drivers:
        args.drivers = 1
    device = local_rank if args.device is None else args.device

    if args.dry_run:
        # placement rehearsal (CPU only): same process group, same exchange call, no engine
        from maniac_mc_amd import exchange
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo" if args.dist_backend != "gloo" else args.dist_backend)
        pts = isotherm_points_of_rank(rank, world) if wl == "co2_isotherm" else []
        row = [float(rank), float(local_rank), float(device), float(args.host_threads), float(args.replicas),
               float(len(pts)), float(pts[0] if pts else -1)]
        table, _ = exchange.gather_block_stats(row)
        if rank == 0:
            # the exchange block a real run's line carries (values are placeholders here: no engine ran)
            want = args.exchange or ("c-abi" if (args.host == "fortran" and args.dist_backend == "nccl") else "torch")
            print(json.dumps({"dry_run": True, "n_gpus": world, "workload": wl,
                              "exchange": {"path": "torch", "would_default_to": want, "ranks_seen": int(table.shape[0]),
                                           "per_rank_value": [0.0] * int(table.shape[0]),
                                           "per_rank_device": [int(r[2]) for r in table]},
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
    if wl not in NVT_WORKLOADS and args.host != "fortran":
        sys.exit("bench.py: the grand-canonical workloads run on the Fortran farm")
    if args.device_build is None:
        args.device_build = WORKLOADS[wl].get("device_build", 0)
    if args.device_accept is None:
        args.device_accept = WORKLOADS[wl].get("device_accept", 0)
    args.device_accept = int(bool(args.device_accept and args.device_build))
    kw = dict(n_threads=args.host_threads, n_lanes=args.lanes, n_drivers=args.drivers,
              device_build=bool(args.device_build), device_accept=bool(args.device_accept)) if args.host == "fortran" else {}
    R = args.replicas
    iso_pts, fug_grid, point_of_chain = None, None, None
    t_act, p_move, fug_one = 0, 1.0, None        # active residue type, share of translation + rotation, the fugacity
    if wl in NVT_WORKLOADS:
        if wl == "spce":
            system = synth.spce_box(args.n_side)
        elif wl == "spce_triclinic":
            system = synth.spce_box(args.n_side)
            L = float(system.box_matrix[0, 0])
            # rows a = (lx, 0, 0), b = (xy, ly, 0), c = (xz, yz, lz), as the reference's reader stores them (readers_utils.f90:242-245)
            system.box_matrix = np.array([[L, 0.0, 0.0], [3.0, L, 0.0], [-2.0, 1.5, L]])
            # the molecules' centres sheared with the cell (fractional coordinates kept; the distance routine's cell vectors
            # are the matrix's columns, geometry_utils.f90:126-129), so that the periodic images still fit each other
            frac = (system.com[0] - system.bounds_lo[None, :]) / L
            system.com[0] = system.bounds_lo[None, :] + frac @ system.box_matrix.T
        else:
            system = synth.rigid_adsorbate_box(n_mol=64, L=60.0, seed=17)
        t_step, r_step = 0.3, 0.3
        moves = "50% translation / 50% rotation, 0.3 A / 0.3 rad, 300 K"
        farm = Farm(system, R, device=device, seed=1000 + rank,
                    translation_step=t_step, rotation_step=r_step, p_translation=0.5, **kw)
    elif wl == "framework_water":
        # configs[3] stand-in: inactive 2208-atom framework, 4-site water adsorbate, full move set
        system = synth.framework_water_box()
        volume = float(np.prod(np.diag(system.box_matrix)))
        t_act, p_move = 1, 0.5
        fug_one = (args.fugacity_n if args.fugacity_n else 40.0) / volume
        t_step, r_step = 0.5, 0.5
        moves = "25% translation / 25% rotation / 50% insertion-deletion, 0.5 A / 0.5 rad, 300 K"
        farm = Farm(system, R, device=device, seed=1000 + rank, translation_step=t_step, rotation_step=r_step,
                    mol_capacity=[1, 256], gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=fug_one), **kw)
    else:
        # configs[2] / [4]: rigid 3-site CO2, empty-ish cubic 50 A box (Nk = 2975)
        system = synth.co2_box(64, seed=13)
        volume = float(np.prod(np.diag(system.box_matrix)))
        t_step, r_step = 1.0, 0.6
        if wl == "co2_gcmc":
            # configs[2]: insertion / deletion only, one fugacity
            p_move = 0.0
            fug_one = (args.fugacity_n if args.fugacity_n else 100.0) / volume
            fug = fug_one
            moves = "100% insertion / deletion (50 / 50), 300 K"
        else:
            # configs[4]: 25 % translation, 25 % rotation, 50 % insertion / deletion; this rank's chains are split evenly
            # over the fugacity points it holds
            p_move = 0.5
            fug_grid = isotherm_fugacities(volume)
            iso_pts = isotherm_points_of_rank(rank, world)
            point_of_chain = np.array([iso_pts[(i * len(iso_pts)) // R] for i in range(R)])
            fug = fug_grid[point_of_chain]
            moves = "25% translation / 25% rotation / 50% insertion-deletion, 300 K"
        farm = Farm(system, R, device=device, seed=1000 + rank, translation_step=t_step, rotation_step=r_step,
                    mol_capacity=[400], gcmc=dict(p_translation=p_move / 2, p_rotation=p_move / 2, fugacity=fug), **kw)
    eng = farm.eng
    N0, Nk = system.n_atoms, eng.nk
    n1 = int(system.topo.atoms_in_res[t_act])

    def evals_done():
        """Delta-E evaluations so far: a translation / rotation trial costs two (old, new), an insertion or a
        deletion one"""
        if not hasattr(farm, "counters"):
            return 2.0 * farm.trials if hasattr(farm, "trials") else 0.0
        c = farm.counters()
        return 2.0 * (c["trial_translations"] + c["trial_rotations"]) + c["trial_creations"] + c["trial_deletions"]

    # profiling on BEFORE the warm-up: the first event-carrying dispatch of a stream costs ~7 ms once
    eng.profile_enable(bool(args.kernel_timing))
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
    comm = None
    exchange_note = None
    if args.exchange is None:
        args.exchange = "c-abi" if (args.host == "fortran" and (world == 1 or args.dist_backend == "nccl")) else "torch"
    if args.exchange == "c-abi":
        uid = [exchange.CAbiComm.unique_id() if (rank == 0 and world > 1) else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        # RCCL's communicator creation is collective: made on a side thread with a deadline, and the ranks then AGREE (over
        # the torch.distributed group) whether every one of them has it -- a rank that failed or is still waiting makes all
        # of them use the torch.distributed gather instead, and the line says so
        import threading
        box = {}

        def make():
            try:
                box["comm"] = exchange.CAbiComm(device=device, rank=rank, world=world, unique_id=uid[0])
            except Exception as exc:                       # noqa: BLE001 -- reported in the line
                box["error"] = str(exc)
        th = threading.Thread(target=make, daemon=True)
        th.start()
        th.join(90.0)
        ok = 1.0 if "comm" in box else 0.0
        all_ok = exchange.min_over_ranks(ok) if world > 1 else ok
        if all_ok >= 1.0:
            comm = box["comm"]
        else:
            exchange_note = box.get("error") or ("this rank's communicator was not ready after 90 s" if not ok else "another rank has no communicator")
    gather = comm.gather_block_stats if comm is not None else exchange.gather_block_stats

    def fence():
        exchange.barrier()
        torch.cuda.synchronize()

    fence()
    trials0 = farm.trials if hasattr(farm, "trials") else 0
    evals0 = evals_done()
    t0 = time.perf_counter()
    accepted = farm.run(args.steps)
    # the path's one real exchange step: per-block all-gather of every rank's counters and
    # molecule-count histogram (NVT: a single bin per rank, the message size is the same; grand-canonical: one
    # histogram of the chains' current N per fugacity point, ISOTHERM_POINTS x 5001 bins for the isotherm)
    nbins = 5001
    trials_now = float(farm.trials - trials0) if hasattr(farm, "trials") else float(args.steps * R)
    evals_now = evals_done() - evals0
    counts = None
    if wl in NVT_WORKLOADS:
        hist = exchange.molecule_count_histogram(np.full(R, int(system.n_mol[0]), dtype=np.int64), nbins)   # (an array: a 16 384-element Python list cost 0.8 ms of the timed region)
    else:
        counts = farm.counts()[:, 0]
        if wl == "co2_isotherm":
            hist = np.concatenate([exchange.molecule_count_histogram(counts[point_of_chain == p], nbins)
                                   for p in range(ISOTHERM_POINTS)])
        else:
            hist = exchange.molecule_count_histogram(counts, nbins)
    sums_by_rank, hist_by_rank = gather([float(accepted), trials_now, evals_now, float(device)], hist)
    fence()
    elapsed = exchange.max_over_ranks(time.perf_counter() - t0)
    tot_acc, tot_trials, tot_evals = (float(sums_by_rank[:, k].sum()) for k in range(3))
    assert int(hist_by_rank.sum()) == R * world
    if args.dump_counts and wl == "co2_isotherm":
        os.makedirs(args.dump_counts, exist_ok=True)
        np.savez(os.path.join(args.dump_counts, f"rank{rank}.npz"), counts=counts, point_of_chain=point_of_chain)

    n_pair, ms_pair = eng.profile_get(_lib.KERNEL_PAIR)
    n_rec, ms_rec = eng.profile_get(_lib.KERNEL_RECIP)
    n_com, ms_com = eng.profile_get(_lib.KERNEL_COMMIT)
    timers1 = farm.timers() if hasattr(farm, "timers") else None

    # A second, self-timed window when the timed region is a short sample (the driver's --steps 20 is ~17 ms):
    # the same step for --sustained-steps more steps, same fences, reported beside the headline figure.
    sustained = None
    if args.sustained_steps > args.steps:
        fence()
        ts0 = time.perf_counter()
        acc_s = farm.run(args.sustained_steps)
        acc_s = float(gather([float(acc_s)], hist)[0][:, 0].sum())
        fence()
        el_s = exchange.max_over_ranks(time.perf_counter() - ts0)
        sustained = {"steps": args.sustained_steps, "value": acc_s / el_s, "unit": "accepted MC moves/s",
                     "ms_per_step": el_s / args.sustained_steps * 1e3, "timed_region_s": el_s,
                     "note": "second self-timed window after the headline region (same step, same fences)"}
    elif args.sustained_steps > 0:
        sustained = {"steps": args.steps, "value": tot_acc / elapsed, "unit": "accepted MC moves/s",
                     "ms_per_step": elapsed / args.steps * 1e3, "timed_region_s": elapsed, "note": "the timed region itself"}

    # The dominant kernel's batch launched alone (outside the timed region, rank 0): with several lanes in flight
    # the kernels of different lanes share the CUs, which raises throughput but stretches every kernel's own
    # duration; this gives the kernel's un-shared time for comparison.
    iso_us = None
    if not args.kernel_timing:
        eng.profile_enable(True)           # the isolated legs below are timed either way
    if rank == 0 and args.host == "fortran" and wl in NVT_WORKLOADS:
        rng = np.random.default_rng(5)
        n_l = max(1, R // farm.n_lanes)
        m_iso = rng.integers(0, int(system.n_mol[0]), n_l).astype(np.int32)
        sites_iso = system.all_sites(0)[m_iso] + rng.uniform(-0.15, 0.15, (n_l, 1, 3))
        eng.profile_reset()
        for _ in range(10):
            eng.trial_energy_candidates(np.arange(n_l, dtype=np.int32), np.zeros(n_l, np.int32), m_iso, sites_iso)
        n_iso, ms_iso = eng.profile_get(_lib.KERNEL_PAIR)
        iso_us = ms_iso / max(1, n_iso) * 1e3
    iso_gc = None
    if rank == 0 and wl not in NVT_WORKLOADS and args.device_build:
        # grand-canonical workloads: one lane's batch of device-built trials (the farm's move mix on the chains' current
        # states) launched alone, evaluation only
        rng = np.random.default_rng(5)
        n_l = max(1, R // farm.n_lanes)
        cnt_l = farm.counts()[:n_l, 0]
        draw = rng.random(n_l)
        mv = np.where(draw < p_move / 2, 1, np.where(draw < p_move, 2, np.where(rng.random(n_l) < 0.5, 3, 4))).astype(np.int32)
        mv[(cnt_l == 0) & (mv != 3)] = 3
        m_iso = np.minimum((rng.random(n_l) * np.maximum(cnt_l, 1)).astype(np.int32), np.maximum(cnt_l, 1) - 1).astype(np.int32)
        u_iso = rng.random((n_l, 5))
        eng.profile_reset()
        for _ in range(10):
            eng.move_trial(np.arange(n_l, dtype=np.int32), np.full(n_l, t_act, np.int32), m_iso, mv, u_iso, t_step, r_step)
        ev_iso = float(2 * (mv <= 2).sum() + (mv > 2).sum())
        n_ip, ms_ip = eng.profile_get(_lib.KERNEL_PAIR)
        n_ik, ms_ik = eng.profile_get(_lib.KERNEL_RECIP)
        iso_gc = {"candidates": int(n_l), "evaluations": ev_iso, "pair_sweep_us": ms_ip / 10 * 1e3, "k_sweep_us": ms_ik / max(1, n_ik) * 1e3}
    eng.profile_enable(False)

    if rank == 0:
        n_lanes = farm.n_lanes if args.host == "fortran" else 1
        evals_rank = evals_now                                   # this rank's evaluations in the timed region
        # mean atom count of this rank's chains now (the grand-canonical boxes fill / drain during the run)
        n_act_mean = float(np.mean(counts)) if counts is not None else float(system.n_mol[t_act])
        N = N0 + (n_act_mean - float(system.n_mol[t_act])) * n1
        # ---- algorithmic work per Delta-E evaluation: 36 N + 52 Nk bytes; 64 flop per site-atom pair
        bytes_pair_eval, bytes_k_eval = 36.0 * N, 52.0 * Nk
        flop_eval = FLOP_PER_PAIR * n1 * max(0.0, N - n1)
        us = lambda ms, n: ms / max(1, n) * 1e3
        kernels = {"pair_sweep": {"launches": n_pair, "avg_launch_us": us(ms_pair, n_pair), "total_ms": ms_pair},
                   "k_sweep": {"launches": n_rec, "avg_launch_us": us(ms_rec, n_rec), "total_ms": ms_rec},
                   "commit": {"launches": n_com, "avg_launch_us": us(ms_com, n_com), "total_ms": ms_com}}
        ms_all = ms_pair + ms_rec + ms_com
        for k in kernels.values():
            k["share_of_kernel_time"] = k["total_ms"] / ms_all if ms_all else None
        k_launches = max(1, n_rec)
        evals_per_k_launch = evals_rank / k_launches             # one k sweep per lane step covers every candidate of it
        pair_dominant = wl in NVT_WORKLOADS or wl == "framework_water"
        if pair_dominant:
            # Pair sweep: bound by fp64 vector issue, not by HBM (measured HBM-side traffic is a fraction of the
            # algorithmic bytes).  achieved = ALGORITHMIC flops (64 per site-atom pair term) over time.
            pmc = pmc_summary(wl, "::pair_")
            fresh = bool(pmc) and not pmc["stale"] and pmc.get("valu_instr_per_eval") is not None
            job_tflops = evals_rank * flop_eval / elapsed / 1e12
            launch_tflops = evals_rank * flop_eval / (ms_pair * 1e-3) / 1e12 if ms_pair else None
            roof = {"bound": "valu", "basis": "algorithmic_flops", "bound_note": "fp64 vector issue (the contract's hbm / mfma do not apply: measured HBM-side "
                                                   "traffic is a fraction of the algorithmic bytes and there is no MFMA-shaped work; SURVEY 8(d))",
                    "kernel": "pair_frozen_kernel<4,...> (framework box: 64 candidates in the lanes of a wave against chunks of 32 framework atoms "
                              "held as scalars, then each lane's own adsorbates; a group's last workgroup adds the chunk partials in order)" if wl == "framework_water"
                              else {"spce": "pair_sweep_kernel<3,false,false,true,true> (old + new state of a trial move in one sweep, two-instruction fold)",
                                    "spce_triclinic": "the triclinic pair sweep of this build (see DESIGN 4.1: image search of ComputeDistance, geometry_utils.f90:397-411)",
                                    "adsorbate24": "pair_sweep_kernel<0,false,false> (24 sites staged through LDS in chunks: the generic site count)"}[wl],
                    "achieved": job_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": job_tflops / FP64_VECTOR_PEAK_TFLOPS,
                    "achieved_basis": f"ALGORITHMIC flops: {FLOP_PER_PAIR:g} per site-atom pair term x {n1} sites x (N - {n1}) atoms = "
                                      f"{flop_eval / 1e6:.3f} MFLOP per evaluation (SURVEY 8(d)), N = {N:.0f} (mean over the chains); "
                                      "job level: all evaluations of the timed region / timed_region_s (every other kernel, every gap "
                                      "and the host's share included)",
                    "flop_per_evaluation": flop_eval, "evaluations": evals_rank,
                    "per_launch": {"achieved": launch_tflops, "frac": launch_tflops / FP64_VECTOR_PEAK_TFLOPS if launch_tflops else None,
                                   "avg_launch_us": us(ms_pair, n_pair), "launches": n_pair,
                                   "note": "same flops over the SUM of the pair-sweep launches' begin-to-end times (dispatch events in the "
                                           f"timed region, what rocprofv3 --kernel-trace shows): with {n_lanes} lanes in flight a launch's "
                                           "time includes waiting for CUs held by the other lanes' kernels"},
                    "traffic": pmc["hbm_bytes_per_eval"] * evals_rank / max(1, n_pair) if fresh else None,
                    "traffic_per_evaluation": pmc["hbm_bytes_per_eval"] if fresh else None,
                    "traffic_source": (f"static: {pmc['path']} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, build {pmc['build']}, "
                                       f"libmaniac_hip.so sha256 {pmc['lib_sha256'][:16]} = the library loaded now; per evaluation x "
                                       "this run's evaluations per launch)") if fresh else
                                      ("STALE: " + pmc["path"] + " was taken from another build of libmaniac_hip.so" if pmc else "none"),
                    "valu_issue": {"instr_per_evaluation": pmc["valu_instr_per_eval"], "valu_busy": pmc["valu_busy"], "lds_busy": pmc["lds_busy"],
                                   "issue_slots_frac_of_peak": pmc["valu_instr_per_eval"] * evals_rank * 128.0 / elapsed / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                   "note": "VALU wave-instructions (SQ_INSTS_VALU, static PMC) x 128 flop-slots / timed_region_s / fp64 vector "
                                           "peak: issue-slot utilisation (integer, compare and conversion instructions count as slots), "
                                           "not a flop rate"} if fresh else None,
                    "hbm": {"algorithmic_bytes_per_evaluation": bytes_pair_eval,
                            "algorithmic_equivalent_GBs": bytes_pair_eval * evals_rank / (ms_pair * 1e-3) / 1e9 if ms_pair else None,
                            "peak": HBM_PEAK_GBS,
                            "note": "36 N bytes per evaluation (SURVEY 8(d)) / pair-sweep time: an algorithmic figure served mostly from "
                                    "L2 / Infinity Cache, NOT an HBM utilisation"}}
            if iso_gc:
                roof["isolated"] = {"avg_launch_us": iso_gc["pair_sweep_us"], "evaluations": iso_gc["evaluations"],
                                    "achieved": iso_gc["evaluations"] * flop_eval / (iso_gc["pair_sweep_us"] * 1e-6) / 1e12,
                                    "frac": iso_gc["evaluations"] * flop_eval / (iso_gc["pair_sweep_us"] * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                    "note": "one lane's batch of device-built trials launched alone after the timed region (pair-sweep "
                                            "time per launch group)"}
            if iso_us:
                ev_l = 2.0 * max(1, R // n_lanes)
                roof["isolated"] = {"avg_launch_us": iso_us, "achieved": ev_l * flop_eval / (iso_us * 1e-6) / 1e12,
                                    "frac": ev_l * flop_eval / (iso_us * 1e-6) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                    "note": "one lane's batch launched alone after the timed region (the kernel's own efficiency)"}
        else:
            # k sweep: one pass over A(k) per candidate, HBM-bound: 52 Nk algorithmic bytes per evaluation
            pmc = pmc_summary(wl if wl != "co2_isotherm" else "co2_gcmc", "recip_rows_kernel<false")
            fresh = bool(pmc) and not pmc["stale"] and pmc.get("hbm_bytes_per_eval") is not None
            gbs = bytes_k_eval * evals_rank / (ms_rec * 1e-3) / 1e9 if ms_rec else None
            k_item_bytes = pmc["hbm_bytes_per_eval"] * pmc["evals"] / pmc["candidates"] if fresh and pmc.get("candidates") else 0.0
            roof = {"bound": "hbm", "basis": "algorithmic_bytes", "kernel": ("recip_rows_kernel<false,true,true> (k sweep: old and new reciprocal energy of every candidate from one pass "
                                               "over A(k); its workgroup then applies the acceptance rule and commits an accepted candidate -- a second "
                                               "pass A <- A + delta -- so avg_launch_us covers sweep AND commit)") if args.device_accept else
                                              "recip_rows_kernel<false,true> (k sweep: old and new reciprocal energy of every candidate from one pass over A(k))",
                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                    "achieved_basis": f"ALGORITHMIC bytes: 52 Nk = {bytes_k_eval:.0f} B per evaluation (SURVEY 8(d)) x "
                                      f"{evals_per_k_launch:.0f} evaluations per launch / the k sweep's average launch time (dispatch events "
                                      "in the timed region)",
                    "avg_launch_us": us(ms_rec, n_rec), "launches": n_rec, "evaluations_per_launch": evals_per_k_launch,
                    "traffic": pmc["hbm_bytes_per_eval"] * evals_per_k_launch if fresh else None,
                    "traffic_per_evaluation": pmc["hbm_bytes_per_eval"] if fresh else None,
                    "traffic_source": (f"static: {pmc['path']} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, build {pmc['build']}, "
                                       f"libmaniac_hip.so sha256 {pmc['lib_sha256'][:16]} = the library loaded now; per evaluation x "
                                       "this run's evaluations per launch)") if fresh else
                                      ("STALE: " + pmc["path"] + " was taken from another build of libmaniac_hip.so" if pmc else "none"),
                    "isolated": ({"avg_launch_us": iso_gc["k_sweep_us"], "evaluations": iso_gc["evaluations"],
                                  "achieved": bytes_k_eval * iso_gc["evaluations"] / (iso_gc["k_sweep_us"] * 1e-6) / 1e9,
                                  "frac": bytes_k_eval * iso_gc["evaluations"] / (iso_gc["k_sweep_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  "measured_traffic_GBs": k_item_bytes * iso_gc["candidates"] / (iso_gc["k_sweep_us"] * 1e-6) / 1e9 if fresh else None,
                                  "measured_traffic_frac": k_item_bytes * iso_gc["candidates"] / (iso_gc["k_sweep_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS if fresh else None,
                                  "note": "one lane's batch launched alone after the timed region, evaluation only (no acceptance, no commit "
                                          "pass); measured_traffic = the PMC bytes per "
                                          "CANDIDATE (one pass over A(k) serves both evaluations of a move; A(k) is 32 B per +-kz pair "
                                          "and is only read by the k sweep: a third of the 52 Nk algorithmic figure) over the same time"} if iso_gc else None),
                    "frac_note": "the algorithmic 52 Nk per evaluation is three times what the sweep moves (it only READS A(k), 32 B per +-kz "
                                 "pair, once per candidate), so this fraction can pass 1: the kernel's efficiency is "
                                 "isolated.measured_traffic_frac (PMC bytes over the launch's own time); `measured` is the same bytes over "
                                 "the launch's begin-to-end time in the four-lane pipeline",
                    "job_frac": evals_rank * (bytes_pair_eval + bytes_k_eval) / elapsed / 1e9 / HBM_PEAK_GBS,
                    "job_frac_note": "all evaluations x (36 N + 52 Nk) algorithmic bytes / timed_region_s / HBM peak"}
            # The k sweep only READS A(k) (32 B per +-kz pair): the bytes it really moves are a third of the 52 Nk
            # algorithmic figure, so the algorithmic rate can exceed the HBM peak.  Where the committed PMC summary is of
            # THIS build, `measured` carries the rate on the bytes the counters saw, and that is the fraction to read.
            if fresh and ms_rec and pmc.get("candidates"):
                cand_per_launch = (tot_trials / world) / max(1, n_rec)
                m_gbs = k_item_bytes * cand_per_launch / (us(ms_rec, n_rec) * 1e-6) / 1e9
                roof["measured"] = {"basis": "measured_traffic", "achieved": m_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": m_gbs / HBM_PEAK_GBS, "bytes_per_candidate": k_item_bytes,
                                    "candidates_per_launch": cand_per_launch, "avg_launch_us": us(ms_rec, n_rec)}
        roof["kernels"] = kernels
        out = {
            "metric": "MC moves/sec", "value": tot_acc / elapsed, "unit": "accepted MC moves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "settle_steps": settle_steps, "timed_region_s": elapsed,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": {"spce": f"spce_{system.n_mol[0]}mol_{N0}atoms_lj_cut_coul_long_ewald_Nk{Nk}",
                                    "co2_gcmc": f"co2_gcmc_50A_box_Nk{Nk}_fV{fug_one * volume:g}" if fug_one else None,
                                    "framework_water": f"framework2208_water4site_gcmc_Nk{Nk}_fV{fug_one * volume:g}" if fug_one else None,
                                    "co2_isotherm": f"co2_isotherm_{ISOTHERM_POINTS}fugacities_50A_box_Nk{Nk}",
                                    "spce_triclinic": f"spce_{system.n_mol[0]}mol_{N0}atoms_triclinic_tilt_3.0_-2.0_1.5_Nk{Nk}",
                                    "adsorbate24": f"rigid_adsorbate_24site_{system.n_mol[0]}mol_{N0}atoms_60A_box_Nk{Nk}"}[wl],
                       "baseline_config": WORKLOADS[wl]["config"],
                       "replicas_per_gpu": R, "host_driver": args.host, "host_threads": args.host_threads if args.host == "fortran" else 1,
                       "lanes": n_lanes, "host_drivers": args.drivers, "trial_geometry": "device-built" if args.device_build else "host-built",
                       "acceptance": "device (k sweep decides and commits)" if args.device_accept else "host driver + commit launch",
                       "host_cores": pinned, "moves": moves,
                       "trials_per_step": R * world, "parallelism": f"replicas x{world}"},
            "trial_moves_per_s": tot_trials / elapsed,
            "acceptance": tot_acc / max(1.0, tot_trials),
            "dE_evals_per_s": tot_evals / elapsed,
            "ns_per_dE_eval": elapsed / max(1.0, tot_evals) * 1e9 * world,
            "ns_per_dE_eval_note": "wall time per Delta-E evaluation per GPU (pair sweep + k sweep), host loop included",
            "roofline": roof,
        }
        if sustained:
            out["sustained"] = sustained
        if counts is not None:
            out["molecules_per_chain"] = {"mean": float(np.mean(counts)), "min": int(np.min(counts)), "max": int(np.max(counts)),
                                          "initial": int(system.n_mol[t_act]), "atoms_mean": N}
        if wl == "co2_isotherm":
            # configs[4]: GCMC isotherm.  Trials are move SELECTIONS that reach the engine (no-op selections of the
            # reference -- empty type, full type -- are skipped by the host)
            hist_pts = hist_by_rank.reshape(world, ISOTHERM_POINTS, nbins).sum(axis=0)
            nn = np.arange(nbins)
            out["config"]["parallelism"] = f"replicas x{world}; fugacity points dealt round-robin over the ranks"
            out["isotherm"] = [{"fugacity_molecules_per_A3": float(fug_grid[p]), "chains": int(hist_pts[p].sum()),
                                "mean_N": float((hist_pts[p] * nn).sum() / max(1, hist_pts[p].sum())),
                                "N_min": int(nn[hist_pts[p] > 0].min()) if hist_pts[p].sum() else None,
                                "N_max": int(nn[hist_pts[p] > 0].max()) if hist_pts[p].sum() else None}
                               for p in range(ISOTHERM_POINTS)]
        # what the gathered table says about the run itself: one row per rank, in rank order
        out["exchange"] = {"collective": "all_gather", "backend": args.dist_backend if world > 1 else None,
                           "path": "c-abi" if comm is not None else "torch",
                           "through": "C ABI (mgpu_allgather_block_stats, RCCL from libmaniac_hip.so)" if comm is not None else "torch.distributed",
                           "ranks_seen": int(sums_by_rank.shape[0]),
                           "per_rank_value": [float(sums_by_rank[r, 0]) / elapsed for r in range(sums_by_rank.shape[0])],
                           "per_rank_device": [int(sums_by_rank[r, 3]) for r in range(sums_by_rank.shape[0])],
                           "bytes_per_rank": int(hist.nbytes + 32), "per": "block (= the timed region)"}
        if exchange_note:
            out["exchange"]["c_abi_fallback"] = exchange_note
        if timers0 is not None:
            out["host_seconds"] = {k: v - timers0[k] for k, v in timers1.items()}   # timed region only
    farm.close()
    if comm is not None:
        comm.close()
    if rank == 0:
        # GPU legs first, CPU baselines last (short), so that whoever samples the GPU from outside sees it busy
        if world == 1 and args.replicas_sweep and wl in ("spce", "co2_gcmc") and args.host == "fortran":
            pts = [int(x) for x in args.replicas_sweep.split(",") if x.strip()]
            gc_kw = None if wl == "spce" else dict(mol_capacity=[400], gcmc=dict(p_translation=0.0, p_rotation=0.0, fugacity=fug_one))
            try:                                           # an extra leg must never take the bench line down
                out["replicas_sweep"] = replicas_sweep(system, pts, device, args.host_threads, args.sweep_seconds, t_step, r_step, gc_kw)
            except Exception as exc:
                out["replicas_sweep"] = {"error": str(exc)}
        if world == 1 and args.replicas_sweep and wl == "spce" and args.host == "fortran":
            try:
                out["single_chain"] = single_chain_leg(system, device, t_step, r_step)
            except Exception as exc:
                out["single_chain"] = {"error": str(exc)}
        if world == 1 and args.configs:
            try:
                out["configs"] = config_legs(args, device)
            except Exception as exc:
                out["configs"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            if wl in NVT_WORKLOADS:
                out["cpu_baseline"] = cpu_baseline(system, t_step, r_step, budget_s=args.cpu_budget,
                                                    all_cores_budget_s=args.cpu_all_cores_budget)
            else:
                f_cpu = fug_one if fug_one is not None else float(fug_grid[ISOTHERM_POINTS // 2])
                out["cpu_baseline"] = cpu_baseline_gcmc(system, t_act, p_move, t_step, r_step, f_cpu, budget_s=args.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
