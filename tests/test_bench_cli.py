"""bench.py's multi-rank path: `--gpus N` starts its own ranks (one process per GPU) and the isotherm workload
(BASELINE.json configs[4]) gathers the uptake histogram over the process group."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_gpus_2_dry_run_spawns_two_ranks_and_gathers_a_rank_ordered_table():
    # no launcher around it: bench.py itself must start the two ranks (gloo, CPU only, no engine)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--dry-run",
                          "--workload", "co2_isotherm", "--replicas", "64"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["dry_run"] and d["n_gpus"] == 2
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert [r["device"] for r in d["ranks"]] == [0, 1]                 # one device per rank
    assert [r["first_point"] for r in d["ranks"]] == [0, 1]            # fugacity points dealt round-robin
    assert sum(r["fugacity_points"] for r in d["ranks"]) == 8
    assert all(r["host_threads"] >= 1 for r in d["ranks"])
    # the keys a scaling run's line carries: which gather ran, how many ranks it saw, every rank's device
    ex = d["exchange"]
    assert ex["ranks_seen"] == 2 and ex["per_rank_device"] == [0, 1] and len(ex["per_rank_value"]) == 2
    assert ex["would_default_to"] == "torch"                           # gloo rehearsal; with nccl + the Fortran host: c-abi


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode != 0 and "launcher started 1 rank" in (out.stderr + out.stdout)


def test_isotherm_points_cover_the_grid_for_every_world_size():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        pts = sorted(p for r in range(world) for p in bench.isotherm_points_of_rank(r, world))
        assert pts == list(range(8))
    assert bench.isotherm_points_of_rank(9, 16) == [1]


@pytest.mark.gpu
def test_isotherm_two_gloo_ranks_on_one_gpu_histogram_matches_rank_counts(tmp_path):
    """Two ranks (gloo) share the one GPU of the test box: each runs a small CO2 GCMC farm at its four fugacity
    points; the gathered per-point histogram must equal the histogram of the ranks' own final molecule counts."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--workload", "co2_isotherm",
                          "--replicas", "64", "--steps", "120", "--warmup", "10", "--settle-s", "0", "--device", "0",
                          "--no-pin", "--host-threads", "2", "--sustained-steps", "0", "--dump-counts", str(tmp_path)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("co2_isotherm")
    assert d["value"] > 0 and len(d["isotherm"]) == 8
    # the gathered table describes the run: both ranks seen, in rank order, each with its own rate and device
    ex = d["exchange"]
    assert ex["path"] == "torch" and ex["ranks_seen"] == 2 and ex["per_rank_device"] == [0, 0]
    assert len(ex["per_rank_value"]) == 2 and all(v > 0 for v in ex["per_rank_value"])
    assert abs(sum(ex["per_rank_value"]) - d["value"]) <= 1e-6 * d["value"]
    counts = {p: [] for p in range(8)}
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        for n, p in zip(z["counts"], z["point_of_chain"]):
            counts[int(p)].append(int(n))
    for p in range(8):
        row = d["isotherm"][p]
        assert row["chains"] == len(counts[p]) == 16
        assert abs(row["mean_N"] - np.mean(counts[p])) < 1e-12
        assert row["N_min"] == min(counts[p]) and row["N_max"] == max(counts[p])
    # uptake grows with fugacity (8x in fugacity from the first to the last point)
    assert d["isotherm"][-1]["mean_N"] > d["isotherm"][0]["mean_N"]


@pytest.mark.gpu
def test_default_line_carries_the_other_configs_the_chain_count_curve_and_both_baselines():
    """The driver's command shape (`bench.py --gpus 1 --steps K --warmup W`, nothing else): ONE JSON line with the headline
    metric, `roofline`, `cpu_baseline`, and -- round 4 -- `configs` (BASELINE.json configs[2]-[4] as short legs with their own
    roofline and CPU baseline), `replicas_sweep` and `single_chain`.  Short legs here; the structure is what is checked."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "5", "--warmup", "2", "--settle-s", "0", "--sustained-steps", "0",
                          "--config-steps", "40", "--sweep-seconds", "0.05", "--replicas-sweep", "1,64", "--cpu-budget", "0.5",
                          "--cpu-all-cores-budget", "0", "--configs", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "MC moves/sec" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["value"] > 1e5 and d["config"]["workload"].startswith("spce_3375mol_10125atoms")
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert set(d["configs"]) == {"co2_gcmc", "framework_water", "co2_isotherm", "spce_triclinic", "adsorbate24"}
    for name, leg in d["configs"].items():
        assert "error" not in leg, (name, leg)
        assert leg["value"] > 1e5 and leg["steps"] == 40 and leg["roofline"]["frac"] is not None and leg["roofline"]["basis"]
        assert leg["cpu_baseline"]["value"] > 0
    assert [r["replicas"] for r in d["replicas_sweep"]] == [1, 64] and all(r["value"] > 0 for r in d["replicas_sweep"])
    assert all(r["path"].startswith("window") for r in d["replicas_sweep"])                 # round 5: one launch per lane step
    assert d["exchange"]["ranks_seen"] == 1 and d["exchange"]["per_rank_device"] == [0]
    assert d["single_chain"]["value"] > 0 and d["single_chain"]["windows"] > 0
