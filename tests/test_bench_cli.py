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
