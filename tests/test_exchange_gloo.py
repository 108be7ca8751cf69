"""The replica farm's per-block exchange (maniac_mc_amd/exchange.py) with world_size 2 on CPU (gloo):
the N > 1 path of bench.py minus the kernels."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from maniac_mc_amd import exchange
dist.init_process_group("gloo")
rank, world = exchange.world()
assert world == 2
sums = np.array([100.0 * (rank + 1), 7.0 + rank, 0.5 * rank])
hist = exchange.molecule_count_histogram([3 + rank, 3 + rank, 9, 5000 + 7 * rank], 5001)
exchange.barrier()
s, h = exchange.gather_block_stats(sums, hist)
t = exchange.max_over_ranks(1.0 + rank)
if rank == 0:
    print(json.dumps({"sums": s.tolist(), "hist_nonzero": [[int(i), int(v)] for r in range(2) for i, v in enumerate(h[r]) if v],
                      "hist_shape": list(h.shape), "tmax": t}))
dist.destroy_process_group()
'''


def test_block_exchange_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["sums"] == [[100.0, 7.0, 0.0], [200.0, 8.0, 0.5]]
    assert d["hist_shape"] == [2, 5001]
    assert d["hist_nonzero"] == [[3, 2], [9, 1], [5000, 1], [4, 2], [9, 1], [5000, 1]]
    assert d["tmax"] == 2.0


def test_exchange_single_process_is_identity():
    from maniac_mc_amd import exchange
    s, h = exchange.gather_block_stats([1.0, 2.0], exchange.molecule_count_histogram([1, 1, 2], 4))
    assert s.shape == (1, 2) and np.array_equal(h, [[0, 2, 1, 0]])
    assert exchange.world() == (0, 1) and exchange.max_over_ranks(3.5) == 3.5
