"""The path's one exchange step behind the C ABI (mgpu_comm_* / mgpu_allgather_block_stats: RCCL called from
libmaniac_hip.so) and the Fortran farm's block exchange on top of it (mfarm_exchange_block).  A one-GPU box can only hold
ONE rank -- RCCL refuses two ranks on one device -- so what is tested here is the single-rank identity, the id plumbing and
the farm-side histogram; more ranks are the driver's multi-GPU run (bench.py --exchange c-abi), with the torch.distributed
exchange (maniac_mc_amd/exchange.py, tests/test_exchange_gloo.py) as the default there.  SURVEY section 8(e);
src/write_utils.f90:144-150 (the per-chain record the histogram aggregates)."""
import numpy as np
import pytest

from maniac_mc_amd import exchange, synth

pytestmark = pytest.mark.gpu


def test_single_rank_gather_is_the_identity():
    c = exchange.CAbiComm(device=0, rank=0, world=1)
    sums = np.array([3.0, 5.5, -1.25])
    hist = np.arange(5001, dtype=np.int64)
    s, h = c.gather_block_stats(sums, hist)
    assert s.shape == (1, 3) and h.shape == (1, 5001)
    assert np.array_equal(s[0], sums) and np.array_equal(h[0], hist)
    s, h = c.gather_block_stats(sums)
    assert h is None and np.array_equal(s[0], sums)
    c.close()


def test_unique_id_and_argument_checks():
    a, b = exchange.CAbiComm.unique_id(), exchange.CAbiComm.unique_id()
    assert len(a) == 128 and a != b and any(a)
    with pytest.raises(Exception, match="unique id"):
        exchange.CAbiComm(device=0, rank=0, world=2)            # more than one rank needs rank 0's id
    with pytest.raises(Exception, match="out of range"):
        exchange.CAbiComm(device=0, rank=3, world=2, unique_id=a)


def test_farm_block_exchange_histogram():
    from maniac_mc_amd.fortran_host import FortranFarm
    s = synth.co2_box(24, seed=5)
    volume = float(np.prod(np.diag(s.box_matrix)))
    farm = FortranFarm(s, 64, seed=3, translation_step=1.0, rotation_step=0.6, mol_capacity=[120], n_threads=2, n_lanes=2,
                       gcmc=dict(p_translation=0.25, p_rotation=0.25, fugacity=60.0 / volume), device_build=True)
    farm.run(150)
    comm = exchange.CAbiComm(device=0, rank=0, world=1)
    sums, hist = farm.exchange_block(comm)
    counts = farm.counts()[:, 0]
    assert np.array_equal(hist[0, 0], np.bincount(counts, minlength=5001))
    assert hist.sum() == 64 and len(set(counts.tolist())) > 3               # the chains have spread out
    assert sums[0, 0] == farm.accepted and sums[0, 1] == farm.trials
    comm.close()
    farm.close()
