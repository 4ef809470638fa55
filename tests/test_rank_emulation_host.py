"""Host side of the single-GPU rank emulation (dedalus_amd/parallel.py::LoopbackComm, tools/rank_emulation.py): the
decomposition a Distributor builds with DDH_EMULATE_RANK=r/P is rank r's share of the P-rank run (reference:
Layout.local_chunks, core/distributor.py:357-385) -- no GPU, no process group."""
import numpy as np
import pytest


def test_emulated_rank_is_parsed_and_checked(monkeypatch):
    from dedalus_amd import parallel
    monkeypatch.delenv("DDH_EMULATE_RANK", raising=False)
    assert parallel.emulated_rank() is None
    monkeypatch.setenv("DDH_EMULATE_RANK", "3/8")
    assert parallel.emulated_rank() == (3, 8)
    c = parallel.Comm(8)
    assert isinstance(c, parallel.LoopbackComm) and (c.rank, c.size, c.backend) == (3, 8, "loopback")
    with pytest.raises(ValueError):
        parallel.Comm(4)                       # the mesh must have the emulated number of ranks
    monkeypatch.setenv("DDH_EMULATE_RANK", "8/8")
    with pytest.raises(ValueError):
        parallel.emulated_rank()


def test_loopback_reductions_and_gathers_stand_for_one_rank(monkeypatch):
    from dedalus_amd import parallel
    monkeypatch.setenv("DDH_EMULATE_RANK", "1/4")
    c = parallel.Comm(4)
    assert c.allreduce_sum(2.5) == 2.5 and c.allreduce_max(-1.0) == -1.0 and c.bcast_float(7.0) == 7.0
    a = np.arange(6.0).reshape(2, 3)
    g = c.all_gather_host(a, axis=0)
    assert g.shape == (8, 3) and np.array_equal(g[2:4], a)
    c.barrier()


def test_distributor_takes_the_emulated_rank_share(monkeypatch):
    """rank 2 of 4 owns kx modes [2 * 16, 3 * 16) of a 128-mode axis: the offset every pencil kernel receives"""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    monkeypatch.setenv("DDH_EMULATE_RANK", "2/4")
    coords = d3.CartesianCoordinates('x', 'y', 'z')
    dist = d3.Distributor(coords, dtype=np.float64, mesh=(4,), executor=NumpyExecutor())
    assert (dist.rank, dist.size) == (2, 4)
    assert dist.pcomm.backend == "loopback"
