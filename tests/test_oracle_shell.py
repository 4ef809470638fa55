"""CPU: the regularity intertwiner Q(ell) (dedalus_amd/tools/sphere.py) equals the reference's, and the
oracle restatement of the regularity recombination (oracle/np_swsh.py) fed with OUR Q reproduces the
reference's ShellBasis outputs (tests/golden/shell.npz, made by oracle/make_golden.py shell)."""
import os

import numpy as np
import pytest

from dedalus_amd.tools import sphere

TAGS = ["8x6x5_k0", "16x10x6_k1"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "shell.npz"))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def q_table(rows, rank):
    nell = int(rows[:, 0].max()) + 1
    return np.array([sphere.intertwiner(l, rank) for l in range(nell)])


@pytest.mark.parametrize("tag", TAGS)
def test_intertwiner_matches_reference(gold, tag):
    for rank in (1, 2):
        ells = gold[tag + "__Q%d_ells" % rank]
        for Qref, l in zip(gold[tag + "__Q%d" % rank], ells):
            assert np.max(np.abs(sphere.intertwiner(int(l), rank) - Qref)) < 1e-15
    for l in range(1, 6):          # orthogonality where every component exists
        Q = sphere.intertwiner(l + 2, 2)
        assert np.max(np.abs(Q @ Q.T - np.eye(9))) < 1e-14


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_oracle_recombination_matches_reference(gold, tag, rank):
    from oracle import np_swsh
    from dedalus_amd.core.curvilinear import RegularityRecombination
    from oracle.np_executor import NumpyExecutor
    rows = gold[tag + "__ellrows"]
    Q = q_table(rows, rank)
    data = gold[tag + "__r%d__in" % rank]
    plan = RegularityRecombination(rows, tuple(int(x) for x in gold[tag + "__shape12"]), rank, executor=NumpyExecutor())
    for forward, key in ((True, "fwd"), (False, "bwd")):
        d = data.copy()
        np_swsh.regularity_recombine(d, rows, Q, forward)                  # sequential restatement
        assert rel(d, gold[tag + "__r%d__%s" % (rank, key)]) < 1e-14
        d2 = data.copy()                                                    # composed per-slot tables (product path)
        (plan.forward if forward else plan.backward)(d2)
        assert rel(d2, gold[tag + "__r%d__%s" % (rank, key)]) < 1e-14
