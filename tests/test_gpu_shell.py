"""GPU parity (through the C ABI): ddh_regularity_recombine == the reference's ShellBasis regularity
recombination (tests/golden/shell.npz), tensor ranks 0-2, forward and backward, with and without the fused
radial factor (dR/r)^(-+k).  Tolerance rel-L2 <= 1e-14."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAGS = ["8x6x5_k0", "16x10x6_k1"]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "shell.npz"))


@pytest.fixture(scope="module")
def hex_():
    from dedalus_amd.executor import HipExecutor
    return HipExecutor()


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_recombination_matches_reference(gold, hex_, tag, rank):
    from dedalus_amd.core.curvilinear import RegularityRecombination
    rows = gold[tag + "__ellrows"]
    plan = RegularityRecombination(rows, tuple(int(x) for x in gold[tag + "__shape12"]), rank, executor=hex_)
    data = gold[tag + "__r%d__in" % rank]
    fac = gold[tag + "__fac1"]
    for forward, key in ((True, "fwd"), (False, "bwd")):
        ref = gold[tag + "__r%d__%s" % (rank, key)]
        d = hex_.from_host(data)
        (plan.forward if forward else plan.backward)(d)
        assert rel(hex_.download(d), ref) < 1e-14, (key,)
        # fused radial factor: forward multiplies by (dR/r)^-k before, backward by (dR/r)^k after the mixing
        f = fac ** (-1.0 if forward else 1.0)
        d = hex_.from_host(data)
        (plan.forward if forward else plan.backward)(d, hex_.from_host(f))
        assert rel(hex_.download(d), ref * f.reshape(1, 1, 1, -1)) < 1e-14, (key, "factor")


def test_recombination_large(hex_):
    """Shell-config sized slots (rank 2, 192 radial points) against the numpy oracle executor."""
    from dedalus_amd.core.curvilinear import RegularityRecombination
    from oracle.np_executor import NumpyExecutor
    rng = np.random.default_rng(3)
    n1, n2, n3 = 12, 20, 192
    rows = np.array([(l, 0, n1, l, l + 1) for l in range(n2)] + [(3, 2, 6, 5, 6)], dtype=np.int64)
    data = rng.standard_normal((9, n1, n2, n3))
    ref = data.copy()
    RegularityRecombination(rows, (n1, n2), 2, executor=NumpyExecutor()).backward(ref)
    d = hex_.from_host(data)
    RegularityRecombination(rows, (n1, n2), 2, executor=hex_).backward(d)
    assert rel(hex_.download(d), ref) < 1e-14
