"""Field-level transforms of ShellBasis tensor fields (row a13) on the numpy oracle executor against the
reference's own `field['g']` / `field['c']` (tests/golden/shellfields.npz): ranks 0-2, k = 0 and k = 1 shells,
scales 1 and 3/2."""
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shellfields.npz"))
CASES = [((16, 12, 6), 0), ((8, 8, 5), 0), ((16, 10, 6), 1)]


def run_case(executor, shape, k, rank):
    import dedalus_amd.public as d3
    tag = "%dx%dx%d_k%d__" % (shape + (k,))
    coords = d3.SphericalCoordinates("phi", "theta", "r")
    dist = d3.Distributor(coords, dtype=np.float64, **({"executor": executor} if executor is not None else {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=tuple(GOLD[tag + "radii"]), dealias=3 / 2, dtype=np.float64, k=k)
    f = dist.TensorField((coords,) * rank, bases=shell) if rank else dist.Field(bases=shell)
    errs = {}
    phi, theta, r = dist.local_grids(shell)
    errs["r_grid"] = np.abs(np.ravel(r) - GOLD[tag + "r_grid"]).max()
    f['c'] = GOLD[tag + "r%d__cin" % rank]
    errs["g1"] = np.abs(f['g'] - GOLD[tag + "r%d__g1" % rank]).max() / np.abs(GOLD[tag + "r%d__g1" % rank]).max()
    f.change_scales(3 / 2)
    errs["g15"] = np.abs(f['g'] - GOLD[tag + "r%d__g15" % rank]).max() / np.abs(GOLD[tag + "r%d__g15" % rank]).max()
    f.change_scales(1)
    f['g'] = GOLD[tag + "r%d__gin" % rank]
    errs["cout"] = np.abs(f['c'] - GOLD[tag + "r%d__cout" % rank]).max() / np.abs(GOLD[tag + "r%d__cout" % rank]).max()
    return errs


@pytest.mark.parametrize("shape,k", CASES)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_shell_field_transforms_oracle(shape, k, rank):
    from oracle.np_executor import NumpyExecutor
    errs = run_case(NumpyExecutor(), shape, k, rank)
    assert all(v < 1e-11 for v in errs.values()), errs


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k", CASES)
@pytest.mark.parametrize("rank", [0, 1, 2])
def test_shell_field_transforms_gpu(shape, k, rank):
    errs = run_case(None, shape, k, rank)
    assert all(v < 1e-11 for v in errs.values()), errs
