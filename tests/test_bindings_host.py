"""CPU: host-side logic of the reference-side binding classes (dedalus_amd/bindings.py) -- the JacobiMMT matrices
(including dealias_before_converting=False and non-Chebyshev grids) against the reference's outputs, the reduced
shapes, and the registry hook.  The device calls are covered by tests/test_gpu_boundary.py."""
import os

import numpy as np
import pytest

from dedalus_amd import bindings


@pytest.fixture(scope="module")
def extra(golden_dir):
    return np.load(os.path.join(golden_dir, "transforms_extra.npz"))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_forward_matrix_without_dealias_before_converting(extra):
    for key in extra["nd_cases"]:
        alpha, N, M, axis = [int(v) for v in str(key).split("_")[1:]]
        fwd, bwd = bindings.jacobi_mmt_matrices(N, M, alpha - 0.5, alpha - 0.5, -0.5, -0.5, dealias_before_converting=False)
        g = np.moveaxis(extra[key + "_g"], axis, 0)
        c = np.moveaxis(np.tensordot(fwd, g, axes=(1, 0)), 0, axis)
        assert rel(c, extra[key + "_c_mmt"]) < 1e-12, key
        assert rel(c, extra[key + "_c"]) < 1e-12, key                    # the reference's fast plan agrees with its matrix plan
        gb = np.moveaxis(np.tensordot(bwd, np.moveaxis(extra[key + "_cin"], axis, 0), axes=(1, 0)), 0, axis)
        assert rel(gb, extra[key + "_gb"]) < 1e-12, key


def test_general_jacobi_matrices(extra):
    for key in extra["jac_cases"]:
        a0, b0, a, b, N, M = extra[str(key) + "_par"]
        N, M = int(N), int(M)
        fwd, bwd = bindings.jacobi_mmt_matrices(N, M, a, b, a0, b0)
        c = np.einsum("ij,ajb->aib", fwd, extra[key + "_g"])
        gb = np.einsum("ij,ajb->aib", bwd, extra[key + "_cin"])
        assert rel(c, extra[key + "_c"]) < 1e-12, key
        assert rel(gb, extra[key + "_gb"]) < 1e-12, key


def test_install_registers_plan_classes():
    class Basis:
        transforms = {}

    class A(Basis):
        transforms = {}

    class B(Basis):
        transforms = {}

    def register_transform(basis, name):            # shape of core/transforms.py:27-32
        def wrapper(cls):
            basis.transforms[name] = cls
            return cls
        return wrapper

    bindings.install(register_transform, RealFourier=A, Jacobi=B)
    assert A.transforms["hip"] is bindings.HipRealFFT and B.transforms["hip"] is bindings.HipJacobi
    assert bindings.HipSWSHColatitude._reduced((3, 16, 12, 5), 2) == (3, 16, 12, 5)
    assert bindings.HipSWSHColatitude._reduced((2, 3, 16, 12), 3) == (6, 16, 12, 1)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks on
    127.0.0.1 (the driver may call it either way); with WORLD_SIZE set it does not"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DRY_LAUNCH="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
