"""GPU: the drop-in boundary as the reference would use it.  The plan classes of
dedalus_amd/bindings.py are registered in a stand-in registry with the shape of the reference's
(register_transform(basis_cls, name) -> basis_cls.transforms[name], core/transforms.py:27-32), constructed with the
reference's signatures (:102, 194, 371, 1254) and called the way IntervalBasis.forward_transform / backward_transform do
(core/basis.py:416-428): forward(gdata, cdata, axis) on two C-contiguous views OF THE SAME BUFFER
(core/basis.py:185-193).  Expected values: the reference's own outputs (tests/golden/transforms*.npz, swsh.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


class _Registry:
    """stand-in for the reference's basis classes: each carries a `transforms` dict filled by register_transform"""

    class RealFourier:
        transforms = {}

    class ComplexFourier:
        transforms = {}

    class Jacobi:
        transforms = {}

    class SphereBasis:
        transforms = {}

    @staticmethod
    def register_transform(basis, name):
        def wrapper(cls):
            if not hasattr(basis, "transforms"):
                basis.transforms = {}
            basis.transforms[name] = cls
            return cls
        return wrapper


@pytest.fixture(scope="module")
def reg():
    from dedalus_amd import bindings
    from dedalus_amd.device import Device
    Device.get()
    bindings.install(_Registry.register_transform, _Registry.RealFourier, _Registry.ComplexFourier, _Registry.Jacobi,
                     _Registry.SphereBasis)
    return _Registry


def field_views(gshape, cshape, dtype=np.float64):
    """grid and coefficient views of ONE device buffer, both starting at its first byte (Field.preset_layout,
    core/field.py:517-526)"""
    import torch
    td = torch.float64 if dtype == np.float64 else torch.complex128
    n = max(int(np.prod(gshape)), int(np.prod(cshape)))
    buf = torch.zeros(n, dtype=td, device="cuda")
    g = buf[:int(np.prod(gshape))].view(*gshape)
    c = buf[:int(np.prod(cshape))].view(*cshape)
    assert g.data_ptr() == c.data_ptr()
    return buf, g, c


def run_pair(plan, g_in, c_in, axis, gshape, cshape, dtype=np.float64):
    import torch
    buf, g, c = field_views(gshape, cshape, dtype)
    g.copy_(torch.from_numpy(np.ascontiguousarray(g_in)))
    plan.forward(g, c, axis)
    torch.cuda.synchronize()
    c_out = c.cpu().numpy().copy()
    c.copy_(torch.from_numpy(np.ascontiguousarray(c_in)))
    plan.backward(c, g, axis)
    torch.cuda.synchronize()
    return c_out, g.cpu().numpy().copy()


def test_real_fourier_plans_in_place(reg, golden_dir):
    gold = np.load(os.path.join(golden_dir, "transforms.npz"))
    cls = reg.RealFourier.transforms["hip"]
    for key in gold["rf_cases"]:
        N, M, axis = [int(v) for v in str(key).split("_")[1:]]
        plan = cls(N, M)
        c, gb = run_pair(plan, gold[key + "_g"], gold[key + "_cin"], axis, gold[key + "_g"].shape, gold[key + "_c"].shape)
        assert rel(c, gold[key + "_c"]) < 1e-12, key
        assert rel(gb, gold[key + "_gb"]) < 1e-12, key


def test_complex_fourier_plans_in_place(reg, golden_dir):
    gold = np.load(os.path.join(golden_dir, "transforms.npz"))
    cls = reg.ComplexFourier.transforms["hip"]
    for key in gold["cf_cases"]:
        N, M, axis = [int(v) for v in str(key).split("_")[1:]]
        plan = cls(N, M)
        c, gb = run_pair(plan, gold[key + "_g"], gold[key + "_cin"], axis, gold[key + "_g"].shape, gold[key + "_c"].shape,
                         dtype=np.complex128)
        assert rel(c, gold[key + "_c"]) < 1e-12, key
        assert rel(gb, gold[key + "_gb"]) < 1e-12, key


def test_jacobi_plans_in_place(reg, golden_dir):
    gold = np.load(os.path.join(golden_dir, "transforms.npz"))
    cls = reg.Jacobi.transforms["hip"]
    for key in gold["ch_cases"]:
        alpha, N, M, axis = [int(v) for v in str(key).split("_")[1:]]
        plan = cls(N, M, alpha - 0.5, alpha - 0.5, -0.5, -0.5)
        c, gb = run_pair(plan, gold[key + "_g"], gold[key + "_cin"], axis, gold[key + "_g"].shape, gold[key + "_c"].shape)
        assert rel(c, gold[key + "_c"]) < 1e-12, key
        assert rel(gb, gold[key + "_gb"]) < 1e-12, key


def test_jacobi_without_dealias_before_converting_and_general_grids(reg, golden_dir):
    gold = np.load(os.path.join(golden_dir, "transforms_extra.npz"))
    cls = reg.Jacobi.transforms["hip"]
    for key in gold["nd_cases"]:
        alpha, N, M, axis = [int(v) for v in str(key).split("_")[1:]]
        plan = cls(N, M, alpha - 0.5, alpha - 0.5, -0.5, -0.5, dealias_before_converting=False)
        c, gb = run_pair(plan, gold[key + "_g"], gold[key + "_cin"], axis, gold[key + "_g"].shape, gold[key + "_c"].shape)
        assert rel(c, gold[key + "_c"]) < 1e-12, key
        assert rel(gb, gold[key + "_gb"]) < 1e-12, key
    for key in gold["jac_cases"]:
        a0, b0, a, b, N, M = gold[str(key) + "_par"]
        plan = cls(int(N), int(M), a, b, a0, b0)
        c, gb = run_pair(plan, gold[key + "_g"], gold[key + "_cin"], 1, gold[key + "_g"].shape, gold[key + "_c"].shape)
        assert rel(c, gold[key + "_c"]) < 1e-12, key
        assert rel(gb, gold[key + "_gb"]) < 1e-12, key


def test_swsh_colatitude_plan_in_place(reg, golden_dir):
    gold = np.load(os.path.join(golden_dir, "swsh.npz"))
    cls = reg.SphereBasis.transforms["hip"]
    for tag in ("16x8", "32x16", "24x20"):
        Ntheta, Lmax, n1g, n1c, n2c = [int(v) for v in gold[tag + "__dims"]]
        for s in (0, 1, -1, 2):
            plan = cls(Ntheta, Lmax, gold[tag + "__groups"], s)
            for (N0, N3) in ((1, 1), (2, 3)):
                key = "%s__s%d__%d_%d" % (tag, s, N0, N3)
                g_in, c_ref, c_in, g_ref = gold[key + "__g"], gold[key + "__c"], gold[key + "__cin"], gold[key + "__gout"]
                c, gb = run_pair(plan, g_in, c_in, 2, g_in.shape, c_ref.shape)
                # forward leaves slots of |m| > Lmax untouched (they hold whatever the shared buffer held): compare
                # where the reference wrote
                mask = c_ref != 0
                assert rel(c[mask], c_ref[mask]) < 1e-11, key
                fin = np.isfinite(g_ref)             # (the golden was pre-filled with NaN to expose slots nobody writes)
                assert fin.any() and rel(gb[fin], g_ref[fin]) < 1e-11, key


def test_host_arrays_are_staged(reg, golden_dir):
    """NumPy arrays (an unmodified host-resident reference) go through ddh_memcpy_h2d / d2h"""
    gold = np.load(os.path.join(golden_dir, "transforms.npz"))
    key = str(gold["rf_cases"][1])
    N, M, axis = [int(v) for v in key.split("_")[1:]]
    plan = reg.RealFourier.transforms["hip"](N, M)
    buf = np.zeros(max(gold[key + "_g"].size, gold[key + "_c"].size))
    g = buf[:gold[key + "_g"].size].reshape(gold[key + "_g"].shape)
    c = buf[:gold[key + "_c"].size].reshape(gold[key + "_c"].shape)
    g[...] = gold[key + "_g"]
    plan.forward(g, c, axis)
    assert rel(c, gold[key + "_c"]) < 1e-12


def test_transpose_planner_signature_one_rank():
    """HipTranspose(global_shape, chunk_shape, dtype, axis, comm) on a 1-rank RCCL communicator"""
    import torch
    from dedalus_amd import bindings
    comm = bindings.HipCommunicator(0, 1, lambda b: b)
    plan = bindings.HipTranspose((3, 8, 6, 4), (1, 2, 2, 1), np.float64, 1, comm)
    a = torch.randn(3, 8, 6, 4, dtype=torch.float64, device="cuda")
    b, c = torch.empty_like(a), torch.empty_like(a)
    plan.localize_rows(a, b)
    plan.localize_columns(b, c)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    # a buffer of another chunk convention is refused on the host (not read out of bounds on the device)
    with pytest.raises(ValueError, match="elements"):
        plan.localize_rows(a[:, :4].contiguous(), b)
    with pytest.raises(ValueError, match="elements"):
        plan.localize_columns(b, c[:, :, :3].contiguous())


def test_matsolver_adapter_vs_superlu():
    """B2 (libraries/matsolvers.py:10-27): bindings.HipBandMatsolver(matrix, solver).solve(vector) registered through a
    registry of the reference's shape, against scipy's SuperLU -- a banded matrix (device band LU), a tau-bordered
    Chebyshev matrix with boundary rows and tau columns (the engine's dense path), and a complex one."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from dedalus_amd import bindings
    from dedalus_amd.tools import jacobi
    registry = {}

    def add_solver(cls):                                   # libraries/matsolvers.py:10-13
        registry[cls.__name__.lower()] = cls
        return cls

    bindings.install_matsolver(add_solver)
    cls = registry["hipbandmatsolver"]
    rng = np.random.default_rng(3)
    # (1) banded, kl = 3, ku = 5, not symmetric, needs pivoting here and there
    N = 200
    A = sp.diags([rng.standard_normal(N - abs(o)) for o in range(-3, 6)], list(range(-3, 6)), format="csr")
    A = A + sp.diags(0.3 * rng.standard_normal(N), 0)
    s = cls(A, None)
    assert s.band
    b = rng.standard_normal(N)
    assert rel(s.solve(b), spla.splu(A.tocsc()).solve(b)) < 1e-10
    B = rng.standard_normal((N, 3))
    assert rel(s.solve(B), spla.splu(A.tocsc()).solve(B)) < 1e-10
    # (2) tau-bordered second-order Chebyshev problem u'' - 4 u = f, u(-1) = u(1) = 0 (two tau columns, two dense rows)
    Nz = 64
    D1 = jacobi.differentiation_matrix(Nz, -0.5, -0.5)
    D2 = jacobi.differentiation_matrix(Nz, 0.5, 0.5) @ D1
    C2 = jacobi.conversion_matrix(Nz, 0.5, 0.5, 1.5, 1.5) @ jacobi.conversion_matrix(Nz, -0.5, -0.5, 0.5, 0.5)
    Lz = (D2 - 4.0 * C2).tolil()
    taus = sp.lil_matrix((Nz, 2))
    taus[Nz - 1, 0] = 1.0
    taus[Nz - 2, 1] = 1.0
    left = jacobi.polynomials(Nz, -0.5, -0.5, np.array([-1.0]))[:, 0]
    right = jacobi.polynomials(Nz, -0.5, -0.5, np.array([1.0]))[:, 0]
    bc = sp.lil_matrix((2, Nz + 2))
    bc[0, :Nz] = left
    bc[1, :Nz] = right
    T = sp.vstack([bc, sp.hstack([Lz, taus])]).tocsr()
    s2 = cls(T, None)
    assert not s2.band
    f = rng.standard_normal(Nz + 2)
    assert rel(s2.solve(f), spla.splu(T.tocsc()).solve(f)) < 1e-9
    # (3) complex banded
    Ac = (A + 1j * sp.diags(rng.standard_normal(N - 1), 1)).tocsr()
    s3 = cls(Ac, None)
    bc_ = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    assert rel(s3.solve(bc_), spla.splu(Ac.tocsc()).solve(bc_)) < 1e-10
