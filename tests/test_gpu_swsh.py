"""GPU parity (through the C ABI): grouped dense transform = SWSHColatitudeTransform of the reference
(tests/golden/swsh.npz), forward and backward, spins 0, +-1, 2, GEMV path (few columns) and tiled GEMM path
(many columns), folded (reversed) ell slices, zero-filled groups.  Tolerance rel-L2 <= 1e-13."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "swsh.npz"))


@pytest.fixture(scope="module")
def hex_():
    from dedalus_amd.executor import HipExecutor
    return HipExecutor()


@pytest.mark.parametrize("tag", ["16x8", "8x8", "4x8", "32x16", "24x20"])
@pytest.mark.parametrize("s", [0, 1, -1, 2])
def test_swsh_matches_reference(gold, hex_, tag, s):
    from dedalus_amd.core.curvilinear import SWSHColatitudeTransform
    Ntheta, Lmax = [int(x) for x in gold[tag + "__dims"][:2]]
    plan = SWSHColatitudeTransform(Ntheta, Lmax, gold[tag + "__groups"], s, executor=hex_)
    for dims in ("1_1", "2_3"):
        key = "%s__s%d__%s" % (tag, s, dims)
        g, cref = gold[key + "__g"], gold[key + "__c"]
        dg = hex_.from_host(g)
        dc = hex_.zeros(cref.shape)
        plan.forward_reduced(dg, dc)
        assert rel(hex_.download(dc), cref) < 1e-13, (key, "forward")
        cin, gref = gold[key + "__cin"], gold[key + "__gout"]
        dci = hex_.from_host(cin)
        dgo = hex_.empty(gref.shape)
        dgo.fill_(float("nan"))
        plan.backward_reduced(dci, dgo)
        got = hex_.download(dgo)
        assert np.array_equal(np.isnan(got), np.isnan(gref)), (key, "coverage")
        mask = ~np.isnan(gref)
        assert rel(got[mask], gref[mask]) < 1e-13, (key, "backward")


def test_swsh_wide_columns_against_oracle(hex_):
    """Shell-like shapes (a radial axis behind theta: hundreds of columns) through the tiled GEMM path."""
    from dedalus_amd.core.curvilinear import SWSHColatitudeTransform
    from oracle.np_executor import NumpyExecutor
    Ntheta, Lmax = 24, 14
    rows = [(0, 0, 0, 2, 0, 1, 15), (14, 6, 2, 2, 0, -1, 1), (1, 4, 2, 2, 1, 1, 14), (15, 2, 0, 2, 0, 1, 0)]
    groups = np.array(rows, dtype=np.int64)
    rng = np.random.default_rng(5)
    g = rng.standard_normal((3, 8, Ntheta, 40))
    c = np.zeros((3, 4, 15, 40))
    for s in (0, -1):
        ref_plan = SWSHColatitudeTransform(Ntheta, Lmax, groups, s, executor=NumpyExecutor())
        plan = SWSHColatitudeTransform(Ntheta, Lmax, groups, s, executor=hex_)
        cref = c.copy()
        ref_plan.forward_reduced(g, cref)
        dc = hex_.zeros(c.shape)
        plan.forward_reduced(hex_.from_host(g), dc)
        assert rel(hex_.download(dc), cref) < 1e-13
        gref = np.full(g.shape, np.nan)
        ref_plan.backward_reduced(cref, gref)
        dg = hex_.empty(g.shape)
        dg.fill_(float("nan"))
        plan.backward_reduced(hex_.from_host(cref), dg)
        got = hex_.download(dg)
        assert np.array_equal(np.isnan(got), np.isnan(gref))
        mask = ~np.isnan(gref)
        assert rel(got[mask], gref[mask]) < 1e-13


@pytest.mark.parametrize("n0,n3", [(1, 16), (3, 40), (2, 192)])
@pytest.mark.parametrize("s", [0, 1, -2])
def test_many_columns_mfma_gemm_matches_matrix_products(n0, n3, s):
    """Shell-like use: the radial axis rides behind theta (n3 columns per slice) -> the FP64 MFMA grouped GEMM
    (grouped_gemm_mfma_kernel).  Reference: the per-m matrix products of SWSHColatitudeTransform.forward_reduced /
    backward_reduced (core/transforms.py:1258-1288) in NumPy with the same matrices, folded groups and |m| > Lmax
    groups included."""
    from dedalus_amd.core.curvilinear import SWSHColatitudeTransform
    from dedalus_amd.executor import HipExecutor
    from dedalus_amd.tools import sphere
    ex = HipExecutor()
    Lmax, Nth = 37, 60
    # groups (m, g_start, c_start, count, ell_start, ell_step, n_ell): plain ones, a folded one (step -1), one beyond Lmax
    rows = []
    for m in range(0, Lmax + 1, 3):
        rows.append((m, 2 * m, 2 * m, 2, m, 1, Lmax + 1 - m))
    rows.append((5, 2 * (Lmax + 1), 2 * (Lmax + 1), 2, Lmax, -1, Lmax + 1 - 5))
    rows.append((Lmax + 3, 2 * (Lmax + 2), 2 * (Lmax + 2), 2, 0, 1, 0))
    rows = np.array(rows, dtype=np.int64)
    n1 = 2 * (Lmax + 3)
    plan = SWSHColatitudeTransform(Nth, Lmax, rows, s, executor=ex)
    rng = np.random.default_rng(5 + n3)
    g = rng.standard_normal((n0, n1, Nth, n3))
    c_in = rng.standard_normal((n0, n1, Lmax + 1, n3))
    c_dev = ex.from_host(np.full((n0, n1, Lmax + 1, n3), 7.0))
    plan.forward_reduced(ex.from_host(g), c_dev)
    g_dev = ex.from_host(np.full(g.shape, 7.0))
    plan.backward_reduced(ex.from_host(c_in), g_dev)
    c_out, g_out = ex.download(c_dev), ex.download(g_dev)
    for (m, gs, cs, cnt, l0, step, ne) in rows:
        if abs(m) > Lmax:
            assert np.all(c_out[:, cs:cs + cnt] == 7.0)              # forward leaves the slot alone
            assert np.all(g_out[:, gs:gs + cnt] == 0.0)              # backward zero-fills (transforms.py:1280-1288)
            continue
        F, B = sphere.swsh_matrices(Nth, Lmax, int(m), s)
        ell = l0 + step * np.arange(ne)
        ref_c = np.einsum("lt,ajtx->ajlx", F, g[:, gs:gs + cnt])
        assert np.linalg.norm(c_out[:, cs:cs + cnt][:, :, ell] - ref_c) <= 1e-13 * np.linalg.norm(ref_c), (m, "fwd")
        ref_g = np.einsum("tl,ajlx->ajtx", B, c_in[:, cs:cs + cnt][:, :, ell])
        assert np.linalg.norm(g_out[:, gs:gs + cnt] - ref_g) <= 1e-13 * np.linalg.norm(ref_g), (m, "bwd")
