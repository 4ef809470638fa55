"""CPU: (1) the host-side spin-weighted spherical harmonics (dedalus_amd/tools/sphere.py) reproduce the
reference's matrices and harmonics, incl. |m| ~ 250 where the envelope under/overflows in double;
(2) the oracle restatement of SWSHColatitudeTransform (oracle/np_swsh.py) fed with OUR matrices reproduces
the reference's forward / backward outputs (tests/golden/swsh.npz, made by oracle/make_golden.py swsh)."""
import os

import numpy as np
import pytest

from dedalus_amd.tools import sphere


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "swsh.npz"))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def cases(gold):
    tags = sorted({k.split("__")[0] for k in gold.files if k.endswith("__groups")})
    return tags


def matrices_for(gold, tag, s):
    Ntheta, Lmax = [int(x) for x in gold[tag + "__dims"][:2]]
    fwd, bwd = {}, {}
    for row in gold[tag + "__groups"]:
        m = int(row[0])
        if m in fwd:
            continue
        if abs(m) > Lmax:
            fwd[m] = bwd[m] = None
        else:
            fwd[m], bwd[m] = sphere.swsh_matrices(Ntheta, Lmax, m, s)
    return fwd, bwd


def test_quadrature_and_large_m_harmonics(gold):
    z, w = sphere.quadrature(384)
    assert np.max(np.abs(np.asarray(z, dtype=np.float64) - gold["big__z"])) < 2e-16
    assert rel(np.asarray(w, dtype=np.float64), gold["big__w"]) < 1e-14
    for (m, s) in ((200, 0), (254, 1), (37, -2)):
        Y = np.asarray(sphere.harmonics(254, m, s, z), dtype=np.float64)
        ref = gold["big__Y_m%d_s%d" % (m, s)]
        assert Y.shape == ref.shape
        assert rel(Y, ref) < 1e-12, (m, s, rel(Y, ref))   # two long-double recurrences, 200+ steps


def test_matrices_match_reference(gold):
    n = 0
    for key in gold.files:
        if "__fwdmat_m" not in key:
            continue
        tag, stag, mtag = key.split("__")
        s, m = int(stag[1:]), int(mtag.split("_m")[1])
        Ntheta, Lmax = [int(x) for x in gold[tag + "__dims"][:2]]
        fwd, bwd = sphere.swsh_matrices(Ntheta, Lmax, m, s)
        assert fwd.shape == gold[key].shape
        assert rel(fwd, gold[key]) < 1e-13
        assert rel(bwd, gold[key.replace("fwdmat", "bwdmat")]) < 1e-13
        assert np.array_equal(fwd == 0, gold[key] == 0)            # same zero padding / truncation
        n += 1
    assert n >= 12


@pytest.mark.parametrize("s", [0, 1, -1, 2])
def test_oracle_matches_reference(gold, s):
    from oracle import np_swsh
    for tag in cases(gold):
        groups = gold[tag + "__groups"]
        fwd, bwd = matrices_for(gold, tag, s)
        for dims in ("1_1", "2_3"):
            key = "%s__s%d__%s" % (tag, s, dims)
            g, cref = gold[key + "__g"], gold[key + "__c"]
            c = np.zeros_like(cref)
            np_swsh.forward_reduced(g, c, groups, fwd)
            assert rel(c, cref) < 1e-13, (key, rel(c, cref))
            cin, gref = gold[key + "__cin"], gold[key + "__gout"]
            gout = np.full_like(gref, np.nan)
            np_swsh.backward_reduced(cin, gout, groups, bwd)
            assert not np.isnan(gout).any() or np.array_equal(np.isnan(gout), np.isnan(gref))
            mask = ~np.isnan(gref)
            assert rel(gout[mask], gref[mask]) < 1e-13, (key,)


@pytest.mark.parametrize("s", [1, 2])
def test_opposite_spin_weights_share_matrices(s):
    """The identity the paired device transform rests on (ddh_grouped_mmt_set_pairs):
    F_{-s}[l, j] = (-1)^(l + m) F_{+s}[l, N-1-j] and the same for the backward matrices."""
    from dedalus_amd.tools import sphere as sph
    N, Lmax = 36, 22
    for m in (0, 1, 2, 5, 11, 22):
        fp, bp = sph.swsh_matrices(N, Lmax, m, s)
        fm, bm = sph.swsh_matrices(N, Lmax, m, -s)
        assert fp.shape == fm.shape and bp.shape == bm.shape
        ells = Lmax + 1 - fp.shape[0] + np.arange(fp.shape[0])          # rows are ell = max(m, |s|) .. Lmax
        sg = (-1.0) ** (ells + m)
        assert np.abs(fm - sg[:, None] * fp[:, ::-1]).max() < 1e-14
        assert np.abs(bm - sg[None, :] * bp[::-1, :]).max() < 1e-14
