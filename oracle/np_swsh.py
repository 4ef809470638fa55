"""
TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the spin-weighted spherical harmonic colatitude
transform, SWSHColatitudeTransform.forward_reduced / backward_reduced (core/transforms.py:1258-1288):
a Python loop over the local azimuthal wavenumbers m, each applying one dense matrix along the
colatitude axis of a slice of the data (apply_matrix, tools/array.py).
Pinned against the reference's own outputs by tests/test_oracle_swsh.py (tests/golden/swsh.npz).
"""
import numpy as np


def forward_reduced(gdata, cdata, groups, fwd_mats):
    """gdata [N0, N1g, Ntheta, N3] -> cdata [N0, N1c, N2c, N3].
    groups: rows (m, g_start, c_start, count, ell_start, ell_step, n_ell); fwd_mats[m]: (n_ell, Ntheta)
    or None when |m| > Lmax (skipped, :1268-1272)."""
    for (m, g0, c0, cnt, e0, es, ne) in groups:
        M = fwd_mats.get(int(m))
        if M is None:
            continue
        grm = gdata[:, g0:g0 + cnt, :, :]
        res = np.einsum("lt,ajtx->ajlx", M, grm)
        idx = e0 + es * np.arange(ne)
        cdata[:, c0:c0 + cnt, idx, :] = res


def backward_reduced(cdata, gdata, groups, bwd_mats):
    """cdata -> gdata; groups whose |m| > Lmax get zeros (:1281-1283)."""
    for (m, g0, c0, cnt, e0, es, ne) in groups:
        M = bwd_mats.get(int(m))
        if M is None:
            gdata[:, g0:g0 + cnt, :, :] = 0
            continue
        idx = e0 + es * np.arange(ne)
        crm = cdata[:, c0:c0 + cnt, idx, :]
        gdata[:, g0:g0 + cnt, :, :] = np.einsum("tl,ajlx->ajtx", M, crm)
