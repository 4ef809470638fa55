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


def regularity_recombine(data, ell_maps, Q, forward, radial_factor=None):
    """In place on data[ncomp, n1, n2, n3]: forward/backward_regularity_recombination (core/basis.py:3595-3626):
    for every entry (ell, m slice, ell slice) of ell_maps IN ORDER, apply_matrix(Q[ell].T) (forward) or
    apply_matrix(Q[ell]) (backward) on the flattened tensor axis of that block -- the entries are bounding boxes
    and may overlap, a slot covered twice is transformed twice -- and the radial factor of
    ShellBasis.forward/backward_transform_radius (:4474-4508).  ell_maps rows: (ell, m0, m1, l0, l1)."""
    nc = data.shape[0]
    if radial_factor is not None:
        data *= np.asarray(radial_factor).reshape(1, 1, 1, -1)
    if nc == 1:
        return
    for (ell, m0, m1, l0, l1) in ell_maps:
        M = Q[int(ell)].T if forward else Q[int(ell)]
        blk = data[:, m0:m1, l0:l1, :]
        data[:, m0:m1, l0:l1, :] = np.einsum("rc,cmlx->rmlx", M, blk)
