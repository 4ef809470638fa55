"""
TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy / scipy.fft) of the reference's spectral
transform plans.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Pinned (tests/test_oracle_transforms.py) against fixtures produced by running the reference itself
through oracle/refshim (tests/golden/transforms.npz, transforms_extra.npz) and against the reference's own
matrix-multiply definitions restated in *_mmt_matrices below.

Each function names the reference code it follows (paths under /root/reference/dedalus).
"""

import numpy as np
import scipy.fft as sf
from scipy import sparse
from scipy.sparse.linalg import spsolve_triangular


def _axslice(axis, start, stop, ndim):
    sl = [slice(None)] * ndim
    sl[axis] = slice(start, stop)
    return tuple(sl)


# ----------------------------------------------------------------------------------------------
# RealFourier: FFTWRealFFT core/transforms.py:537-565, unpack_rescale :469-487,
# repack_rescale :489-509, wavenumber cut-off :371-378
# ----------------------------------------------------------------------------------------------

def rfft_kmax(N, M):
    return min((N - 1) // 2, (M - 1) // 2)


def rfft_forward(g, axis, M):
    N = g.shape[axis]
    K = rfft_kmax(N, M)
    t = sf.rfft(g, axis=axis)                       # unnormalised r2c, exp(-i...)
    shape = list(g.shape)
    shape[axis] = M
    c = np.zeros(shape, dtype=g.dtype)
    nd = g.ndim
    t = np.moveaxis(t, axis, 0)
    cm = np.moveaxis(c, axis, 0)
    cm[0] = t[0].real / N                            # a0 ; b0 (msin of k=0) stays zero
    cm[2:2 * (K + 1):2] = t[1:K + 1].real * (2.0 / N)
    cm[3:2 * (K + 1):2] = t[1:K + 1].imag * (2.0 / N)
    return c


def rfft_backward(c, axis, N):
    M = c.shape[axis]
    K = rfft_kmax(N, M)
    cm = np.moveaxis(c, axis, 0)
    tshape = (N // 2 + 1,) + cm.shape[1:]
    t = np.zeros(tshape, dtype=np.complex128)
    t[0] = cm[0]
    t[1:K + 1] = 0.5 * (cm[2:2 * (K + 1):2] + 1j * cm[3:2 * (K + 1):2])
    g = sf.irfft(t, n=N, axis=0) * N                # FFTW c2r is unnormalised
    return np.ascontiguousarray(np.moveaxis(g, 0, axis))


def real_fourier_mmt_matrices(N, M):
    """RealFourierMMT core/transforms.py:387-424 restated: (forward M x N, backward N x M)."""
    K = rfft_kmax(N, M)
    x = 2 * np.pi * np.arange(N) / N
    F = np.zeros((M, N))
    Bm = np.zeros((N, M))
    F[0] = 1.0 / N
    Bm[:, 0] = 1.0
    for k in range(1, K + 1):
        F[2 * k] = (2.0 / N) * np.cos(k * x)
        F[2 * k + 1] = -(2.0 / N) * np.sin(k * x)
        Bm[:, 2 * k] = np.cos(k * x)
        Bm[:, 2 * k + 1] = -np.sin(k * x)
    return F, Bm


# ----------------------------------------------------------------------------------------------
# ComplexFourier: FFTWComplexFFT core/transforms.py:292-330, resize_coeffs :243-267,
# wavenumbers :201-208
# ----------------------------------------------------------------------------------------------

def cfft_forward(g, axis, M):
    N = g.shape[axis]
    K = rfft_kmax(N, M)
    t = np.moveaxis(sf.fft(g, axis=axis), axis, 0) / N
    c = np.zeros((M,) + t.shape[1:], dtype=np.complex128)
    c[:K + 1] = t[:K + 1]
    if K > 0:
        c[M - K:] = t[N - K:]
    return np.ascontiguousarray(np.moveaxis(c, 0, axis))


def cfft_backward(c, axis, N):
    M = c.shape[axis]
    K = rfft_kmax(N, M)
    cm = np.moveaxis(c, axis, 0)
    t = np.zeros((N,) + cm.shape[1:], dtype=np.complex128)
    t[:K + 1] = cm[:K + 1]
    if K > 0:
        t[N - K:] = cm[M - K:]
    g = sf.ifft(t, axis=0) * N
    return np.ascontiguousarray(np.moveaxis(g, 0, axis))


# ----------------------------------------------------------------------------------------------
# Chebyshev-T grid with optional ultraspherical output:
# FastChebyshevTransform core/transforms.py:801-902, FastCosineTransform :715-746, FFTWDCT :771-798
# ----------------------------------------------------------------------------------------------

def cheb_scales(N, L):
    k = np.arange(L)
    sgn = np.where(k % 2 == 0, 1.0, -1.0)
    fs = sgn * np.sqrt(np.pi / 2) / N
    fs[0] = np.sqrt(np.pi) / (2 * N)
    bs = sgn / (2 * np.sqrt(np.pi / 2))
    bs[0] = 1 / np.sqrt(np.pi)
    return fs, bs


def cheb_forward(g, axis, M, conv=None):
    """conv: optional (M x M) upper-banded scipy sparse conversion matrix applied after truncation
    (dealias_before_converting=True, dedalus.cfg:41)."""
    N = g.shape[axis]
    d = np.moveaxis(sf.dct(g, type=2, axis=axis), axis, 0)      # REDFT10
    Mk = min(N, M)
    fs, _ = cheb_scales(N, Mk)
    c = np.zeros((M,) + d.shape[1:], dtype=g.dtype)
    c[:Mk] = d[:Mk] * fs.reshape((-1,) + (1,) * (d.ndim - 1))
    if conv is not None:
        c = (sparse.csr_matrix(conv) @ c.reshape(M, -1)).reshape(c.shape)
    return np.ascontiguousarray(np.moveaxis(c, 0, axis))


def cheb_backward(c, axis, N, conv=None):
    M = c.shape[axis]
    cm = np.moveaxis(c, axis, 0)
    if conv is not None:
        if M > N:                                   # transforms.py:878-881: truncate input first
            cm = cm.copy()
            cm[N:] = 0.0
        cm = spsolve_triangular(sparse.csr_matrix(conv), cm.reshape(M, -1).copy(), lower=False).reshape(cm.shape)
    Mk = min(N, M)
    _, bs = cheb_scales(N, Mk)
    e = np.zeros((N,) + cm.shape[1:], dtype=c.dtype)
    e[:Mk] = cm[:Mk] * bs.reshape((-1,) + (1,) * (cm.ndim - 1))
    g = sf.dct(e, type=3, axis=0)                                # REDFT01
    return np.ascontiguousarray(np.moveaxis(g, 0, axis))


def chebyshev_mmt_matrices(N, M, conv=None):
    """JacobiMMT core/transforms.py:114-158 for a0=b0=-1/2, restated in closed form:
    orthonormal p_0 = 1/sqrt(pi), p_n = sqrt(2/pi) T_n; ascending Gauss grid z_j = -cos(pi (j+1/2)/N);
    weights pi/N.  Returns (forward M x N, backward N x M) for grid basis == coefficient basis;
    with conv (M x M) the forward matrix is conv @ forward (truncate-then-convert)."""
    j = np.arange(N)
    theta = np.pi * (j + 0.5) / N
    n = np.arange(M)[:, None]
    P = np.sqrt(2 / np.pi) * np.cos(n * (np.pi - theta[None, :]))
    P[0] = 1 / np.sqrt(np.pi)
    F = P * (np.pi / N)
    F[N:] = 0.0
    Bm = P.T.copy()
    Bm[:, N:] = 0.0
    if conv is not None:
        F = sparse.csr_matrix(conv) @ F
    return F, Bm


def apply_matrix_along_axis(mat, a, axis):
    """apply_dense tools/array.py:104-129"""
    am = np.moveaxis(a, axis, 0)
    out = (mat @ am.reshape(am.shape[0], -1)).reshape((mat.shape[0],) + am.shape[1:])
    return np.ascontiguousarray(np.moveaxis(out, 0, axis))
