"""
Orthonormal Jacobi polynomial toolkit (host-side setup; float64 out, long double inside).

p_n^{(a,b)} = P_n^{(a,b)} / sqrt(h_n^{(a,b)}),  int_{-1}^{1} (1-z)^a (1+z)^b p_m p_n dz = delta_mn.

Everything here is derived from the classical identities (DLMF 18.9.5, 18.9.15, 18.3.1):

    (2n+a+b+1) P_n^{(a,b)} = (n+a+b+1) P_n^{(a+1,b)} - (n+b) P_{n-1}^{(a+1,b)}
    d/dz P_n^{(a,b)}       = (n+a+b+1)/2 * P_{n-1}^{(a+1,b+1)}
    h_n^{(a,b)}            = 2^{a+b+1} G(n+a+1) G(n+b+1) / ((2n+a+b+1) G(n+a+b+1) n!)

which give, for the orthonormal family, the squared band entries used below.  The conventions
(what is a conversion / differentiation matrix, which way it acts on coefficient vectors) mirror
the reference so matrices can be compared entry by entry:
    dedalus/tools/jacobi.py:217-263  and  dedalus/libraries/dedalus_sphere/jacobi.py:30-133.
"""

import numpy as np
from scipy import sparse
from scipy.special import gammaln

LD = np.longdouble


def mass(a, b):
    """int (1-z)^a (1+z)^b dz over [-1, 1]."""
    return float(np.exp((a + b + 1) * np.log(2.0) + gammaln(a + 1) + gammaln(b + 1) - gammaln(a + b + 2)))


def _step_a(N, a, b):
    """Coefficient map (a,b) -> (a+1,b) as (diag, superdiag) of an N x N upper-bidiagonal matrix."""
    n = np.arange(N, dtype=LD)
    a = LD(a)
    b = LD(b)
    s = 2 * n + a + b
    r = np.ones(N, dtype=LD)
    if N > 1:
        r[1:] = (n[1:] + a + b + 1) / (s[1:] + 1)
    diag = np.sqrt(2 * (n + a + 1) / (s + 2) * r)
    sup = np.zeros(N, dtype=LD)        # sup[n] = C[n-1, n]
    if N > 1:
        m = n[1:]
        sup[1:] = -np.sqrt(2 * m * (m + b) / ((s[1:]) * (s[1:] + 1)))
    return diag, sup


def _bidiag(diag, sup):
    N = len(diag)
    return sparse.diags([diag.astype(LD), sup[1:].astype(LD)], [0, 1], shape=(N, N), dtype=LD).tocsr()


def _conv_a(N, a, b):
    return _bidiag(*_step_a(N, a, b))


def _conv_b(N, a, b):
    # parity: P_n^{(a,b)}(-z) = (-1)^n P_n^{(b,a)}(z)  ->  same magnitudes with a<->b, superdiagonal +
    diag, sup = _step_a(N, b, a)
    return _bidiag(diag, -sup)


def conversion_matrix(N, a0, b0, a1, b1, dtype=np.float64):
    """Matrix taking (a0,b0) coefficients to (a1,b1) coefficients, a1-a0 and b1-b0 non-negative integers."""
    da, db = a1 - a0, b1 - b0
    if not (float(da).is_integer() and float(db).is_integer()) or da < 0 or db < 0:
        raise ValueError("conversion needs integer, non-negative parameter increments")
    C = sparse.identity(N, dtype=LD, format="csr")
    a, b = a0, b0
    # same order as the reference (A**da @ B**db acting on the input): B steps first, then A steps
    for _ in range(int(db)):
        C = _conv_b(N, a, b) @ C
        b += 1
    for _ in range(int(da)):
        C = _conv_a(N, a, b) @ C
        a += 1
    C = sparse.csr_matrix(C.astype(dtype))
    # products of the symmetric (a == b) steps cancel on the odd diagonals only up to long-double
    # round-off (~1e-20): remove that noise so the band structure is exact
    C.data[np.abs(C.data) < 1e-17] = 0.0
    C.eliminate_zeros()
    return C


def differentiation_matrix(N, a, b, dtype=np.float64):
    """d/dz: (a,b) coefficients -> (a+1,b+1) coefficients; single superdiagonal sqrt(n (n+a+b+1))."""
    n = np.arange(N, dtype=LD)
    sup = np.sqrt(n * (n + LD(a) + LD(b) + 1))
    D = sparse.diags([sup[1:]], [1], shape=(N, N), dtype=LD)
    return sparse.csr_matrix(D.astype(dtype))


def jacobi_matrix(N, a, b):
    """Symmetric tridiagonal matrix of multiplication by z in the orthonormal (a,b) basis (long double)."""
    A = _conv_a(N + 1, a, b)
    B = _conv_b(N + 1, a, b)
    J = ((B.T @ B) - (A.T @ A)) / LD(2)
    return J[:N, :N]


def polynomials(N, a, b, z, dtype=np.float64):
    """p_0..p_{N-1} evaluated at z (array), shape (N, len(z)); three-term recurrence in long double."""
    z = np.atleast_1d(np.asarray(z, dtype=LD))
    P = np.zeros((max(N, 1), z.size), dtype=LD)
    P[0] = 1 / np.sqrt(LD(mass(a, b)))
    if a == b == -0.5:
        P[0] = 1 / np.sqrt(LD(np.pi))
    if N > 1:
        J = jacobi_matrix(N + 1, a, b).toarray()
        d = np.diag(J)
        e = np.diag(J, 1)
        P[1] = (z - d[0]) * P[0] / e[0]
        for n in range(1, N - 1):
            P[n + 1] = ((z - d[n]) * P[n] - e[n - 1] * P[n - 1]) / e[n]
    return P[:N].astype(dtype)


def quadrature(N, a, b, dtype=np.float64):
    """Gauss-Jacobi nodes (ascending) and weights for the (a,b) weight, exact to degree 2N-1."""
    if a == b == -0.5:
        j = np.arange(N, dtype=LD)
        z = -np.cos(LD(np.pi) * (j + LD(0.5)) / N)
        w = np.full(N, LD(np.pi) / N)
        return z.astype(dtype), w.astype(dtype)
    J = jacobi_matrix(N, a, b).toarray()
    from scipy.linalg import eigvalsh_tridiagonal
    if N == 1:
        z = np.array([J[0, 0]], dtype=LD)
    else:
        z = eigvalsh_tridiagonal(np.diag(J).astype(np.float64), np.diag(J, 1).astype(np.float64)).astype(LD)
    # Newton polish on p_N in long double
    d = np.diag(jacobi_matrix(N + 2, a, b).toarray())
    e = np.diag(jacobi_matrix(N + 2, a, b).toarray(), 1)
    for _ in range(3):
        P = np.zeros((N + 1, N), dtype=LD)
        dP = np.zeros((N + 1, N), dtype=LD)
        P[0] = 1 / np.sqrt(LD(mass(a, b)))
        P[1] = (z - d[0]) * P[0] / e[0]
        dP[1] = P[0] / e[0]
        for n in range(1, N):
            P[n + 1] = ((z - d[n]) * P[n] - e[n - 1] * P[n - 1]) / e[n]
            dP[n + 1] = ((z - d[n]) * dP[n] + P[n] - e[n - 1] * dP[n - 1]) / e[n]
        z = z - P[N] / dP[N]
    Pn = polynomials(N, a, b, z, dtype=LD)
    w = 1 / np.sum(Pn ** 2, axis=0)
    return z.astype(dtype), w.astype(dtype)


def build_grid(N, a, b):
    return quadrature(N, a, b)[0]


def build_weights(N, a, b):
    return quadrature(N, a, b)[1]


def interpolation_vector(N, a, b, z):
    """Row vector of p_n(z), n < N (InterpolateJacobi, core/basis.py:713-735)."""
    return polynomials(N, a, b, np.array([z]))[:, 0]


def integration_vector(N, a, b):
    """int_{-1}^{1} p_n^{(a,b)}(z) dz (unweighted), n < N, via Gauss-Legendre quadrature.
    Entries at round-off level are zeroed like the reference does (tools/jacobi.py:249-263)."""
    zl, wl = quadrature(N, 0, 0, dtype=LD)
    P = polynomials(N, a, b, zl, dtype=LD)
    integ = (P @ wl).astype(np.float64)
    integ[np.abs(integ) <= np.finfo(np.float64).resolution] = 0.0
    return integ
