"""
Spin-weighted spherical harmonics on the Gauss-Legendre colatitude grid, and the per-(m, s) matrices
of the colatitude transform (what SWSHColatitudeTransform builds, core/transforms.py:1290-1340, from
libraries/dedalus_sphere/sphere.py:8-66).

Derived from the Jacobi toolkit in tools/jacobi.py:  with a = |m+s|, b = |m-s|, Lmin = max(|m|, |s|),

    Y^s_{l,m}(z) = (-1)^max(m,-s) * sqrt((1-z)^a (1+z)^b) * p^{(a,b)}_{l-Lmin}(z),     z = cos(theta),

where p^{(a,b)}_k are the Jacobi polynomials orthonormal for the weight (1-z)^a (1+z)^b, so that
int Y_l Y_l' dz = delta.  The three-term recurrence is started from the envelope-scaled p_0 in long
double and in log space, which keeps |m| of several hundred free of over/underflow.
"""

import numpy as np
from scipy.special import gammaln

from . import jacobi

LD = np.longdouble


_QUAD = {}


def quadrature(Ntheta):
    """Gauss-Legendre nodes cos(theta) (ascending) and weights, exact to degree 2 Ntheta - 1 (cached: every
    (m, s) matrix of a transform plan uses the same grid)."""
    if Ntheta not in _QUAD:
        _QUAD[Ntheta] = jacobi.quadrature(Ntheta, 0, 0, dtype=LD)
    return _QUAD[Ntheta]


def _log_mass(a, b):
    # int (1-z)^a (1+z)^b dz = 2^(a+b+1) B(a+1, b+1)
    return (a + b + 1) * np.log(LD(2)) + LD(gammaln(a + 1) + gammaln(b + 1) - gammaln(a + b + 2))


def harmonics(Lmax, m, s, z):
    """Y^s_{l,m}(z) for l = Lmin..Lmax: array (Lmax + 1 - Lmin, len(z)) in long double."""
    z = np.atleast_1d(np.asarray(z, dtype=LD))
    Lmin = max(abs(m), abs(s))
    n = Lmax + 1 - Lmin
    a, b = abs(m + s), abs(m - s)
    if n < 1:
        return np.zeros((0, z.size), dtype=LD)
    log_env = 0.5 * (a * np.log1p(-z) + b * np.log1p(z) - _log_mass(a, b))
    P = np.zeros((n, z.size), dtype=LD)
    P[0] = np.exp(log_env) * LD((-1.0) ** max(m, -s))
    if n > 1:
        J = jacobi.jacobi_matrix(n + 1, a, b).toarray()
        d, e = np.diag(J), np.diag(J, 1)
        P[1] = (z - d[0]) * P[0] / e[0]
        for k in range(1, n - 1):
            P[k + 1] = ((z - d[k]) * P[k] - e[k - 1] * P[k - 1]) / e[k]
    return P


def swsh_matrices(Ntheta, Lmax, m, s):
    """(forward, backward) matrices of the colatitude transform for one (m, s):
    forward (Lmax+1-|m|, Ntheta) = Y * weights, backward (Ntheta, Lmax+1-|m|) = Y^T, both padded with zero
    rows / columns for l < Lmin and zeroed for l >= Ntheta (core/transforms.py:1300-1316, 1326-1340)."""
    z, w = quadrature(Ntheta)
    Y = harmonics(Lmax, m, s, z)
    Lmin = max(abs(m), abs(s))
    am = abs(m)
    fwd = np.zeros((Lmax + 1 - am, Ntheta))
    bwd = np.zeros((Ntheta, Lmax + 1 - am))
    fwd[Lmin - am:, :] = (Y * w).astype(np.float64)
    bwd[:, Lmin - am:] = Y.T.astype(np.float64)
    if Ntheta - am < Lmax + 1 - am:
        fwd[max(Ntheta - am, 0):, :] = 0
        bwd[:, max(Ntheta - am, 0):] = 0
    return fwd, bwd


# ---- regularity <-> spin intertwiner ------------------------------------------------------------------
def intertwiner(ell, rank, indexing=(-1, +1, 0)):
    """Q(ell)[spin, regularity] for tensors of the given rank: the orthogonal map between regularity
    components (radial behaviour r^(ell+a)) and spin components at degree ell, both indexed over
    `indexing`^rank in C order (what dedalus_sphere.spin_operators.Intertwiner(ell, indexing)(rank) returns,
    libraries/dedalus_sphere/spin_operators.py:276-362; used by radial_recombinations, core/basis.py:3549-3560).

    Recursive definition (Vasil, Lecoanet, Burns, Oishi & Brown 2019, tensor calculus in spherical
    coordinates): peel the first index (sigma, a) off the spin / regularity tuples,
        Q_{sigma tau, a b} = C_a( Q_{tau b} [sigma = 0],  R ),      J = ell + sum(b),
        R = sum_i ( [tau_i = 0] Q_{tau(i->sigma), b} - [tau_i = -sigma] Q_{tau(i->0), b} ) - k(sigma, sum tau) Q_{tau b},
        k(mu, s) = -mu sqrt((ell - s mu)(ell + s mu + 1) / 2),
        C_-1 = (Q J - R) / sqrt(J (2J+1)),  C_0 = sigma R / sqrt(J (J+1)),  C_+1 = (Q (J+1) + R) / sqrt((J+1)(2J+1)).
    Components that cannot exist at this ell (|sum spin| > ell, or a regularity walk that leaves ell >= 0)
    are zero."""
    from itertools import product
    L = int(ell)
    cache = {}

    def forbidden_spin(spin):
        return L < abs(sum(spin))

    def forbidden_reg(reg):
        if L >= len(reg):
            return False
        walk = [L]
        for r in reg[::-1]:
            walk.append(walk[-1] + r)
            if walk[-1] < 0 or (walk[-1] == 0 and walk[-2] == 0):
                return True
        return False

    def kfun(mu, s):
        return -mu * np.sqrt((L - s * mu) * (L + s * mu + 1) / 2)

    def q(spin, reg):
        key = (spin, reg)
        if key in cache:
            return cache[key]
        if len(spin) == 0:
            val = 1.0
        elif forbidden_spin(spin) or forbidden_reg(reg):
            val = 0.0
        else:
            sigma, a = spin[0], reg[0]
            tau, b = spin[1:], reg[1:]
            R = 0.0
            for i, t in enumerate(tau):
                if t + sigma == 0:
                    R -= q(tau[:i] + (0,) + tau[i + 1:], b)
                if t == 0:
                    R += q(tau[:i] + (sigma,) + tau[i + 1:], b)
            Q0 = q(tau, b)
            R -= kfun(sigma, sum(tau)) * Q0
            J = L + sum(b)
            Qs = Q0 if sigma == 0 else 0.0
            if a == -1:
                val = (Qs * J - R) / np.sqrt(J * (2 * J + 1))
            elif a == 0:
                val = sigma * R / np.sqrt(J * (J + 1))
            else:
                val = (Qs * (J + 1) + R) / np.sqrt((J + 1) * (2 * J + 1))
        cache[key] = val
        return val

    idx = list(product(*(rank * (tuple(indexing),))))
    out = np.zeros((len(idx), len(idx)))
    for i, spin in enumerate(idx):
        for j, reg in enumerate(idx):
            out[i, j] = q(spin, reg)
    return out
