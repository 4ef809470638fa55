"""
Radial operators of spherical shells on the weighted Jacobi bases
    psi_n(r) = (dR / r)^k  p_n^{(k + alpha0, k + alpha1)}(z),      r = dR/2 (z + rho),  rho = (Ro + Ri) / dR,
what libraries/dedalus_sphere/shell.py:10-66 assembles from its Jacobi operator algebra and
ShellRadialBasis.operator_matrix / conversion_matrix / jacobi_conversion return (core/basis.py:3846-3875).
Restated directly from the calculus, with the orthonormal-Jacobi toolkit of tools/jacobi.py:

  multiplying a basis function by r/dR raises k by one, so with f the polynomial part
      d/dr [(dR/r)^k f]            = (dR/r)^(k+1) (1/dR) [ (z + rho) f' - k f ],
      (d/dr - l/r)     [...]       = (dR/r)^(k+1) (1/dR) [ (z + rho) f' - (k + l) f ]          ("D+", raises regularity)
      (d/dr + (l+1)/r) [...]       = (dR/r)^(k+1) (1/dR) [ (z + rho) f' + (l + 1 - k) f ]      ("D-", lowers regularity)
  where (z + rho) f' + f = d/dz[(z + rho) f] is evaluated exactly through the (N+1) x N multiplication and the
  N x (N+1) differentiation matrices, and every result is expressed in the (k+1) polynomial family.
All matrices are N x N, real, banded.
"""
import functools

import numpy as np

from . import jacobi


def _arr(m):
    return np.asarray(m.toarray() if hasattr(m, "toarray") else m, dtype=np.float64)


def _zrect(N, a, b, rho):
    """(N+1) x N: multiplication by (z + rho) of p_0..p_{N-1} in the (a, b) family."""
    J = _arr(jacobi.jacobi_matrix(N + 1, a, b))[:, :N]
    Z = J.copy()
    Z[:N, :N] += rho * np.eye(N)
    return Z


def conversion(N, a, b):
    """N x N: (a, b) -> (a+1, b+1) coefficients ('AB')."""
    return _arr(jacobi.conversion_matrix(N, a, b, a + 1, b + 1))


@functools.lru_cache(maxsize=64)
def _D_parts(N, k, radii, alpha):
    """The ell-independent pieces of D: d/dz[(z + rho) .] and the (k -> k + 1) conversion.  D is requested once per
    operator and ell (hundreds of times while the matrices of a shell problem are built)."""
    dR = radii[1] - radii[0]
    rho = (radii[1] + radii[0]) / dR
    a, b = k + alpha[0], k + alpha[1]
    D1 = _arr(jacobi.differentiation_matrix(N + 1, a, b))[:N, :N + 1]
    P, C = D1 @ _zrect(N, a, b, rho), conversion(N, a, b)
    P.setflags(write=False)
    C.setflags(write=False)
    return P, C


def D(dl, ell, N, k, radii, alpha=(-0.5, -0.5)):
    """shell.operator(3, radii, 'D')(dl, ell)(N, k): k -> k + 1."""
    dR = radii[1] - radii[0]
    P, C = _D_parts(int(N), int(k), (float(radii[0]), float(radii[1])), (float(alpha[0]), float(alpha[1])))
    K = k + 1 + dl * ell - (1 if dl == -1 else 0)
    return (P - K * C) / dR


def E(N, k, radii, alpha=(-0.5, -0.5)):
    """'E': multiplication by r / dR, k -> k + 1 (conversion_matrix of the reference uses its powers)."""
    dR = radii[1] - radii[0]
    rho = (radii[1] + radii[0]) / dR
    a, b = k + alpha[0], k + alpha[1]
    return 0.5 * (conversion(N + 1, a, b) @ _zrect(N, a, b, rho))[:N, :N]


def E_power(N, k, dk, radii, alpha=(-0.5, -0.5)):
    """E^dk: k -> k + dk (ShellRadialBasis.conversion_matrix, core/basis.py:3868-3872).  Each factor raises the
    polynomial degree by one, so the product runs through rectangular (N+j+1) x (N+j) factors and is truncated
    to N x N only at the end, like the reference's operator algebra."""
    dR = radii[1] - radii[0]
    rho = (radii[1] + radii[0]) / dR
    M = np.eye(N)
    for j in range(dk):
        a, b = k + j + alpha[0], k + j + alpha[1]
        M = 0.5 * (conversion(N + j + 1, a, b) @ _zrect(N + j, a, b, rho)) @ M
    return M[:N, :N]


def R(N, k, radii, alpha=(-0.5, -0.5)):
    """'R': multiplication by r within the same k (truncated to N x N)."""
    dR = radii[1] - radii[0]
    rho = (radii[1] + radii[0]) / dR
    return (0.5 * dR) * _zrect(N, k + alpha[0], k + alpha[1], rho)[:N, :N]


def operator_matrix(op, ell, regtotal, N, k, radii, alpha=(-0.5, -0.5)):
    """ShellRadialBasis.operator_matrix (core/basis.py:3846-3858)."""
    l = ell + regtotal
    if op == "D+":
        return D(+1, l, N, k, radii, alpha)
    if op == "D-":
        return D(-1, l, N, k, radii, alpha)
    if op == "L":
        return D(-1, l + 1, N, k + 1, radii, alpha) @ D(+1, l, N, k, radii, alpha)
    if op == "E":
        return E(N, k, radii, alpha)
    if op == "R":
        return R(N, k, radii, alpha)
    if op == "AB":
        return conversion(N, k + alpha[0], k + alpha[1])
    if op == "Id":
        return np.eye(N)
    raise ValueError(op)


def xi(mu, l):
    """RegularityBasis.xi (core/basis.py:3545-3546)"""
    return np.sqrt((l + (mu + 1) // 2) / (2 * l + 1))


def interpolation(position, N, k, radii, alpha=(-0.5, -0.5)):
    """ShellRadialBasis.interpolation (core/basis.py:3799-3805): row vector of basis values at r = position."""
    dR = radii[1] - radii[0]
    rho = (radii[1] + radii[0]) / dR
    z = position * 2 / dR - rho
    return (dR / position) ** k * np.asarray(jacobi.polynomials(N, k + alpha[0], k + alpha[1], np.array([z]))).reshape(-1)
