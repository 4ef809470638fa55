"""d3-style public namespace: `import dedalus_amd.public as d3` (mirrors dedalus/public.py:1-15)."""

from .core.coords import Coordinate, CartesianCoordinates
from .core.distributor import Distributor
from .core.basis import (RealFourier, Jacobi, Legendre, Ultraspherical, ChebyshevT, ChebyshevU, ChebyshevV,
                         Chebyshev)
from .core.field import Field
from .core.operators import (Gradient, Divergence, Laplacian, Differentiate, Integrate, Average, Interpolate,
                             Lift, Trace, TransposeComponents, Skew, Convert, TimeDerivative, DotProduct,
                             CrossProduct, Multiply, Add, Power, Component,
                             grad, div, lap, trace, dot, cross, skew, transpose, integ, ave, interp, lift, curl, dt)
from .core.problems import IVP, LBVP, InitialValueProblem, LinearBoundaryValueProblem
from .core.timesteppers import (schemes, CNAB1, SBDF1, CNAB2, MCNAB2, SBDF2, CNLF2, SBDF3, SBDF4,
                                RK111, RK222, RK443, RKSMR, RKGFY)
from .extras.flow_tools import CFL, GlobalFlowProperty


# ---- the sphere (core/sphere.py): same names, dispatched on the coordinate system / operand type -----------------
from .core import sphere as _sphere
from .core.sphere import S2Coordinates, SphereBasis
from .core import shell as _shell
from .core.shell import SphericalCoordinates, ShellBasis

_CartesianDistributor = Distributor


def Distributor(coordsystems, *args, **kw):
    if isinstance(coordsystems, SphericalCoordinates):
        return _shell.ShellDistributor(coordsystems, *args, **kw)
    if isinstance(coordsystems, S2Coordinates):
        return _sphere.SphereDistributor(coordsystems, *args, **kw)
    return _CartesianDistributor(coordsystems, *args, **kw)


def _dispatch(name, cart):
    sph = getattr(_sphere, name, None)
    shl = getattr(_shell, name, None)

    def f(operand, *args, **kw):
        if sph is not None and isinstance(operand, _sphere.SOperand):
            return sph(operand, *args, **kw)
        if shl is not None and isinstance(operand, _shell.ShOperand):
            return shl(operand, *args, **kw)
        return cart(operand, *args, **kw)
    f.__name__ = name
    return f


grad, div, lap, skew, ave, dt, trace, integ = (_dispatch(n, c) for n, c in (
    ("grad", grad), ("div", div), ("lap", lap), ("skew", skew), ("ave", ave), ("dt", dt), ("trace", trace),
    ("integ", integ)))
MulCosine = _sphere.MulCosine
_CartesianIVP, _CartesianLBVP = IVP, LBVP


_CartesianLift = Lift


def Lift(operand, *args, **kw):
    if isinstance(operand, _shell.ShOperand):
        return _shell.Lift(operand, *args, **kw)
    return _CartesianLift(operand, *args, **kw)


def IVP(variables, *args, **kw):
    if isinstance(variables[0], _shell.ShOperand):
        return _shell.ShellIVP(variables, *args, **kw)
    if isinstance(variables[0], _sphere.SOperand):
        return _sphere.SphereIVP(variables, *args, **kw)
    return _CartesianIVP(variables, *args, **kw)


def LBVP(variables, *args, **kw):
    if isinstance(variables[0], _shell.ShOperand):
        return _shell.ShellLBVP(variables, *args, **kw)
    if isinstance(variables[0], _sphere.SOperand):
        return _sphere.SphereLBVP(variables, *args, **kw)
    return _CartesianLBVP(variables, *args, **kw)


InitialValueProblem, LinearBoundaryValueProblem = IVP, LBVP
