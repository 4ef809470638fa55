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
