"""CPU, build container only (skipped where /root/reference is absent, i.e. on the GPU box): `bindings.install` and
`bindings.install_matsolver` against the REAL registries of the unmodified reference, imported through oracle/refshim
-- `dedalus.core.transforms.register_transform` (core/transforms.py:27-32) and `dedalus.libraries.matsolvers.add_solver`
(libraries/matsolvers.py:10-13) -- and the reference's own bases then select the plan classes with `library='hip'`
(core/basis.py:485-509, 849-922).  Constructing a plan needs the GPU: that half is tests/test_gpu_boundary.py."""
import numpy as np
import pytest

from oracle import refshim

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference checkout not present")


def test_install_into_the_reference_registries():
    d3 = refshim.load_reference()
    from dedalus.core import basis as rbasis, transforms as rtransforms
    from dedalus.libraries import matsolvers as rmatsolvers
    from dedalus_amd import bindings
    bindings.install(rtransforms.register_transform, rbasis.RealFourier, rbasis.ComplexFourier, rbasis.Jacobi,
                     rbasis.SphereBasis)
    assert rbasis.RealFourier.transforms["hip"] is bindings.HipRealFFT
    assert rbasis.ComplexFourier.transforms["hip"] is bindings.HipComplexFFT
    assert rbasis.Jacobi.transforms["hip"] is bindings.HipJacobi
    assert rbasis.SphereBasis.transforms["hip"] is bindings.HipSWSHColatitude
    # the reference's plans are still there
    assert "fftw" in rbasis.RealFourier.transforms and "matrix" in rbasis.Jacobi.transforms
    bindings.install_matsolver(rmatsolvers.add_solver)
    assert rmatsolvers.matsolvers["hipbandmatsolver"] is bindings.HipBandMatsolver
    # the reference's bases accept the library name and would construct the plan classes with their own arguments
    coords = d3.CartesianCoordinates('x', 'z')
    xb = d3.RealFourier(coords['x'], size=16, bounds=(0, 1), dealias=3 / 2, library='hip')
    zb = d3.ChebyshevT(coords['z'], size=12, bounds=(0, 1), dealias=3 / 2, library='hip')
    assert xb.library == 'hip' and zb.library == 'hip'
    assert type(xb).transforms[xb.library] is bindings.HipRealFFT
    assert type(zb).transforms[zb.library] is bindings.HipJacobi
    # the constructor signatures line up with what the reference passes (core/basis.py:509, 922)
    import inspect
    assert list(inspect.signature(bindings.HipRealFFT.__init__).parameters)[1:3] == ["grid_size", "coeff_size"]
    assert list(inspect.signature(bindings.HipJacobi.__init__).parameters)[1:7] == ["grid_size", "coeff_size", "a", "b", "a0", "b0"]
    # matsolver interface: cls(matrix, solver), .solve(vector), class attribute config (core/solvers.py:112-116)
    assert list(inspect.signature(bindings.HipBandMatsolver.__init__).parameters)[1:3] == ["matrix", "solver"]
    assert isinstance(bindings.HipBandMatsolver.config, dict) and hasattr(bindings.HipBandMatsolver, "solve")
