"""Problem scripts shared by the golden generator (run against the reference) and the tests (run
against dedalus_amd).  Each takes the d3 namespace to use, so the SAME script text drives both --
the bodies are the reference's example scripts (examples/ivp_1d_kdv_burgers/kdv_burgers.py:25-55,
examples/ivp_2d_rayleigh_benard/rayleigh_benard.py:32-89) and the 3-D extension of SURVEY.md
Appendix B."""
import numpy as np


def kdv_burgers(d3, Nx=64, timestepper="SBDF2", dist_kw=None):
    Lx, a, b, dealias = 10, 1e-4, 2e-4, 3 / 2
    xcoord = d3.Coordinate('x')
    dist = d3.Distributor(xcoord, dtype=np.float64, **(dist_kw or {}))
    xbasis = d3.RealFourier(xcoord, size=Nx, bounds=(0, Lx), dealias=dealias)
    u = dist.Field(name='u', bases=xbasis)
    dx = lambda A: d3.Differentiate(A, xcoord)
    problem = d3.IVP([u], namespace=locals())
    problem.add_equation("dt(u) - a*dx(dx(u)) - b*dx(dx(dx(u))) = - u*dx(u)")
    x = dist.local_grid(xbasis)
    n = 20
    u['g'] = np.log(1 + np.cosh(n) ** 2 / np.cosh(n * (x - 0.2 * Lx)) ** 2) / (2 * n)
    solver = problem.build_solver(getattr(d3, timestepper))
    return solver, dict(u=u)


def rayleigh_benard_2d(d3, Nx=32, Nz=16, timestepper="RK222", dist_kw=None):
    Lx, Lz = 4, 1
    Rayleigh, Prandtl, dealias = 2e6, 1, 3 / 2
    coords = d3.CartesianCoordinates('x', 'z')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    xbasis = d3.RealFourier(coords['x'], size=Nx, bounds=(0, Lx), dealias=dealias)
    zbasis = d3.ChebyshevT(coords['z'], size=Nz, bounds=(0, Lz), dealias=dealias)
    p = dist.Field(name='p', bases=(xbasis, zbasis))
    b = dist.Field(name='b', bases=(xbasis, zbasis))
    u = dist.VectorField(coords, name='u', bases=(xbasis, zbasis))
    tau_p = dist.Field(name='tau_p')
    tau_b1 = dist.Field(name='tau_b1', bases=xbasis)
    tau_b2 = dist.Field(name='tau_b2', bases=xbasis)
    tau_u1 = dist.VectorField(coords, name='tau_u1', bases=xbasis)
    tau_u2 = dist.VectorField(coords, name='tau_u2', bases=xbasis)
    kappa = (Rayleigh * Prandtl) ** (-1 / 2)
    nu = (Rayleigh / Prandtl) ** (-1 / 2)
    x, z = dist.local_grids(xbasis, zbasis)
    ex, ez = coords.unit_vector_fields(dist)
    lift_basis = zbasis.derivative_basis(1)
    lift = lambda A: d3.Lift(A, lift_basis, -1)
    grad_u = d3.grad(u) + ez * lift(tau_u1)
    grad_b = d3.grad(b) + ez * lift(tau_b1)
    problem = d3.IVP([p, b, u, tau_p, tau_b1, tau_b2, tau_u1, tau_u2], namespace=locals())
    problem.add_equation("trace(grad_u) + tau_p = 0")
    problem.add_equation("dt(b) - kappa*div(grad_b) + lift(tau_b2) = - u@grad(b)")
    problem.add_equation("dt(u) - nu*div(grad_u) + grad(p) - b*ez + lift(tau_u2) = - u@grad(u)")
    problem.add_equation("b(z=0) = Lz")
    problem.add_equation("u(z=0) = 0")
    problem.add_equation("b(z=Lz) = 0")
    problem.add_equation("u(z=Lz) = 0")
    problem.add_equation("integ(p) = 0")
    solver = problem.build_solver(getattr(d3, timestepper))
    b.fill_random('g', seed=42, distribution='normal', scale=1e-3)
    b['g'] *= z * (Lz - z)
    b['g'] += Lz - z
    return solver, dict(p=p, b=b, u=u, tau_b1=tau_b1, tau_b2=tau_b2, tau_u1=tau_u1, tau_u2=tau_u2)


def rayleigh_benard_3d(d3, Nx=8, Ny=12, Nz=8, timestepper="RK222", dist_kw=None, Lx=4, Ly=4):
    Lz = 1
    Rayleigh, Prandtl, dealias = 2e6, 1, 3 / 2
    coords = d3.CartesianCoordinates('x', 'y', 'z')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    xbasis = d3.RealFourier(coords['x'], size=Nx, bounds=(0, Lx), dealias=dealias)
    ybasis = d3.RealFourier(coords['y'], size=Ny, bounds=(0, Ly), dealias=dealias)
    zbasis = d3.ChebyshevT(coords['z'], size=Nz, bounds=(0, Lz), dealias=dealias)
    B, Bh = (xbasis, ybasis, zbasis), (xbasis, ybasis)
    p = dist.Field(name='p', bases=B)
    b = dist.Field(name='b', bases=B)
    u = dist.VectorField(coords, name='u', bases=B)
    tau_p = dist.Field(name='tau_p')
    tau_b1 = dist.Field(name='tau_b1', bases=Bh)
    tau_b2 = dist.Field(name='tau_b2', bases=Bh)
    tau_u1 = dist.VectorField(coords, name='tau_u1', bases=Bh)
    tau_u2 = dist.VectorField(coords, name='tau_u2', bases=Bh)
    kappa = (Rayleigh * Prandtl) ** (-1 / 2)
    nu = (Rayleigh / Prandtl) ** (-1 / 2)
    x, y, z = dist.local_grids(*B)
    ex, ey, ez = coords.unit_vector_fields(dist)
    lift_basis = zbasis.derivative_basis(1)
    lift = lambda A: d3.Lift(A, lift_basis, -1)
    grad_u = d3.grad(u) + ez * lift(tau_u1)
    grad_b = d3.grad(b) + ez * lift(tau_b1)
    problem = d3.IVP([p, b, u, tau_p, tau_b1, tau_b2, tau_u1, tau_u2], namespace=locals())
    problem.add_equation("trace(grad_u) + tau_p = 0")
    problem.add_equation("dt(b) - kappa*div(grad_b) + lift(tau_b2) = - u@grad(b)")
    problem.add_equation("dt(u) - nu*div(grad_u) + grad(p) - b*ez + lift(tau_u2) = - u@grad(u)")
    problem.add_equation("b(z=0) = Lz")
    problem.add_equation("u(z=0) = 0")
    problem.add_equation("b(z=Lz) = 0")
    problem.add_equation("u(z=Lz) = 0")
    problem.add_equation("integ(p) = 0")
    solver = problem.build_solver(getattr(d3, timestepper))
    b.fill_random('g', seed=42, distribution='normal', scale=1e-3)
    b['g'] *= z * (Lz - z)
    b['g'] += Lz - z
    return solver, dict(p=p, b=b, u=u, tau_b1=tau_b1, tau_b2=tau_b2, tau_u1=tau_u1, tau_u2=tau_u2)


def shear_flow_2d(d3, Nx=32, Nz=64, timestepper="RK222", dist_kw=None):
    """The reference's example examples/ivp_2d_shear_flow/shear_flow.py:28-75 (periodic Fourier x Fourier shear layer
    with a passive tracer, pressure gauge through tau_p), parameterised by resolution only."""
    Lx, Lz = 1, 2
    Reynolds, Schmidt, dealias = 5e4, 1, 3 / 2
    coords = d3.CartesianCoordinates('x', 'z')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    xbasis = d3.RealFourier(coords['x'], size=Nx, bounds=(0, Lx), dealias=dealias)
    zbasis = d3.RealFourier(coords['z'], size=Nz, bounds=(-Lz / 2, Lz / 2), dealias=dealias)
    p = dist.Field(name='p', bases=(xbasis, zbasis))
    s = dist.Field(name='s', bases=(xbasis, zbasis))
    u = dist.VectorField(coords, name='u', bases=(xbasis, zbasis))
    tau_p = dist.Field(name='tau_p')
    nu = 1 / Reynolds
    D = nu / Schmidt
    x, z = dist.local_grids(xbasis, zbasis)
    ex, ez = coords.unit_vector_fields(dist)
    problem = d3.IVP([u, s, p, tau_p], namespace=locals())
    problem.add_equation("dt(u) + grad(p) - nu*lap(u) = - u@grad(u)")
    problem.add_equation("dt(s) - D*lap(s) = - u@grad(s)")
    problem.add_equation("div(u) + tau_p = 0")
    problem.add_equation("integ(p) = 0")
    solver = problem.build_solver(getattr(d3, timestepper))
    u['g'][0] = 1 / 2 + 1 / 2 * (np.tanh((z - 0.5) / 0.1) - np.tanh((z + 0.5) / 0.1))
    s['g'] = u['g'][0]
    u['g'][1] += 0.1 * np.sin(2 * np.pi * x / Lx) * np.exp(-(z - 0.5) ** 2 / 0.01)
    u['g'][1] += 0.1 * np.sin(2 * np.pi * x / Lx) * np.exp(-(z + 0.5) ** 2 / 0.01)
    return solver, dict(u=u, s=s, p=p)


def poisson_2d(d3, Nx=64, Ny=32, dist_kw=None):
    """The reference's example examples/lbvp_2d_poisson/poisson.py:27-64 (Fourier x Chebyshev Poisson LBVP with
    tau lifts, a Dirichlet and a Neumann condition, low-pass filtered random forcing)."""
    Lx, Ly = 2 * np.pi, np.pi
    coords = d3.CartesianCoordinates('x', 'y')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    xbasis = d3.RealFourier(coords['x'], size=Nx, bounds=(0, Lx))
    ybasis = d3.Chebyshev(coords['y'], size=Ny, bounds=(0, Ly))
    u = dist.Field(name='u', bases=(xbasis, ybasis))
    tau_1 = dist.Field(name='tau_1', bases=xbasis)
    tau_2 = dist.Field(name='tau_2', bases=xbasis)
    x, y = dist.local_grids(xbasis, ybasis)
    f = dist.Field(bases=(xbasis, ybasis))
    g = dist.Field(bases=xbasis)
    h = dist.Field(bases=xbasis)
    f.fill_random('g', seed=40)
    f.low_pass_filter(shape=(Nx // 4, Ny // 4))
    g['g'] = np.sin(8 * x) * 0.025
    h['g'] = 0
    dy = lambda A: d3.Differentiate(A, coords['y'])
    lift_basis = ybasis.derivative_basis(2)
    lift = lambda A, n: d3.Lift(A, lift_basis, n)
    problem = d3.LBVP([u, tau_1, tau_2], namespace=locals())
    problem.add_equation("lap(u) + lift(tau_1,-1) + lift(tau_2,-2) = f")
    problem.add_equation("u(y=0) = g")
    problem.add_equation("dy(u)(y=Ly) = h")
    solver = problem.build_solver()
    solver.solve()
    return solver, dict(u=u, f=f, tau_1=tau_1, tau_2=tau_2)


IVP_CASES = {
    # name: (builder, kwargs, timestep, number of steps)
    "kdv64_sbdf2": (kdv_burgers, dict(Nx=64, timestepper="SBDF2"), 2e-3, 20),
    "kdv64_rk443": (kdv_burgers, dict(Nx=64, timestepper="RK443"), 2e-3, 8),
    "rb2d_32x16_rk222": (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper="RK222"), 1e-3, 6),
    "rb2d_32x16_sbdf2": (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper="SBDF2"), 1e-3, 6),
    "rb2d_64x32_rk222": (rayleigh_benard_2d, dict(Nx=64, Nz=32, timestepper="RK222"), 1e-3, 10),
    "rb3d_8x12x8_rk222": (rayleigh_benard_3d, dict(Nx=8, Ny=12, Nz=8, timestepper="RK222"), 1e-3, 4),
    "rb3d_16x16x16_rk222": (rayleigh_benard_3d, dict(Nx=16, Ny=16, Nz=16, timestepper="RK222"), 1e-3, 3),
    "shear2d_32x64_rk222": (shear_flow_2d, dict(Nx=32, Nz=64, timestepper="RK222"), 5e-3, 8),
    "shear2d_32x64_sbdf2": (shear_flow_2d, dict(Nx=32, Nz=64, timestepper="SBDF2"), 5e-3, 8),
}


def heat_forced(d3, N=16, timestepper="SBDF2", dist_kw=None):
    """The reference's timestepper test problem (tests/test_ivp.py:20-49: forced 1-D heat equation, analytic solution
    (1 - exp(-t)) sin x) on the real Fourier basis, plus a second forced mode so that both parities carry data."""
    c = d3.Coordinate('x')
    dist = d3.Distributor(c, dtype=np.float64, **(dist_kw or {}))
    b = d3.RealFourier(c, size=N, bounds=(0, 2 * np.pi), dealias=1)
    x = dist.local_grid(b, scale=1)
    u = dist.Field(name='u', bases=b)
    F = dist.Field(name='F', bases=b)
    F['g'] = np.sin(x) + 0.5 * np.cos(3 * x)
    dx = lambda A: d3.Differentiate(A, c)
    problem = d3.IVP([u], namespace=locals())
    problem.add_equation("dt(u) - dx(dx(u)) = F")
    solver = problem.build_solver(getattr(d3, timestepper))
    return solver, dict(u=u)


ALL_SCHEMES = ("CNAB1", "SBDF1", "CNAB2", "MCNAB2", "SBDF2", "CNLF2", "SBDF3", "SBDF4", "RK111", "RK222", "RK443", "RKSMR")

# Every registered scheme stepped end to end (the reference pins all of them, tests/test_ivp.py:20-49).  A timestep
# SEQUENCE per case: the constant ones reach the steady coefficients past the start-up ramp of SBDF3 / SBDF4 / CNLF2,
# the varying ones exercise the variable-step coefficient formulas and the refactorization on every change.
SCHEME_CASES = {}
for _ts in ALL_SCHEMES:
    SCHEME_CASES["heat16_" + _ts] = (heat_forced, dict(N=16, timestepper=_ts), [1e-3] * 12)
    SCHEME_CASES["kdv64_" + _ts] = (kdv_burgers, dict(Nx=64, timestepper=_ts), [2e-3] * 10)
    SCHEME_CASES["rb2d_32x16_" + _ts] = (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper=_ts), [1e-3] * 7)
    SCHEME_CASES["rb2d_32x16_vardt_" + _ts] = (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper=_ts),
                                               [1e-3, 1e-3, 1.5e-3, 1.5e-3, 0.8e-3, 1.2e-3, 1.2e-3, 1.2e-3])


# Tolerances of the scheme cases: rel-L2, except that the Crank-Nicolson family (CNAB1/2, CNLF2) leaves the pressure and
# tau amplitudes of the index-2 constraints undamped -- they alternate from step to step in the reference too and pass
# through ~0 (|p| = 7.6e-8 where the damped schemes have 0.325) -- so their error is measured against the amplitude these
# fields have in the damped schemes (SCHEME_FLOOR) when their own norm is smaller.
SCHEME_TOL = {"p": 1e-10, "b": 1e-10, "u": 1e-9, "tau_b1": 1e-5, "tau_b2": 1e-5, "tau_u1": 1e-8, "tau_u2": 1e-8}
SCHEME_FLOOR = {"p": 0.325, "tau_b1": 7.6e-4, "tau_b2": 1.6e-4, "tau_u1": 2.4e-5, "tau_u2": 3.7e-5}


def scheme_error(key, v, ref):
    return np.linalg.norm(v - ref) / max(np.linalg.norm(ref), SCHEME_FLOOR.get(key, 1e-300))


def run_scheme_case(d3, name, dist_kw=None, before_step=None):
    builder, kw, dts = SCHEME_CASES[name]
    solver, fields = builder(d3, dist_kw=dist_kw, **kw)
    for i, dt in enumerate(dts):
        if before_step is not None:
            before_step(solver, i)
        solver.step(dt)
    return solver, {k: np.array(f['c']) for k, f in fields.items()}


def run_case(d3, name, dist_kw=None):
    builder, kw, dt, nsteps = IVP_CASES[name]
    solver, fields = builder(d3, dist_kw=dist_kw, **kw)
    for _ in range(nsteps):
        solver.step(dt)
    return solver, {k: np.array(f['c']) for k, f in fields.items()}


def run_cfl_case(d3, dist_kw=None, nsteps=45):
    """2-D RB with an O(1) initial flow and the example's adaptive-timestep loop
    (examples/ivp_2d_rayleigh_benard/rayleigh_benard.py:97-113)."""
    solver, f = rayleigh_benard_2d(d3, Nx=32, Nz=16, timestepper="RK222", dist_kw=dist_kw)
    u = f["u"]
    dist = u.dist
    xb, zb = [b for b in u.domain.bases]
    x, z = dist.local_grids(xb, zb)
    ug = np.zeros((2,) + np.broadcast(x, z).shape)
    ug[0] = 0.5 * np.sin(2 * np.pi * x / 4) * z * (1 - z) * 4
    ug[1] = 0.3 * np.cos(4 * np.pi * x / 4) * z * (1 - z) * 4
    u['g'] = ug
    CFL = d3.CFL(solver, initial_dt=0.02, cadence=3, safety=0.5, threshold=0.05, max_change=1.5,
                 min_change=0.5, max_dt=0.125)
    CFL.add_velocity(u)
    dts = []
    for _ in range(nsteps):
        dt = CFL.compute_timestep()
        solver.step(dt)
        dts.append(dt)
    return solver, np.array(dts), {k: np.array(v['c']) for k, v in f.items()}


# ---- sphere (BASELINE config 4) ------------------------------------------------

def sphere_operator_cases(d3, Nphi=16, Ntheta=12, radius=1.3, seed=5, dist_kw=None):
    """Fields and operator results on the sphere for seeded grid data (the operator set of the reference's
    shallow-water example, examples/ivp_sphere_shallow_water/shallow_water.py).  Returns dict name -> operand."""
    dealias = 3 / 2
    coords = d3.S2Coordinates('phi', 'theta')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    basis = d3.SphereBasis(coords, (Nphi, Ntheta), radius=radius, dealias=dealias, dtype=np.float64)
    u = dist.VectorField(coords, name='u', bases=basis)
    h = dist.Field(name='h', bases=basis)
    phi, theta = dist.local_grids(basis)
    rng = np.random.default_rng(seed)
    # band-limited smooth data: combinations of low-order harmonics in grid space
    x, y, z = np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta) + 0 * phi
    a = rng.standard_normal(10)
    h['g'] = a[0] + a[1] * x + a[2] * y + a[3] * z + a[4] * x * y + a[5] * y * z + a[6] * z * z + a[7] * x * x * z
    psi = dist.Field(name='psi', bases=basis)
    chi = dist.Field(name='chi', bases=basis)
    b = rng.standard_normal(10)
    psi['g'] = b[0] * x + b[1] * z + b[2] * x * z + b[3] * y * y + b[4] * x * y * z
    chi['g'] = b[5] * y + b[6] * z * z + b[7] * x * y + b[8] * x * z * z
    u0 = (d3.grad(psi) + d3.skew(d3.grad(chi))).evaluate()
    u0.change_scales(1)
    u['g'] = u0['g']
    zcross = lambda A: d3.MulCosine(d3.skew(A))
    exprs = dict(
        grad_h=d3.grad(h), lap_h=d3.lap(h), laplap_h=d3.lap(d3.lap(h)), mulcos_h=d3.MulCosine(h),
        div_u=d3.div(u), lap_u=d3.lap(u), laplap_u=d3.lap(d3.lap(u)), skew_u=d3.skew(u), zcross_u=zcross(u),
        grad_u=d3.grad(u), u_grad_u=u @ d3.grad(u), hu=h * u, div_hu=d3.div(h * u),
        vort=-d3.div(d3.skew(u)), rhs_lbvp=-d3.div(u @ d3.grad(u) + 2 * 0.7 * zcross(u)),
    )
    return dist, basis, dict(u=u, h=h), exprs


def sphere_operator_results(d3, **kw):
    dist, basis, fields, exprs = sphere_operator_cases(d3, **kw)
    res = {}
    for k, f in fields.items():
        f.change_scales(1)
        res[k + "__g"] = np.array(f['g'])
        res[k + "__c"] = np.array(f['c'])
    for k, e in exprs.items():
        f = e.evaluate()
        f.change_scales(1)
        res[k + "__g"] = np.array(f['g'])
        res[k + "__c"] = np.array(f['c'])
    return res


def shallow_water(d3, Nphi=32, Ntheta=16, timestepper="RK222", dist_kw=None):
    """The reference's example examples/ivp_sphere_shallow_water/shallow_water.py:24-92 (Galewsky et al. jet:
    balanced height from an LBVP, perturbation, IVP), parameterised by resolution only."""
    meter = 1 / 6.37122e6
    hour = 1
    second = hour / 3600
    dealias = 3 / 2
    R = 6.37122e6 * meter
    Omega = 7.292e-5 / second
    nu = 1e5 * meter ** 2 / second / 32 ** 2
    g = 9.80616 * meter / second ** 2
    H = 1e4 * meter
    dtype = np.float64
    coords = d3.S2Coordinates('phi', 'theta')
    dist = d3.Distributor(coords, dtype=dtype, **(dist_kw or {}))
    basis = d3.SphereBasis(coords, (Nphi, Ntheta), radius=R, dealias=dealias, dtype=dtype)
    u = dist.VectorField(coords, name='u', bases=basis)
    h = dist.Field(name='h', bases=basis)
    zcross = lambda A: d3.MulCosine(d3.skew(A))
    phi, theta = dist.local_grids(basis)
    lat = np.pi / 2 - theta + 0 * phi
    umax = 80 * meter / second
    lat0 = np.pi / 7
    lat1 = np.pi / 2 - lat0
    en = np.exp(-4 / (lat1 - lat0) ** 2)
    jet = (lat0 <= lat) * (lat <= lat1)
    u_jet = umax / en * np.exp(1 / (lat[jet] - lat0) / (lat[jet] - lat1))
    u['g'][0][jet] = u_jet
    c = dist.Field(name='c')
    problem = d3.LBVP([h, c], namespace=locals())
    problem.add_equation("g*lap(h) + c = - div(u@grad(u) + 2*Omega*zcross(u))")
    problem.add_equation("ave(h) = 0")
    solver = problem.build_solver()
    solver.solve()
    h_balanced = np.array(h['g'])
    lat2 = np.pi / 4
    hpert = 120 * meter
    alpha = 1 / 3
    beta = 1 / 15
    h['g'] += hpert * np.cos(lat) * np.exp(-(phi / alpha) ** 2) * np.exp(-((lat2 - lat) / beta) ** 2)
    problem = d3.IVP([u, h], namespace=locals())
    problem.add_equation("dt(u) + nu*lap(lap(u)) + g*grad(h) + 2*Omega*zcross(u) = - u@grad(u)")
    problem.add_equation("dt(h) + nu*lap(lap(h)) + H*div(u) = - div(h*u)")
    solver = problem.build_solver(getattr(d3, timestepper))
    return solver, dict(u=u, h=h), dict(h_balanced=h_balanced, timestep=600 * second)


def run_shallow_water(d3, steps=5, **kw):
    solver, fields, extra = shallow_water(d3, **kw)
    for _ in range(steps):
        solver.step(extra["timestep"])
    res = dict(h_balanced=extra["h_balanced"])
    for k, f in fields.items():
        f.change_scales(1)
        res[k + "__g"] = np.array(f['g'])
        res[k + "__c"] = np.array(f['c'])
    return solver, res


# ---- spherical shell ----------------------------------------------------------------

def shell_heat(d3, shape=(16, 12, 10), timestepper="SBDF2", dist_kw=None):
    """Heat equation in a spherical shell with tau lifts and Dirichlet conditions on both spheres: the scalar core of
    the reference's shell-convection example (examples/ivp_shell_convection/shell_convection.py:41-77) in second-order
    form, as in examples/lbvp_2d_poisson/poisson.py:55-57."""
    Ri, Ro, kappa = 0.7, 1.9, 0.05
    coords = d3.SphericalCoordinates('phi', 'theta', 'r')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=(Ri, Ro), dealias=3 / 2, dtype=np.float64)
    sphere = shell.outer_surface
    T = dist.Field(name='T', bases=shell)
    tau1 = dist.Field(name='tau1', bases=sphere)
    tau2 = dist.Field(name='tau2', bases=sphere)
    lift_basis = shell.derivative_basis(2)
    lift = lambda A, n: d3.Lift(A, lift_basis, n)
    phi, theta, r = dist.local_grids(shell)
    problem = d3.IVP([T, tau1, tau2], namespace=locals())
    problem.add_equation("dt(T) - kappa*lap(T) + lift(tau1,-1) + lift(tau2,-2) = 0")
    problem.add_equation("T(r=Ri) = 0")
    problem.add_equation("T(r=Ro) = 0")
    solver = problem.build_solver(getattr(d3, timestepper))
    T['g'] = (r - Ri) * (Ro - r) * (1 + 0.5 * np.sin(theta) * np.cos(phi) + 0.3 * np.cos(theta) * r
                                    + 0.2 * np.sin(theta) ** 2 * np.sin(2 * phi))
    return solver, dict(T=T, tau1=tau1, tau2=tau2)


def run_shell_heat(d3, steps=5, dt=0.01, **kw):
    solver, fields = shell_heat(d3, **kw)
    for _ in range(steps):
        solver.step(dt)
    res = {}
    for k, f in fields.items():
        res[k + "__c"] = np.array(f['c'])
    fields["T"].change_scales(1)
    res["T__g"] = np.array(fields["T"]['g'])
    return solver, res


def shell_operator_results(d3, shape=(16, 12, 8), dist_kw=None):
    """lap / grad / div of a smooth scalar field in the shell (evaluated, coefficient data)."""
    Ri, Ro = 0.7, 1.9
    coords = d3.SphericalCoordinates('phi', 'theta', 'r')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=(Ri, Ro), dealias=3 / 2, dtype=np.float64)
    T = dist.Field(name='T', bases=shell)
    phi, theta, r = dist.local_grids(shell)
    x, y, z = r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta) + 0 * phi
    T['g'] = 1 / r + 0.3 * x * y + 0.2 * z * z * x - 0.5 * y + 0.1 * x * x * z / r
    exprs = dict(lap_T=d3.lap(T), grad_T=d3.grad(T), div_grad_T=d3.div(d3.grad(T)), T_inner=T(r=Ri), T_outer=T(r=Ro))
    res = {"T__c": np.array(T['c'])}
    for k, e in exprs.items():
        res[k + "__c"] = np.array(e.evaluate()['c'])
    return res


def shell_analysis_results(d3, shape=(16, 12, 8), dist_kw=None):
    """The output tasks of the reference's shell example (examples/ivp_shell_convection/shell_convection.py:93-99):
    a radial NCC contracted with a flux on the grid, radial and azimuthal interpolations, at the dealias scales."""
    Ri, Ro = 0.7, 1.9
    coords = d3.SphericalCoordinates('phi', 'theta', 'r')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=(Ri, Ro), dealias=3 / 2, dtype=np.float64)
    b = dist.Field(name='b', bases=shell)
    u = dist.VectorField(coords, name='u', bases=shell)
    phi, theta, r = dist.local_grids(shell)
    x, y, z = r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta) + 0 * phi
    b['g'] = 1 / r + 0.3 * x * y + 0.2 * z * z * x - 0.5 * y + 0.1 * x * x * z / r
    ug = np.zeros((3,) + np.broadcast(phi, theta, r).shape)
    ug[0] = 0.8 * np.sin(theta) * (r - Ri) * (Ro - r) * (1 + 0.2 * np.sin(2 * phi))
    ug[1] = 0.5 * np.sin(theta) * np.cos(phi) * (r - Ri) * (Ro - r)
    ug[2] = 0.3 * np.cos(theta) * (r - Ri) * (Ro - r) + 0.1 * np.sin(theta) * np.sin(phi)
    u['g'] = ug
    er = dist.VectorField(coords, bases=shell.radial_basis)
    er['g'][2] = 1
    kappa = 0.3
    flux = er @ (-kappa * d3.grad(b) + u * b)
    tasks = dict(bmid=b(r=(Ri + Ro) / 2), flux_r_outer=flux(r=Ro), flux_r_inner=flux(r=Ri), flux_phi_start=flux(phi=0),
                 flux_phi_end=flux(phi=3 * np.pi / 2), flux=flux)
    res = {}
    for k, e in tasks.items():
        out = e.evaluate()
        out.change_scales(3 / 2)
        res[k] = np.array(out['g'])
    return res


def shell_convection(d3, shape=(16, 12, 8), timestepper="SBDF2", dist_kw=None):
    """The reference's example examples/ivp_shell_convection/shell_convection.py:31-91 (Boussinesq convection in a
    spherical shell: first-order tau formulation with radial NCCs, no-slip fixed-temperature walls, pressure gauge),
    parameterised by resolution; fixed timestep instead of the CFL loop."""
    Ri, Ro = 14, 15
    Rayleigh, Prandtl, dealias = 3500, 1, 3 / 2
    coords = d3.SphericalCoordinates('phi', 'theta', 'r')
    dist = d3.Distributor(coords, dtype=np.float64, **(dist_kw or {}))
    shell = d3.ShellBasis(coords, shape=shape, radii=(Ri, Ro), dealias=dealias, dtype=np.float64)
    sphere = shell.outer_surface
    p = dist.Field(name='p', bases=shell)
    b = dist.Field(name='b', bases=shell)
    u = dist.VectorField(coords, name='u', bases=shell)
    tau_p = dist.Field(name='tau_p')
    tau_b1 = dist.Field(name='tau_b1', bases=sphere)
    tau_b2 = dist.Field(name='tau_b2', bases=sphere)
    tau_u1 = dist.VectorField(coords, name='tau_u1', bases=sphere)
    tau_u2 = dist.VectorField(coords, name='tau_u2', bases=sphere)
    kappa = (Rayleigh * Prandtl) ** (-1 / 2)
    nu = (Rayleigh / Prandtl) ** (-1 / 2)
    phi, theta, r = dist.local_grids(shell)
    er = dist.VectorField(coords, bases=shell.radial_basis)
    er['g'][2] = 1
    rvec = dist.VectorField(coords, bases=shell.radial_basis)
    rvec['g'][2] = r
    lift_basis = shell.derivative_basis(1)
    lift = lambda A: d3.Lift(A, lift_basis, -1)
    grad_u = d3.grad(u) + rvec * lift(tau_u1)
    grad_b = d3.grad(b) + rvec * lift(tau_b1)
    problem = d3.IVP([p, b, u, tau_p, tau_b1, tau_b2, tau_u1, tau_u2], namespace=locals())
    problem.add_equation("trace(grad_u) + tau_p = 0")
    problem.add_equation("dt(b) - kappa*div(grad_b) + lift(tau_b2) = - u@grad(b)")
    problem.add_equation("dt(u) - nu*div(grad_u) + grad(p) - b*er + lift(tau_u2) = - u@grad(u)")
    problem.add_equation("b(r=Ri) = 1")
    problem.add_equation("u(r=Ri) = 0")
    problem.add_equation("b(r=Ro) = 0")
    problem.add_equation("u(r=Ro) = 0")
    problem.add_equation("integ(p) = 0")
    solver = problem.build_solver(getattr(d3, timestepper))
    b.fill_random('g', seed=42, distribution='normal', scale=1e-3)
    b['g'] *= (r - Ri) * (Ro - r)
    b['g'] += (Ri - Ri * Ro / r) / (Ri - Ro)
    return solver, dict(p=p, b=b, u=u, tau_p=tau_p, tau_b1=tau_b1, tau_b2=tau_b2, tau_u1=tau_u1, tau_u2=tau_u2)


def shell_band_limited_state(fields, grids):
    """A state of few modes for the shell-convection problem, the same FUNCTION at any resolution: Cartesian polynomials
    restricted to the shell (angular degree <= 3, polynomial in r), the velocity given through its spherical components.
    The explicit right-hand sides -u.grad(b), -u.grad(u) of such a state populate a handful of low (m, ell, n) modes whose
    coefficients do not depend on the number of modes carried (radial dependence of the 1/r factors: resolved to round-off
    by ~12 radial modes in the example's thin shell, radii 14 .. 15)."""
    phi, theta, r = grids
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    x, y, z = r * st * cp, r * st * sp, r * ct + 0 * phi
    s = 1.0 / 15.0                                      # (coordinates are O(15) in the example's shell: keep values O(1))
    x, y, z = s * x, s * y, s * z
    fields["b"]["g"] = 0.4 + 0.3 * x * y + 0.2 * z * z * x - 0.5 * y + 0.1 * x * x * z
    U = (0.5 * y * z - 0.2 * x, -0.4 * x * z + 0.3 * y * y, 0.25 * x * y + 0.1 * z)          # Cartesian components
    e_phi = (-sp + 0 * theta, cp + 0 * theta, 0 * phi + 0 * theta)
    e_theta = (ct * cp, ct * sp, -st + 0 * phi)
    e_r = (st * cp, st * sp, ct + 0 * phi)
    u = fields["u"]
    ug = np.zeros((3,) + np.broadcast(phi, theta, r).shape)
    for c, e in enumerate((e_phi, e_theta, e_r)):
        ug[c] = sum(U[k] * e[k] for k in range(3))
    u["g"] = ug


def shell_explicit_results(d3, shape, labels, dist_kw=None):
    """F of the shell-convection problem (the explicit half of a step: -u.grad(b), -u.grad(u), boundary constants) for the
    band-limited state above, evaluated through the SOLVER's right-hand-side path, as a resolution-independent table
    {(equation, component, m, part, ell, n): value} of the coefficients above 1e-13 of the largest.
    labels(field) -> (m, ell, n) integer arrays of the field's coefficient layout (the reference's
    local_group_arrays; this package's packed_groups)."""
    solver, f = shell_convection(d3, shape=shape, timestepper="SBDF2", dist_kw=dist_kw)
    b = f["b"]
    basis = b.domain.bases[0] if hasattr(b, "domain") else b.basis
    shell_band_limited_state(f, b.dist.local_grids(basis))
    if hasattr(solver, "evaluator") and hasattr(solver, "F") and not hasattr(solver, "evaluate_F"):
        solver.evaluator.evaluate_group("F", iteration=0, wall_time=0.0, sim_time=0.0, timestep=1.0)      # the reference
        Fs = list(solver.F)
    else:
        Fs = [(eq["F"].evaluate() if hasattr(eq["F"], "evaluate") else None) for eq in solver.problem.equations]
    table = {}
    for i, F in enumerate(Fs):
        if F is None or isinstance(F, (int, float)):
            continue
        c = np.array(F["c"])
        if c.size == 0 or not np.abs(c).max() > 0:
            continue
        m, ell, n = labels(F)
        c = c.reshape((-1,) + c.shape[-3:])
        scale = np.abs(c).max()
        for idx in zip(*np.nonzero(np.abs(c) > 1e-13 * scale)):
            comp, a, b_, k = (int(v) for v in idx)
            key = (i, comp, int(m[a, b_, k]), a % 2, int(ell[a, b_, k]), int(n[a, b_, k]))
            table[key] = float(c[idx])
    return table


def run_shell_convection(d3, steps=4, dt=0.05, **kw):
    solver, fields = shell_convection(d3, **kw)
    for _ in range(steps):
        solver.step(dt)
    return solver, {k: np.array(f['c']) for k, f in fields.items()}


def run_shell_cfl_case(d3, dist_kw=None, nsteps=20):
    """Shell convection with an O(1) initial flow and the example's adaptive-timestep loop
    (examples/ivp_shell_convection/shell_convection.py:97-118): the dt sequence pins the spherical CFL reduction,
    its scheduling and the refactorization on every dt change."""
    solver, f = shell_convection(d3, shape=(16, 12, 8), timestepper="SBDF2", dist_kw=dist_kw)
    u = f["u"]
    dist = u.dist
    shell = [v for v in (f["b"],)][0]
    coords = dist.coords
    phi, theta, r = dist.local_grids(_shell_basis_of(f["b"]))
    ug = np.zeros((3,) + np.broadcast(phi, theta, r).shape)
    ug[0] = 8.0 * np.sin(theta) * (r - 14) * (15 - r) * 4
    ug[1] = 5.0 * np.sin(theta) * np.cos(phi) * (r - 14) * (15 - r) * 4
    ug[2] = 3.0 * np.cos(theta) * (r - 14) * (15 - r) * 4
    u['g'] = ug
    CFL = d3.CFL(solver, initial_dt=0.006, cadence=2, safety=0.5, threshold=0.05, max_change=1.5, min_change=0.5,
                 max_dt=0.1)
    CFL.add_velocity(u)
    flow = d3.GlobalFlowProperty(solver, cadence=2)
    flow.add_property(np.sqrt(u @ u), name='speed')
    dts, speeds = [], []
    for _ in range(nsteps):
        dt = CFL.compute_timestep()
        solver.step(dt)
        dts.append(dt)
        if (solver.iteration - 1) % 2 == 0:
            speeds.append(flow.max('speed'))
    return solver, np.array(dts), np.array(speeds), {k: np.array(f[k]['c']) for k in ("p", "b", "u")}


def _shell_basis_of(field):
    b = getattr(field, "basis", None)
    if b is not None:
        return b
    return field.domain.bases[0]
