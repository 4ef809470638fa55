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


def rayleigh_benard_3d(d3, Nx=8, Ny=12, Nz=8, timestepper="RK222", dist_kw=None):
    Lx, Ly, Lz = 4, 4, 1
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


IVP_CASES = {
    # name: (builder, kwargs, timestep, number of steps)
    "kdv64_sbdf2": (kdv_burgers, dict(Nx=64, timestepper="SBDF2"), 2e-3, 20),
    "kdv64_rk443": (kdv_burgers, dict(Nx=64, timestepper="RK443"), 2e-3, 8),
    "rb2d_32x16_rk222": (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper="RK222"), 1e-3, 6),
    "rb2d_32x16_sbdf2": (rayleigh_benard_2d, dict(Nx=32, Nz=16, timestepper="SBDF2"), 1e-3, 6),
    "rb2d_64x32_rk222": (rayleigh_benard_2d, dict(Nx=64, Nz=32, timestepper="RK222"), 1e-3, 10),
    "rb3d_8x12x8_rk222": (rayleigh_benard_3d, dict(Nx=8, Ny=12, Nz=8, timestepper="RK222"), 1e-3, 4),
    "rb3d_16x16x16_rk222": (rayleigh_benard_3d, dict(Nx=16, Ny=16, Nz=16, timestepper="RK222"), 1e-3, 3),
}


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
