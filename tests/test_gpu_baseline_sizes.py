"""GPU known-answer tests at BASELINE.json configuration sizes.

The expected numbers are the reference's own end-state checksums measured during the survey by
running the unmodified reference:
  KdV-Burgers N=1024, SBDF2, dt=2e-3, 200 steps      sum(u_g^2)  = 1.017795766558e+02
  2-D RB 512x256, RK222, dt=1e-3, 13 steps           ||b_c||_2   = 1.085400858051743e+00
  2-D RB 256x64,  RK222, dt=1e-3, 23 steps           ||b_c||_2   = 1.085400744860160e+00
  3-D RB 32x32x32, RK222, dt=1e-3, 5 steps           ||b_c||_2   = 1.085405276538733e+00
plus size-independent properties at the full 3-D benchmark size (run by bench.py, not here)."""
import numpy as np
import pytest

import problems

pytestmark = pytest.mark.gpu


def test_kdv_1024_known_answer():
    import dedalus_amd.public as d3
    solver, f = problems.kdv_burgers(d3, Nx=1024, timestepper="SBDF2")
    for _ in range(200):
        solver.step(2e-3)
    f["u"].change_scales(3 / 2)          # the reference's state fields sit at the dealias scales after a step
    val = float(np.sum(np.asarray(f["u"]["g"]) ** 2))
    assert abs(val - 1.017795766558e+02) / 1.017795766558e+02 < 1e-10, val


@pytest.mark.parametrize("Nx,Nz,steps,expect", [(512, 256, 13, 1.085400858051743e+00),
                                                (256, 64, 23, 1.085400744860160e+00)])
def test_rb2d_known_answer(Nx, Nz, steps, expect):
    import dedalus_amd.public as d3
    solver, f = problems.rayleigh_benard_2d(d3, Nx=Nx, Nz=Nz, timestepper="RK222")
    for _ in range(steps):
        solver.step(1e-3)
    val = float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2)))
    assert abs(val - expect) / expect < 1e-12, val


def test_rb3d_32_known_answer():
    import dedalus_amd.public as d3
    solver, f = problems.rayleigh_benard_3d(d3, Nx=32, Ny=32, Nz=32, timestepper="RK222")
    for _ in range(5):
        solver.step(1e-3)
    val = float(np.sqrt(np.sum(np.asarray(f["b"]["c"]) ** 2)))
    assert abs(val - 1.085405276538733e+00) / 1.085405276538733e+00 < 1e-12, val


def test_rb3d_properties_medium_size():
    """Size-independent properties on a larger 3-D run: boundary conditions hold, the k=0 msin parts
    stay zero (valid-mode structure), divergence-free to solver precision."""
    import dedalus_amd.public as d3
    solver, f = problems.rayleigh_benard_3d(d3, Nx=64, Ny=64, Nz=32, timestepper="RK222")
    for _ in range(3):
        solver.step(1e-3)
    u, b = f["u"], f["b"]
    uc = np.asarray(u["c"])
    # msin parts of k = 0 are not stored modes: they stay at round-off of round-off (the two conjugate
    # systems of a ky = 0 pencil are solved separately; |u| ~ 1e-6 here)
    assert np.abs(uc[:, 1, :, :]).max() < 1e-20, np.abs(uc[:, 1, :, :]).max()
    assert np.abs(uc[:, :, 1, :]).max() < 1e-20, np.abs(uc[:, :, 1, :]).max()
    div = d3.div(u).evaluate()
    assert np.abs(np.asarray(div["c"])[..., :-2]).max() < 1e-10      # tau terms live in the last modes
    top = b(z=1).evaluate()
    bot = b(z=0).evaluate()
    assert np.abs(np.asarray(top["c"])).max() < 1e-12
    bc = np.asarray(bot["c"]).ravel()
    assert abs(bc[0] - 1.0) < 1e-12 and np.abs(bc[1:]).max() < 1e-12


def _projection_is_idempotent(d3, dist, coords, basis, rank, seed):
    """grid -> coefficients -> grid -> coefficients at the dealiased grid: the second coefficient set equals the first
    (the transform pair is a projection; no reference data needed, any size)."""
    f = dist.VectorField(coords, bases=basis) if rank == 1 else dist.Field(bases=basis)
    f.change_scales(3 / 2)
    rng = np.random.default_rng(seed)
    f["g"] = rng.standard_normal(np.asarray(f["g"]).shape)
    c1 = np.array(f["c"])
    f.change_scales(3 / 2)
    g1 = np.array(f["g"])
    f["g"] = g1
    c2 = np.array(f["c"])
    assert np.isfinite(c2).all()
    err = np.linalg.norm(c2 - c1) / np.linalg.norm(c1)
    assert err < 1e-12, err
    # and the grid data are reproduced by their own coefficients
    f.change_scales(3 / 2)
    g2 = np.array(f["g"])
    assert np.linalg.norm(g2 - g1) / np.linalg.norm(g1) < 1e-12


@pytest.mark.parametrize("rank", [0, 1])
def test_sphere_full_size_transform_projection(rank):
    """BASELINE config 4 size: SphereBasis(512, 256), Lmax = 254, dealias 3/2 (768 x 384 grid)."""
    import dedalus_amd.public as d3
    coords = d3.S2Coordinates('phi', 'theta')
    dist = d3.Distributor(coords, dtype=np.float64)
    basis = d3.SphereBasis(coords, (512, 256), radius=1.0, dealias=3 / 2, dtype=np.float64)
    _projection_is_idempotent(d3, dist, coords, basis, rank, 11 + rank)


@pytest.mark.parametrize("rank", [0])
def test_shell_full_size_transform_projection(rank):
    """BASELINE config 5 size: ShellBasis(256, 128, 128), Lmax = 126, dealias 3/2 (384 x 192 x 192 grid).  Scalars only:
    for tensors the reference's overlapping ell_maps boxes (reproduced here) make the pair
    forward / backward a non-projection on the few doubly covered slots -- in the reference as well."""
    import dedalus_amd.public as d3
    coords = d3.SphericalCoordinates('phi', 'theta', 'r')
    dist = d3.Distributor(coords, dtype=np.float64)
    basis = d3.ShellBasis(coords, shape=(256, 128, 128), radii=(14, 15), dealias=3 / 2, dtype=np.float64)
    _projection_is_idempotent(d3, dist, coords, basis, rank, 21 + rank)


def _gold_cartesian():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_cartesian.npz"))


def test_kdv_1024_every_mode_against_the_reference():
    """BASELINE config 1 at its size: the end state of the unmodified reference (oracle/make_golden_config.py), every
    coefficient and every grid value -- not one checksum."""
    import dedalus_amd.public as d3
    G = _gold_cartesian()
    solver, f = problems.kdv_burgers(d3, Nx=1024, timestepper="SBDF2")
    for _ in range(200):
        solver.step(2e-3)
    c = np.asarray(f["u"]["c"])
    assert np.linalg.norm(c - G["kdv1024__u_c"]) < 1e-10 * np.linalg.norm(G["kdv1024__u_c"])
    assert np.abs(c - G["kdv1024__u_c"]).max() < 1e-10 * np.abs(G["kdv1024__u_c"]).max()
    f["u"].change_scales(3 / 2)
    g = np.asarray(f["u"]["g"])
    assert np.abs(g - G["kdv1024__u_g"]).max() < 1e-10 * np.abs(G["kdv1024__u_g"]).max()


def test_rb2d_512x256_every_mode_against_the_reference():
    """BASELINE config 2: every mode of b, p and u after 13 RK222 steps against the unmodified reference's arrays
    (tolerances as in tests/test_gpu_ivp.py: u is O(1e-6) of b here and carries the solve's absolute error)."""
    import dedalus_amd.public as d3
    G = _gold_cartesian()
    solver, f = problems.rayleigh_benard_2d(d3, Nx=512, Nz=256, timestepper="RK222")
    for _ in range(13):
        solver.step(1e-3)
    for k, tol in (("b", 1e-12), ("p", 1e-9), ("u", 1e-8)):
        a, ref = np.asarray(f[k]["c"]), G["rb2d_512x256__" + k]
        assert a.shape == ref.shape
        err = np.linalg.norm(a - ref) / np.linalg.norm(ref)
        worst = np.abs(a - ref).max() / np.abs(ref).max()
        assert err < tol and worst < 10 * tol, (k, err, worst)


@pytest.mark.parametrize("shape,tol", [((24, 20, 18), 1e-12), ((512, 512, 256), 1e-11)])
def test_explicit_half_against_the_reference_at_full_size(shape, tol):
    """The explicit half of a stage -- backward z / x transforms, fused y stage with derivatives at load, forward
    transforms writing the equation rows of F -- at the METRIC's size against the unmodified reference: the reference's F
    for a band-limited state (tests/golden/config_explicit.npz, 16^3 modes) holds the coefficients of that state at any
    resolution; every other mode of the 512 x 512 x 256 result must be zero (tests/explicit_check.py)."""
    import dedalus_amd.public as d3
    import explicit_check
    solver, f = problems.rayleigh_benard_3d(d3, Nx=shape[0], Ny=shape[1], Nz=shape[2], timestepper="RK222")
    assert solver.ex.name == "hip"
    worst = explicit_check.check(solver, f, tol=tol)
    if shape[0] == 512:
        assert solver.F_direct is not None      # the fused path with direct F writes is what ran
    print("explicit half %s: max error / max |F| = %.2e" % (shape, worst))


def test_shell_explicit_half_against_the_reference_at_config_size():
    """BASELINE config 5 at its size, ShellBasis(256, 128, 128): the explicit half of a convection step -- backward radial /
    colatitude / azimuth transforms of u and b, the operator term lists of grad, the grid products u.grad(b), u.grad(u),
    forward transforms into the equations' bases -- against the table the UNMODIFIED reference produces for the same
    band-limited functions at a small shell (tests/golden/config_shell_explicit.npz, oracle/make_golden_config.py):
    every populated (m, ell, n) mode equal, every other mode empty."""
    import dedalus_amd.public as d3
    import explicit_check
    worst = explicit_check.check_shell(d3, (256, 128, 128), tol=1e-11)
    print("shell explicit half at 256 x 128 x 128: max error", worst)
    assert worst < 1e-11


def test_full_spectrum_transforms_at_the_metric_size_against_the_series():
    """512 x 512 x 256 with EVERY mode populated (the band-limited explicit-half check fills 16^3 of them): the backward
    transforms (z Chebyshev, x and y real Fourier, 3/2 padding) of seeded random coefficients, compared at sampled grid
    points with the direct summation of the series -- the reference's matrix definitions of the transforms
    (core/transforms.py:114-158, 387-424, restated in oracle/np_transforms.py and pinned to the reference's goldens) --
    and the forward transforms through the round trip of the same full-spectrum data.  A vector field exercises the
    dual / multi-component launches."""
    import dedalus_amd.public as d3
    from oracle import np_transforms as T
    N = (512, 512, 256)
    coords = d3.CartesianCoordinates('x', 'y', 'z')
    dist = d3.Distributor(coords, dtype=np.float64)
    bases = (d3.RealFourier(coords['x'], size=N[0], bounds=(0, 4), dealias=3 / 2),
             d3.RealFourier(coords['y'], size=N[1], bounds=(0, 4), dealias=3 / 2),
             d3.ChebyshevT(coords['z'], size=N[2], bounds=(0, 1), dealias=3 / 2))
    u = dist.VectorField(coords, name='u', bases=bases)
    rng = np.random.default_rng(20)
    c = rng.standard_normal((3,) + N)
    c[:, 1, :, :] = 0.0                    # the msin parts of k = 0 are not modes of a real field
    c[:, :, 1, :] = 0.0
    u["c"] = c
    u.change_scales(3 / 2)
    g = np.array(u["g"])                   # [3][768][768][384]
    assert g.shape == (3, 768, 768, 384)
    Bx = T.real_fourier_mmt_matrices(768, 512)[1]
    By = T.real_fourier_mmt_matrices(768, 512)[1]
    Bz = T.chebyshev_mmt_matrices(384, 256)[1]
    scale = np.abs(g).max()
    worst = 0.0
    for comp in range(3):
        for k in rng.choice(384, size=3, replace=False):
            plane = c[comp] @ Bz[k]                                   # [512][512]: the series summed along z
            for i, j in zip(rng.choice(768, size=6), rng.choice(768, size=6)):
                want = Bx[i] @ plane @ By[j]
                worst = max(worst, abs(g[comp, i, j, k] - want) / scale)
    print("full-spectrum backward transforms at 512 x 512 x 256 vs the series: max error", worst)
    assert worst < 1e-12, worst
    # forward: grid -> coefficients returns the full spectrum
    u["g"] = g
    c2 = np.array(u["c"])
    err = np.abs(c2 - c).max() / np.abs(c).max()
    print("round trip of the full spectrum:", err)
    assert err < 1e-12, err


def _wave_launches():
    import ctypes as C
    from dedalus_amd import libhip
    n = C.c_long(0)
    libhip.call("ddh_fft_wave_launches", C.byref(n))
    return n.value


@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 8, 256), (16, 512, 256), (512, 16, 256)])
def test_rb3d_end_state_through_the_headline_kernels(shape, monkeypatch):
    """End states of the UNMODIFIED reference (oracle/make_golden_config.py::config_rb3d_endstate: three RK222 steps of the
    benchmark script; strided samples of the ARRAYS, not norms) at sizes whose steps run through the kernels of the
    headline configuration, composed:
      128 x 128 x 64   partner pencils (x <-> y symmetric), per-thread lean sweeps, real FFT 192 <- 128 on the wave kernel
      256 x 8 x 256    Chebyshev 384 <- 256 wave kernels (dual / conversion / forward writing F tile-major), real FFT
                       384 <- 256 wave kernels, x-blocked stage arrays, per-thread lean sweeps on tile-major terms
      16 x 512 x 256   the second-generation fused y stage (768-point lines, LDS-DMA operands) + the same z kernels
      512 x 16 x 256   real FFT 768 <- 512 wave kernels (plain, dual, forward) + x-blocked stage + the same z kernels
    The per-thread sweeps are forced (`set_solve_variant(0)`: these sizes have few pencils and would take the cooperative
    sweeps) and the tile-major right-hand sides and state vector are allowed below their size threshold, so that what runs
    is what the 512 x 512 x 256 benchmark runs.  Tolerances: rel-L2 1e-10 (b, p), 1e-9 (u) on the sampled arrays."""
    import os
    import dedalus_amd.public as d3
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_rb3d_endstate_%dx%dx%d.npz" % shape)
    if not os.path.exists(path):
        pytest.skip("golden not generated: " + os.path.basename(path))
    G = np.load(path)
    monkeypatch.setenv("DDH_RHS_TILING_MIN", "0")
    monkeypatch.setenv("DDH_PAIR_MIN", "0")              # partner pencils below their default size threshold (65 536 systems)
    monkeypatch.setenv("DDH_X_TILED_MIN", "0")           # the state vector tile-major, as at the benchmark's size
    Nx, Ny, Nz = shape
    solver, f = problems.rayleigh_benard_3d(d3, Nx=Nx, Ny=Ny, Nz=Nz, timestepper="RK222")
    assert solver.ex.name == "hip"
    assert solver.x_tiled == (Ny if Nz == 256 else 0)    # (tile-major where the backward z transforms read it in place: 384 <- 256)
    solver.pack.set_solve_variant(0)
    w0 = _wave_launches()
    for _ in range(int(G["steps"])):
        solver.step(float(G["dt"]))
    nwave = _wave_launches() - w0
    lus = sorted(solver._lu_params)
    infos = [solver.pack.lu_info(lu) for lu in lus]
    assert all(i["forward"] == "lean" and i["real"] for i in infos), infos
    if shape != (128, 128, 64):
        assert solver.rhs_tiling(lus) == Ny              # tile-major M.X / F buffers
    assert nwave > 0
    assert bool(infos[0]["pair"]) == (Nx == Ny)          # x <-> y symmetric problems share factorizations between (kx, ky) and (ky, kx)
    sa = G["sample_a"] if "sample_a" in G.files else np.array([0, 4, 0, 4])
    sb = G["sample_b"] if "sample_b" in G.files else np.array([1, 8, 3, 8])
    worst = {}
    for k, tol in (("b", 1e-10), ("p", 1e-10), ("u", 1e-9)):
        c = np.asarray(f[k]["c"])
        assert abs(np.linalg.norm(c) - float(G["end__%s_norm" % k])) <= tol * float(G["end__%s_norm" % k])
        for tag, s in (("a", sa), ("b", sb)):
            mine = c[..., int(s[0])::int(s[1]), int(s[2])::int(s[3]), :]
            ref = G["end__%s_%s" % (k, tag)]
            assert mine.shape == ref.shape, (k, tag, mine.shape, ref.shape)
            if ref.size == 0:
                continue
            err = np.linalg.norm(mine - ref) / np.linalg.norm(ref)
            worst[k] = max(worst.get(k, 0.0), err)
            assert err < tol, (shape, k, tag, err)
    print("rb3d %s end state vs the reference: %s, %d wave-kernel transforms, pairing %s"
          % (shape, {k: "%.1e" % v for k, v in worst.items()}, nwave, infos[0].get("pair")))
