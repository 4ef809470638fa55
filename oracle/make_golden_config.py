"""
TEST INFRASTRUCTURE ONLY.  Config-size goldens of the curvilinear BASELINE configurations, made by the UNMODIFIED
reference through oracle/refshim (run in the build container; the GPU box has no reference):

  S  ivp_sphere_shallow_water at SphereBasis(512, 256): the reference's own per-m subproblem matrices M_min / L_min
     (core/subsystems.py:497-596) for sampled m, the order of their unknowns / equations, and the end state of two
     RK222 steps of the example (sub-sampled arrays + norms).
  H  ivp_shell_convection at ShellBasis(256, 128, 128): the reference's per-(m, ell) subproblem matrices for sampled
     ell.  Only those subproblems are BUILT (the reference's build_matrices is called with the sampled list instead of
     all ~8000 subproblems: building every one takes the reference more than half an hour); the matrix code itself is
     the reference's.

The order of a subproblem's unknowns is recorded with tags: every variable's coefficient array is filled with its own
flat indices, `Subproblem.gather_inputs` (core/subsystems.py:340-349) then returns, per unknown, which entry of which
field it is (pre_right_pinv is a selection); likewise `gather_outputs` on the F fields for the equations.  The tests
(tests/test_config_matrices.py, tests/test_gpu_sphere.py, tests/test_gpu_shell.py) tag this package's fields the same
way, so no knowledge of either layout is needed to line the two up.

    python oracle/make_golden_config.py [sphere] [shell]      ->  tests/golden/config_sphere.npz, config_shell.npz
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import refshim  # noqa: E402

TAG = 2 ** 40


def _tag(fields):
    """fill every field's coefficient array with TAG * index + flat position; returns the saved data"""
    saved = []
    for i, f in enumerate(fields):
        c = np.array(f['c'])
        saved.append(c)
        f['c'] = (i * TAG + np.arange(c.size, dtype=np.float64)).reshape(c.shape)
    return saved


def _untag(fields, saved):
    for f, c in zip(fields, saved):
        f['c'] = c


def _ids(vec):
    """(unknowns, subsystems) tags -> field index and flat coefficient position, same shape"""
    v = np.asarray(vec).real
    v = v.reshape(v.shape[0], -1)
    ids = np.rint(v).astype(np.int64)
    assert np.all(ids == v), "a subproblem entry is not a pure selection of one coefficient"
    return (ids // TAG).astype(np.int32), (ids % TAG).astype(np.int64)


def _csr(out, tag, m):
    m = m.tocsr()
    m.sort_indices()
    out[tag + "_indptr"] = m.indptr.astype(np.int32)
    out[tag + "_indices"] = m.indices.astype(np.int32)
    out[tag + "_data"] = np.asarray(m.data)
    out[tag + "_shape"] = np.array(m.shape)


def _subproblem(out, tag, solver, sp):
    for name in ("M_min", "L_min"):
        _csr(out, tag + name, getattr(sp, name))
    state, F = list(solver.state), list(solver.F)
    s1 = _tag(state)
    iv, ix = _ids(sp.gather_inputs(state).copy())
    _untag(state, s1)
    s2 = _tag(F)
    ov, ox = _ids(sp.gather_outputs(F).copy())
    _untag(F, s2)
    out[tag + "in_var"], out[tag + "in_flat"] = iv, ix
    out[tag + "out_eq"], out[tag + "out_flat"] = ov, ox


def config_sphere(Nphi=512, Ntheta=256, ms=(0, 1, 2, 37, 128, 200, 253, 254), steps=2):
    import problems
    d3 = refshim.load_reference()
    t0 = time.time()
    solver, fields, extra = problems.shallow_water(d3, Nphi=Nphi, Ntheta=Ntheta, timestepper="RK222")
    print("reference sphere %d x %d built in %.1f s, %d subproblems" % (Nphi, Ntheta, time.time() - t0, len(solver.subproblems)))
    out = {"shape": np.array([Nphi, Ntheta]), "ms": np.array(ms), "steps": np.array(steps),
           "variables": np.array([v.name for v in solver.problem.variables])}
    out["coeff_shapes"] = np.array([len(np.array(f['c']).shape) for f in solver.state])
    for i, f in enumerate(solver.state):
        out["cshape_%d" % i] = np.array(np.array(f['c']).shape)
    for i, f in enumerate(solver.F):
        out["fshape_%d" % i] = np.array(np.array(f['c']).shape)
    by_m = {int(sp.group[0]): sp for sp in solver.subproblems}
    for m in ms:
        _subproblem(out, "m%d__" % m, solver, by_m[m])
    out["h_balanced_norm"] = np.array(np.linalg.norm(extra["h_balanced"]))
    out["h_balanced_sub"] = np.array(extra["h_balanced"])[::16, ::8]
    t0 = time.time()
    for _ in range(steps):
        solver.step(extra["timestep"])
    print("%d reference steps in %.1f s" % (steps, time.time() - t0))
    for k, f in fields.items():
        f.change_scales(1)
        c = np.array(f['c'])
        out["end__%s_norm" % k] = np.array(np.linalg.norm(c))
        out["end__%s_sub" % k] = c[..., ::4, :]                  # every 4th row of the packed azimuthal axis, all ell
        print(k, c.shape, float(np.linalg.norm(c)))
    path = os.path.join(GOLD, "config_sphere.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def config_shell(shape=(256, 128, 128), ells=(0, 1, 2, 50, 126)):
    import problems
    d3 = refshim.load_reference()
    import dedalus.core.solvers as rsolvers
    # build only the sampled subproblems' matrices: the reference's build_matrices with a shorter list
    wanted = {}
    orig = rsolvers.SolverBase.build_matrices

    def build_some(self, subproblems=None, matrices=None):
        subs = self.subproblems if subproblems is None else subproblems
        pick = []
        for sp in subs:
            m, ell = sp.group[0], sp.group[1]
            if ell in ells and (ell not in wanted):
                wanted[ell] = sp
                pick.append(sp)
        print("building %d of %d subproblems: groups %s" % (len(pick), len(subs), [sp.group for sp in pick]), flush=True)
        from scipy import sparse
        for sp in subs:
            if sp not in pick:
                sp.pre_left = sparse.csr_matrix((1, 1))      # only its shape is read (the solver's mode count)
        return orig(self, pick, matrices)

    # no step is taken: a do-nothing timestepper keeps the solver constructor from sizing its work arrays on the
    # subproblems that were not built
    d3._NoStepper = type("_NoStepper", (), {"__init__": lambda self, solver: None, "steps": 1, "stages": 1})
    rsolvers.SolverBase.build_matrices = build_some
    try:
        t0 = time.time()
        solver, f = problems.shell_convection(d3, shape=shape, timestepper="_NoStepper")
        print("reference shell %s: solver object in %.1f s" % (shape, time.time() - t0), flush=True)
    finally:
        rsolvers.SolverBase.build_matrices = orig
    out = {"shape": np.array(shape), "ells": np.array(ells),
           "variables": np.array([v.name for v in solver.problem.variables])}
    for i, fld in enumerate(solver.state):
        out["cshape_%d" % i] = np.array(np.array(fld['c']).shape)
    for i, fld in enumerate(solver.F):
        out["fshape_%d" % i] = np.array(np.array(fld['c']).shape)
    for ell in ells:
        sp = wanted[ell]
        out["ell%d__group" % ell] = np.array([-1 if g is None else g for g in sp.group])
        _subproblem(out, "ell%d__" % ell, solver, sp)
        print("ell", ell, "group", sp.group, "size", sp.M_min.shape, flush=True)
    path = os.path.join(GOLD, "config_shell.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def config_shell_endstate(shape=(256, 128, 128), steps=2, dt=0.05):
    """H end state at configuration size: the UNMODIFIED reference builds all subproblems of ShellBasis(256, 128, 128)
    (hours of host time: that is why config_shell only samples matrices) and takes `steps` SBDF2 steps of the example
    from its seeded initial condition; stored: norms and sub-sampled coefficient arrays of every state field."""
    import problems
    d3 = refshim.load_reference()
    t0 = time.time()
    solver, fields = problems.shell_convection(d3, shape=shape, timestepper="SBDF2")
    t_build = time.time() - t0
    print("reference shell %s built in %.1f s, %d subproblems" % (shape, t_build, len(solver.subproblems)), flush=True)
    out = {"shape": np.array(shape), "steps": np.array(steps), "dt": np.array(dt), "build_seconds": np.array(t_build),
           "sub_stride": np.array(32)}
    t0 = time.time()
    for i in range(steps):
        solver.step(dt)
        print("step", i, "done after %.1f s" % (time.time() - t0), flush=True)
    out["step_seconds"] = np.array(time.time() - t0)
    for k, f in fields.items():
        f.change_scales(1)
        c = np.array(f['c'])
        out["end__%s_norm" % k] = np.array(np.linalg.norm(c))
        if c.ndim >= 3 and c.shape[-3] >= 32:
            c = c[..., ::32, :, :]                      # every 32nd row of the packed azimuthal axis, all ell, all n
        out["end__%s_sub" % k] = c
        print(k, np.array(f['c']).shape, float(out["end__%s_norm" % k]), flush=True)
    path = os.path.join(GOLD, "config_shell_endstate.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def config_cartesian():
    """K (KdV-Burgers N = 1024, SBDF2, 200 steps of 2e-3) and R2 (2-D Rayleigh-Benard 512 x 256, RK222, 13 steps of 1e-3)
    end states of the unmodified reference as ARRAYS: every mode of K's u and of R2's b, p and u."""
    import problems
    d3 = refshim.load_reference()
    out = {}
    solver, f = problems.kdv_burgers(d3, Nx=1024, timestepper="SBDF2")
    for _ in range(200):
        solver.step(2e-3)
    out["kdv1024__u_c"] = np.array(f["u"]["c"])
    f["u"].change_scales(3 / 2)
    out["kdv1024__u_g"] = np.array(f["u"]["g"])
    print("kdv sum u_g^2 =", repr(float(np.sum(out["kdv1024__u_g"] ** 2))))
    solver, f = problems.rayleigh_benard_2d(d3, Nx=512, Nz=256, timestepper="RK222")
    for _ in range(13):
        solver.step(1e-3)
    for k in ("p", "b", "u"):
        out["rb2d_512x256__" + k] = np.array(f[k]["c"])
    print("rb2d |b_c| =", repr(float(np.linalg.norm(out["rb2d_512x256__b"]))))
    path = os.path.join(GOLD, "config_cartesian.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def band_limited_state(fields, grids, Lx=4.0, Ly=4.0, Lz=1.0):
    """A state of few modes, the same FUNCTION at any resolution (used by config_explicit and by the GPU test at the
    metric's size): products of such fields populate a handful of low modes whose coefficients do not depend on the
    number of modes carried."""
    x, y, z = grids
    kx, ky = 2 * np.pi / Lx, 2 * np.pi / Ly
    w = z * (Lz - z)
    b = (Lz - z) + 0.3 * np.sin(kx * x) * np.cos(2 * ky * y) * w + 0.1 * np.cos(2 * kx * x) * z ** 2 + 0 * y
    ux = 0.5 * np.sin(kx * x) * np.cos(ky * y) * w
    uy = -0.4 * np.cos(2 * kx * x) * np.sin(ky * y) * z * w
    uz = 0.25 * np.cos(kx * x) * np.cos(ky * y) * w ** 2
    fields["b"]["g"] = b
    u = fields["u"]
    ug = np.zeros((3,) + b.shape)
    ug[0], ug[1], ug[2] = ux, uy, uz
    u["g"] = ug


def config_explicit(N=16):
    """The explicit half of a 3-D Rayleigh-Benard stage -- F = (0, -u.grad(b), -u.grad(u), boundary constants) in the
    equations' bases -- of the unmodified reference for the band-limited state above at N^3 modes.  The coefficients are
    those of the same state at ANY resolution (zero beyond the populated modes): tests/test_gpu_baseline_sizes.py
    compares the 512 x 512 x 256 evaluation of this package (z / x transforms, fused y stage, direct F writes) with them."""
    import problems
    d3 = refshim.load_reference()
    solver, f = problems.rayleigh_benard_3d(d3, Nx=N, Ny=N, Nz=N, timestepper="RK222")
    dist = solver.dist
    grids = dist.local_grids(*f["b"].domain.bases)
    band_limited_state(f, grids)
    solver.evaluator.evaluate_group("F", iteration=0, wall_time=0.0, sim_time=0.0, timestep=1e-3)
    out = {"N": np.array(N)}
    for i, F in enumerate(solver.F):
        c = np.array(F["c"])
        out["F%d" % i] = c
        print("equation", i, c.shape, float(np.abs(c).max()), int(np.sum(np.abs(c) > 1e-14)), "modes above 1e-14")
    path = os.path.join(GOLD, "config_explicit.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def config_shell_explicit(shape=(32, 16, 16)):
    """The explicit half of a shell-convection step -- F_b = -u.grad(b), F_u = -u.grad(u) in the equations' bases,
    evaluated by the unmodified reference's solver (evaluate_group("F"), core/solvers.py:683-711 -> operators.py:3136-3175,
    basis.py:4474-4508) -- for the band-limited state of tests/problems.py::shell_band_limited_state at a SMALL shell.
    The table {(equation, component, m, part, ell, n): value}, labelled with the reference's own group arrays, holds the
    coefficients of the same functions at ANY resolution: tests/explicit_check.py::check_shell compares this package's
    evaluation at ShellBasis(256, 128, 128) with it (every other mode must vanish)."""
    import problems
    d3 = refshim.load_reference()

    def labels(F):
        g = F.dist.coeff_layout.local_group_arrays(F.domain, scales=1)
        return tuple(np.ma.filled(a, -1) for a in g)

    tab = problems.shell_explicit_results(d3, shape, labels)
    keys = np.array(sorted(tab), dtype=np.int64)
    vals = np.array([tab[tuple(k)] for k in keys])
    print("shell explicit half at", shape, ":", len(keys), "modes; max ell", int(keys[:, 4].max()), "max n", int(keys[:, 5].max()))
    path = os.path.join(GOLD, "config_shell_explicit.npz")
    np.savez_compressed(path, shape=np.array(shape), keys=keys, values=vals)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def config_rb3d_endstate(shape, steps=3, dt=1e-3):
    """3-D Rayleigh-Benard end state of the UNMODIFIED reference at a size whose transforms and solves take the
    product's headline code paths (wave transforms, per-thread lean sweeps, tile-major right-hand sides): `steps` RK222
    steps of the benchmark script from its seeded initial condition.  Stored: the norm of every state array and two
    strided samples of the ARRAYS themselves (at most 24 rows per Fourier axis starting at 0 -- which keeps the
    k = 0 rows -- and a second, coarser set starting at (1, 3): `sample_a`, `sample_b` = start, stride per axis), all z modes.  128 x 128 x 64 takes the reference ~9 min to build
    and ~8 s per step on one core."""
    import problems
    d3 = refshim.load_reference()
    Nx, Ny, Nz = shape
    t0 = time.time()
    solver, f = problems.rayleigh_benard_3d(d3, Nx=Nx, Ny=Ny, Nz=Nz, timestepper="RK222")
    t_build = time.time() - t0
    out = {"shape": np.array(shape), "steps": np.array(steps), "dt": np.array(dt)}
    t0 = time.time()
    for i in range(steps):
        solver.step(dt)
        if i == 0:
            out["first_step_seconds"] = np.array(time.time() - t0)
        print("step", i, "done after %.1f s" % (time.time() - t0), flush=True)
    out["build_seconds"] = np.array(t_build)
    out["step_seconds"] = np.array(time.time() - t0)
    sx, sy = max(1, -(-Nx // 24)), max(1, -(-Ny // 24))     # at most 24 samples per Fourier axis, all z modes
    out["sample_a"] = np.array([0, sx, 0, sy])
    out["sample_b"] = np.array([1, 2 * sx, min(3, Ny - 1), 2 * sy])
    for k in ("p", "b", "u"):
        f[k].change_scales(1)
        c = np.array(f[k]["c"])
        out["end__%s_norm" % k] = np.array(np.linalg.norm(c))
        out["end__%s_max" % k] = np.array(np.abs(c).max())
        out["end__%s_a" % k] = c[..., ::sx, ::sy, :]
        out["end__%s_b" % k] = c[..., 1::2 * sx, min(3, Ny - 1)::2 * sy, :]
        print(k, c.shape, repr(float(out["end__%s_norm" % k])), flush=True)
    path = os.path.join(GOLD, "config_rb3d_endstate_%dx%dx%d.npz" % shape)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB, build %.0f s, steps %.0f s" % (t_build, float(out["step_seconds"])))


if __name__ == "__main__":
    which = sys.argv[1:] or ["sphere", "shell", "cartesian", "explicit", "shell_explicit"]
    if "shell_explicit" in which:
        config_shell_explicit()
    if "cartesian" in which:
        config_cartesian()
    if "sphere" in which:
        config_sphere()
    if "shell" in which:
        config_shell()
    if "explicit" in which:
        config_explicit()
    if "shell_endstate" in which:
        config_shell_endstate()
    for w in which:
        if w.startswith("rb3d_"):                      # e.g. rb3d_128x128x64
            config_rb3d_endstate(tuple(int(v) for v in w[5:].split("x")))
