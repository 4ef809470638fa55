"""GPU parity of the whole IMEX hot path: the same problem scripts the reference ran
(tests/problems.py), stepped by the HIP path, against the reference's own end states
(tests/golden/ivp.npz).  float64; tolerance: rel-L2 <= 1e-10 on the dynamical fields after the
fixed number of steps, looser on the tau amplitudes (tiny, ill-conditioned)."""
import os

import numpy as np
import pytest

import problems

pytestmark = pytest.mark.gpu

TOL = {"p": 1e-10, "b": 1e-10, "u": 1e-9, "tau_b1": 1e-5, "tau_b2": 1e-5, "tau_u1": 1e-8, "tau_u2": 1e-8}


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ivp.npz"))


@pytest.mark.parametrize("name", list(problems.IVP_CASES))
def test_hip_path_matches_reference(gold, name):
    import dedalus_amd.public as d3
    solver, res = problems.run_case(d3, name)
    assert solver.ex.name == "hip"
    for k, v in res.items():
        ref = gold[name + "__" + k]
        assert np.isfinite(v).all()
        assert rel(v, ref) < TOL.get(k, 1e-10), (name, k, rel(v, ref))


def test_hip_vs_oracle_executor_larger_3d():
    """A size the reference fixtures do not cover, against the CPU oracle on identical inputs."""
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    kw = dict(Nx=16, Ny=24, Nz=16, timestepper="RK222")
    s1, f1 = problems.rayleigh_benard_3d(d3, **kw)
    s2, f2 = problems.rayleigh_benard_3d(d3, dist_kw=dict(executor=NumpyExecutor()), **kw)
    for _ in range(3):
        s1.step(1e-3)
        s2.step(1e-3)
    for k in ("p", "b", "u"):
        assert rel(np.array(f1[k]['c']), np.array(f2[k]['c'])) < TOL[k], k


def test_refactorization_on_dt_change_and_grid_access():
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    s1, f1 = problems.rayleigh_benard_2d(d3, Nx=32, Nz=16)
    s2, f2 = problems.rayleigh_benard_2d(d3, Nx=32, Nz=16, dist_kw=dict(executor=NumpyExecutor()))
    for dt in (1e-3, 1e-3, 2e-3, 1.5e-3):
        s1.step(dt)
        s2.step(dt)
    g1 = np.array(f1["b"]["g"])
    g2 = np.array(f2["b"]["g"])
    assert rel(g1, g2) < 1e-10
    assert abs(s1.sim_time - 5.5e-3) < 1e-15


def test_cfl_on_device_matches_reference(gold):
    """ddh_grid_cfl + adaptive stepping + device refactorization vs the reference's dt sequence."""
    import dedalus_amd.public as d3
    solver, dts, res = problems.run_cfl_case(d3)
    assert solver.ex.name == "hip"
    assert np.allclose(dts, gold["cfl__dts"], rtol=1e-11, atol=0), np.max(np.abs(dts - gold["cfl__dts"]))
    for k in ("p", "b", "u"):
        assert rel(res[k], gold["cfl__" + k]) < 1e-9, (k, rel(res[k], gold["cfl__" + k]))


@pytest.mark.parametrize("shape", [(64, 32), (32, 24)])
def test_poisson_lbvp_on_device_matches_reference(gold, shape):
    import dedalus_amd.public as d3
    solver, fields = problems.poisson_2d(d3, Nx=shape[0], Ny=shape[1])
    assert solver.ex.name == "hip"
    for k, f in fields.items():
        ref = gold["poisson_%dx%d__%s" % (shape + (k,))]
        tol = 1e-10 if k in ("u", "f") else 1e-6          # the tau amplitudes are ~1e-20 (spectrally small residuals)
        assert rel(np.array(f['c']), ref) < tol, (k, rel(np.array(f['c']), ref))


def test_flow_properties_reduce_on_device():
    """GlobalFlowProperty min / max / grid_average through ddh_grid_reduce (extras/flow_tools.py:49-111) against NumPy
    on the downloaded grid, and the raw kernel on odd sizes."""
    import ctypes as C
    import dedalus_amd.public as d3
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    solver, f = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    for _ in range(3):
        solver.step(1e-3)
    u, b = f["u"], f["b"]
    flow = d3.GlobalFlowProperty(solver, cadence=1)
    flow.add_property(u @ u, name="uu")
    flow.add_property(b, name="b")
    g = np.asarray((u @ u).evaluate()["g"])
    assert abs(flow.max("uu") - g.max()) <= 1e-15 * abs(g.max())
    assert abs(flow.min("uu") - g.min()) <= 1e-15 * abs(g.max())
    assert abs(flow.grid_average("uu") - g.mean()) <= 1e-13 * abs(g.mean())
    gb = np.asarray(b["g"])
    # (b['g'] may re-transform the field: equal up to the round-off of a transform round trip, not bit for bit)
    assert abs(flow.grid_average("b") - gb.mean()) <= 1e-13
    assert abs(flow.max("b") - gb.max()) <= 4e-16 * abs(gb.max()) and abs(flow.min("b") - gb.min()) <= 1e-15
    dev = solver.ex.dev
    rng = np.random.default_rng(0)
    for n in (1, 63, 1025, 300007):
        x = rng.standard_normal(n)
        mn, mx, sm = solver.ex.reduce3(dev.from_host(x))
        assert mn == x.min() and mx == x.max() and abs(sm - x.sum()) <= 1e-12 * np.abs(x).sum()
    # a blown-up field must be visible: NaN propagates like np.min / np.max (fmin / fmax would drop it)
    x = rng.standard_normal(70001)
    x[12345] = np.nan
    mn, mx, sm = solver.ex.reduce3(dev.from_host(x))
    assert np.isnan(mn) and np.isnan(mx) and np.isnan(sm)


def test_file_output_and_restart_from_device_fields(tmp_path):
    """Analysis sets written from DEVICE fields through the asynchronous staging path (device snapshot + pinned host
    copy on a side stream, flushed at the next output / close), read back with h5lite; load_state restart on the GPU
    reproduces the uninterrupted run (core/evaluator.py:366-618, core/solvers.py:632-673)."""
    import dedalus_amd.public as d3
    from dedalus_amd.tools import h5lite
    solver, f = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    snap = solver.evaluator.add_file_handler(str(tmp_path / "snap"), iter=2, max_writes=10)
    snap.add_task(f["b"], name="b")
    snap.add_task(f["u"], layout="c", name="u_c")
    chk = solver.evaluator.add_file_handler(str(tmp_path / "chk"), iter=4, max_writes=10)
    chk.add_tasks(solver.state, layout="g")
    assert snap.async_staging
    seen = {}
    for i in range(9):
        if i % 2 == 0:
            f["b"]["c"]
            seen[i] = (np.array(f["b"]["g"]), np.array(f["u"]["c"]))
        solver.step(1e-3)
    end = {k: np.array(v["c"]) for k, v in f.items()}
    snap.close()
    chk.close()
    r = h5lite.read(str(tmp_path / "snap" / "snap_s1.h5"))
    assert np.array_equal(r["scales/iteration"].read(), [0, 2, 4, 6, 8])
    for k, it in enumerate((0, 2, 4, 6, 8)):
        assert np.allclose(r["tasks/b"].read(k), seen[it][0], rtol=0, atol=1e-13)
        assert np.allclose(r["tasks/u_c"].read(k), seen[it][1], rtol=0, atol=1e-13)
    # restart from the checkpoint written at iteration 4 and redo steps 4..8
    solver2, f2 = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    write, dt = solver2.load_state(str(tmp_path / "chk" / "chk_s1.h5"), index=1)
    assert solver2.iteration == 4 and abs(dt - 1e-3) < 1e-18
    for _ in range(5):
        solver2.step(1e-3)
    for k in ("p", "b", "u"):
        assert rel(np.array(f2[k]["c"]), end[k]) < 1e-9, k


def test_fixed_dt_step_graph_matches_ordinary_launches():
    """RK222 steps replayed from a captured HIP graph (core/ivp_common.py) == the same steps launched one by one;
    a timestep change and a host access of the state in between fall back and re-capture."""
    import dedalus_amd.public as d3
    a, fa = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    b, fb = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    a.enable_step_graph(False)              # (small problems replay by default)
    b.enable_step_graph(True)
    seq = [1e-3] * 8 + [2e-3] * 6
    for i, dt in enumerate(seq):
        a.step(dt)
        b.step(dt)
        if i == 10:
            assert rel(np.array(fb["b"]["c"]), np.array(fa["b"]["c"])) < 1e-13      # host access: state not clean next step
    assert b._graph["graphs"] and not b._graph["failed"]
    assert abs(b.sim_time - a.sim_time) < 1e-15 and b.iteration == a.iteration
    for k in ("p", "b", "u"):
        assert rel(np.array(fb[k]["c"]), np.array(fa[k]["c"])) < 1e-13, k


@pytest.fixture(scope="module")
def gold_schemes(golden_dir):
    return np.load(os.path.join(golden_dir, "ivp_schemes.npz"))


@pytest.mark.parametrize("name", list(problems.SCHEME_CASES))
def test_every_scheme_on_the_device_matches_reference(gold_schemes, name):
    """All 12 registered IMEX schemes stepped by the HIP path against the reference's own end states
    (tests/golden/ivp_schemes.npz <- oracle/make_golden.py::golden_schemes; reference: tests/test_ivp.py:20-49,
    core/timesteppers.py:190-725): forced heat equation, KdV-Burgers, tau-bordered 2-D Rayleigh-Benard with constant
    and with varying timesteps.  float64, rel-L2 (problems.SCHEME_TOL)."""
    import dedalus_amd.public as d3
    solver, res = problems.run_scheme_case(d3, name)
    assert solver.ex.name == "hip"
    assert abs(solver.sim_time - float(gold_schemes[name + "__sim_time"])) < 1e-15
    for k, v in res.items():
        assert np.isfinite(v).all()
        err = problems.scheme_error(k, v, gold_schemes[name + "__" + k])
        assert err < problems.SCHEME_TOL.get(k, 1e-10), (name, k, err)


@pytest.mark.parametrize("scheme", problems.ALL_SCHEMES)
def test_every_scheme_replayed_from_step_graphs(gold_schemes, scheme):
    """The same reference end states with the steps replayed from HIP graphs wherever the pattern allows (after the
    start-up orders, constant timestep), and bit-identity with ordinary launches over a longer run that visits every
    phase of the history rotation."""
    import dedalus_amd.public as d3
    for case in ("kdv64_", "rb2d_32x16_"):
        name = case + scheme
        solver, res = problems.run_scheme_case(d3, name, before_step=lambda s, i: s.enable_step_graph(True) if i == 0 else None)
        if scheme not in ("SBDF4",) or case == "kdv64_":
            assert solver._graph["graphs"] and not solver._graph["failed"], name
        for k, v in res.items():
            err = problems.scheme_error(k, v, gold_schemes[name + "__" + k])
            assert err < problems.SCHEME_TOL.get(k, 1e-10), (name, k, err)
    a, fa = problems.kdv_burgers(d3, Nx=128, timestepper=scheme)
    b, fb = problems.kdv_burgers(d3, Nx=128, timestepper=scheme)
    a.enable_step_graph(False)              # (small problems replay by default)
    b.enable_step_graph(True)
    for h in [2e-3] * 16 + [1e-3] * 12:
        a.step(h)
        b.step(h)
    assert b._graph["graphs"] and not b._graph["failed"]
    assert abs(b.sim_time - a.sim_time) < 1e-15 and b.iteration == a.iteration
    assert np.array_equal(np.array(fb["u"]["c"]), np.array(fa["u"]["c"]))


@pytest.mark.parametrize("scheme", ["SBDF2", "SBDF3", "CNAB2", "RK222"])
def test_failed_graph_capture_leaves_the_history_intact(scheme):
    """A capture runs the host side of step() (history rotation, iteration count) without executing a kernel: when
    it fails the timestepper's host state is rolled back before the step is taken by ordinary launches
    (MultistepIMEX.graph_snapshot / graph_rollback) -- the run equals one that never tried."""
    import torch
    import dedalus_amd.public as d3
    a, fa = problems.kdv_burgers(d3, Nx=128, timestepper=scheme)
    b, fb = problems.kdv_burgers(d3, Nx=128, timestepper=scheme)
    a.enable_step_graph(False)              # (small problems replay by default)
    b.enable_step_graph(True)
    real = b.timestepper.step

    def flaky(dt, wall_time=None):
        real(dt, wall_time)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("forced capture failure")
    b.timestepper.step = flaky
    for h in [2e-3] * 12:
        a.step(h)
        b.step(h)
    assert b._graph["failed"] and not b._graph["graphs"]
    assert abs(b.sim_time - a.sim_time) < 1e-15 and b.iteration == a.iteration
    assert np.array_equal(np.array(fb["u"]["c"]), np.array(fa["u"]["c"]))


@pytest.mark.parametrize("case", ["kdv_SBDF2", "rb2d_SBDF2", "kdv_CNAB2"])
def test_multistep_step_graphs_match_ordinary_launches(case):
    """Multistep schemes rotate their history buffers: one HIP graph per phase of the rotation, replayed in turn
    (core/ivp_common.py::_graph_replay, MultistepIMEX.graph_phase); a timestep change falls back to ordinary launches
    through the start-up of the new history and captures again."""
    import dedalus_amd.public as d3
    name, ts = case.split("_")
    make = (lambda: problems.kdv_burgers(d3, Nx=256, timestepper=ts)) if name == "kdv" else \
        (lambda: problems.rayleigh_benard_2d(d3, Nx=64, Nz=32, timestepper=ts))
    a, fa = make()
    b, fb = make()
    a.enable_step_graph(False)              # (small problems replay by default)
    b.enable_step_graph(True)
    dt = 2e-3 if name == "kdv" else 1e-3
    for i, h in enumerate([dt] * 11 + [0.5 * dt] * 9):
        a.step(h)
        b.step(h)
        if i == 10:
            assert len(b._graph["graphs"]) >= 2 and not b._graph["failed"]
    assert len(b._graph["graphs"]) >= 2 and not b._graph["failed"]
    assert abs(b.sim_time - a.sim_time) < 1e-15 and b.iteration == a.iteration
    for k in fa:
        assert np.array_equal(np.array(fb[k]["c"]), np.array(fa[k]["c"])), k


def test_direct_right_hand_side_path_on_the_device(monkeypatch):
    """The forward z transform writing F's equation rows (conversion in its band apply) against the gather mat-vec path
    (DDH_NO_DIRECT_F=1), 3-D and 2-D Rayleigh-Benard, several steps."""
    import dedalus_amd.public as d3
    for builder, kw in ((problems.rayleigh_benard_3d, dict(Nx=16, Ny=24, Nz=16, timestepper="RK222")),
                        (problems.rayleigh_benard_2d, dict(Nx=64, Nz=32, timestepper="SBDF2"))):
        monkeypatch.delenv("DDH_NO_DIRECT_F", raising=False)
        s1, f1 = builder(d3, **kw)
        monkeypatch.setenv("DDH_NO_DIRECT_F", "1")
        s0, f0 = builder(d3, **kw)
        assert s1.F_direct is not None and s0.F_direct is None
        for _ in range(4):
            s1.step(1e-3)
            s0.step(1e-3)
        for k in ("p", "b", "u"):
            a1, a0 = np.array(f1[k]['c']), np.array(f0[k]['c'])
            assert np.isfinite(a1).all() and rel(a1, a0) < 1e-12, (k, rel(a1, a0))


def test_reference_style_loop_loses_no_output(tmp_path):
    """A script that never closes its handlers (`while solver.proceed: solver.step(dt)`, the reference's examples) still
    finds every scheduled write in its file: the asynchronously staged last write is flushed when `proceed` turns False
    (the reference writes inside process(), core/evaluator.py:366-700), and each write records the timestep of ITS step
    (core/timesteppers.py:150, 608), not the previous one's."""
    import dedalus_amd.public as d3
    from dedalus_amd.tools import h5lite
    solver, f = problems.rayleigh_benard_2d(d3, Nx=64, Nz=32)
    snap = solver.evaluator.add_file_handler(str(tmp_path / "snap"), iter=2, max_writes=10)
    snap.add_task(f["b"], name="b")
    assert snap.async_staging
    solver.stop_iteration = 5
    dts = [1e-3, 2e-3, 3e-3, 4e-3, 5e-3]
    while solver.proceed:
        solver.step(dts[solver.iteration])
    # no close(), no flush() by the script
    r = h5lite.read(str(tmp_path / "snap" / "snap_s1.h5"))
    assert np.array_equal(r["scales/iteration"].read(), [0, 2, 4])
    assert np.allclose(r["scales/timestep"].read(), [dts[0], dts[2], dts[4]], rtol=0, atol=0)
    assert np.isfinite(r["tasks/b"].read(2)).all()


def test_zero_row_mask_of_the_runge_kutta_right_hand_side(monkeypatch):
    """Rows that are zero in M.X and in F (the continuity equation) are not read by the lean forward sweep
    (ddh_pencil_solve_recombined_sparse): same end state bit for bit as with the mask switched off, and the mask is what
    the problem structure says."""
    import dedalus_amd.public as d3

    def run(masked):
        for var in ("DDH_NO_ZERO_ROWS", "DDH_NO_SKIP_ROWS"):
            if not masked:
                monkeypatch.setenv(var, "1")
            else:
                monkeypatch.delenv(var, raising=False)
        solver, f = problems.rayleigh_benard_3d(d3, Nx=32, Ny=32, Nz=16, timestepper="RK222")
        solver.pack.set_solve_variant(0)                 # one thread per system: the lean forward sweep
        for _ in range(3):
            solver.step(1e-3)
        z = solver.mx_f_zero_rows()
        return solver, {k: np.array(f[k]["c"]) for k in ("p", "b", "u")}, z

    s1, a, z1 = run(True)
    s0, b, z0 = run(False)
    assert z0 is None and z1 is not None
    mask = z1[0].cpu().numpy()
    cont = s1.eq_info[0]                                  # "trace(grad_u) + tau_p = 0"
    assert mask[cont["row0"]:cont["row0"] + cont["rows"]].all()
    for i in (1, 2):                                      # the b and u equations carry dt and F
        e = s1.eq_info[i]
        assert not mask[e["row0"]:e["row0"] + e["rows"]].any()
    # intermediate stages do not store what nothing reads before the last stage: the pressure and the tau variables
    k1, k0 = s1.intermediate_skip_rows(), s0.intermediate_skip_rows()
    assert k0 is None and k1 is not None
    skip = k1[0].cpu().numpy()
    for info in s1.var_info:
        want = info["field"].name not in ("b", "u")
        assert skip[info["row0"]:info["row0"] + info["rows"]].all() == want, info["field"].name
        assert skip[info["row0"]:info["row0"] + info["rows"]].any() == want, info["field"].name
    assert abs(z1[1] - mask.mean()) < 1e-15 and 0.15 < z1[1] < 0.35     # (16 + 8 of 89 rows here, 264 of 1289 at 512 x 512 x 256)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_unknowns_skipped_by_intermediate_stages_are_never_consumed(monkeypatch):
    """An intermediate Runge-Kutta stage does not store the unknowns nothing reads before the last stage (skip_rows).
    DDH_POISON_SKIPPED overwrites exactly those rows of the state with NaN after every such solve: if anything -- F, M.X,
    a transform, a hook -- consumed them, the end state would not be finite; it is bit-identical to the ordinary run."""
    import dedalus_amd.public as d3

    def run(poison):
        if poison:
            monkeypatch.setenv("DDH_POISON_SKIPPED", "1")
        else:
            monkeypatch.delenv("DDH_POISON_SKIPPED", raising=False)
        solver, f = problems.rayleigh_benard_3d(d3, Nx=32, Ny=32, Nz=16, timestepper="RK222")
        solver.pack.set_solve_variant(0)
        assert solver.intermediate_skip_rows() is not None
        for _ in range(3):
            solver.step(1e-3)
        return {k: np.array(f[k]["c"]) for k in ("p", "b", "u")}

    a, b = run(True), run(False)
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), k


def test_tile_major_right_hand_sides_change_addresses_not_values(monkeypatch):
    """The timestepper's M.X and F buffers in the tile-major layout (ddh_pencil_matvec_update_tiled,
    ddh_cheb_forward_tiled, ddh_pencil_solve_recombined_tiled) against the natural layout (DDH_NO_RHS_TILING): the same
    values are summed in the same order, so the end states agree bit for bit.  256 x 256 x 256: the smallest size at which
    every tiled producer runs (window mat-vec, strided wave transform, lean sweep)."""
    import dedalus_amd.public as d3

    def run(tiled):
        if tiled:
            monkeypatch.delenv("DDH_NO_RHS_TILING", raising=False)
        else:
            monkeypatch.setenv("DDH_NO_RHS_TILING", "1")
        solver, f = problems.rayleigh_benard_3d(d3, Nx=256, Ny=256, Nz=256, timestepper="RK222")
        for _ in range(2):
            solver.step(1e-3)
        assert bool(solver.timestepper._tiled) == tiled
        out = {k: np.array(f[k]["c"]) for k in ("p", "b", "u")}
        del solver, f
        return out

    a, b = run(True), run(False)
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), k


def test_tile_major_layout_follows_the_sweep_variant(monkeypatch):
    """The tile-major decision belongs to the factorizations and the sweep variant of the moment (SolverBase.rhs_tiling is
    asked again after PencilPack.set_solve_variant and after every refactorization): a solver that starts tiled and is
    switched to the cooperative sweeps (no tiled form) goes back to the natural layout instead of failing inside
    ddh_pencil_solve_recombined_tiled, returns to tile-major with the default variant, and reaches bit for bit the end state
    of a run that never tiled."""
    import dedalus_amd.public as d3

    def run(tiled):
        if tiled:
            monkeypatch.delenv("DDH_NO_RHS_TILING", raising=False)
        else:
            monkeypatch.setenv("DDH_NO_RHS_TILING", "1")
        solver, f = problems.rayleigh_benard_3d(d3, Nx=256, Ny=256, Nz=256, timestepper="RK222")
        seen = []
        solver.step(1e-3)
        seen.append(bool(solver.timestepper._tiled))
        solver.pack.set_solve_variant(2)
        solver.step(1e-3)
        seen.append(bool(solver.timestepper._tiled))
        solver.pack.set_solve_variant(1)
        solver.step(5e-4)                                  # (a refactorization on the way back)
        seen.append(bool(solver.timestepper._tiled))
        out = {k: np.array(f[k]["c"]) for k in ("p", "b", "u")}
        del solver, f
        return out, seen

    (a, sa), (b, sb) = run(True), run(False)
    assert sa == [True, False, True] and sb == [False, False, False]
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), k


def test_x_blocked_stage_layout_changes_addresses_not_values(monkeypatch):
    """The arrays between the z and the x transforms stored [kx / 64][z][kx % 64][ky] (ddh_fft_set_stage_layout,
    Transformer.stage_xb) against the natural layout (DDH_NO_STAGE_XB): same arithmetic, bit-identical end states; the
    Hermitian-symmetry round trip of the first steps and a full-grid read-back go through the same stage arrays."""
    import dedalus_amd.public as d3

    def run(xb):
        if xb:
            monkeypatch.delenv("DDH_NO_STAGE_XB", raising=False)
        else:
            monkeypatch.setenv("DDH_NO_STAGE_XB", "1")
        solver, f = problems.rayleigh_benard_3d(d3, Nx=256, Ny=256, Nz=256, timestepper="RK222")
        dom = f["b"].domain
        assert (solver.dist.transformer.stage_xb(dom, dom.dealias) is not None) == xb
        for _ in range(2):
            solver.step(1e-3)
        out = {k: np.array(f[k]["c"]) for k in ("p", "b", "u")}
        f["b"].change_scales(1.5)
        out["b_grid"] = np.array(f["b"]["g"])
        del solver, f
        return out

    a, b = run(True), run(False)
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("ts", ["RK222", "SBDF2"])
def test_block_inverses_of_few_pencils_match_the_sweeps(ts, monkeypatch):
    """2-D problems (a few hundred pencils) apply explicit inverses of the diagonal blocks instead of sweeping them
    (SolverBase._block_inverses, blockinv_solve_kernel; inverses from unit solves of the band LU): same end state as the
    sweeps (DDH_BLOCK_INVERSE=0) through timestep changes, and the reference's own end state."""
    import dedalus_amd.public as d3
    seq = [1e-3] * 5 + [2e-3] * 3 + [1.5e-3] * 3
    monkeypatch.setenv("DDH_BLOCK_INVERSE", "0")
    a, fa = problems.rayleigh_benard_2d(d3, Nx=128, Nz=64, timestepper=ts)
    assert not a._block_inverse_plan()
    monkeypatch.delenv("DDH_BLOCK_INVERSE")
    b, fb = problems.rayleigh_benard_2d(d3, Nx=128, Nz=64, timestepper=ts)
    for dt in seq:
        a.step(dt)
        b.step(dt)
    bi = b._block_inverse_plan()
    assert bi and bi["x"], "the block-inverse path must be active by default for this size"
    for k, tol in (("b", 1e-10), ("u", 1e-9), ("p", 1e-8)):
        x, y = np.array(fb[k]["c"]), np.array(fa[k]["c"])
        assert np.isfinite(x).all()
        assert rel(x, y) < tol, (k, rel(x, y))
