"""Analysis output: the dependency-free HDF5 writer / reader, the file handlers' on-disk format
and schedule (core/evaluator.py:206-645 of the reference), and restarts through load_state (core/solvers.py:632-673).
The files are additionally opened with a real h5py when the image's conda interpreter offers one."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dedalus_amd.tools import h5lite          # noqa: E402

H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    r = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, env={"PATH": os.environ.get("PATH", "")})
    return r.returncode == 0


def _h5py(script, cwd):
    r = subprocess.run([H5PY_PYTHON, "-c", script], capture_output=True, text=True, cwd=cwd,
                       env={"PATH": os.environ.get("PATH", "")})
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _write_sample(path, nwrites):
    f = h5lite.File(path)
    f.attrs['set_number'] = 1
    f.attrs['handler_name'] = "sample"
    f.attrs['writes'] = 0
    sc = f.create_group('scales')
    t = sc.create_dataset('sim_time', shape=(0,), maxshape=(None,), dtype=np.float64)
    t.make_scale('sim_time')
    it = sc.create_dataset('iteration', shape=(0,), maxshape=(None,), dtype=np.int64)
    it.make_scale('iteration')
    x = sc.create_dataset('x_hash_0', data=np.linspace(0, 1, 40))
    x.make_scale('x')
    tk = f.create_group('tasks')
    big = tk.create_dataset('big', shape=(0, 3, 40, 20), maxshape=(None, 3, 40, 20), dtype=np.float64)   # 1 row per chunk
    small = tk.create_dataset('small', shape=(0, 2), maxshape=(None, 2), dtype=np.float64)               # many rows per chunk
    for d in (big, small):
        d.set_label(0, 't')
        d.attach_scale(0, t)
        d.attach_scale(0, it)
    big.set_label(2, 'x')
    big.attach_scale(2, x)
    big.attrs['grid_space'] = np.array([True, True])
    big.attrs['scales'] = (1.5, 1.0)
    f.commit()
    for w in range(nwrites):
        t.append(0.25 * w)
        it.append(10 * w)
        big.append(np.arange(2400).reshape(3, 40, 20) + 1e4 * w)
        small.append([w, -w])
        f.set_scalar_attr(f, 'writes', w + 1)
    f.close()


def test_h5lite_round_trip_single_and_multi_node_index(tmp_path):
    for n in (0, 3, 200):            # 200 rows of 'big' need several B-tree leaves and a root node
        path = str(tmp_path / ("s%d.h5" % n))
        _write_sample(path, n)
        r = h5lite.read(path)
        assert r.attrs['writes'] == n and r.attrs['handler_name'] == "sample"
        assert sorted(r.keys()) == ['scales', 'tasks']
        big = r['tasks/big']
        assert big.shape == (n, 3, 40, 20)
        a = big.read()
        assert np.array_equal(a, np.arange(2400).reshape(1, 3, 40, 20) + 1e4 * np.arange(n).reshape(n, 1, 1, 1))
        assert np.array_equal(r['tasks/small'].read(), np.stack([np.arange(n), -np.arange(n)], axis=1).astype(float))
        assert np.array_equal(r['scales/iteration'].read(), 10 * np.arange(n))
        if n:
            assert np.array_equal(big.read(-1), a[-1]) and np.array_equal(big.read(1), a[1])
        assert np.array_equal(big.attrs['grid_space'], [True, True]) and np.allclose(big.attrs['scales'], [1.5, 1.0])
        assert big.attrs['DIMENSION_LABELS'] == ['t', '', 'x', '']
        assert r['scales/x_hash_0'].attrs['CLASS'] == "DIMENSION_SCALE" and r['scales/x_hash_0'].attrs['NAME'] == "x"


@pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py in this image")
def test_h5lite_files_open_in_h5py(tmp_path):
    _write_sample(str(tmp_path / "s.h5"), 150)
    out = _h5py("""
import h5py, numpy as np
f = h5py.File('s.h5', 'r')
big = f['tasks/big']
n = big.shape[0]
assert big.maxshape == (None, 3, 40, 20)
assert np.array_equal(big[:], np.arange(2400).reshape(1, 3, 40, 20) + 1e4 * np.arange(n).reshape(n, 1, 1, 1))
assert [d.label for d in big.dims] == ['t', '', 'x', '']
assert list(big.dims[0].keys()) == ['sim_time', 'iteration'] and list(big.dims[2].keys()) == ['x']
assert np.array_equal(big.dims[0]['sim_time'][:], 0.25 * np.arange(n))
assert np.array_equal(big.dims[2]['x'][:], np.linspace(0, 1, 40))
assert np.array_equal(f['tasks/small'][:, 1], -np.arange(n))
assert f.attrs['writes'] == n and f.attrs['handler_name'] == 'sample'
assert np.array_equal(big.attrs['grid_space'], [True, True])
print('ok', n)
""", str(tmp_path))
    assert out.split() == ["ok", "150"]


def _rb(ts="RK222"):
    import problems
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    return d3, problems.rayleigh_benard_2d(d3, Nx=32, Nz=16, timestepper=ts, dist_kw=dict(executor=NumpyExecutor()))


def test_file_handler_format_and_schedule(tmp_path):
    d3, (solver, f) = _rb()
    u, b = f["u"], f["b"]
    snap = solver.evaluator.add_file_handler(str(tmp_path / "snap"), sim_dt=0.02, max_writes=3)
    snap.add_task(b, name="buoyancy")
    snap.add_task("-div(skew(u))", name="vorticity", scales=1.5)
    snap.add_task(u, layout='c', name="u_c")
    probe = solver.evaluator.add_dictionary_handler(iter=3)
    probe.add_task(d3.grad(b), name="gb")
    seen = []
    for i in range(9):
        if i % 2 == 0:
            b['c']                              # outputs pass through coefficient space (drops the Nyquist mode of the noise)
            seen.append(np.array(b['g']))       # the handler sees the pre-step state of iterations 0, 2, 4, ...
        solver.step(0.01)
    assert probe['gb']['g'].shape == (2, 32, 16)
    s1 = h5lite.read(str(tmp_path / "snap" / "snap_s1.h5"))
    s2 = h5lite.read(str(tmp_path / "snap" / "snap_s2.h5"))
    assert s1.attrs['writes'] == 3 and s1.attrs['set_number'] == 1 and s2.attrs['set_number'] == 2
    assert np.allclose(s1['scales/sim_time'].read(), [0.0, 0.02, 0.04]) and np.allclose(s2['scales/sim_time'].read(), [0.06, 0.08])
    assert np.array_equal(s1['scales/iteration'].read(), [0, 2, 4]) and np.array_equal(s2['scales/write_number'].read(), [4, 5])
    assert np.allclose(s1['scales/timestep'].read(), 0.01)
    bo = s1['tasks/buoyancy']
    assert bo.shape == (3, 32, 16) and bo.attrs['DIMENSION_LABELS'] == ['t', 'x', 'z']
    for k in range(3):
        assert np.allclose(bo.read(k), seen[k], rtol=0, atol=1e-13)
    assert np.allclose(s2['tasks/buoyancy'].read(1), seen[4], rtol=0, atol=1e-13)
    vo = s1['tasks/vorticity']
    assert vo.shape == (3, 48, 24) and np.allclose(vo.attrs['scales'], 1.5)
    uc = s1['tasks/u_c']
    assert uc.shape == (3, 2, 32, 16) and not uc.attrs['grid_space'].any() and uc.attrs['DIMENSION_LABELS'] == ['t', '', 'kx', 'kz']
    names = s1['scales'].keys()
    assert sum(n.startswith("x_hash_") for n in names) == 2 and sum(n.startswith("kx_hash_") for n in names) == 1
    if _have_h5py():
        out = _h5py("""
import h5py, numpy as np
f = h5py.File('snap/snap_s1.h5', 'r')
d = f['tasks/vorticity']
assert [x.label for x in d.dims] == ['t', 'x', 'z'] and d.dims[1][0].shape == (48,) and d.dims[2][0].shape == (24,)
assert list(d.dims[0].keys()) == ['sim_time', 'wall_time', 'timestep', 'iteration', 'write_number']
assert np.allclose(d.dims[0]['sim_time'][:], [0, 0.02, 0.04])
print('ok')
""", str(tmp_path))
        assert out.strip() == "ok"


def test_restart_from_checkpoint_reproduces_the_run(tmp_path):
    d3, (solver, f) = _rb("RK222")
    chk = solver.evaluator.add_file_handler(str(tmp_path / "chk"), iter=4, max_writes=10)
    for v in solver.state:
        chk.add_task(v, layout='g')
    for _ in range(8):
        solver.step(0.01)
    end = {k: np.array(v['c']) for k, v in f.items()}
    path = str(tmp_path / "chk" / "chk_s1.h5")
    assert np.array_equal(h5lite.read(path)['scales/iteration'].read(), [0, 4])
    d3, (solver2, f2) = _rb("RK222")
    write, dt = solver2.load_state(path)                      # the last write: iteration 4
    assert (write, dt, solver2.iteration) == (2, 0.01, 4) and abs(solver2.sim_time - 0.04) < 1e-15
    for _ in range(4):
        solver2.step(dt)
    for k in ("b", "u", "p"):
        err = np.linalg.norm(np.array(f2[k]['c']) - end[k]) / np.linalg.norm(end[k])
        assert err < 1e-11, (k, err)
    # append mode continues the set and write numbering
    chk2 = solver2.evaluator.add_file_handler(str(tmp_path / "chk"), iter=1, mode="append")
    chk2.add_task(f2["b"])
    solver2.step(dt)
    r = h5lite.read(str(tmp_path / "chk" / "chk_s2.h5"))
    assert r.attrs['set_number'] == 2 and np.array_equal(r['scales/write_number'].read(), [3])


def test_shell_file_handler(tmp_path):
    import problems
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, f = problems.shell_convection(d3, dist_kw=dict(executor=NumpyExecutor()))
    snap = solver.evaluator.add_file_handler(str(tmp_path / "slices"), iter=2, max_writes=10)
    snap.add_task(f["b"], name="b", scales=1.5)
    snap.add_task(f["u"], name="u")
    snap.add_task(f["tau_b1"], name="tau")
    f["b"].change_scales(1.5)
    before = np.array(f["b"]["g"])
    f["b"].change_scales(1)
    for _ in range(3):
        solver.step(0.05)
    r = h5lite.read(str(tmp_path / "slices" / "slices_s1.h5"))
    assert np.array_equal(r['scales/iteration'].read(), [0, 2])
    assert r['tasks/b'].shape == (2, 24, 18, 12) and r['tasks/u'].shape == (2, 3, 16, 12, 8) and r['tasks/tau'].shape == (2, 16, 12, 1)
    assert r['tasks/u'].attrs['DIMENSION_LABELS'] == ['t', '', 'phi', 'theta', 'r']
    assert r['tasks/tau'].attrs['DIMENSION_LABELS'] == ['t', 'phi', 'theta', 'constant']
    assert np.allclose(r['tasks/b'].read(0), before, rtol=0, atol=1e-13)


def test_h5lite_index_growth_and_capacity(tmp_path, monkeypatch):
    """Leaf split, root creation and the capacity limit of the chunk index, exercised with a tiny node size
    (own reader only: libhdf5 assumes the default node size of a version-0 superblock)."""
    monkeypatch.setattr(h5lite, "CHUNK_K", 2)                 # 4 entries per node -> 16 chunks per dataset
    path = str(tmp_path / "k2.h5")
    f = h5lite.File(path)
    d = f.create_dataset("x", shape=(0, 600), maxshape=(None, 600), dtype=np.float64)     # 4800 B rows: 1 row per chunk
    f.commit()
    assert d.chunk_rows == 1 and d.capacity == 16
    for w in range(16):
        d.append(np.full(600, float(w)))
        f.flush()
        r = h5lite.read(path)["x"]
        assert r.shape == (w + 1, 600) and np.array_equal(r.read()[:, 0], np.arange(w + 1.0))
    with pytest.raises(RuntimeError):
        d.append(np.zeros(600))
    f.close()


def test_file_handler_rolls_over_when_a_dataset_is_full(tmp_path, monkeypatch):
    monkeypatch.setattr(h5lite, "CHUNK_K", 1)                 # 2 entries per node -> 4 chunks per dataset
    d3, (solver, f) = _rb()
    h = solver.evaluator.add_file_handler(str(tmp_path / "roll"), iter=1)          # no max_writes
    h.add_task(f["b"], name="b")                                                    # 4 KB rows: one row per chunk
    for _ in range(6):
        solver.step(0.01)
    s1 = h5lite.read(str(tmp_path / "roll" / "roll_s1.h5"))
    s2 = h5lite.read(str(tmp_path / "roll" / "roll_s2.h5"))
    assert s1["tasks/b"].shape[0] == 4 and s2["tasks/b"].shape[0] == 2
    assert np.array_equal(s2["scales/write_number"].read(), [5, 6])
