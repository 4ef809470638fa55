"""GPU: the solver's state vector kept tile-major (round 6; SolverBase._enable_state_tiling, ddh_pencil_set_state_tiled,
ddh_fft_set_coeff_tiled, ddh_tile_rows).  The layout changes addresses, never values: every check here is bit for bit
against the natural layout.  Reference semantics: the state lives in the fields, gathered / scattered per solve
(core/subsystems.py:497-596, core/timesteppers.py:588-643)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tile(a):
    """natural [R][nx][ny] -> tile-major [R][nx/8][ny/8][8][8] (flattened back to [R][nx][ny])"""
    R, nx, ny = a.shape
    return np.ascontiguousarray(a.reshape(R, nx // 8, 8, ny // 8, 8).transpose(0, 1, 3, 2, 4)).reshape(R, nx, ny)


def _band(a):
    """natural [R][nx][ny] -> kx-band-major [nx/8][R][ny/8][8][8] (flattened back to [R][nx][ny])"""
    R, nx, ny = a.shape
    return np.ascontiguousarray(a.reshape(R, nx // 8, 8, ny // 8, 8).transpose(1, 0, 3, 2, 4)).reshape(R, nx, ny)


@pytest.mark.parametrize("shape", [(3, 8, 8), (5, 64, 40), (2, 256, 512)])
def test_tile_rows_is_the_documented_permutation_both_ways(shape):
    from dedalus_amd.device import Device
    from dedalus_amd.executor import HipExecutor
    dev = Device.get()
    ex = HipExecutor(dev)
    R, nx, ny = shape
    a = np.random.default_rng(1).standard_normal(shape)
    ad = dev.from_host(a)
    td, bd = dev.empty(shape), dev.empty(shape)
    ex.tile_rows(ad, td, R, nx, ny, True)
    assert np.array_equal(td.cpu().numpy(), _tile(a))
    ex.tile_rows(td, bd, R, nx, ny, False)
    assert np.array_equal(bd.cpu().numpy(), a)
    # kx-band-major: a block of rows of a vector of R + 3 rows, starting at row 2
    Rt = R + 3
    full = np.random.default_rng(2).standard_normal((Rt, nx, ny))
    fd = dev.from_host(_band(full))
    base = fd.reshape(-1)[2 * 8 * ny:]
    ex.tile_rows(ad, base, R, nx, ny, True, Rt)
    full[2:2 + R] = a
    assert np.array_equal(fd.cpu().numpy(), _band(full))
    bd.zero_()
    ex.tile_rows(base, bd, R, nx, ny, False, Rt)
    assert np.array_equal(bd.cpu().numpy(), a)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("noband", [False, True])
def test_mat_vec_reads_a_tile_major_state_bit_for_bit(monkeypatch, noband, mode):
    """x tile-major (ddh_pencil_set_state_tiled) with y natural / y tile-major (threads following the tiles), window-form and
    term-list kernels: the values of the natural run, at the tiled addresses."""
    from dedalus_amd.device import Device
    from dedalus_amd.pencilpack import PencilPack, TermList
    dev = Device.get()
    rng = np.random.default_rng(5)
    nvar, Nz, ncx, ncy = 3, 24, 128, 64
    N = nvar * Nz + 3
    rows, cols, vals = [], [], []
    for v in range(1, nvar):
        for kz in range(Nz):
            for off in (0, 2, 4):
                if kz + off < Nz and (off == 0 or rng.random() < 0.8):
                    rows.append(v * Nz + kz); cols.append(v * Nz + kz + off); vals.append(rng.standard_normal())
    M = TermList(N, N, rows, cols, vals)
    nx, ny = 2 * ncx, 2 * ncy
    pack = PencilPack(dev, 2, N, nx, ny, 0.7 * np.arange(ncx), 1.3 * np.arange(ncy))
    if noband:
        monkeypatch.setenv("DDH_MV_NOBAND", "1")
    mid = pack.add_matrix(M)
    x = rng.standard_normal((N, nx, ny))
    xn, xt = dev.from_host(x), dev.from_host(_tile(x) if mode == 1 else _band(x))
    y_nat = dev.zeros((N, nx, ny))
    pack.matvec(mid, xn, y_nat)
    want = y_nat.cpu().numpy()
    pack.set_state_tiled(mode)
    try:
        y1 = dev.zeros((N, nx, ny))
        pack.matvec(mid, xt, y1)                                    # x tiled, y natural
        assert np.array_equal(y1.cpu().numpy(), want)
        if not noband:
            y2 = dev.zeros((N, nx, ny))
            pack.matvec(mid, xt, y2, owned=True, tiled=True)        # x tiled, y tiled: threads follow the tiles
            assert np.array_equal(y2.cpu().numpy(), _tile(want))
    finally:
        pack.set_state_tiled(False)
    y3 = dev.zeros((N, nx, ny))
    pack.matvec(mid, xn, y3, owned=True, tiled=True) if not noband else pack.matvec(mid, xn, y3)
    assert np.array_equal(y3.cpu().numpy(), _tile(want) if not noband else want)


WORKER = r"""
import os, sys, json, hashlib
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import problems
import dedalus_amd.public as d3
Nx, Ny, Nz = %(size)s
solver, f = problems.rayleigh_benard_3d(d3, Nx=Nx, Ny=Ny, Nz=Nz, timestepper=%(ts)r)
out = {"x_tiled": int(solver.x_tiled)}
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
for _ in range(2):
    solver.step(1e-3)
out["u_grid_mid"] = sha(np.array(f["u"]["g"]))            # user access to grid data between steps (natural shadow + transforms)
c = np.array(f["b"]["c"]); c *= 1.0 + 2.0 ** -20
f["b"]["c"] = c                                             # the user rewrites a state field: tiled again before the next step
for _ in range(2):
    solver.step(1e-3)
for k in ("p", "b", "u", "tau_u1", "tau_b2"):
    if k in f:
        out[k] = sha(np.array(f[k]["c"]))
out["state"] = sha(np.asarray(solver.ex.download(solver.state_natural())))
print(json.dumps(out))
"""


def _run(size, ts, env):
    code = WORKER % dict(root=ROOT, size=repr(tuple(size)), ts=ts)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DDH_X_TILED_MIN="0", DDH_RHS_TILING_MIN="0", **env))
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("size,ts,mode", [((256, 256, 128), "RK222", "1"), ((64, 48, 128), "RK443", "1"), ((128, 128, 128), "SBDF2", "1"),
                                          ((256, 256, 128), "RK222", "2")])
def test_tile_major_state_changes_addresses_not_values(size, ts, mode):
    """3-D Rayleigh-Benard stepped with the state vector stored with tile-major rows (mode 1: the default from 65 536 storage
    entries per row on; forced here) or kx-band-major (mode 2, opt-in) and natural (DDH_X_TILED=0), with a grid-space read and a coefficient-space rewrite of state fields between
    the steps: the same end state bit for bit -- fields, the un-aliased tau variable, the whole state vector -- also with the
    right-hand sides natural (DDH_NO_RHS_TILING: tile-major x, natural y in the mat-vec).  Nz = 128: the state is only kept
    tile-major where the backward z transforms read it in place (Chebyshev 192 <- 128 is a wave-kernel size); 256 x 256 x 64
    (96 <- 64 is not) keeps the natural layout."""
    a = _run(size, ts, {"DDH_X_TILED": mode})
    b = _run(size, ts, {"DDH_X_TILED": "0"})
    c = _run(size, ts, {"DDH_X_TILED": mode, "DDH_NO_RHS_TILING": "1"})
    assert a["x_tiled"] == size[1] and b["x_tiled"] == 0 and c["x_tiled"] == size[1]
    for k in a:
        if k != "x_tiled":
            assert a[k] == b[k] == c[k], (k, a[k], b[k], c[k])
    if ts == "RK222" and mode == "1":
        assert _run((256, 256, 64), ts, {})["x_tiled"] == 0
