"""GPU, 2 processes sharing the one GPU of the test box: the HIP multi-rank path (pencils sharded on kx,
z-sharded grid space, ddh_a2a_pack/unpack around the exchange, fused grid stage on the exchanged
layout) reproduces the reference's end state.  RCCL needs one GPU per rank, so the exchange itself runs
through gloo on host copies here; with one GPU per rank the same code issues it through "nccl"."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("case,world", [("rb3d_8x12x8_rk222", 2), ("rb2d_32x16_rk222", 2)])
def test_sharded_hip_run_matches_reference(golden_dir, case, world):
    gold = np.load(os.path.join(golden_dir, "ivp.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "mp_worker.py"), case, tmp, "hip"]
        env = dict(os.environ, OMP_NUM_THREADS="1", DDH_DIST_BACKEND="gloo")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(world)]
        for key, tol in (("p", 1e-9), ("b", 1e-9), ("u", 1e-8)):
            ref = gold[case + "__" + key]
            xaxis = ref.ndim - (3 if "3d" in case else 2)
            full = np.concatenate([p[key] for p in parts], axis=xaxis)
            assert full.shape == ref.shape
            assert rel(full, ref) < tol, (key, rel(full, ref))


def _run_sharded(case, world, env_extra):
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "mp_worker.py"), case, tmp, "hip"]
        env = dict(os.environ, OMP_NUM_THREADS="1", DDH_DIST_BACKEND="gloo", **env_extra)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return [dict(np.load(os.path.join(tmp, "rank%d.npz" % k))) for k in range(world)]


def test_sharded_run_with_the_blocked_exchange_layout_matches_the_reference():
    """3-D RB 128 x 128 x 64 on 2 ranks (64 kx rows each: blocks of 64): the x transforms read the exchanged layout
    [p][z_loc][nx / P][ky] in place and write what leaves -- no unpack / pack kernel (Transformer.stage_xb, round 6) -- and
    the end state is the UNMODIFIED reference's (tests/golden/config_rb3d_endstate_128x128x64.npz, three RK222 steps)."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "config_rb3d_endstate_128x128x64.npz"))
    parts = _run_sharded("rb3dsize_128x128x64", 2, {})
    assert list(parts[0]["stage_xb"]) == [0, 96 // 2, 64], parts[0]["stage_xb"]      # z side natural, x side (gz_loc, rows)
    sa = G["sample_a"] if "sample_a" in G else np.array([0, 4, 0, 4])
    for key, tol in (("b", 1e-10), ("p", 1e-10), ("u", 1e-9)):
        full = np.concatenate([p[key] for p in parts], axis=parts[0][key].ndim - 3)
        mine = full[..., int(sa[0])::int(sa[1]), int(sa[2])::int(sa[3]), :]
        ref = G["end__%s_a" % key]
        assert mine.shape == ref.shape
        assert rel(mine, ref) < tol, (key, rel(mine, ref))


@pytest.mark.parametrize("nx,world,rows", [(256, 2, 128), (512, 2, 256), (256, 4, 64)])
def test_blocked_and_unpacked_exchanges_take_the_same_steps(nx, world, rows):
    """Blocks of nx / world = 128, 256 and 64 rows: the blocked exchange (x transforms on the exchanged layout, no pack /
    unpack kernels: the default) and the round-5 pack / exchange / unpack path (DDH_A2A_BLOCKED=0) give bit-identical
    states -- and so do the blocked exchange's schedules: waits deferred to the consumer (default) or not, the x steps
    component by component behind the arrivals, every field's z step issued first, the grid stage in windows of z planes."""
    case = "rb3dsize_%dx16x32" % nx
    a = _run_sharded(case, world, {})
    b = _run_sharded(case, world, {"DDH_A2A_BLOCKED": "0"})
    assert int(a[0]["stage_xb"][2]) == rows and int(b[0]["stage_xb"][0]) == -1
    others = [b]
    if rows == 128:
        others += [_run_sharded(case, world, env) for env in ({"DDH_A2A_DEFER": "0"}, {"DDH_A2A_SPLIT_X": "1"},
                                                              {"DDH_A2A_PREFETCH": "1", "DDH_A2A_SPLIT_X": "1"})]
    # the grid stage in 2 / 4 windows of this rank's z planes, the exchanges in parts (Transformer windows)
    assert int(a[0]["windows"]) == 2                     # (the default)
    for K in ("1", "4"):
        w = _run_sharded(case, world, {"DDH_A2A_WINDOWS": K})
        assert int(w[0]["windows"]) == int(K), w[0]["windows"]
        others.append(w)
    for other in others:
        for ra, rb in zip(a, other):
            for key in ("p", "b", "u"):
                assert np.array_equal(ra[key], rb[key]), (rows, key)


@pytest.mark.parametrize("ts", ["SBDF2"])
def test_m_sharded_shell_hip_run_matches_reference(golden_dir, ts):
    """Shell convection, azimuthal wavenumbers sharded over 2 processes that share the GPU: local SWSH / regularity /
    radial plans, ell-term kernels on the local slots, a2a pack/unpack around the (host-staged) exchange."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_multiprocess import _run_worker, check_shell_parts, SHELL_TOL
    gold = np.load(os.path.join(golden_dir, "shellfields.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        parts = _run_worker("shell_conv_" + ts, 2, tmp, extra=["hip"], env_extra=dict(DDH_DIST_BACKEND="gloo"))
        check_shell_parts(parts, gold, "conv_%s__" % ts, list(SHELL_TOL))
