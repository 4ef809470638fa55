"""CPU, 2 processes (gloo): pencils sharded over ranks + all-to-all transposes reproduce the reference
end state (and therefore the single-process result) -- the direct multi-rank-vs-serial equality
test the reference's own suite lacks."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("case,world,overlap", [("rb3d_8x12x8_rk222", 2, "0"), ("rb2d_32x16_rk222", 2, "0"),
                                                ("rb3d_8x12x8_rk222", 2, "1"), ("rb3d_8x12x8_rk222", 4, "1"),
                                                ("rb3d_16x16x16_rk222", 8, "1")])
def test_sharded_run_matches_reference(golden_dir, case, world, overlap, monkeypatch):
    # overlap = "1": the per-component exchange pipeline (DDH_A2A_OVERLAP, the default); 4 ranks / 8 ranks (the target
    # node's rank count): one kx group per rank, 3 z grid planes per rank at 8
    monkeypatch.setenv("DDH_A2A_OVERLAP", overlap)
    gold = np.load(os.path.join(golden_dir, "ivp.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "mp_worker.py"), case, tmp]
        env = dict(os.environ, OMP_NUM_THREADS="1")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(world)]
        for key in ("p", "b", "u"):
            ref = gold[case + "__" + key]
            # coefficient space is sharded along x (first spatial axis in the user's order)
            xaxis = ref.ndim - (3 if "3d" in case else 2)
            full = np.concatenate([p[key] for p in parts], axis=xaxis)
            assert full.shape == ref.shape
            assert rel(full, ref) < 1e-9, (key, rel(full, ref))
        # gathered analysis output (rank 0 writes the global arrays)
        from dedalus_amd.tools import h5lite
        r = h5lite.read(os.path.join(tmp, "final", "final_s1.h5"))
        ref = gold[case + "__b"]
        assert r["tasks/b_c"].shape == (1,) + ref.shape and rel(r["tasks/b_c"].read(0), ref) < 1e-9
        assert r["tasks/b_g"].shape == (1,) + ref.shape and np.isfinite(r["tasks/b_g"].read(0)).all()


def _run_worker(case, world, tmp, extra=(), env_extra=None, timeout=900):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "mp_worker.py"), case, tmp] + list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(world)]


SHELL_TOL = dict(p=1e-10, b=1e-10, u=1e-9, tau_p=1e-8, tau_b1=1e-8, tau_b2=1e-8, tau_u1=1e-8, tau_u2=1e-8)


def check_shell_parts(parts, gold, prefix, keys):
    for key in keys:
        ref = gold[prefix + key]
        if key == "tau_p":
            full = parts[0][key]                         # a constant: replicated
            assert all(np.array_equal(p[key], full) for p in parts)
            assert abs(float(full.reshape(-1)[0]) - float(ref.reshape(-1)[0])) < 1e-10
            continue
        else:
            # the packed azimuthal axis is block-distributed (third axis from the end)
            full = np.concatenate([p[key] for p in parts], axis=ref.ndim - 3)
        assert full.shape == ref.shape, (key, full.shape, ref.shape)
        assert rel(full, ref) < SHELL_TOL[key], (key, rel(full, ref))


@pytest.mark.parametrize("ts,world", [("SBDF2", 2), ("RK222", 2)])
def test_m_sharded_shell_convection_matches_reference(golden_dir, ts, world):
    """Shell convection (config 5) with the azimuthal wavenumbers block-distributed over the ranks, colatitudes
    distributed in grid space, one all-to-all per transform: the reference's (serial) end state."""
    gold = np.load(os.path.join(golden_dir, "shellfields.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        parts = _run_worker("shell_conv_" + ts, world, tmp)
        check_shell_parts(parts, gold, "conv_%s__" % ts, list(SHELL_TOL))
        # the analysis set written by rank 0 from the gathered blocks holds the global arrays
        from dedalus_amd.tools import h5lite
        r = h5lite.read(os.path.join(tmp, "final", "final_s1.h5"))
        ref = gold["conv_%s__b" % ts]
        assert r["tasks/b_c"].shape == (1,) + ref.shape and rel(r["tasks/b_c"].read(0), ref) < SHELL_TOL["b"]
        assert r["tasks/u_g"].shape == (1, 3, 24, 18, 12) and np.isfinite(r["tasks/u_g"].read(0)).all()
        assert np.array_equal(r["scales/iteration"].read(), [4])


def test_m_sharded_shell_cfl_sequence(golden_dir):
    gold = np.load(os.path.join(golden_dir, "shellfields.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        parts = _run_worker("shell_cfl", 2, tmp)
        for p in parts:
            assert np.allclose(p["dts"], gold["shellcfl__dts"], rtol=1e-10, atol=0)
            assert np.allclose(p["speeds"], gold["shellcfl__speeds"], rtol=1e-10, atol=0)
        check_shell_parts(parts, gold, "shellcfl__", ["p", "b", "u"])


def test_sharded_solve_against_the_references_pencil_matrices(tmp_path):
    """2 ranks (kx-sharded pencils, gloo): every solve on the 16 sampled pencils of the Nz = 256 problem against the
    reference's own M_min / L_min (tests/golden/pencils_nz256.npz) -- the multi-rank twin of
    tests/test_reference_pencils.py (mode offsets of the local pencils, SolverBase.gather_pencil on a shard)"""
    parts = _run_worker("refpencils", 2, str(tmp_path))
    assert sum(int(p["npencils"]) for p in parts) == 16 and all(int(p["npencils"]) == 8 for p in parts)
    for p in parts:
        assert p["residual"].max() < 1e-13 and p["solution"].max() < 1e-11 and p["dropped"].max() == 0.0


@pytest.mark.parametrize("world", [2, 4])
def test_partial_all_to_all_of_several_components(world):
    """all_to_all_start(part=(offset, count), batch=n) -- the exchange of one window of z planes of a field's n components
    (ddh_comm_alltoall_part on the GPUs) -- through its gloo stand-in: rank r receives from every peer q exactly the
    elements [offset, offset + count) of q's block r of every component, and nothing else of its buffer changes."""
    with tempfile.TemporaryDirectory() as tmp:
        parts = _run_worker("a2a_part", world, tmp)
    nb, blk, off, cnt = 3, 10, 4, 5
    for r, me in enumerate(parts):
        for key in ("recv", "recv_t"):
            got = me[key]
            want = np.full((nb, world, blk), -7.0)
            for q in range(world):
                want[:, q, off:off + cnt] = parts[q]["send"][:, r, off:off + cnt]
            assert np.array_equal(got, want), (world, r, key)
