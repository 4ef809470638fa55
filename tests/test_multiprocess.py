"""CPU, 2 processes (gloo): pencils sharded over ranks + all-to-all transposes reproduce the reference
end state (and therefore the single-process result) -- the direct multi-rank-vs-serial equality
test the reference's own suite lacks."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("case,world", [("rb3d_8x12x8_rk222", 2), ("rb2d_32x16_rk222", 2)])
def test_sharded_run_matches_reference(golden_dir, case, world):
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
