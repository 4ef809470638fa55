"""GPU: every selectable variant of the per-thread sweeps takes bit-identical steps (csrc/ddh_pencil.hip, round 6):
the plain kernels, the deep-prefetch kernels for few systems (solve_forward_deep_kernel / solve_backward_deep_kernel, two and
four register sets in the backward sweep) and the backward sweep through the per-wave LDS-DMA ring
(solve_backward_ring_kernel + solve_backward_kernel<..., 64> for the waves whose lane quads do not share a factorization),
two, three and four rows deep.  The switches are read once per process, so every variant steps in a subprocess; the plain
variant is also compared with the reference's end state (tests/golden/config_rb3d_endstate_128x128x64.npz).
Same arithmetic in the same order in all of them: the reference's per-pencil LU solves (libraries/matsolvers.py:126-149)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import problems
import dedalus_amd.public as d3
solver, f = problems.rayleigh_benard_3d(d3, Nx=%(nx)d, Ny=%(ny)d, Nz=%(nz)d, timestepper="RK222")
solver.pack.set_solve_variant(0)                     # one thread per (system, block): the kernels under test
for _ in range(3):
    solver.step(1e-3)
lu = sorted(solver._lu_params)[0]
info = solver.pack.lu_info(lu)
out = {k: hashlib.sha256(np.ascontiguousarray(np.asarray(f[k]["c"])).tobytes()).hexdigest() for k in ("p", "b", "u")}
out["norm_b"] = float(np.linalg.norm(np.asarray(f["b"]["c"])))
out["pair"] = info["pair"]; out["forward"] = info["forward"]; out["nsplit"] = info["nsplit"]
print("RESULT " + json.dumps(out))
"""


def _run(env_extra, shape):
    env = dict(os.environ, **env_extra)
    code = SCRIPT % dict(root=ROOT, nx=shape[0], ny=shape[1], nz=shape[2])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (env_extra, r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


VARIANTS = [{"DDH_SWEEP_DEEP": "1"}, {"DDH_SWEEP_DEEP": "1", "DDH_BWD_DEEP_PD": "4"},
            {"DDH_SWEEP_DEEP": "0", "DDH_BWD_RING": "2"}, {"DDH_SWEEP_DEEP": "0", "DDH_BWD_RING": "3"},
            {"DDH_SWEEP_DEEP": "0", "DDH_BWD_RING": "4"}]


@pytest.mark.parametrize("shape", [(128, 128, 64), (64, 96, 32)])
def test_sweep_variants_take_bit_identical_steps(shape):
    """128 x 128 x 64: partner pencils (x <-> y symmetric; the ring kernel's case); 64 x 96 x 32: unpaired factorizations (the
    ring kernel must stand aside, the deep kernels run)."""
    common = {"DDH_PAIR_MIN": "0"}                      # (pairing is reserved for >= 65 536 systems by default)
    base = _run(dict(common, DDH_SWEEP_DEEP="0", DDH_BWD_RING="0"), shape)
    assert base["forward"] == "lean" and base["nsplit"] == 2
    assert bool(base["pair"]) == (shape[0] == shape[1])
    for env in VARIANTS:
        got = _run(dict(common, **env), shape)
        for k in ("p", "b", "u"):
            assert got[k] == base[k], (shape, env, k, got["norm_b"], base["norm_b"])
    if shape == (128, 128, 64):
        G = np.load(os.path.join(ROOT, "tests", "golden", "config_rb3d_endstate_128x128x64.npz"))
        assert abs(base["norm_b"] - float(G["end__b_norm"])) < 1e-11 * float(G["end__b_norm"])
