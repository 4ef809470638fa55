"""Radial operators of the shell (dedalus_amd/tools/shellops.py) against the reference's
ShellRadialBasis.operator_matrix / conversion_matrix / jacobi_conversion / interpolation values
(tests/golden/shellops.npz: radii (0.7, 1.9), N = 7, k = 0..2, ell in {0, 1, 3}, regtotal in {-1, 0, 1})."""
import os

import numpy as np

from dedalus_amd.tools import shellops as so

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shellops.npz"))


def test_radial_operators_match_reference():
    radii, N = (0.7, 1.9), 7
    checked = 0
    for key in GOLD.files:
        parts = key.split("_")
        k = int(parts[0][1:])
        if parts[1].startswith("l"):
            ell, rt, op = int(parts[1][1:]), int(parts[2][2:]), parts[3]
            if op == "Z":
                continue
            mine = so.operator_matrix(op, ell, rt, N, k, radii)
        elif parts[1] == "conv1":
            mine = so.E(N, k, radii)
        elif parts[1] == "jconv1":
            mine = so.conversion(N, k - 0.5, k - 0.5)
        else:
            mine = so.interpolation(0.7 if parts[2] == "in" else 1.9, N, k, radii)
        ref = GOLD[key].reshape(mine.shape)
        assert np.abs(mine - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1.0), key
        checked += 1
    assert checked > 150
