"""
TEST INFRASTRUCTURE ONLY.

Compile the two dependency-free Cython units of the reference from where they lie:
    /root/reference/dedalus/tools/linalg.pyx                   (CSR apply / upper solve)
    /root/reference/dedalus/libraries/spin_recombination.pyx   (spin recombination)
Outputs (generated .c, objects, .so) go ONLY to ``oracle/_ref/`` which is git-ignored.
No reference source is copied into the repository.  Does nothing where the reference
is absent (GPU box).
"""

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("DEDALUS_REFERENCE_PATH", "/root/reference")

UNITS = {
    "linalg": os.path.join(REF, "dedalus", "tools", "linalg.pyx"),
    "spin_recombination": os.path.join(REF, "dedalus", "libraries", "spin_recombination.pyx"),
}


def build(verbose=False):
    if not os.path.isdir(os.path.join(REF, "dedalus")):
        return False
    try:
        import Cython  # noqa: F401
        import numpy as np
    except ImportError:
        return False
    os.makedirs(OUT, exist_ok=True)
    ext_suffix = sysconfig.get_config_var("EXT_SUFFIX")
    inc_py = sysconfig.get_paths()["include"]
    ok = True
    for stem, pyx in UNITS.items():
        so = os.path.join(OUT, stem + ext_suffix)
        if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(pyx):
            continue
        c_file = os.path.join(OUT, stem + ".c")
        cmds = [
            [sys.executable, "-m", "cython", "-3", pyx, "-o", c_file],
            ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-Wno-unused-function",
             "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
             "-I", inc_py, "-I", np.get_include(), c_file, "-o", so],
        ]
        for cmd in cmds:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                ok = False
                if verbose:
                    print("FAILED:", " ".join(cmd), "\n", r.stderr[-2000:])
                break
    return ok


if __name__ == "__main__":
    print("oracle/_ref built:", build(verbose=True))
