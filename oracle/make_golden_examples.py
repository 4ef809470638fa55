"""
TEST INFRASTRUCTURE ONLY.  The reference's own example scripts as parity fixtures.

    python oracle/make_golden_examples.py [kdv_burgers] [rayleigh_benard] [shallow_water] [shell_convection]

For each of the four BASELINE example scripts (configs 1, 2, 4, 5) this
  1. copies the script VERBATIM from /root/reference/examples/... to tests/golden/examples/<name>.py -- they are the
     workload definitions `north_star` says must run unmodified, kept as fixtures (the GPU box has no reference) with
     their sha256 in tests/golden/examples/MANIFEST.json; nothing under dedalus_amd/ imports them;
  2. runs the same file under the UNMODIFIED reference (oracle/refshim) for a fixed number of main-loop iterations
     (`solver.proceed` also stops at iteration K; writes of the analysis handlers are disabled -- h5py does not exist
     here and they do not touch the state) and stores the end state in tests/golden/examples.npz.
tests/test_gpu_examples.py runs the vendored files through dedalus_amd.compat on the GPU, stopped at the same iteration,
and compares with these arrays.
"""

import hashlib
import json
import os
import runpy
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
EXDIR = os.path.join(GOLD, "examples")

# name -> (path under /root/reference/examples, iterations, {saved name: (script variable, layout)})
CASES = {
    "kdv_burgers": ("ivp_1d_kdv_burgers/kdv_burgers.py", 60, {"u": ("u", "c")}),
    "rayleigh_benard": ("ivp_2d_rayleigh_benard/rayleigh_benard.py", 25, {"b": ("b", "c"), "u": ("u", "c"), "p": ("p", "c")}),
    "shallow_water": ("ivp_sphere_shallow_water/shallow_water.py", 3, {"u": ("u", "g"), "h": ("h", "g")}),
    "shell_convection": ("ivp_shell_convection/shell_convection.py", 3, {"u": ("u", "g"), "b": ("b", "g"), "p": ("p", "g")}),
}


def vendor(name):
    src = os.path.join("/root/reference/examples", CASES[name][0])
    dst = os.path.join(EXDIR, name + ".py")
    os.makedirs(EXDIR, exist_ok=True)
    shutil.copyfile(src, dst)
    return dict(source="examples/" + CASES[name][0], sha256=hashlib.sha256(open(dst, "rb").read()).hexdigest(),
                iterations=CASES[name][1])


def run_reference(name):
    from oracle import refshim
    refshim.load_reference()
    from dedalus.core import evaluator as ev
    from dedalus.core import solvers
    for n in dir(ev):                                   # analysis-set writes need h5py: disabled (output only)
        c = getattr(ev, n)
        if isinstance(c, type) and hasattr(c, "get_file") and hasattr(c, "process"):
            c.process = lambda self, **kw: None
    K = CASES[name][1]
    if not hasattr(solvers.InitialValueSolver, "_orig_proceed"):
        solvers.InitialValueSolver._orig_proceed = solvers.InitialValueSolver.proceed
    orig = solvers.InitialValueSolver._orig_proceed
    solvers.InitialValueSolver.proceed = property(lambda self: orig.fget(self) and self.iteration < K)
    os.environ.setdefault("MPLBACKEND", "Agg")
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            t0 = time.time()
            ns = runpy.run_path(os.path.join(EXDIR, name + ".py"), run_name="__main__")
            print("%s: %d iterations of the reference in %.1f s" % (name, ns["solver"].iteration, time.time() - t0))
        finally:
            os.chdir(cwd)
    out = {name + "__iteration": np.array(ns["solver"].iteration), name + "__sim_time": np.array(ns["solver"].sim_time)}
    for key, (var, layout) in CASES[name][2].items():
        f = ns[var]
        f.change_scales(1)
        out["%s__%s_%s" % (name, key, layout)] = np.array(f[layout])
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    path = os.path.join(GOLD, "examples.npz")
    data = dict(np.load(path)) if os.path.exists(path) else {}
    mpath = os.path.join(EXDIR, "MANIFEST.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    for name in names:
        manifest[name] = vendor(name)
        data.update(run_reference(name))
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    np.savez_compressed(path, **data)
    print("wrote", path, {k: v.shape for k, v in data.items()})
