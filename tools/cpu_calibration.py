"""Calibration of bench.py's `cpu_baseline` ("port" = oracle/np_executor, the restated reference algorithm) against the
TRUE reference (unmodified Dedalus through oracle/refshim) on the same problems, same machine, one core.
Works only where /root/reference exists (the build container); writes profiles/<tag>_cpu_port_vs_reference.json.

    OMP_NUM_THREADS=1 python tools/cpu_calibration.py [tag]
"""
import json
import os
import platform
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import problems  # noqa: E402

CASES = [("rb3d", dict(Nx=32, Ny=32, Nz=32)), ("rb3d", dict(Nx=64, Ny=64, Nz=32)), ("rb3d", dict(Nx=64, Ny=64, Nz=64)),
         ("rb2d", dict(Nx=512, Nz=256)), ("rb3d", dict(Nx=128, Ny=128, Nz=64))]      # (the last one: SURVEY 8d)


def host():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cpu_model=model, cores_total=os.cpu_count(), machine=platform.machine())


def time_case(d3, kind, kw, dist_kw=None, budget=15.0, max_steps=40):
    build = problems.rayleigh_benard_3d if kind == "rb3d" else problems.rayleigh_benard_2d
    t0 = time.time()
    solver, f = build(d3, timestepper="RK222", dist_kw=dist_kw, **kw)
    solver.step(1e-3)                                  # factorizations
    setup = time.time() - t0
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < budget and n < max_steps):
        solver.step(1e-3)
        n += 1
    el = time.time() - t0
    modes = 5 * int(np.prod(list(kw.values()))) if kind == "rb3d" else 4 * int(np.prod(list(kw.values())))
    return dict(case=kind, size=kw, setup_s=setup, steps=n, seconds=el, steps_per_s=n / el,
                mode_stages_per_cpu_s=modes * 2 * n / el, norm_b=float(np.linalg.norm(np.asarray(f["b"]["c"]))))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
    import dedalus_amd.public as d3p
    from oracle.np_executor import NumpyExecutor
    from oracle import refshim
    d3r = refshim.load_reference()
    out = dict(host=host(), threads=1, cases=[])
    for kind, kw in CASES:
        port = time_case(d3p, kind, kw, dist_kw=dict(executor=NumpyExecutor()))
        ref = time_case(d3r, kind, kw)
        out["cases"].append(dict(case=kind, size=kw, port=port, reference=ref,
                                 port_vs_reference=port["steps_per_s"] / ref["steps_per_s"]))
        print(kind, kw, "port %.3f steps/s, reference %.3f steps/s, ratio %.3f" %
              (port["steps_per_s"], ref["steps_per_s"], port["steps_per_s"] / ref["steps_per_s"]), flush=True)
    out["port_vs_reference_geomean"] = float(np.exp(np.mean([np.log(c["port_vs_reference"]) for c in out["cases"]])))
    path = os.path.join(ROOT, "profiles", "%s_cpu_port_vs_reference.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "geomean", out["port_vs_reference_geomean"])


if __name__ == "__main__":
    main()
