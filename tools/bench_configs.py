"""Secondary BASELINE.json configurations on one GPU (not the headline metric):
   K  = KdV-Burgers N=1024 SBDF2,  R2 = 2-D Rayleigh-Benard 512x256 RK222,
   S  = sphere shallow water SphereBasis(512, 256) RK222 (`python tools/bench_configs.py sphere`),
   H  = shell convection ShellBasis(256, 128, 128) SBDF2 on one GPU (`python tools/bench_configs.py shell`).
   Prints steps/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import problems  # noqa: E402
import dedalus_amd.public as d3  # noqa: E402


def run(name, builder, kw, dt, warm, steps, graph=False):
    solver, f = builder(d3, **kw)
    if graph:
        solver.enable_step_graph(True)
        name += " [hipGraph]"
    for _ in range(warm):
        solver.step(dt)
    solver.ex.sync()
    t0 = time.time()
    for _ in range(steps):
        solver.step(dt)
    solver.ex.sync()
    el = time.time() - t0
    print("%-28s %8.1f steps/s  (%.3f ms/step, %d steps)" % (name, steps / el, 1e3 * el / steps, steps), flush=True)


def run_sphere(name, kw, warm, steps):
    t0 = time.time()
    solver, f, extra = problems.shallow_water(d3, **kw)
    dt = extra["timestep"]
    solver.step(dt)
    solver.ex.sync()
    print("%-28s setup + LBVP + first step: %.1f s" % (name, time.time() - t0), flush=True)
    for _ in range(warm):
        solver.step(dt)
    solver.ex.sync()
    t0 = time.time()
    for _ in range(steps):
        solver.step(dt)
    solver.ex.sync()
    el = time.time() - t0
    print("%-28s %8.1f steps/s  (%.3f ms/step, %d steps)" % (name, steps / el, 1e3 * el / steps, steps), flush=True)
    refactor_time(name, solver, dt, el / steps)


def refactor_time(name, solver, dt, step_s):
    """cost of a timestep change: the LHS of every subproblem is re-formed and re-factored on the device"""
    ts = []
    for k in range(3):
        solver.ex.sync()
        t0 = time.time()
        solver.step(dt * (1.0 + 0.01 * (k + 1)))
        solver.ex.sync()
        ts.append(time.time() - t0 - step_s)
    print("%-34s refactorization on a timestep change: %.1f ms (min of 3; step time subtracted)" % (name, 1e3 * min(ts)),
          flush=True)


def run_shell(name, kw, dt, warm, steps):
    """Under torch.distributed.run (one rank per GPU) the azimuthal wavenumbers are sharded over the ranks
    (mesh = (WORLD_SIZE,)): `python -m torch.distributed.run --nproc-per-node 4 ... tools/bench_configs.py shell`."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        kw = dict(kw, dist_kw=dict(mesh=(world,)))
        name += " x%d ranks" % world
    t0 = time.time()
    solver, f = problems.shell_convection(d3, **kw)
    solver.step(dt)
    solver.ex.sync()
    print("%-34s setup + first step: %.1f s" % (name, time.time() - t0), flush=True)
    for _ in range(warm):
        solver.step(dt)
    solver.ex.sync()
    t0 = time.time()
    for _ in range(steps):
        solver.step(dt)
    solver.ex.sync()
    el = time.time() - t0
    b = np.asarray(f["b"]['c'])
    nrm2 = float(np.sum(b * b))
    if world > 1:
        nrm2 = solver.dist.pcomm.allreduce_sum(nrm2)
        el = solver.dist.pcomm.allreduce_max(el)
    if int(os.environ.get("RANK", "0")) == 0:
        print("%-34s %8.2f steps/s  (%.2f ms/step, %d steps)  |b_c| = %.12f finite=%s" % (
            name, steps / el, 1e3 * el / steps, steps, np.sqrt(nrm2), bool(np.isfinite(b).all())), flush=True)
    if world == 1:
        refactor_time(name, solver, dt, el / steps)


if __name__ == "__main__":
    if "shell" in sys.argv[1:]:
        shape = tuple(int(x) for x in os.environ.get("SHELL_SHAPE", "256,128,128").split(","))
        run_shell("H  shell convection %dx%dx%d SBDF2" % shape, dict(shape=shape, timestepper="SBDF2"), 0.05, 3, 10)
        sys.exit(0)
    if "r2" in sys.argv[1:]:
        run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 50)
        run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 200, graph=True)
        run("rb3d 64x64x32 RK222", problems.rayleigh_benard_3d, dict(Nx=64, Ny=64, Nz=32), 1e-3, 5, 100)
        run("rb3d 64x64x32 RK222", problems.rayleigh_benard_3d, dict(Nx=64, Ny=64, Nz=32), 1e-3, 5, 100, graph=True)
        sys.exit(0)
    if "sphere" in sys.argv[1:]:
        run_sphere("S  shallow water 512x256 RK222", dict(Nphi=512, Ntheta=256), 5, 50)
        sys.exit(0)
    run("K  kdv N=1024 SBDF2", problems.kdv_burgers, dict(Nx=1024, timestepper="SBDF2"), 2e-3, 20, 200)
    run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 50)
    run("rb3d 128x128x64 RK222", problems.rayleigh_benard_3d, dict(Nx=128, Ny=128, Nz=64), 1e-3, 3, 20)
