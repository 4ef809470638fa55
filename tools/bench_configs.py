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
    solver.enable_step_graph(bool(graph))          # (the default replays launch-bound problems: pinned either way here)
    name += " [hipGraph]" if graph else " [ordinary launches]"
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


def run_shell(name, kw, dt, warm, steps, graph=False):
    """Under torch.distributed.run (one rank per GPU) the azimuthal wavenumbers are sharded over the ranks
    (mesh = (WORLD_SIZE,)): `python -m torch.distributed.run --nproc-per-node 4 ... tools/bench_configs.py shell`."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        kw = dict(kw, dist_kw=dict(mesh=(world,)))
        name += " x%d ranks" % world
    t0 = time.time()
    solver, f = problems.shell_convection(d3, **kw)
    if graph:
        solver.enable_step_graph(True)
        name += " [hipGraph]"
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


# ---- --json: steps/s + where the step goes + roofline of the dominant entry point, per secondary configuration -------------
FP64_MFMA_PEAK_TFLOPS = 78.6        # MI355X FP64 matrix peak (AMD spec; v_mfma_f64_16x16x4: 256 flop / clk / CU at 2.4 GHz)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured copy)


class EventProfiler:
    """libhip.profiler: HIP events around every C-ABI entry point (torch's current stream = the library's launch stream)."""

    def __init__(self, torch):
        self.torch, self.rec = torch, {}

    def __call__(self, name, fn, args):
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        st = fn(*args)
        e1.record()
        self.rec.setdefault(name, []).append((e0, e1))
        return st

    def summary(self):
        self.torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.rec.items()}


def measure_json(tag, make, step, warm, steps, repeats=3):
    """make() -> solver; step(solver) advances one timestep.  Clean timing first (median of `repeats`), then one more pass
    with the entry-point profiler and the wrappers' algorithmic cost notes attached."""
    import json
    import torch
    from dedalus_amd import libhip
    from dedalus_amd.executor import KernelTimer
    solver = make()
    for _ in range(warm):
        step(solver)
    rates = []
    for _ in range(repeats):
        solver.ex.sync()
        t0 = time.time()
        for _ in range(steps):
            step(solver)
        solver.ex.sync()
        rates.append(steps / (time.time() - t0))
    prof, costs = EventProfiler(torch), {}
    import ctypes as C
    w0 = C.c_long(0)
    libhip.call("ddh_fft_wave_launches", C.byref(w0))

    def cost_log(name, flops, nbytes):
        c = costs.setdefault(name, [0.0, 0.0])
        c[0] += flops
        c[1] += nbytes
    libhip.profiler, libhip.cost_log = prof, cost_log
    timer = KernelTimer(torch)
    solver.ex.timer = timer                         # (Cartesian executors: algorithmic bytes per kernel family)
    try:
        for _ in range(steps):
            step(solver)
        ents = prof.summary()
        fam = timer.summary()
    finally:
        libhip.profiler = libhip.cost_log = None
        solver.ex.timer = None
    table = {}
    for name, (n, ms) in sorted(ents.items(), key=lambda kv: -kv[1][1]):
        if ms / steps < 1e-3:
            continue
        row = dict(launches_per_step=n / steps, ms_per_step=ms / steps, ms_per_launch=ms / n)
        if name in costs and n:
            fl, by = costs[name][0] / n, costs[name][1] / n
            row.update(algorithmic_GFLOP_per_launch=fl / 1e9, algorithmic_GB_per_launch=by / 1e9)
            sec = ms / n / 1e3
            if by > 0 and fl / by >= FP64_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
                row.update(bound="mfma", achieved=fl / sec / 1e12, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s")
            elif by > 0:
                row.update(bound="hbm", achieved=by / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            if "achieved" in row:
                row["frac"] = row["achieved"] / row["peak"]
        table[name] = row
    out = dict(config=tag, steps_per_s=float(np.median(rates)), ms_per_step=1e3 / float(np.median(rates)),
               repeats=rates, steps=steps, launches_per_step=sum(n for n, _ in ents.values()) / steps,
               entry_point_ms_per_step=sum(ms for _, ms in ents.values()) / steps, entry_points=table)
    w1 = C.c_long(0)
    libhip.call("ddh_fft_wave_launches", C.byref(w1))
    out["wave_kernel_transforms_per_step"] = (w1.value - w0.value) / steps     # (of the Chebyshev / real-Fourier launches)
    bi = getattr(solver, "_binv", None)
    if isinstance(bi, dict) and "residual" in bi:
        out["block_inverse_residual_max"] = bi["residual"]                    # ||B^T X - I||_max of the sampled blocks
        out["block_inverse_left_residual_max"] = bi.get("left_residual")      # ||X B^T - I||_max ~ cond(B) eps
    if fam:
        out["kernel_families"] = {k: dict(launches_per_step=v["launches"] / steps, ms_per_launch=v["avg_ms"],
                                          algorithmic_GB_per_launch=v["bytes_per_launch"] / 1e9, GBps=v["gbps"],
                                          frac_of_hbm_peak=v["gbps"] / HBM_PEAK_GBS) for k, v in fam.items()}
    rated = [(k, v) for k, v in table.items() if "frac" in v]
    if rated:
        k, v = max(rated, key=lambda kv: kv[1]["ms_per_step"])
        out["dominant"] = dict(entry_point=k, **v)
        out["roofline"] = dict(bound=v["bound"], kernel=k, achieved=v["achieved"], peak=v["peak"], unit=v["unit"], frac=v["frac"])
    elif fam:
        k, v = max(fam.items(), key=lambda kv: kv[1]["total_ms"])
        out["roofline"] = dict(bound="hbm", kernel=k, achieved=v["gbps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=v["gbps"] / HBM_PEAK_GBS)
    print(json.dumps(out), flush=True)
    return out


def all_json():
    def k_make():
        return problems.kdv_burgers(d3, Nx=1024, timestepper="SBDF2")[0]
    measure_json("K kdv_burgers N=1024 SBDF2", k_make, lambda s: s.step(2e-3), 20, 200)

    def r2_make():
        return problems.rayleigh_benard_2d(d3, Nx=512, Nz=256)[0]
    measure_json("R2 rayleigh_benard 2-D 512x256 RK222", r2_make, lambda s: s.step(1e-3), 5, 100)
    sw = {}

    def s_make():
        solver, f, extra = problems.shallow_water(d3, Nphi=512, Ntheta=256)
        sw["dt"] = extra["timestep"]
        return solver
    measure_json("S shallow_water SphereBasis(512,256) RK222", s_make, lambda s: s.step(sw["dt"]), 5, 50)

    def h_make():
        return problems.shell_convection(d3, shape=(256, 128, 128), timestepper="SBDF2")[0]
    measure_json("H shell_convection ShellBasis(256,128,128) SBDF2", h_make, lambda s: s.step(0.05), 3, 10)


def offsize_json():
    """3-D Rayleigh-Benard at sizes OTHER than the benchmark's: which transform families run on the wave kernels and at what
    fraction of the HBM rate (`kernel_families`: algorithmic bytes / HIP-event time)."""
    for (nx, ny, nz), warm, steps in (((128, 128, 128), 3, 20), ((256, 256, 256), 3, 10), ((384, 384, 192), 3, 10),
                                      ((1024, 1024, 128), 2, 5)):
        def make(nx=nx, ny=ny, nz=nz):
            return problems.rayleigh_benard_3d(d3, Nx=nx, Ny=ny, Nz=nz, timestepper="RK222")[0]
        measure_json("3-D rayleigh_benard %dx%dx%d RK222" % (nx, ny, nz), make, lambda s: s.step(1e-3), warm, steps)


if __name__ == "__main__":
    if "--offsize" in sys.argv[1:]:
        offsize_json()
        sys.exit(0)
    if "--json" in sys.argv[1:]:
        all_json()
        sys.exit(0)
    if "shell" in sys.argv[1:]:
        shape = tuple(int(x) for x in os.environ.get("SHELL_SHAPE", "256,128,128").split(","))
        run_shell("H  shell convection %dx%dx%d SBDF2" % shape, dict(shape=shape, timestepper="SBDF2"), 0.05, 3, 10)
        if int(os.environ.get("WORLD_SIZE", "1")) == 1:
            run_shell("H  shell convection %dx%dx%d SBDF2" % shape, dict(shape=shape, timestepper="SBDF2"), 0.05, 8, 20, graph=True)
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
    if "graphs" in sys.argv[1:]:         # launch-bound configurations with the fixed-timestep steps replayed from HIP graphs
        run("K  kdv N=1024 SBDF2", problems.kdv_burgers, dict(Nx=1024, timestepper="SBDF2"), 2e-3, 20, 2000)
        run("K  kdv N=1024 SBDF2", problems.kdv_burgers, dict(Nx=1024, timestepper="SBDF2"), 2e-3, 20, 2000, graph=True)
        run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 200)
        run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 200, graph=True)
        os.environ["DDH_STEP_GRAPH"] = "1"
        run_sphere("S  shallow water 512x256 RK222 [hipGraph]", dict(Nphi=512, Ntheta=256), 5, 200)
        sys.exit(0)
    run("K  kdv N=1024 SBDF2", problems.kdv_burgers, dict(Nx=1024, timestepper="SBDF2"), 2e-3, 20, 200)
    run("R2 rb2d 512x256 RK222", problems.rayleigh_benard_2d, dict(Nx=512, Nz=256), 1e-3, 5, 50)
    run("rb3d 128x128x64 RK222", problems.rayleigh_benard_3d, dict(Nx=128, Ny=128, Nz=64), 1e-3, 3, 20)
