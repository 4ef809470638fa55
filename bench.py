"""
Benchmark of the IMEX hot path: timesteps/sec of 3-D Rayleigh-Benard (Fourier x Fourier x Chebyshev,
RK222, fixed dt) -- BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--size NX,NY,NZ]

N=1 runs the largest configuration that fits one MI355X (the metric's 512x512x256 by default).
For N>1 the pencils are sharded across N ranks, one per GPU, exchanging through RCCL (the metric's problem size is
fixed, so this is "strong" scaling: total work constant).  `python bench.py --gpus N` launches the N ranks itself
(re-executing under torch.distributed.run on 127.0.0.1) when it was not already started by a launcher.

Rank 0 prints ONE JSON line with the timing, the roofline of the dominant kernel (algorithmic bytes
per launch / HIP-event time, measured inside the timed region) and a CPU baseline (the numpy/scipy
oracle executor -- the reference's per-pencil algorithm -- on a bounded sample).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PMC_FAMILY = {   # kernel family in this file -> device kernel name prefixes in profiles/r5_pmc_traffic.json
    "pencil_solve": ["solve_forward_", "solve_backward_"],
    "pencil_matvec": ["band_matvec_kernel"],
    "rfft_bilinear_fused": ["gw2::gridwave2_bilinear_kernel", "gw::gridwave_bilinear_kernel", "fused_rfft_bilinear_kernel"],
    "rfft_backward_contig": ["fft_axis_kernel<1, false"],
    "rfft_backward_strided": ["wave_rfft_kernel<0"],
    "rfft_backward_strided_dual": ["wave_rfft_kernel<2"],
    "rfft_forward_contig": ["fft_axis_kernel<0, false"],
    "rfft_forward_strided": ["wave_rfft_kernel<3"],
    "cheb_forward_strided": ["wave_cheb_kernel<3"],
    "cheb_backward_strided": ["wave_cheb_kernel<0"],
    "cheb_backward_strided_dual": ["wave_cheb_kernel<1"],
    "grid_bilinear": ["bilinear_kernel"],
    "lincomb": ["lincomb_kernel"],
}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes
    (profiles/r6_pmc_summary.txt, else the newest older one: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes,
    full-size run; a family made of several kernels per call sums their per-launch means).
    Returns None when no measurement is on file."""
    path = None
    for tag in ("r6", "r5", "r4"):
        cand = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)
        if os.path.exists(cand):
            path = cand
            break
    if path is None or family not in PMC_FAMILY:
        return None
    data = json.load(open(path))
    tot = 0.0
    prefixes = PMC_FAMILY[family]
    if family == "rfft_bilinear_fused":         # one of the two implementations runs
        prefixes = [p for p in prefixes if any(k.startswith(p) for k in data)][:1]
        if not prefixes:
            return None
    for prefix in prefixes:
        hits = [v for k, v in data.items() if k.startswith(prefix)]
        if not hits:
            return None
        n = sum(h["launches"] for h in hits)
        tot += sum((h["read_GB_mean"] + h["write_GB_mean"]) * h["launches"] for h in hits) / n * 1e9
    return tot


HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=10)     # (the reference's warmup_iterations, core/solvers.py:546)
    ap.add_argument("--repeats", type=int, default=3,
                    help="the K timed steps are repeated this many times back to back (each repeat bracketed by its own "
                         "barrier + synchronize); the line reports the MEDIAN repeat and lists all of them")
    ap.add_argument("--size", type=str, default=os.environ.get("BENCH_SIZE", "512,512,256"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=120.0,
                    help="budget of the one-core cpu_baseline samples; >= 100 (default 120) includes 128x128x64, the sample the value "
                         "extrapolates from (~1 min of setup + ~7 s per step)")
    ap.add_argument("--cpu-replicas", type=int, default=None,
                    help="one-core replicas of the 64^3 sample run at once (default: min(32, usable cores / 2))")
    ap.add_argument("--cpu-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-seconds", type=float, default=5.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-which", type=str, default="port", help=argparse.SUPPRESS)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--cfl", action="store_true", help="(kept for old command lines: the adaptive loop now runs by default)")
    ap.add_argument("--no-cfl", action="store_true",
                    help="skip the second reported mode: after the fixed-dt measurement (the metric) the example's adaptive "
                         "loop -- d3.CFL(cadence=10, threshold=0.05) on an O(1) flow, i.e. with LHS refactorizations -- is "
                         "timed and reported as `cfl_mode` beside the fixed-dt value (one GPU only)")
    ap.add_argument("--cfl-steps", type=int, default=40)
    ap.add_argument("--emulate-rank", type=str, default=None, metavar="r/P",
                    help="time rank r's share of the P-rank run on ONE GPU with a loop-back exchange (tools/rank_emulation.py); "
                         "not the metric")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("BENCH_DRY_LAUNCH"):          # (tests: show the launch line instead of running it)
        print(json.dumps(cmd))
        sys.exit(0)
    os.execvpe(cmd[0], cmd, env)


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return dict(cpu_model=model, cores_total=os.cpu_count(), cores_usable=usable)


CPU_CASES = {   # name -> (kind, size kwargs, share of the budget)
    "rb3d 32x32x32": ("rb3d", dict(Nx=32, Ny=32, Nz=32), 0.15),
    "rb3d 64x64x32": ("rb3d", dict(Nx=64, Ny=64, Nz=32), 0.2),
    "rb3d 64x64x64": ("rb3d", dict(Nx=64, Ny=64, Nz=64), 0.3),
    "rb2d 512x256": ("rb2d", dict(Nx=512, Nz=256), 0.25),
    "rb3d 128x128x64": ("rb3d", dict(Nx=128, Ny=128, Nz=64), 1.0),       # the sample `value` extrapolates from
}


def cpu_sample(name, seconds, which="port"):
    """One bounded CPU sample on ONE core: build, first step (factorizations), then >= 2 timed RK222 steps for about
    `seconds`.  which = "port": the oracle executor (oracle/np_executor.py: the reference's per-pencil algorithm restated
    -- scipy CSR + SuperLU per pencil, scipy.fft + NumPy pack passes); "reference": the UNMODIFIED reference through
    oracle/refshim (only where /root/reference exists, i.e. never on the GPU box)."""
    import problems
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    kind, kw, _ = CPU_CASES[name]
    if which == "reference":
        from oracle import refshim
        d3 = refshim.load_reference()
        dist_kw = None
    else:
        import dedalus_amd.public as d3
        from oracle.np_executor import NumpyExecutor
        dist_kw = dict(executor=NumpyExecutor())
    build = problems.rayleigh_benard_3d if kind == "rb3d" else problems.rayleigh_benard_2d
    t0 = time.time()
    solver, fields = build(d3, timestepper="RK222", dist_kw=dist_kw, **kw)
    solver.step(1e-3)                      # factorizations happen here
    setup = time.time() - t0
    t0 = time.time()
    steps = 0
    while steps < 2 or (time.time() - t0 < seconds and steps < 100):
        solver.step(1e-3)
        steps += 1
    el = time.time() - t0
    nvar = 5 if kind == "rb3d" else 4
    modes = nvar * int(np.prod(list(kw.values())))
    return dict(case=name, which=which, steps=steps, seconds=round(el, 2), setup_s=round(setup, 2), steps_per_s=steps / el,
                modes=modes, mode_stages_per_cpu_s=modes * 2 * steps / el,
                norm_b=float(np.linalg.norm(np.asarray(fields["b"]["c"]))))


def cpu_replicas(name, seconds, n, which="port"):
    """n independent one-core replicas of a sample running AT THE SAME TIME (one process each, `python bench.py
    --cpu-worker`): what n cores of this host deliver together on the reference's algorithm when nothing is
    communicated -- an upper bound for an n-rank MPI run of the reference, and the measure of the memory-bandwidth
    contention a single-core sample hides."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", name, "--cpu-worker-seconds",
                               str(seconds), "--cpu-worker-which", which], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True, env=env) for _ in range(n)]
    res = []
    for p in procs:
        out, _ = p.communicate()
        try:
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pass
    return res


def cpu_baseline(full_shape, budget_s=120.0, replicas=None):
    """CPU baseline on this box's host cores (rank 0, N = 1 only), bounded samples of the same problem family:
    3-D RB 32^3, 64x64x32, 64^3, **128x128x64** (the largest the budget allows: ~1 min of setup + three steps of ~7 s) and
    the 2-D config 512x256:
      * one core, one sample after the other ("port" = the oracle executor; plus the unmodified reference on small
        samples wherever /root/reference exists: `reference_here`);
      * `size_trend`: mode-stages per cpu-second of every 3-D sample -- the port's (and the reference's) rate FALLS with
        size (SuperLU fill, cache misses), so the extrapolation starts from the LARGEST sample (`extrapolated_from`) and is
        still an over-estimate of the CPU at the metric's size;
      * `replicas`: n one-core copies of the 64^3 sample at once -> what n cores deliver together when nothing is
        communicated (an upper bound for an n-rank MPI run); `per_core_vs_alone` is the memory-contention factor.
    The metric's size is far beyond host memory for the reference's algorithm, so
        value = n cores x per_core_vs_alone x (mode-stages per cpu-second of the largest sample) / (2 stages x modes of the metric)
    with the reference's own speed figure (core/solvers.py:755-778).  `port_vs_reference` is the calibration of this port
    against the unmodified reference AT THE LARGEST SAMPLE (profiles/r5_cpu_port_vs_reference.json, tools/cpu_calibration.py:
    same machine, same sample, one core) -- not a mean over small cases."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    host = host_info()
    names = ["rb3d 32x32x32", "rb3d 64x64x32", "rb3d 64x64x64", "rb2d 512x256"]
    not_run = None
    if budget_s >= 100:
        names.append("rb3d 128x128x64")
    else:
        not_run = ("rb3d 128x128x64 (SURVEY 8d): ~1 min of setup + ~7 s per step on one core, beyond --cpu-budget %g s "
                   "(default 120)" % budget_s)
    base = min(budget_s, 30.0)
    samples = [cpu_sample(n, CPU_CASES[n][2] * base * 0.5) for n in names]
    s3 = [x for x in samples if x["case"].startswith("rb3d")]
    big = max(s3, key=lambda x: x["modes"])                                  # the largest 3-D sample
    s64 = [x for x in samples if x["case"] == "rb3d 64x64x64"][0]
    full_modes = 5 * int(np.prod(full_shape))
    one_core = big["mode_stages_per_cpu_s"] / (2 * full_modes)
    size_trend = [dict(case=x["case"], modes=x["modes"], mode_stages_per_cpu_s=round(x["mode_stages_per_cpu_s"], 1)) for x in s3]
    # the reference itself, where it exists (the build container; the GPU box has no /root/reference)
    reference_here = None
    try:
        from oracle import refshim
        if refshim.available():
            reference_here = [cpu_sample(n, CPU_CASES[n][2] * base * 0.5, which="reference")
                              for n in ("rb3d 32x32x32", "rb3d 64x64x64")]     # (its setup alone is ~80 s at 64^3)
    except Exception as e:                                   # noqa: BLE001 -- a checker that fails must not fail the bench
        reference_here = "failed: %s" % (e,)
    # n replicas at once
    if replicas is None:
        replicas = max(1, min(32, host["cores_usable"] // 2))
    rep = None
    cores, value, kind = 1, one_core, "port"
    if replicas > 1:
        rs = cpu_replicas("rb3d 64x64x64", CPU_CASES["rb3d 64x64x64"][2] * base * 0.5, replicas)
        if rs:
            agg = sum(r["mode_stages_per_cpu_s"] for r in rs)
            eff = (agg / len(rs)) / s64["mode_stages_per_cpu_s"]
            rep = dict(case="rb3d 64x64x64", replicas_started=replicas, replicas_finished=len(rs),
                       per_replica_steps_per_s=[round(r["steps_per_s"], 4) for r in rs],
                       aggregate_mode_stages_per_s=agg, per_core_vs_alone=eff)
            cores, value = len(rs), len(rs) * eff * big["mode_stages_per_cpu_s"] / (2 * full_modes)
    calib = None
    for tag in ("r6", "r5"):
        cpath = os.path.join(ROOT, "profiles", "%s_cpu_port_vs_reference.json" % tag)
        if os.path.exists(cpath):
            cj = json.load(open(cpath))
            per_case = {"%s %s" % (c["case"], "x".join(str(v) for v in c["size"].values())): round(c["port_vs_reference"], 3)
                        for c in cj["cases"]}
            at_big = [v for k, v in per_case.items() if "128x128x64" in k]
            calib = dict(at_largest_sample=(at_big[0] if at_big else None), geomean_all_cases=cj["port_vs_reference_geomean"],
                         measured_on=cj["host"]["cpu_model"], per_case=per_case,
                         source="profiles/%s_cpu_port_vs_reference.json" % tag)
            break
    ratio = (calib or {}).get("at_largest_sample") if big["case"].endswith("128x128x64") else None
    return dict(value=value, unit="timesteps/sec (extrapolated to the metric's size with mode-stages per cpu-second)",
                cores=cores, kind=kind, host=host, cores_total=os.cpu_count(), one_core_value=one_core,
                extrapolated_from=big["case"], size_trend=size_trend,
                size_trend_note="the rate falls with size: extrapolating from the largest sample still overstates the CPU at "
                                "the metric's size (64 x more modes per pencil system and 16 x more pencils)",
                replicas=rep, not_run=not_run, reference_here=reference_here,
                sample="%d bounded samples on 1 core, RK222 dt=1e-3: " % len(samples) + "; ".join(
                    "%s: %d steps in %.1f s = %.3f steps/s" % (x["case"], x["steps"], x["seconds"], x["steps_per_s"]) for x in samples)
                       + ("; then %d one-core replicas of 64x64x64 at once (contention factor %.2f)" % (cores, rep["per_core_vs_alone"]) if rep else "")
                       + "; value = %d core(s) x contention factor x mode-stages/s of the %s sample / (2 stages x %d modes)"
                       % (cores, big["case"], full_modes),
                samples=samples, port_vs_reference=calib,
                reference_estimate=(value / ratio) if ratio else None,
                reference_estimate_note="value / (port-vs-reference ratio measured at the extrapolation's own sample)")


def parity_check(solver, dt):
    """One extra (untimed) step at the metric's size with every solve of the sampled pencils compared with the
    REFERENCE's own pencil matrices (tests/pencil_check.py, tests/golden/pencils_nz256.npz)."""
    import pencil_check
    ref = pencil_check.ReferencePencils()
    lo = solver.dist._mx_offset
    hi = lo + solver.nx // 2
    mine = [g for g in ref.groups if lo <= ref.modes(g)[0] < hi]
    solver.solve_probe = dict(groups=[ref.modes(g) for g in mine], records=[])
    solver.step(dt)
    recs = solver.solve_probe["records"]
    solver.solve_probe = None
    res = pencil_check.check_records(ref, recs, mine) if mine else []
    return dict(pencils_checked=len(mine), solves=len(recs), solve_paths=sorted({r.get("path") for r in recs}),
                max_residual=max((r["residual"] for r in res), default=0.0),
                max_solution_error=max((r["solution"] for r in res), default=0.0),
                max_invalid_mode=max((r["dropped_max"] for r in res), default=0.0))


def adaptive_loop(d3, solver, fields, args, torch):
    """The reference example's main loop (examples/ivp_2d_rayleigh_benard/rayleigh_benard.py:97-113): the timestep comes
    from d3.CFL(cadence=10, safety=0.5, threshold=0.05, max_change=1.5, min_change=0.5) and every change re-forms and
    re-factors every pencil's LHS (core/timesteppers.py:630-640).  The example's initial velocity is zero (the CFL limit
    never binds), so an O(1) cellular flow is put into u first: the timestep then moves at (nearly) every cadence.
    Not the metric -- the fixed-dt value is -- but what an adaptive run of the same problem costs."""
    u = fields["u"]
    ug = np.array(u["g"])
    nx, ny, nz = ug.shape[1:]
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    sz = np.sin(np.pi * (iz + 0.5) / nz)
    ug[0] = 0.5 * np.sin(2 * np.pi * ix / nx) * sz * np.cos(2 * np.pi * iy / ny)
    ug[1] = 0.5 * np.cos(2 * np.pi * ix / nx) * sz * np.sin(2 * np.pi * iy / ny)
    ug[2] = 0.25 * np.cos(2 * np.pi * ix / nx) * np.cos(2 * np.pi * iy / ny) * sz ** 2
    u["g"] = ug
    cfl = d3.CFL(solver, initial_dt=args.dt, cadence=10, safety=0.5, threshold=0.05, max_change=1.5, min_change=0.5,
                 max_dt=0.125)
    cfl.add_velocity(u)
    for _ in range(2):
        solver.step(cfl.compute_timestep())
    torch.cuda.synchronize()
    t0 = time.time()
    dts = []
    for _ in range(args.cfl_steps):
        dt = cfl.compute_timestep()
        dts.append(float(dt))
        solver.step(dt)
    torch.cuda.synchronize()
    el = time.time() - t0
    changes = int(sum(1 for a, b in zip(dts[:-1], dts[1:]) if a != b))
    finite = bool(np.isfinite(np.asarray(fields["b"]["c"])).all())
    return dict(steps=args.cfl_steps, ms_per_step=1e3 * el / args.cfl_steps, value=args.cfl_steps / el,
                unit="timesteps/sec", timestep_changes=changes, dt_first=dts[0], dt_last=dts[-1], finite=finite,
                what="d3.CFL(cadence=10, threshold=0.05, safety=0.5) loop of the reference example on an O(1) flow; every "
                     "timestep change refactors all pencils (factor kernel + device inversion of the mean-mode pencil)")


def main():
    args = parse()
    if args.cpu_worker:                          # one replica of cpu_replicas(): no GPU, no torch
        print(json.dumps(cpu_sample(args.cpu_worker, args.cpu_worker_seconds, args.cpu_worker_which)))
        return
    if args.emulate_rank:                        # one rank of a sharded run on one GPU: its own tool, its own output
        r, P = args.emulate_rank.split("/")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "rank_emulation.py"), "--ranks", P, "--rank", r, "--size", args.size,
               "--steps", str(args.steps), "--warmup", str(min(args.warmup, 5)), "--dt", str(args.dt)]
        os.execv(cmd[0], cmd)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                   # does not return
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("DDH_FORCE_DEVICE") is not None:      # test aid: several ranks on one GPU (gloo)
            local_rank = int(os.environ["DDH_FORCE_DEVICE"])
            os.environ["LOCAL_RANK"] = str(local_rank)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("DDH_DIST_BACKEND", "nccl"))
    Nx, Ny, Nz = [int(s) for s in args.size.split(",")]
    import problems
    import dedalus_amd.public as d3
    from dedalus_amd.executor import KernelTimer

    dist_kw = {}
    if world > 1:
        dist_kw["mesh"] = (world,)
    t0 = time.time()
    solver, fields = problems.rayleigh_benard_3d(d3, Nx=Nx, Ny=Ny, Nz=Nz, timestepper="RK222", dist_kw=dist_kw)
    ex = solver.ex
    ex.sync()
    build_s = time.time() - t0

    for _ in range(args.warmup):
        solver.step(args.dt)
    ex.sync()
    if world > 1:
        torch.distributed.barrier()
    timer = KernelTimer(torch)
    ex.timer = timer
    if world > 1:
        solver.dist.pcomm.via = {}
        solver.dist.pcomm.wire_events = []
    els = []
    for _ in range(max(1, args.repeats)):               # every repeat: exactly K steps between barrier + synchronize
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.time()
        for _ in range(args.steps):
            solver.step(args.dt)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        e = time.time() - t0
        if world > 1:
            t = torch.tensor([e], device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            e = float(t.item())
        els.append(e)
    ex.timer = None
    if world > 1:
        wire_events = solver.dist.pcomm.wire_events            # (later steps -- parity probe -- add no events)
        solver.dist.pcomm.wire_events = None
        via_counts = {k: list(v) for k, v in solver.dist.pcomm.via.items()}
    el = float(np.median(els))
    nrep = len(els)
    summ = timer.summary()
    ranks_seen, backend, exch = 1, None, None
    if world > 1:
        ranks_seen = torch.distributed.get_world_size()
        backend = torch.distributed.get_backend()
        pc = solver.dist.pcomm
        nst = args.steps * nrep                       # (the per-path counters were reset after the warm-up)
        # the code path every exchange of the timed steps really took (parallel.Comm.via), not what could have been used
        via = {k: dict(exchanges_per_step=v[0] / nst, wire_MB_per_rank_per_step=v[1] / nst / 1e6) for k, v in via_counts.items()}
        wire_bytes = sum(v[1] for v in via_counts.values()) / nst
        wire_ms = None
        if wire_events:
            wire_ms = sum(a.elapsed_time(b) for a, b in wire_events) / nst
        kern_ms = sum(v["total_ms"] for k, v in summ.items()) / nst
        step_ms = 1e3 * el / args.steps
        overlap = None
        if wire_ms:
            overlap = max(0.0, min(1.0, (kern_ms + wire_ms - step_ms) / wire_ms))
        exch = dict(via=via, per_rank_wire_bytes_per_step=wire_bytes, exchanges_per_step=sum(v[0] for v in via_counts.values()) / nst,
                    wire_ms_per_step_side_stream_rank0=wire_ms, timed_kernel_ms_per_step_rank0=kern_ms,
                    overlap_fraction=overlap,
                    grid_stage_windows=int(solver._grid_windows()) if hasattr(solver, "_grid_windows") else 1,
                    grid_stage_windows_note="the grid stage (x backward, fused y, x forward) runs in this many windows of the "
                                            "rank's z planes, each window's exchange parts on the wire while another "
                                            "window computes (DDH_A2A_WINDOWS; DESIGN.md section 6)",
                    overlap_note="(timed kernels + side-stream exchange time - step time) / exchange time on rank 0, clamped to "
                                 "[0, 1]; the exchange time is HIP events around ddh_comm_alltoall(_part) on the side stream",
                    per_link_GBps=(wire_bytes / max(world - 1, 1) / 1e9) / (wire_ms / 1e3) if wire_ms else None)
    chk2 = float(np.sum(np.asarray(fields["b"]["c"]) ** 2))
    if world > 1:
        t = torch.tensor([chk2], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t)
        chk2 = float(t.item())
    parity = None
    if (Nx, Ny, Nz) == (512, 512, 256) and not args.no_parity:
        parity = parity_check(solver, args.dt)
        if world > 1:
            vals = torch.tensor([parity["max_residual"], parity["max_solution_error"], parity["max_invalid_mode"]],
                                device="cuda", dtype=torch.float64)
            cnt = torch.tensor([float(parity["pencils_checked"])], device="cuda", dtype=torch.float64)
            torch.distributed.all_reduce(vals, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(cnt)
            parity.update(max_residual=float(vals[0]), max_solution_error=float(vals[1]),
                          max_invalid_mode=float(vals[2]), pencils_checked=int(cnt.item()))
        parity["what"] = ("every solve of one extra step -- issued through the SAME fused, masked entry point the timed steps "
                          "use (solve_paths) -- on a 4x4 sample of pencils (modes 0, 85, 170, 255) against the reference's own "
                          "M_min/L_min (SuperLU solution and residual); the right-hand side of the check is formed "
                          "independently (unmasked lincomb); an intermediate stage is checked on the unknowns it stores; "
                          "tests/pencil_check.py")
    chk = float(np.sqrt(chk2))
    cfl_mode = adaptive_loop(d3, solver, fields, args, torch) if (not args.no_cfl and world == 1) else None

    if rank == 0:
        steps_per_s = args.steps / el
        # dominant kernel family by total time
        comp = {k: v for k, v in summ.items() if k != "a2a_exchange"}      # (the exchange is reported separately)
        dom = max(comp.items(), key=lambda kv: kv[1]["total_ms"]) if comp else (None, None)
        roof = None
        if dom[0]:
            roof = dict(bound="hbm", kernel=dom[0], achieved=dom[1]["gbps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=dom[1]["gbps"] / HBM_PEAK_GBS,
                        traffic=(pmc_traffic(dom[0]) if (Nx, Ny, Nz) == (512, 512, 256) and world == 1 else None),
                        traffic_note="bytes per launch from rocprofv3 PMC passes committed under profiles/ "
                                     "(r6_pmc_summary.txt); algorithmic bytes and time are measured live",
                        avg_launch_ms=dom[1]["avg_ms"], algorithmic_bytes_per_launch=dom[1]["bytes_per_launch"],
                        launches=dom[1]["launches"])
        # the fused y stage is co-limited by FP64 issue: its algorithmic flops per launch (a real transform of N points =
        # 2.5 N log2 N; 15 backward + 4 forward per line; 12 product terms of 3 flops per grid point) against the vector
        # FP64 peak (MI355X_MICROARCH.md: 78.6 TFLOP/s).  The kernel issues ~1.5 x these flops' worth of instructions
        # (pre- / post-processing, twiddles, address arithmetic; profiles/r5_fused_sq_counters.txt: 6405 VALU per line).
        if roof and roof["kernel"] == "rfft_bilinear_fused" and world == 1:
            gy, nl = 3 * Ny // 2, (3 * Nz // 2) * (3 * Nx // 2)
            fl = nl * (19 * 2.5 * gy * np.log2(gy) + 12 * 3 * gy)
            tf = fl / (roof["avg_launch_ms"] * 1e-3) / 1e12
            roof["valu"] = dict(bound="fp64-valu", algorithmic_flops_per_launch=fl, achieved=tf, peak=78.6, unit="TFLOP/s",
                                frac=tf / 78.6, issued_valu_frac_note="6405 wave instructions per line x 4 cycles on 1024 "
                                "SIMDs = 3.1 ms of pure VALU issue per launch (profiles/r5_fused_sq_counters.txt)")
            roof["power_note"] = ("this kernel runs at the board's power limit: 1362-1368 W and a shader clock of 2.02-2.05 GHz "
                                  "on real operands, 1250 W at 2.39 GHz and 5.3 instead of 6.4 ms on all-zero operands "
                                  "(profiles/r5_fused_clocks.txt, rocm-smi sampled beside tools/bench_fused.py); the peaks above "
                                  "assume 2.4 GHz")
        total_kernel_ms = sum(v["total_ms"] for v in summ.values())
        total_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in summ.values())
        out = {
            "metric": "timesteps/sec, 3D Rayleigh-Benard %dx%dx%d (Fourier x Fourier x Chebyshev, RK222)" % (Nx, Ny, Nz),
            "value": steps_per_s, "unit": "timesteps/sec", "n_gpus": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "repeats": {"n": nrep, "steps_per_s": [args.steps / e for e in els], "reported": "median"},
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D Rayleigh-Benard IVP %dx%dx%d, dealias 3/2, RK222, fixed dt=%g, Ra=2e6 Pr=1, "
                                   "example-script initial condition (fill_random seed 42)" % (Nx, Ny, Nz, args.dt),
                       "pencils": (Nx // 2) * (Ny // 2), "rows_per_pencil_real": 4 * solver.R,
                       "layouts": {"state_vector": "tile-major" if getattr(solver, "x_tiled", 0) else "natural",
                                   "right_hand_sides": "tile-major" if getattr(solver.timestepper, "_tiled", 0) else "natural"},
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs, pencils sharded on kx, RCCL all-to-all" % world},
            "roofline": roof,
            "whole_step": {"algorithmic_GB_per_step": total_bytes / (args.steps * nrep) / 1e9,
                           "kernel_ms_per_step": total_kernel_ms / (args.steps * nrep),
                           "achieved_GBps_all_kernels": (total_bytes / 1e9) / (total_kernel_ms / 1e3) if total_kernel_ms else None},
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["gbps"], 1),
                            "total_ms": round(v["total_ms"], 2)} for k, v in sorted(summ.items())},
            "build_s": build_s, "checksum_b_c_l2": chk, "parity": parity,
            "ranks_seen": ranks_seen, "dist_backend": backend, "exchange": exch,
        }
        if cfl_mode is not None:
            out["cfl_mode"] = cfl_mode
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline((Nx, Ny, Nz), args.cpu_budget, args.cpu_replicas)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
