"""
Benchmark of the IMEX hot path: timesteps/sec of 3-D Rayleigh-Benard (Fourier x Fourier x Chebyshev,
RK222, fixed dt) -- BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--size NX,NY,NZ]

N=1 runs the largest configuration that fits one MI355X (the metric's 512x512x256 by default).
For N>1 (launched by torch.distributed.run, one rank per GPU) the pencils are sharded across ranks
(the metric's problem size is fixed, so this is "strong" scaling: total work constant).

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

PMC_FAMILY = {   # kernel family in this file -> device kernel name prefixes in profiles/r1d_pmc_traffic.json
    "pencil_solve": ["solve_forward_kernel", "solve_backward_kernel"],
    "pencil_matvec": ["matvec_kernel"],
    "rfft_bilinear_fused": ["gw::gridwave_bilinear_kernel", "fused_rfft_bilinear_kernel"],
    "rfft_backward_contig": ["fft_axis_kernel<1, false"],
    "rfft_backward_strided": ["fft_axis_kernel<1, true"],
    "rfft_forward_contig": ["fft_axis_kernel<0, false"],
    "rfft_forward_strided": ["fft_axis_kernel<0, true"],
    "cheb_forward_strided": ["fft_axis_kernel<2, true"],
    "cheb_backward_strided": ["fft_axis_kernel<3, true"],
    "grid_bilinear": ["bilinear_kernel"],
    "lincomb": ["lincomb_kernel"],
}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes
    (profiles/r1d_pmc_summary.txt: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes,
    full-size run; a family made of several kernels per call sums their per-launch means).
    Returns None when no measurement is on file."""
    path = os.path.join(ROOT, "profiles", "r1d_pmc_traffic.json")
    if not os.path.exists(path) or family not in PMC_FAMILY:
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
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=str, default=os.environ.get("BENCH_SIZE", "512,512,256"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dt", type=float, default=1e-3)
    return ap.parse_args()


def cpu_baseline(full_modes, budget_s=20.0):
    """The oracle executor (reference algorithm: per-pencil scipy CSR + SuperLU, scipy.fft + NumPy
    pack passes) on 3-D RB 32x32x32, one core.  Extrapolated to the metric's size by modes."""
    import problems
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    n = (32, 32, 32)
    solver, fields = problems.rayleigh_benard_3d(d3, Nx=n[0], Ny=n[1], Nz=n[2], dist_kw=dict(executor=NumpyExecutor()))
    solver.step(1e-3)                      # factorizations happen here
    t0 = time.time()
    steps = 0
    while time.time() - t0 < budget_s and steps < 200:
        solver.step(1e-3)
        steps += 1
    el = time.time() - t0
    sps = steps / el
    modes = 5 * n[0] * n[1] * n[2]
    extrap = sps * modes / full_modes
    return dict(value=extrap, unit="timesteps/sec (extrapolated to the metric's size by mode count)",
                cores=1, kind="port",
                sample="3-D RB %dx%dx%d RK222 dt=1e-3, %d steps in %.1f s = %.3f steps/s measured on 1 core; "
                       "x (modes %d / %d)" % (n[0], n[1], n[2], steps, el, sps, modes, full_modes),
                measured_steps_per_sec=sps)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        solver.step(args.dt)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.time() - t0
    ex.timer = None
    if world > 1:
        t = torch.tensor([el], device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    summ = timer.summary()
    chk2 = float(np.sum(np.asarray(fields["b"]["c"]) ** 2))
    if world > 1:
        t = torch.tensor([chk2], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t)
        chk2 = float(t.item())
    chk = float(np.sqrt(chk2))

    if rank == 0:
        steps_per_s = args.steps / el
        # dominant kernel family by total time
        dom = max(summ.items(), key=lambda kv: kv[1]["total_ms"]) if summ else (None, None)
        roof = None
        if dom[0]:
            roof = dict(bound="hbm", kernel=dom[0], achieved=dom[1]["gbps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=dom[1]["gbps"] / HBM_PEAK_GBS,
                        traffic=(pmc_traffic(dom[0]) if (Nx, Ny, Nz) == (512, 512, 256) and world == 1 else None),
                        traffic_note="bytes per launch from rocprofv3 PMC passes committed under profiles/ "
                                     "(r1d_pmc_summary.txt); algorithmic bytes and time are measured live",
                        avg_launch_ms=dom[1]["avg_ms"], algorithmic_bytes_per_launch=dom[1]["bytes_per_launch"],
                        launches=dom[1]["launches"])
        total_kernel_ms = sum(v["total_ms"] for v in summ.values())
        total_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in summ.values())
        out = {
            "metric": "timesteps/sec, 3D Rayleigh-Benard %dx%dx%d (Fourier x Fourier x Chebyshev, RK222)" % (Nx, Ny, Nz),
            "value": steps_per_s, "unit": "timesteps/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D Rayleigh-Benard IVP %dx%dx%d, dealias 3/2, RK222, fixed dt=%g, Ra=2e6 Pr=1, "
                                   "example-script initial condition (fill_random seed 42)" % (Nx, Ny, Nz, args.dt),
                       "pencils": (Nx // 2) * (Ny // 2), "rows_per_pencil_real": 4 * solver.R,
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs, pencils sharded on kx, RCCL all-to-all" % world},
            "roofline": roof,
            "whole_step": {"algorithmic_GB_per_step": total_bytes / args.steps / 1e9,
                           "kernel_ms_per_step": total_kernel_ms / args.steps,
                           "achieved_GBps_all_kernels": (total_bytes / 1e9) / (total_kernel_ms / 1e3) if total_kernel_ms else None},
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["gbps"], 1),
                            "total_ms": round(v["total_ms"], 2)} for k, v in sorted(summ.items())},
            "build_s": build_s, "checksum_b_c_l2": chk,
        }
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(5 * Nx * Ny * Nz)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
