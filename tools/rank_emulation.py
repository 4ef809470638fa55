"""Rank r of P of the sharded 3-D Rayleigh-Benard run, timed on ONE GPU (reference layouts
core/transposes.pyx:359-445, core/distributor.py:770-924).

    python tools/rank_emulation.py --ranks 2,4,8 [--rank 0] [--size 512,512,256] [--steps 10] [--warmup 3]
    (also: python bench.py --emulate-rank r/P)

The problem is decomposed exactly as for P ranks (Distributor(mesh=(P,)): this process owns rank r's kx block of the
pencils and rank r's z planes of the grid) and runs rank r's REAL work through the production code path -- z transforms,
the side-stream exchanges (ddh_comm_alltoall_part: a field's components as one group, in windows of z planes; for sizes
without the blocked stage layout pack kernels, ddh_comm_alltoall / the transpose plans ddh_a2a_localize_*, unpack kernels),
x transforms, fused y stage, sharded factor + solve.  The communicator is the library's loop-back one
(ddh_comm_create_loopback): what a peer would have sent is the block this rank sends to it, copied on the device, so every
buffer size, kernel shape and stream dependency is that of the P-GPU run and only the wire is missing.  The wire is
priced separately: bytes per peer / 75 GB/s (one xGMI link per peer and direction, MI355X_MICROARCH.md: 7 links x ~153
GB/s bidirectional).  Values computed this way are NOT those of the P-rank run (tests/test_multiprocess.py and
test_gpu_multirank.py check those); this tool measures time only.

Output: one JSON line per P (and a table on stderr) with the step time, the kernel families' times, pack / unpack GB/s,
the side-stream exchange time (HIP events around ddh_comm_alltoall), and the predicted step time with the wire
un-overlapped / fully overlapped.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LINK_GBPS = 75.0          # one direction of one xGMI link (~153 GB/s bidirectional per link)


def one(rank, P, size, steps, warmup, dt):
    os.environ["DDH_EMULATE_RANK"] = "%d/%d" % (rank, P)
    import numpy as np
    import torch
    import problems
    import dedalus_amd.public as d3
    from dedalus_amd.executor import KernelTimer
    Nx, Ny, Nz = size
    t0 = time.time()
    solver, fields = problems.rayleigh_benard_3d(d3, Nx=Nx, Ny=Ny, Nz=Nz, timestepper="RK222", dist_kw=dict(mesh=(P,)))
    ex = solver.ex
    ex.sync()
    build_s = time.time() - t0
    pc = solver.dist.pcomm
    assert pc.backend == "loopback" and pc.rank == rank and pc.size == P
    for _ in range(warmup):
        solver.step(dt)
    ex.sync()
    timer = KernelTimer(torch)
    ex.timer = timer
    pc.via = {}
    pc.wire_events = []
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        solver.step(dt)
    torch.cuda.synchronize()
    el = time.time() - t0
    ex.timer = None
    wire_events, pc.wire_events = pc.wire_events, None
    summ = timer.summary()
    step_ms = 1e3 * el / steps
    fam = {k: dict(launches_per_step=v["launches"] / steps, ms_per_step=v["total_ms"] / steps, avg_ms=v["avg_ms"],
                   GBps=v["gbps"]) for k, v in sorted(summ.items())}
    via = {k: dict(exchanges_per_step=v[0] / steps, wire_MB_per_step=v[1] / steps / 1e6) for k, v in pc.via.items()}
    wire_bytes = sum(v[1] for v in pc.via.values()) / steps                      # bytes this rank sends to its peers per step
    side_ms = sum(a.elapsed_time(b) for a, b in wire_events) / steps if wire_events else 0.0
    wire_ms = (wire_bytes / max(P - 1, 1) / 1e9) / LINK_GBPS * 1e3               # every peer over its own link, concurrently
    kern_ms = sum(v["total_ms"] for v in summ.values()) / steps
    import hashlib
    Xh = np.ascontiguousarray(np.asarray(ex.download(solver.X))) if hasattr(solver, "X") else None
    finite = bool(np.isfinite(Xh).all()) if Xh is not None else None
    sha = hashlib.sha256(Xh.tobytes()).hexdigest() if Xh is not None else None
    return dict(P=P, rank=rank, size=list(size), steps=steps, ms_per_step_loopback=step_ms, kernel_ms_per_step=kern_ms,
                families=fam, exchange_via=via, wire_MB_per_rank_per_step=wire_bytes / 1e6,
                side_stream_copy_ms_per_step=side_ms, predicted_wire_ms_per_step=wire_ms, link_GBps_assumed=LINK_GBPS,
                predicted_ms_per_step_no_overlap=step_ms + wire_ms, predicted_ms_per_step_full_overlap=max(step_ms, wire_ms),
                predicted_steps_per_s=dict(no_overlap=1e3 / (step_ms + wire_ms), full_overlap=1e3 / max(step_ms, wire_ms)),
                ideal_share_ms=None, build_s=build_s, state_finite=finite, state_sha256=sha,
                grid_stage_windows=int(solver._grid_windows()) if hasattr(solver, "_grid_windows") else 1,
                emulated_link_GBps=float(os.environ.get("DDH_LOOPBACK_LINK_GBPS", 0) or 0),
                pencils_local=(Nx // 2 // P) * (Ny // 2), z_planes_local=(3 * Nz // 2) // P)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=str, default="2,4,8")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--size", type=str, default="512,512,256")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--link-gbps", type=float, default=0.0,
                    help="> 0: the loop-back communicator holds the exchange stream for (bytes to one peer) / this rate after "
                         "every exchange (DDH_LOOPBACK_LINK_GBPS): the step time then INCLUDES an emulated wire and shows how "
                         "much of it the pipeline hides")
    ap.add_argument("--single-gpu-ms", type=float, default=None, help="ms per step of the 1-GPU run (ideal share = that / P)")
    ap.add_argument("--worker", type=str, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    size = tuple(int(v) for v in args.size.split(","))
    if args.link_gbps > 0:
        os.environ["DDH_LOOPBACK_LINK_GBPS"] = str(args.link_gbps)
    if args.worker:                       # one (rank, P) per process: the decomposition is read once per process
        r, P = (int(v) for v in args.worker.split("/"))
        print(json.dumps(one(r, P, size, args.steps, args.warmup, args.dt)))
        return
    rows = []
    for P in [int(v) for v in args.ranks.split(",")]:
        r = min(args.rank, P - 1)
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "%d/%d" % (r, P), "--size", args.size, "--steps",
               str(args.steps), "--warmup", str(args.warmup), "--dt", str(args.dt)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
        if p.returncode != 0:
            print("P=%d failed:\n%s" % (P, p.stderr[-3000:]), file=sys.stderr)
            continue
        d = json.loads(p.stdout.strip().splitlines()[-1])
        if args.single_gpu_ms:
            d["ideal_share_ms"] = args.single_gpu_ms / P
        rows.append(d)
        print(json.dumps(d))
        f = d["families"]

        def ms(k):
            return f.get(k, {}).get("ms_per_step", 0.0)
        print("P=%d rank %d: %.2f ms/step with loop-back exchange%s (kernels %.2f)%s | solve %.2f fused-y %.2f x %.2f z %.2f "
              "matvec %.2f | pack %.2f ms (%.0f GB/s) unpack %.2f ms (%.0f GB/s) plan-exchanges %.2f | side stream %.2f ms | wire "
              "%.1f MB -> %.2f ms at %.0f GB/s per link | predicted %.1f (no overlap) .. %.1f (full overlap) steps/s"
              % (P, r, d["ms_per_step_loopback"], (" + emulated wire at %.0f GB/s per link" % d["emulated_link_GBps"]) if d["emulated_link_GBps"] else "",
                 d["kernel_ms_per_step"],
                 (" ideal share %.2f" % d["ideal_share_ms"]) if d["ideal_share_ms"] else "",
                 ms("pencil_solve"), ms("rfft_bilinear_fused"),
                 sum(ms(k) for k in f if k.startswith("rfft_") and k != "rfft_bilinear_fused"),
                 sum(ms(k) for k in f if k.startswith("cheb_")), ms("pencil_matvec"),
                 ms("a2a_pack"), f.get("a2a_pack", {}).get("GBps", 0.0), ms("a2a_unpack"), f.get("a2a_unpack", {}).get("GBps", 0.0),
                 ms("a2a_exchange"), d["side_stream_copy_ms_per_step"], d["wire_MB_per_rank_per_step"],
                 d["predicted_wire_ms_per_step"], LINK_GBPS, d["predicted_steps_per_s"]["no_overlap"],
                 d["predicted_steps_per_s"]["full_overlap"]), file=sys.stderr)


if __name__ == "__main__":
    main()
