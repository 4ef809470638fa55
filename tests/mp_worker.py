"""Worker for the multi-process sharding tests: every rank runs the same problem script with
mesh=(world,) and torch.distributed all-to-all (gloo), then saves its local coefficient blocks.
Default: the numpy oracle executor on CPU.  With a third argument "hip": the HIP executor, all ranks on
GPU 0 (exercises the sharded pencil packs, a2a pack/unpack kernels and the fused grid stage of the
multi-GPU path; only RCCL itself is replaced by a host-staged exchange)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    case, outdir = sys.argv[1], sys.argv[2]
    use_hip = len(sys.argv) > 3 and sys.argv[3] == "hip"
    import torch.distributed as dist
    backend = os.environ.get("DDH_DIST_BACKEND", "gloo") if use_hip else "gloo"
    if backend == "nccl":
        import torch
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))        # one GPU per rank, RCCL exchanges
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    import problems
    import dedalus_amd.public as d3
    if use_hip:
        if backend != "nccl":
            # all ranks share GPU 0 (RCCL needs one GPU per rank; the exchange is staged through the host)
            os.environ["LOCAL_RANK"] = "0"
        dist_kw = dict(mesh=(world,))
    else:
        from oracle.np_executor import NumpyExecutor
        dist_kw = dict(executor=NumpyExecutor(), mesh=(world,))
    if case == "a2a_part":
        # a PART of every peer's block, several components as one group (parallel.Comm.all_to_all_start(part=, batch=)): the
        # windowed exchanges of the sharded grid stage, here through the gloo stand-in on numpy and torch buffers
        from dedalus_amd.parallel import Comm
        pc = Comm(world)
        nb, blk, off, cnt = 3, 10, 4, 5
        send = np.empty((nb, world, blk))
        for b_ in range(nb):
            for p in range(world):
                send[b_, p] = 1000 * rank + 100 * p + 10 * b_ + np.arange(blk) / 16.0     # (from rank, to p, component b_)
        recv = np.full((nb, world, blk), -7.0)
        pc.all_to_all_start(recv, send, part=(off, cnt), batch=nb).wait()
        import torch
        recv_t = torch.full((nb, world, blk), -7.0, dtype=torch.float64)
        pc.all_to_all_start(recv_t, torch.from_numpy(send.copy()), part=(off, cnt), batch=nb).wait()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), recv=recv, recv_t=recv_t.numpy(), send=send)
        dist.destroy_process_group()
        return
    if case.startswith("shell_conv_"):
        # m-sharded shell convection: local blocks of the packed coefficient arrays (+ tau_p, replicated)
        solver, res = problems.run_shell_convection(d3, steps=4, timestepper=case.split("_")[-1], dist_kw=dist_kw)
        # gathered analysis output of the sharded state: rank 0 writes the global arrays
        fields = {v.name: v for v in solver.variables}
        h = solver.evaluator.add_file_handler(os.path.join(outdir, "final"), iter=1)
        h.add_task(fields["b"], layout='c', name="b_c")
        h.add_task(fields["u"], layout='g', name="u_g", scales=1.5)
        solver.evaluator.evaluate_handlers([h], iteration=solver.iteration, sim_time=solver.sim_time, timestep=0.05,
                                           wall_time=0.0)
        h.close()
    elif case == "refpencils":
        # the sharded solve against the reference's own pencil matrices (tests/pencil_check.py): every rank checks the
        # sampled pencils it owns
        import pencil_check
        ref = pencil_check.ReferencePencils()
        solver, f = problems.rayleigh_benard_3d(d3, Nx=8, Ny=8, Nz=ref.nz, Lx=4 / ref.stride, Ly=4 / ref.stride,
                                                dist_kw=dist_kw)
        lo = solver.dist._mx_offset
        mine = [g for g in ref.groups if lo <= g[0] < lo + solver.nx // 2]
        solver.solve_probe = dict(groups=mine, records=[])
        solver.step(1e-3)
        out = pencil_check.check_records(ref, solver.solve_probe["records"], mine)
        res = dict(npencils=np.array(len(mine)), residual=np.array([r["residual"] for r in out]),
                   solution=np.array([r["solution"] for r in out]), dropped=np.array([r["dropped_max"] for r in out]))
    elif case.startswith("rb3dsize_"):
        # 3-D Rayleigh-Benard at a given size, three RK222 steps: the local coefficient blocks + what the transformer's
        # stage rule says about the exchanged layout (tests/test_gpu_multirank.py: blocked x side of the sharded run)
        nx, ny, nz = (int(v) for v in case.split("_")[1].split("x"))
        solver, f = problems.rayleigh_benard_3d(d3, Nx=nx, Ny=ny, Nz=nz, timestepper="RK222", dist_kw=dist_kw)
        for _ in range(3):
            solver.step(1e-3)
        res = {k: np.array(f[k]['c']) for k in ("p", "b", "u")}
        dom = f["b"].domain
        xb = solver.dist.transformer.stage_xb(dom, dom.dealias)
        res["stage_xb"] = np.array([-1, -1, -1] if xb is None else [int(xb[0]), int(xb[1][0]), int(xb[1][1])] if isinstance(xb[1], tuple)
                                   else [int(xb[0]), int(xb[1]), 64])
        res["via"] = np.array(sorted(solver.dist.pcomm.via)) if solver.dist.pcomm is not None else np.array([])
        res["windows"] = np.array(solver._grid_windows() if hasattr(solver, "_grid_windows") else 1)
    elif case == "shell_cfl":
        solver, dts, speeds, res = problems.run_shell_cfl_case(d3, dist_kw=dist_kw)
        res = dict(res, dts=np.array(dts), speeds=np.array(speeds))
    else:
        solver, res = problems.run_case(d3, case, dist_kw=dist_kw)
        fields = {v.name: v for v in solver.state}
        h = solver.evaluator.add_file_handler(os.path.join(outdir, "final"), iter=1)
        h.add_task(fields["b"], layout='c', name="b_c")
        h.add_task(fields["b"], layout='g', name="b_g")
        solver.evaluator.evaluate_handlers([h], iteration=solver.iteration, sim_time=solver.sim_time, timestep=0.0,
                                           wall_time=0.0)
        h.close()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
