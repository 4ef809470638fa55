"""Worker for the multi-process (gloo, CPU) sharding test: every rank runs the same problem script with
mesh=(world,), the numpy oracle executor and torch.distributed all-to-all, then saves its local
coefficient blocks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    case, outdir = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import problems
    import dedalus_amd.public as d3
    from oracle.np_executor import NumpyExecutor
    solver, res = problems.run_case(d3, case, dist_kw=dict(executor=NumpyExecutor(), mesh=(world,)))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
