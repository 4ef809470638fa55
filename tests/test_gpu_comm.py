"""GPU: the library-owned RCCL communicator and the transpose plans of boundary B3 (ddh_comm_*, ddh_a2a_*).

The test box has ONE GPU and RCCL needs one GPU per rank, so
  * the real RCCL calls (ncclCommInitRank, grouped ncclSend/ncclRecv, ncclAllReduce) are exercised with a 1-rank
    communicator,
  * the index arithmetic of the P-rank transposes is checked by emulating P ranks in one process: the plan's own
    pack / unpack kernel calls with the arguments documented in include/dedalus_hip.h, the exchange done on the host,
    against the definition of FFTWTranspose.localize_rows / localize_columns (core/transposes.pyx:248-266),
  * a 2-rank "nccl" run of the whole solver against the reference goldens is launched when >= 2 GPUs are visible."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _comm1():
    from dedalus_amd import libhip
    from dedalus_amd.device import Device
    dev = Device.get()
    ident = (C.c_ubyte * 128)()
    libhip.call("ddh_comm_unique_id", ident)
    comm = C.c_uint64(0)
    libhip.call("ddh_comm_create", C.byref(comm), 0, 1, ident)
    return dev, comm


def test_one_rank_rccl_communicator_transposes_and_allreduce():
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    dev, comm = _comm1()
    r, n = C.c_int(-1), C.c_int(-1)
    libhip.call("ddh_comm_info", comm, C.byref(r), C.byref(n))
    assert (r.value, n.value) == (0, 1)
    rng = np.random.default_rng(5)
    for shape in [(3, 8, 6, 10), (1, 5, 7, 3), (2, 4, 4, 1)]:
        plan = C.c_uint64(0)
        libhip.call("ddh_a2a_plan", C.byref(plan), comm, *shape)
        a = rng.standard_normal(shape)
        d_a, d_b, d_c = dev.from_host(a), dev.empty(shape), dev.empty(shape)
        libhip.call("ddh_a2a_localize_rows", plan, ptr(d_a), ptr(d_b), dev.stream)       # through ncclSend/Recv to self
        libhip.call("ddh_a2a_localize_columns", plan, ptr(d_b), ptr(d_c), dev.stream)
        dev.sync()
        assert np.array_equal(dev.to_host(d_b), a) and np.array_equal(dev.to_host(d_c), a)
        with pytest.raises(libhip.DdhError):
            libhip.call("ddh_a2a_forward", plan, ptr(d_a), ptr(d_a), dev.stream)         # aliasing is refused
        libhip.call("ddh_destroy", plan)
    x = rng.standard_normal(1000)
    d_x = dev.from_host(x)
    for op in (0, 1, 2):
        libhip.call("ddh_comm_allreduce", comm, ptr(d_x), x.size, op, dev.stream)
    dev.sync()
    assert np.array_equal(dev.to_host(d_x), x)
    libhip.call("ddh_destroy", comm)


@pytest.mark.parametrize("P,shape", [(P, sh) for P in (2, 4) for sh in [(3, 8, 12, 5), (1, 4, 8, 1), (2, 12, 4, 3)]]
                         + [(8, (3, 8, 16, 5)), (8, (1, 24, 8, 2))])          # 8 = the target node's rank count
def test_transpose_index_logic_for_P_ranks(P, shape):
    """Emulated ranks: CL_r = A[:, :, block r of N2, :], RL_r = A[:, block r of N1, :, :] of one global array A."""
    from dedalus_amd import libhip
    from dedalus_amd.device import Device, ptr
    dev = Device.get()
    n0, n1, n2, n3 = shape
    A = np.random.default_rng(P).standard_normal(shape)
    CL = [np.ascontiguousarray(A[:, :, r * (n2 // P):(r + 1) * (n2 // P), :]) for r in range(P)]
    RL = [np.ascontiguousarray(A[:, r * (n1 // P):(r + 1) * (n1 // P), :, :]) for r in range(P)]
    loc = A.size // P

    def kernel(name, src, out_shape, *dims):
        d_s, d_o = dev.from_host(src), dev.empty(out_shape)
        libhip.call(name, ptr(d_s), ptr(d_o), *dims, P, dev.stream)
        dev.sync()
        return dev.to_host(d_o)

    def exchange(sends):           # block p of rank r's buffer -> block r of rank p's buffer
        chunk = loc // P
        return [np.concatenate([sends[q].ravel()[r * chunk:(r + 1) * chunk] for q in range(P)]) for r in range(P)]

    # localize_rows: CL -> RL  (pack: outer n0, na n1, nb n2/P, inner n3; unpack: outer n0, na n1/P, nb n2, inner n3)
    sends = [kernel("ddh_a2a_pack", CL[r], (loc,), n0, n1, n2 // P, n3) for r in range(P)]
    recvs = exchange(sends)
    for r in range(P):
        got = kernel("ddh_a2a_unpack", recvs[r], RL[r].shape, n0, n1 // P, n2, n3)
        assert np.array_equal(got, RL[r]), ("rows", r)
    # localize_columns: RL -> CL (pack: outer n0 n1/P, na n2, nb 1, inner n3; unpack: outer n0, na 1, nb n1, inner n2/P n3)
    sends = [kernel("ddh_a2a_pack", RL[r], (loc,), n0 * (n1 // P), n2, 1, n3) for r in range(P)]
    recvs = exchange(sends)
    for r in range(P):
        got = kernel("ddh_a2a_unpack", recvs[r], CL[r].shape, n0, 1, n1, (n2 // P) * n3)
        assert np.array_equal(got, CL[r]), ("columns", r)


def test_two_rank_nccl_run_matches_reference(golden_dir):
    """One process per GPU, backend nccl, the library's RCCL plans in the transposes: needs 2 GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL: one GPU per rank)")
    gold = np.load(os.path.join(golden_dir, "ivp.npz"))
    case = "rb3d_8x12x8_rk222"
    with tempfile.TemporaryDirectory() as tmp:
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "mp_worker.py"), case, tmp, "hip"]
        env = dict(os.environ, OMP_NUM_THREADS="1", DDH_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % k)) for k in range(2)]
        for key, tol in (("p", 1e-9), ("b", 1e-9), ("u", 1e-8)):
            ref = gold[case + "__" + key]
            full = np.concatenate([p[key] for p in parts], axis=ref.ndim - 3)
            assert np.linalg.norm(full - ref) / np.linalg.norm(ref) < tol, key


def test_two_rank_nccl_bench_at_full_size_matches_the_one_gpu_run():
    """`bench.py --gpus 2` at the metric's size (512 x 512 x 256 sharded over two GPUs, RCCL exchange) against the 1-GPU run
    of the same steps: both ranks seen, the library's RCCL plans in use, the pencil parity block green on both shards and
    the same end-state checksum.  Needs 2 GPUs."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL: one GPU per rank)")
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[n] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["dist_backend"] == "nccl", two
    assert two["exchange"]["exchanges_per_step"] > 0
    assert two["parity"]["max_residual"] < 1e-12 and two["parity"]["max_solution_error"] < 1e-10, two["parity"]
    assert abs(two["checksum_b_c_l2"] - one["checksum_b_c_l2"]) < 1e-11 * abs(one["checksum_b_c_l2"])


def test_comm_bootstrap_through_a_one_rank_nccl_group():
    """parallel.Comm on a 1-rank `nccl` process group: the RCCL unique id travels through the group, the library
    communicator is created, self-checked (a transpose each way) and selected for the exchanges"""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", world_size=1, rank=0)
from dedalus_amd.parallel import Comm
from dedalus_amd.executor import HipExecutor
c = Comm(1)
assert c.backend == "nccl" and c.library_comm() is not None
ex = HipExecutor()
plan = ex.a2a_plan(c, 2, 6, 4, 5)
a = torch.randn(2, 6, 4, 5, dtype=torch.float64, device="cuda")
b, d = torch.empty_like(a), torch.empty_like(a)
ex.a2a_localize_rows(plan, a, b); ex.a2a_localize_columns(plan, b, d)
torch.cuda.synchronize()
assert torch.equal(a, b) and torch.equal(a, d)
assert abs(c.bcast_float(3.25) - 3.25) == 0 and c.allreduce_max(2.0) == 2.0
# the per-component pipeline's exchange (core/distributor.py::_rows_after_z): ddh_comm_alltoall of the library's own
# communicator on the side stream, ordered against the producer / consumer kernels by events only
c.wire_events = []
works = []
for k in range(4):
    send = torch.randn(1 << 16, dtype=torch.float64, device="cuda") * (k + 1)
    recv = torch.zeros_like(send)
    works.append((c.all_to_all_start(recv, send), send, recv))
for w, send, recv in works:
    w.wait()
    out = recv * 2.0                      # consumer on the current stream, after the wait
    assert torch.equal(out, send * 2.0)
torch.cuda.synchronize()
assert list(c.via) == ["ddh_comm_alltoall (library RCCL, side stream)"] and c.via["ddh_comm_alltoall (library RCCL, side stream)"][0] == 4
assert len(c.wire_events) == 4 and all(a.elapsed_time(b) >= 0 for a, b in c.wire_events)
dist.destroy_process_group()
print("OK")
''' % (ROOT, 29600 + os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


def test_two_rank_bench_line_names_the_exchange_paths_taken():
    """`bench.py --gpus 2` with both ranks on the one GPU of the test box (gloo, host-staged exchange): the `exchange`
    block lists the code path every exchange of the timed steps really took, with wire bytes per rank and step."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1", DDH_DIST_BACKEND="gloo", DDH_FORCE_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "32,32,16", "--steps", "2", "--warmup", "1",
           "--repeats", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ex = line["exchange"]
    assert line["n_gpus"] == 2 and line["dist_backend"] == "gloo"
    assert ex["via"] and all(k.startswith("torch.distributed.all_to_all_single (gloo") for k in ex["via"]), ex["via"]
    # 8 + 4 components cross per stage, backward + forward; wire bytes = half of a rank's share at P = 2
    assert ex["exchanges_per_step"] > 0 and ex["per_rank_wire_bytes_per_step"] > 0
    assert abs(sum(v["wire_MB_per_rank_per_step"] for v in ex["via"].values()) * 1e6 - ex["per_rank_wire_bytes_per_step"]) < 1


@pytest.mark.parametrize("P", [2, 3, 4, 8])
@pytest.mark.parametrize("shape", [(3, 7, 10, 5), (1, 5, 9, 1), (2, 13, 3, 3), (2, 5, 5, 2)])
def test_uneven_block_transposes_for_P_ranks(P, shape):
    """Axes that P does not divide (the reference's Alltoallv transposes, core/transposes.pyx:287-445): blocks of
    ceil(n / P), trailing ranks own less or nothing (n = 5, P = 4 -> 2, 2, 1, 0).  Emulated ranks as above: the pack /
    unpack kernels ddh_a2a_plan uses for such shapes, with the exchange (counts and displacements of exchange_v) done
    on the host."""
    from dedalus_amd import libhip
    from dedalus_amd.device import Device, ptr
    dev = Device.get()
    n0, n1, n2, n3 = shape
    A = np.random.default_rng(P + n1).standard_normal(shape)
    B1, B2 = -(-n1 // P), -(-n2 // P)
    lo1 = [min(p * B1, n1) for p in range(P + 1)]
    lo2 = [min(p * B2, n2) for p in range(P + 1)]
    CL = [np.ascontiguousarray(A[:, :, lo2[r]:lo2[r + 1], :]) for r in range(P)]
    RL = [np.ascontiguousarray(A[:, lo1[r]:lo1[r + 1], :, :]) for r in range(P)]

    def kernel(name, src, nout, *dims):
        if src.size == 0 or nout == 0:
            return np.zeros(nout)
        d_s, d_o = dev.from_host(src.ravel()), dev.empty((max(nout, 1),))
        d_o.fill_(float("nan"))
        libhip.call(name, ptr(d_s), ptr(d_o), *dims, P, dev.stream)
        dev.sync()
        return dev.to_host(d_o)[:nout]

    # ---- localize_rows: CL_r [n0][n1][n2_r][n3] -> RL_r [n0][n1_r][n2][n3]
    sends = [kernel("ddh_a2av_pack", CL[r], CL[r].size, n0, n1, CL[r].shape[2] * n3) for r in range(P)]
    recvs = []
    for r in range(P):
        n1r = lo1[r + 1] - lo1[r]
        parts = []
        for q in range(P):                                   # what q sends to r: its block r
            n2q = lo2[q + 1] - lo2[q]
            disp, cnt = n0 * n2q * n3 * lo1[r], n0 * n1r * n2q * n3
            parts.append(sends[q][disp:disp + cnt])
        recvs.append(np.concatenate(parts) if parts else np.zeros(0))
    for r in range(P):
        n1r = lo1[r + 1] - lo1[r]
        got = kernel("ddh_a2av_unpack", recvs[r], RL[r].size, n0 * n1r, n2, n3)
        assert np.array_equal(got.reshape(RL[r].shape), RL[r]), ("rows", r)
    # ---- localize_columns: RL_r -> CL_r
    sends = [kernel("ddh_a2av_pack", RL[r], RL[r].size, n0 * RL[r].shape[1], n2, n3) for r in range(P)]
    recvs = []
    for r in range(P):
        n2r = lo2[r + 1] - lo2[r]
        parts = []
        for q in range(P):
            n1q = lo1[q + 1] - lo1[q]
            disp, cnt = n0 * n1q * n3 * lo2[r], n0 * n1q * n2r * n3
            parts.append(sends[q][disp:disp + cnt])
        recvs.append(np.concatenate(parts) if parts else np.zeros(0))
    for r in range(P):
        n2r = lo2[r + 1] - lo2[r]
        got = kernel("ddh_a2av_unpack", recvs[r], CL[r].size, n0, n1, n2r * n3)
        assert np.array_equal(got.reshape(CL[r].shape), CL[r]), ("columns", r)


def test_one_rank_plan_with_any_shape():
    """ddh_a2a_plan no longer insists on divisible axes; with one rank every shape is a copy through the plan's buffers"""
    from dedalus_amd import libhip
    from dedalus_amd.device import ptr
    dev, comm = _comm1()
    rng = np.random.default_rng(8)
    for shape in [(2, 5, 7, 3), (1, 1, 1, 1)]:
        plan = C.c_uint64(0)
        libhip.call("ddh_a2a_plan", C.byref(plan), comm, *shape)
        a = rng.standard_normal(shape)
        d_a, d_b = dev.from_host(a), dev.empty(shape)
        libhip.call("ddh_a2a_localize_rows", plan, ptr(d_a), ptr(d_b), dev.stream)
        dev.sync()
        assert np.array_equal(dev.to_host(d_b), a)
        libhip.call("ddh_destroy", plan)
    libhip.call("ddh_destroy", comm)


@pytest.mark.parametrize("P,rank", [(2, 1), (4, 0), (8, 5)])
def test_loopback_communicator_returns_the_rank_own_blocks(P, rank):
    """ddh_comm_create_loopback (rank emulation on one GPU): every exchange hands back the blocks this rank sends -- through
    ddh_comm_alltoall and through a transpose plan, whose pack -> exchange -> unpack then equals pack -> unpack."""
    import ctypes as C
    import torch
    from dedalus_amd import libhip
    from dedalus_amd.device import Device
    dev = Device.get()
    h = C.c_uint64(0)
    libhip.call("ddh_comm_create_loopback", C.byref(h), rank, P)
    r, n = C.c_int(-1), C.c_int(-1)
    libhip.call("ddh_comm_info", h, C.byref(r), C.byref(n))
    assert (r.value, n.value) == (rank, P)
    chunk = 1000
    send = torch.arange(P * chunk, dtype=torch.float64, device="cuda")
    recv = torch.full_like(send, float("nan"))
    libhip.call("ddh_comm_alltoall", h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), chunk, dev.stream)
    dev.sync()
    assert torch.equal(send, recv)
    n0, n1, n2, n3 = 2, 3 * P, 2 * P, 4
    plan = C.c_uint64(0)
    libhip.call("ddh_a2a_plan", C.byref(plan), h, n0, n1, n2, n3)
    cl = torch.randn(n0 * n1 * (n2 // P) * n3, dtype=torch.float64, device="cuda")
    rl = torch.full_like(cl, float("nan"))
    libhip.call("ddh_a2a_localize_rows", plan, C.c_void_p(cl.data_ptr()), C.c_void_p(rl.data_ptr()), dev.stream)
    tmp, want = torch.empty_like(cl), torch.empty_like(cl)
    libhip.call("ddh_a2a_pack", C.c_void_p(cl.data_ptr()), C.c_void_p(tmp.data_ptr()), n0, n1, n2 // P, n3, P, dev.stream)
    libhip.call("ddh_a2a_unpack", C.c_void_p(tmp.data_ptr()), C.c_void_p(want.data_ptr()), n0, n1 // P, n2, n3, P, dev.stream)
    dev.sync()
    assert torch.equal(rl, want)
    libhip.call("ddh_destroy", plan)
    libhip.call("ddh_destroy", h)


@pytest.mark.parametrize("P,rank", [(2, 0), (8, 3)])
def test_partial_exchange_on_the_loopback_communicator(P, rank):
    """ddh_comm_alltoall_part (the windowed exchanges of the sharded grid stage): `count` doubles of every peer's block, the
    blocks peer_stride apart, nbatch components batch_stride apart as one group -- on the loop-back communicator the parts
    come back as they were sent and nothing else of the receive buffer is written."""
    import ctypes as C
    import torch
    from dedalus_amd import libhip
    from dedalus_amd.device import Device
    dev = Device.get()
    h = C.c_uint64(0)
    libhip.call("ddh_comm_create_loopback", C.byref(h), rank, P)
    nb, blk, off, cnt = 3, 500, 120, 250
    send = torch.randn(nb, P, blk, dtype=torch.float64, device="cuda")
    recv = torch.full_like(send, -7.0)
    libhip.call("ddh_comm_alltoall_part", h, C.c_void_p(send.data_ptr() + 8 * off), C.c_void_p(recv.data_ptr() + 8 * off), cnt,
                blk, nb, P * blk, dev.stream)
    dev.sync()
    want = torch.full_like(send, -7.0)
    want[:, :, off:off + cnt] = send[:, :, off:off + cnt]
    assert torch.equal(recv, want)
    libhip.call("ddh_destroy", h)


def test_rank_emulation_tool_runs_a_rank_of_a_sharded_problem():
    """tools/rank_emulation.py at a small size: rank 1 of 4 of 3-D Rayleigh-Benard 64 x 32 x 32 runs through the production
    pipeline (per-component side-stream exchanges + transpose plans) on the loop-back communicator and reports a finite
    state, the exchange paths taken and the kernel families of the sharded step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_emulation.py"), "--ranks", "4", "--rank", "1", "--size",
                        "64,32,32", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["P"] == 4 and d["rank"] == 1 and d["state_finite"] is True
    assert d["pencils_local"] == (64 // 2 // 4) * (32 // 2)
    assert any("loop-back" in k for k in d["exchange_via"]), d["exchange_via"]
    assert "pencil_solve" in d["families"] and "a2a_pack" in d["families"] and "a2a_unpack" in d["families"]
    assert d["wire_MB_per_rank_per_step"] > 0 and d["predicted_wire_ms_per_step"] > 0


def test_deferred_exchange_schedules_compute_the_same_state_under_an_emulated_wire():
    """Rank 1 of 2 of 3-D RB 128 x 16 x 32 (blocks of 64 rows: the blocked exchange) on the loop-back communicator with
    every exchange followed by an emulated wire time on its stream (DDH_LOOPBACK_LINK_GBPS, slow enough that a consumer
    that did not wait would read stale data): the schedules of the exchange -- waits at the exchange, waits deferred to the
    consumer, x steps component by component, all z steps first, the grid stage in 2 (default) / 4 windows of z planes
    pipelined against windowed exchanges -- end in bit-identical states."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = {}
    for name, env in (("eager", {"DDH_A2A_DEFER": "0"}), ("deferred", {"DDH_A2A_WINDOWS": "1"}),
                      ("split", {"DDH_A2A_SPLIT_X": "1", "DDH_A2A_WINDOWS": "1"}),
                      ("prefetch+split", {"DDH_A2A_PREFETCH": "1", "DDH_A2A_SPLIT_X": "1", "DDH_A2A_WINDOWS": "1"}),
                      ("2 windows (default)", {}), ("4 windows", {"DDH_A2A_WINDOWS": "4"})):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_emulation.py"), "--ranks", "2", "--rank", "1",
                            "--size", "128,16,32", "--steps", "3", "--warmup", "1", "--link-gbps", "0.5"],
                           capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["state_finite"] is True and d["emulated_link_GBps"] == 0.5
        assert "a2a_pack" not in d["families"] and "a2a_unpack" not in d["families"], sorted(d["families"])
        shas[name] = d["state_sha256"]
    assert len(set(shas.values())) == 1, shas
