"""
One process per GPU; pencils sharded over the ranks of a 1-D mesh.

Coefficient space is block-distributed along the first separable (Fourier) axis, exactly like the
reference's Layout.local_chunks (core/distributor.py:357-385), so every pencil is wholly owned by
one rank and all pencil linear algebra is communication-free.  The only exchange on the hot path is
the all-to-all between "kx-sharded, z local" and "z-sharded, kx local", which replaces the MPI
pencil transposes (core/transposes.pyx:22-445, core/distributor.py:770-924).  It is issued through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) on buffers
packed / unpacked by ddh_a2a_pack / ddh_a2a_unpack.

The exchange is placed right after the z transform, i.e. BEFORE the 3/2 padding of x and y: the
reference transposes padded data (core/distributor.py:58-75 order); moving it earlier sends 4/9 of
the bytes and changes no result.
"""

import os

import numpy as np


class _Done:
    def wait(self):
        return True


class _StreamWork:
    """An exchange issued on the communicator's side stream: wait() makes the CURRENT stream wait for it (no host
    synchronisation), like the Work object of an asynchronous torch.distributed collective."""

    def __init__(self, torch, event, keep):
        self.torch, self.event, self.keep = torch, event, keep

    def wait(self):
        self.torch.cuda.current_stream().wait_event(self.event)
        self.keep = None
        return True


def emulated_rank():
    """(rank, size) when this process emulates ONE rank of a sharded run on one GPU (DDH_EMULATE_RANK="r/P",
    tools/rank_emulation.py), else None."""
    v = os.environ.get("DDH_EMULATE_RANK")
    if not v:
        return None
    r, P = (int(x) for x in v.split("/"))
    if not (0 <= r < P):
        raise ValueError("DDH_EMULATE_RANK=%s: rank out of range" % v)
    return r, P


class LoopbackComm:
    """Rank r of P with no peers (DDH_EMULATE_RANK): the problem is decomposed exactly as for P ranks, this process owns
    rank r's pencils / z planes and runs rank r's kernels -- pack, exchange entry points, unpack, the per-component
    side-stream pipeline -- while every exchange returns the rank's own send blocks (ddh_comm_create_loopback: device
    copies in place of the wire).  For TIMING a rank's share of the sharded problem on one GPU; the values are not those
    of the P-rank run, reductions return this rank's contribution only."""

    backend = "loopback"

    def __init__(self, size):
        import torch
        r, P = emulated_rank()
        if P != size:
            raise ValueError("mesh size %d does not match DDH_EMULATE_RANK (%d ranks)" % (size, P))
        self.torch = torch
        self.size, self.rank = P, r
        self._lib_comm = None
        self.stats = dict(exchanges=0, bytes_sent=0)
        self.via = {}
        self._side = None
        self.wire_events = None

    def library_comm(self):
        if self._lib_comm is None:
            import ctypes as C
            from . import libhip
            h = C.c_uint64(0)
            libhip.call("ddh_comm_create_loopback", C.byref(h), self.rank, self.size)
            self._lib_comm = h
        return self._lib_comm

    note_via = None            # (bound below: shared with Comm)
    all_to_all_start = None

    def all_to_all(self, recv, send):
        self.note_via("loopback copy", send.numel() * 8 * (self.size - 1) // self.size)
        recv.reshape(-1).copy_(send.reshape(-1))

    def all_gather_host(self, a, axis=0):
        return np.concatenate([np.asarray(a)] * self.size, axis=axis)

    def allreduce_sum(self, value):
        return float(value)

    def allreduce_max(self, value):
        return float(value)

    def bcast_float(self, value, src=0):
        return float(value)

    def bcast_scalar(self, view, ex, src_rank=0):
        return view.reshape(-1)[:1].clone()

    def barrier(self):
        pass


class Comm:
    def __new__(cls, size):
        if emulated_rank() is not None:
            return LoopbackComm(size)
        return super().__new__(cls)

    def __init__(self, size):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            backend = os.environ.get("DDH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend)
        self.size = dist.get_world_size()
        self.rank = dist.get_rank()
        if self.size != size:
            raise ValueError("mesh size %d does not match the number of ranks %d" % (size, self.size))
        self.backend = dist.get_backend()
        self._lib_comm = None
        self.stats = dict(exchanges=0, bytes_sent=0)      # per rank, device exchanges only (bench.py reads it)
        # which code path carried how many exchanges / wire bytes (bench.py prints it as exchange.via):
        #   "ddh_a2a_localize (library RCCL plan)"       whole-field transposes, core/distributor.py::_exchange
        #   "ddh_comm_alltoall (library RCCL, side stream)"  the per-component pipeline
        #   "torch.distributed.all_to_all_single"         gloo test configurations, DDH_A2A_VIA=torch, fallback
        self.via = {}
        self._side = None
        self.wire_events = None        # bench.py: list collecting (start, end) events of the side-stream exchanges

    # ---- RCCL communicator owned by libdedalus_hip (ddh_comm_*): the production exchange path on the GPUs ----------
    def library_comm(self):
        """ddh communicator handle spanning the same ranks, or None when the exchange has to go through
        torch.distributed (gloo test configurations, DDH_A2A_VIA=torch).  The RCCL unique id travels from rank 0 to
        the others through the already initialised process group."""
        if self.backend != "nccl" or os.environ.get("DDH_A2A_VIA", "rccl") == "torch":
            return None
        if self._lib_comm is False:
            return None
        if self._lib_comm is None:
            import ctypes as C
            from . import libhip
            t = self.torch
            # Rank 0 makes the unique id.  If it cannot (RCCL does not load there), it broadcasts an all-zero id: every
            # rank then agrees to fall back BEFORE anyone enters ncclCommInitRank, where the others would wait for ever.
            buf = (C.c_ubyte * 128)()
            if self.rank == 0:
                try:
                    libhip.call("ddh_comm_unique_id", buf)
                    if not any(buf):
                        buf[0] = 1                       # (a valid id is never all zero; keep the convention safe)
                except Exception as e:
                    import logging
                    logging.getLogger(__name__).warning("library RCCL communicator unavailable on rank 0 (%s)" % (e,))
                    buf = (C.c_ubyte * 128)()
            ident = t.tensor(list(buf), dtype=t.uint8, device="cuda")
            self.dist.broadcast(ident, src=0)
            have_id = bool(ident.any().item())
            buf = (C.c_ubyte * 128)(*[int(v) for v in ident.cpu().tolist()])
            h = C.c_uint64(0)
            ok = False
            if have_id:
                # every rank enters ncclCommInitRank together (a rank that cannot load RCCL raises before it and the
                # others would block): agree first that the library is loadable everywhere
                try:
                    libhip.call("ddh_comm_probe")
                    can = 1
                except Exception:
                    can = 0
                cflag = t.tensor([can], dtype=t.int32, device="cuda")
                self.dist.all_reduce(cflag, op=self.dist.ReduceOp.MIN)
                if int(cflag.item()) == 1:
                    try:
                        libhip.call("ddh_comm_create", C.byref(h), self.rank, self.size, buf)
                        ok = self._self_check(h)
                    except Exception as e:           # (an RCCL that cannot initialise here: use the process group)
                        ok = False
                        import logging
                        logging.getLogger(__name__).warning("library RCCL communicator unavailable (%s)" % (e,))
            # every rank must take the same path
            flag = t.tensor([1 if ok else 0], dtype=t.int32, device="cuda")
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                import logging
                logging.getLogger(__name__).warning("pencil transposes fall back to torch.distributed all_to_all_single")
                self._lib_comm = False
            else:
                self._lib_comm = h
        return self._lib_comm or None

    def _self_check(self, h):
        """One tiny transpose each way through the library plan against the definition of the two layouts
        (column-local CL[n0][n1][n2/P][n3], row-local RL[n0][n1/P][n2][n3] of one global array)."""
        import ctypes as C
        from . import libhip
        t, P, r = self.torch, self.size, self.rank
        n0, n1, n2, n3 = 2, 2 * P, 3 * P, 2
        A = np.arange(n0 * n1 * n2 * n3, dtype=np.float64).reshape(n0, n1, n2, n3)
        cl = np.ascontiguousarray(A[:, :, r * (n2 // P):(r + 1) * (n2 // P), :])
        rl = np.ascontiguousarray(A[:, r * (n1 // P):(r + 1) * (n1 // P), :, :])
        plan = C.c_uint64(0)
        libhip.call("ddh_a2a_plan", C.byref(plan), h, n0, n1, n2, n3)
        d_cl, d_rl = t.from_numpy(cl).cuda(), t.empty(rl.shape, dtype=t.float64, device="cuda")
        d_back = t.empty(cl.shape, dtype=t.float64, device="cuda")
        st = C.c_void_p(t.cuda.current_stream().cuda_stream)
        libhip.call("ddh_a2a_localize_rows", plan, C.c_void_p(d_cl.data_ptr()), C.c_void_p(d_rl.data_ptr()), st)
        libhip.call("ddh_a2a_localize_columns", plan, C.c_void_p(d_rl.data_ptr()), C.c_void_p(d_back.data_ptr()), st)
        t.cuda.synchronize()
        ok = bool(np.array_equal(d_rl.cpu().numpy(), rl) and np.array_equal(d_back.cpu().numpy(), cl))
        libhip.call("ddh_destroy", plan)
        return ok

    def note_via(self, path, nbytes):
        v = self.via.setdefault(path, [0, 0])
        v[0] += 1
        v[1] += int(nbytes)

    def all_to_all(self, recv, send):
        """Equal-split all-to-all on flat buffers (torch tensors, or numpy arrays for the CPU oracle)."""
        t = self.torch
        self.note_via("torch.distributed.all_to_all_single (%s)" % self.backend,
                      (send.size if isinstance(send, np.ndarray) else send.numel()) * 8 * (self.size - 1) // self.size)
        if isinstance(send, np.ndarray):
            s = t.from_numpy(np.ascontiguousarray(send).reshape(-1))
            r = t.empty_like(s)
            self.dist.all_to_all_single(r, s)
            recv[...] = r.numpy().reshape(recv.shape)
        elif send.is_cuda and self.dist.get_backend() == "gloo":
            # test configuration only (several ranks sharing one GPU, where RCCL cannot run): stage the
            # exchange through the host; the production backend is "nccl" (RCCL) on device buffers
            s = send.reshape(-1).cpu()
            r = t.empty_like(s)
            self.dist.all_to_all_single(r, s)
            recv.reshape(-1).copy_(r)
        else:
            self.dist.all_to_all_single(recv.reshape(-1), send.reshape(-1))

    def all_to_all_start(self, recv, send, part=None, batch=1):
        """Start an equal-split all-to-all and return a handle with .wait(): on device buffers with RCCL the exchange
        runs on the communicator's stream while the caller keeps launching kernels; the numpy / host-staged test
        configurations complete immediately.  part = (offset, count): only `count` elements at `offset` of every peer's
        block are exchanged (ddh_comm_alltoall_part: one window of a component's planes); batch = n: send / recv hold n
        such components one after the other, exchanged as one group."""
        t = self.torch
        if part is not None and (isinstance(send, np.ndarray) or self.library_comm() is None or
                                 (send.is_cuda and self.backend == "gloo")):
            # test configurations: the parts gathered into contiguous buffers around the whole-buffer exchange
            off, cnt = part
            P = self.size
            for bi in range(batch):
                s2 = send.reshape(batch, P, -1)[bi, :, off:off + cnt]
                r2 = recv.reshape(batch, P, -1)[bi]
                if isinstance(send, np.ndarray):
                    tmp = np.empty((P, cnt))
                    self.all_to_all(tmp, np.ascontiguousarray(s2))
                    r2[:, off:off + cnt] = tmp
                else:
                    tmp = t.empty((P, cnt), dtype=send.dtype, device=send.device)
                    self.all_to_all(tmp, s2.contiguous())
                    r2[:, off:off + cnt] = tmp
            return _Done()
        if isinstance(send, np.ndarray) or (send.is_cuda and self.backend == "gloo"):
            self.all_to_all(recv, send)
            return _Done()
        h = self.library_comm()
        if h is not None:
            # the library's own communicator (grouped ncclSend / ncclRecv, ddh_comm_alltoall) on a side stream: ordered
            # after the packing kernel by an event, the unpacking kernel waits for its completion event
            import ctypes as C
            from . import libhip
            if self._side is None:
                self._side = t.cuda.Stream()
            cur = t.cuda.current_stream()
            ready = t.cuda.Event()
            ready.record(cur)
            self._side.wait_event(ready)
            send.record_stream(self._side)
            recv.record_stream(self._side)
            timed = self.wire_events is not None
            if timed:
                e0 = t.cuda.Event(enable_timing=True)
                e0.record(self._side)
            if part is None:
                libhip.call("ddh_comm_alltoall", h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                            send.numel() // self.size, C.c_void_p(self._side.cuda_stream))
                sent = send.numel() * 8 * (self.size - 1) // self.size
            else:
                off, cnt = part
                per = send.numel() // batch
                libhip.call("ddh_comm_alltoall_part", h, C.c_void_p(send.data_ptr() + 8 * off),
                            C.c_void_p(recv.data_ptr() + 8 * off), int(cnt), per // self.size, int(batch), per,
                            C.c_void_p(self._side.cuda_stream))
                sent = batch * cnt * 8 * (self.size - 1)
            done = t.cuda.Event(enable_timing=timed)
            done.record(self._side)
            if timed:
                self.wire_events.append((e0, done))
            self.note_via("ddh_comm_alltoall%s (%s, side stream)" % ("_part" if part is not None else "",
                                                                     "loop-back communicator" if self.backend == "loopback" else "library RCCL"),
                          sent)
            return _StreamWork(t, done, (send, recv))
        self.note_via("torch.distributed.all_to_all_single (%s, async)" % self.backend,
                      send.numel() * 8 * (self.size - 1) // self.size)
        return self.dist.all_to_all_single(recv.reshape(-1), send.reshape(-1), async_op=True)

    def all_gather_host(self, a, axis=0):
        """concatenate equal-shaped host arrays of all ranks along `axis` (user-boundary accesses only)"""
        t = self.torch
        src = t.from_numpy(np.ascontiguousarray(a))
        cuda = t.cuda.is_available() and self.dist.get_backend() == "nccl"
        if cuda:
            src = src.cuda()
        parts = [t.empty_like(src) for _ in range(self.size)]
        self.dist.all_gather(parts, src)
        return np.concatenate([p.cpu().numpy() for p in parts], axis=axis)

    def allreduce_sum(self, value):
        t = self.torch
        dev = "cuda" if (t.cuda.is_available() and self.dist.get_backend() == "nccl") else "cpu"
        x = t.tensor([float(value)], dtype=t.float64, device=dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        return float(x.item())

    def allreduce_max(self, value):
        t = self.torch
        dev = "cuda" if (t.cuda.is_available() and self.dist.get_backend() == "nccl") else "cpu"
        x = t.tensor([float(value)], dtype=t.float64, device=dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return float(x.item())

    def bcast_float(self, value, src=0):
        """rank `src`'s value on every rank (the reference's world_time broadcast, core/solvers.py:603-611)"""
        t = self.torch
        dev = "cuda" if (t.cuda.is_available() and self.dist.get_backend() == "nccl") else "cpu"
        x = t.tensor([float(value)], dtype=t.float64, device=dev)
        self.dist.broadcast(x, src=src)
        return float(x.item())

    def bcast_scalar(self, view, ex, src_rank=0):
        """a one-element array (device tensor or numpy) holding rank `src_rank`'s value of `view` on every rank: an
        asynchronous RCCL broadcast on the device (no host synchronisation), host-staged in the gloo test configurations"""
        t = self.torch
        if isinstance(view, np.ndarray):
            out = np.array([self.bcast_float(float(view.reshape(-1)[0]), src=src_rank)])
            return out
        buf = view.reshape(-1)[:1].clone()
        if self.backend == "nccl":
            self.dist.broadcast(buf, src=src_rank)
        else:
            buf.fill_(self.bcast_float(float(buf.cpu()[0]), src=src_rank))
        return buf

    def barrier(self):
        self.dist.barrier()


LoopbackComm.note_via = Comm.note_via
LoopbackComm.all_to_all_start = Comm.all_to_all_start      # (the production side-stream pipeline around ddh_comm_alltoall)
