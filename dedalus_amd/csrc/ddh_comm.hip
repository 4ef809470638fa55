// Distributed pencil transposes with a library-owned RCCL communicator (boundary B3).
//
// Replaces FFTWTranspose / AlltoallvTranspose (core/transposes.pyx:22-445) and the MPI communicator they are planned
// on (core/distributor.py:696-768): one process per GPU, xGMI point-to-point links underneath.  A plan describes the
// reference's reduced 4-D view (N0, N1, N2, N3) of a block-distributed array and moves it between
//     column-local  CL: [N0][N1     ][N2 / P][N3]     (axis N1 local, N2 distributed)
//     row-local     RL: [N0][N1 / P ][N2    ][N3]     (axis N2 local, N1 distributed)
// as   pack (ddh_a2a_pack kernel: contiguous 16-byte copies)  ->  P grouped ncclSend / ncclRecv pairs on the caller's
// stream  ->  unpack.  RCCL is bound at run time (dlopen of librccl.so.1: the copy torch already loaded when present),
// so the library itself has no link-time dependency and loads on boxes without a GPU.
#include "ddh_common.h"
#include <vector>
#include <cstdlib>

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace ddh {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (torch's), if any
        if (lib) break;
    }
    for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(std::string("RCCL not found: ") + dlerror());
#define DDH_SYM(field, name)                                                        \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));      \
    if (!g_rccl.field) return fail(std::string("RCCL symbol missing: ") + name);
    DDH_SYM(GetUniqueId, "ncclGetUniqueId")
    DDH_SYM(CommInitRank, "ncclCommInitRank")
    DDH_SYM(CommDestroy, "ncclCommDestroy")
    DDH_SYM(GroupStart, "ncclGroupStart")
    DDH_SYM(GroupEnd, "ncclGroupEnd")
    DDH_SYM(Send, "ncclSend")
    DDH_SYM(Recv, "ncclRecv")
    DDH_SYM(AllReduce, "ncclAllReduce")
    DDH_SYM(GetErrorString, "ncclGetErrorString")
#undef DDH_SYM
    g_rccl.lib = lib;
    return 0;
}

static int check_nccl(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return 0;
    return fail(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
#define DDH_NCCL(call)                                   \
    do {                                                 \
        int _s = check_nccl((call), #call);              \
        if (_s) return _s;                               \
    } while (0)

struct Comm : HandleBase {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    // ddh_comm_create_loopback: rank `rank` of `nranks` on ONE GPU, no peers.  Every block a peer would have sent is the
    // block this rank sends to that peer (device copies on the caller's stream): sizes, pack / unpack kernels, stream
    // ordering and the per-rank kernel shapes are those of the real run, the VALUES received are not.
    bool loopback = false;
    double link_gbps = 0.0;     // loop-back only (DDH_LOOPBACK_LINK_GBPS): > 0 = every exchange is followed, on its stream, by a
                                // wait of (bytes to one peer) / link_gbps -- the time the P - 1 concurrent xGMI transfers would take
    ~Comm() override {
        if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
    }
};

// holds a stream for `ns` nanoseconds (one lane spins on the 100 MHz wall clock): the wire time of an emulated exchange
__global__ void wire_delay_kernel(long long ns) {
    const long long t0 = wall_clock64();
    const long long ticks = ns / 10;                     // wall_clock64: 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static int loopback_wire(const Comm *c, size_t bytes_to_one_peer, hipStream_t s) {
    if (c->link_gbps <= 0.0 || c->nranks < 2 || bytes_to_one_peer == 0) return 0;
    const long long ns = (long long)((double)bytes_to_one_peer / c->link_gbps);       // bytes / (GB/s) = ns
    hipLaunchKernelGGL(wire_delay_kernel, dim3(1), dim3(1), 0, s, ns);
    DDH_HIP(hipGetLastError());
    return 0;
}

struct A2aPlan : HandleBase {
    Comm *comm = nullptr;
    long n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    size_t local = 0;          // doubles per rank (buffer size)
    bool even = true;          // both transposed axes divisible by the number of ranks
    long B1 = 0, B2 = 0;       // block sizes ceil(n1 / P), ceil(n2 / P) of the uneven distribution
    double *send = nullptr, *recv = nullptr;
    ~A2aPlan() override {
        (void)hipFree(send);
        (void)hipFree(recv);
    }
};

// equal-split all-to-all of `local` doubles: block p of `send` goes to rank p, block q of `recv` comes from rank q
// (send / recv: the plan's staging buffers, or the caller's arrays where a pack / unpack pass would be the identity)
static int exchange(A2aPlan *pl, const double *send, double *recv, hipStream_t s) {
    const Comm *c = pl->comm;
    const size_t chunk = pl->local / (size_t)c->nranks;
    // the block a rank keeps is a device copy, not a trip through RCCL (1/P of the data never touches the fabric queue)
    DDH_HIP(hipMemcpyAsync(recv + (size_t)c->rank * chunk, send + (size_t)c->rank * chunk, chunk * sizeof(double),
                           hipMemcpyDeviceToDevice, s));
    if (c->nranks == 1) return 0;
    if (c->loopback) {
        for (int p = 0; p < c->nranks; ++p)
            if (p != c->rank)
                DDH_HIP(hipMemcpyAsync(recv + (size_t)p * chunk, send + (size_t)p * chunk, chunk * sizeof(double),
                                       hipMemcpyDeviceToDevice, s));
        return loopback_wire(c, chunk * sizeof(double), s);
    }
    DDH_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) continue;
        DDH_NCCL(g_rccl.Send(send + (size_t)p * chunk, chunk, ncclDouble, p, c->comm, s));
        DDH_NCCL(g_rccl.Recv(recv + (size_t)p * chunk, chunk, ncclDouble, p, c->comm, s));
    }
    DDH_NCCL(g_rccl.GroupEnd());
    return 0;
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_comm_probe(void) { return load_rccl(); }

int ddh_comm_unique_id(unsigned char *id_h) {
    if (!id_h) return fail("ddh_comm_unique_id: null buffer");
    if (int s = load_rccl()) return s;
    ncclUniqueId id;
    DDH_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == DDH_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_h, &id, sizeof(id));
    return 0;
}

int ddh_comm_create(ddh_handle *comm, int rank, int nranks, const unsigned char *id_h) {
    if (!comm || !id_h) return fail("ddh_comm_create: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("ddh_comm_create: rank out of range");
    if (int s = load_rccl()) return s;
    ncclUniqueId id;
    memcpy(&id, id_h, sizeof(id));
    Comm *c = new Comm();
    c->kind = H_COMM;
    c->rank = rank;
    c->nranks = nranks;
    if (int s = check_nccl(g_rccl.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank")) {
        c->comm = nullptr;
        delete c;
        return s;
    }
    *comm = register_handle(c);
    return 0;
}

int ddh_comm_create_loopback(ddh_handle *comm, int rank, int nranks) {
    if (!comm) return fail("ddh_comm_create_loopback: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("ddh_comm_create_loopback: rank out of range");
    Comm *c = new Comm();
    c->kind = H_COMM;
    c->rank = rank;
    c->nranks = nranks;
    c->loopback = true;
    if (const char *g = getenv("DDH_LOOPBACK_LINK_GBPS")) c->link_gbps = atof(g);
    *comm = register_handle(c);
    return 0;
}

int ddh_comm_info(ddh_handle comm, int *rank, int *nranks) {
    Comm *c = (Comm *)lookup_handle(comm, H_COMM);
    if (!c) return -1;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return 0;
}

int ddh_comm_allreduce(ddh_handle comm, double *buf, long count, int op, void *stream) {
    Comm *c = (Comm *)lookup_handle(comm, H_COMM);
    if (!c) return -1;
    if (op < 0 || op > 2) return fail("ddh_comm_allreduce: op must be 0 (sum), 1 (max) or 2 (min)");
    if (c->loopback) return 0;          // (this rank's own value stands for the reduction)
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    DDH_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ops[op], c->comm, as_stream(stream)));
    return 0;
}

// Equal-split all-to-all on caller-owned buffers: block p (`chunk` doubles) of `send` goes to rank p, block q of `recv`
// comes from rank q; asynchronous on `stream` (the caller orders it against its pack / unpack kernels with events).
int ddh_comm_alltoall(ddh_handle comm, const double *send, double *recv, long chunk, void *stream) {
    Comm *c = (Comm *)lookup_handle(comm, H_COMM);
    if (!c) return -1;
    if (!send || !recv || chunk < 0) return fail("ddh_comm_alltoall: bad argument");
    if (send == recv) return fail("ddh_comm_alltoall: send and receive buffers must differ");
    hipStream_t s = as_stream(stream);
    if (chunk == 0) return 0;
    DDH_HIP(hipMemcpyAsync(recv + (size_t)c->rank * chunk, send + (size_t)c->rank * chunk, (size_t)chunk * sizeof(double),
                           hipMemcpyDeviceToDevice, s));
    if (c->nranks == 1) return 0;
    if (c->loopback) {
        for (int p = 0; p < c->nranks; ++p)
            if (p != c->rank)
                DDH_HIP(hipMemcpyAsync(recv + (size_t)p * chunk, send + (size_t)p * chunk, (size_t)chunk * sizeof(double),
                                       hipMemcpyDeviceToDevice, s));
        return loopback_wire(c, (size_t)chunk * sizeof(double), s);
    }
    DDH_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) continue;
        DDH_NCCL(g_rccl.Send(send + (size_t)p * chunk, (size_t)chunk, ncclDouble, p, c->comm, s));
        DDH_NCCL(g_rccl.Recv(recv + (size_t)p * chunk, (size_t)chunk, ncclDouble, p, c->comm, s));
    }
    DDH_NCCL(g_rccl.GroupEnd());
    return 0;
}

// A PART of such an all-to-all: `count` doubles per peer, the blocks `peer_stride` doubles apart (send / recv point at the
// part of block 0).  A component exchanged in windows of its planes: window k of [p][z][rows][ky] is one contiguous range
// per peer, and the grid stage of window k runs while window k + 1 is on the wire (Transformer windows, core/distributor.py).
// `nbatch` such exchanges `batch_stride` doubles apart (the components of a field) go out as ONE group of sends / receives.
int ddh_comm_alltoall_part(ddh_handle comm, const double *send, double *recv, long count, long peer_stride, int nbatch,
                           long batch_stride, void *stream) {
    Comm *c = (Comm *)lookup_handle(comm, H_COMM);
    if (!c) return -1;
    if (!send || !recv || count < 0 || peer_stride < count || nbatch < 1 || (nbatch > 1 && batch_stride < 1))
        return fail("ddh_comm_alltoall_part: bad argument");
    if (send == recv) return fail("ddh_comm_alltoall_part: send and receive buffers must differ");
    hipStream_t s = as_stream(stream);
    if (count == 0) return 0;
    const size_t st = (size_t)peer_stride, bs = (size_t)batch_stride;
    if (c->loopback || c->nranks == 1) {
        // every block -- the rank's own and, in place of the wire, the ones it would have received -- as one strided copy
        for (int b = 0; b < nbatch; ++b)
            DDH_HIP(hipMemcpy2DAsync(recv + b * bs, st * sizeof(double), send + b * bs, st * sizeof(double),
                                     (size_t)count * sizeof(double), (size_t)c->nranks, hipMemcpyDeviceToDevice, s));
        return c->nranks == 1 ? 0 : loopback_wire(c, (size_t)nbatch * (size_t)count * sizeof(double), s);
    }
    for (int b = 0; b < nbatch; ++b)
        DDH_HIP(hipMemcpyAsync(recv + b * bs + (size_t)c->rank * st, send + b * bs + (size_t)c->rank * st,
                               (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, s));
    DDH_NCCL(g_rccl.GroupStart());
    for (int b = 0; b < nbatch; ++b)
        for (int p = 0; p < c->nranks; ++p) {
            if (p == c->rank) continue;
            DDH_NCCL(g_rccl.Send(send + b * bs + (size_t)p * st, (size_t)count, ncclDouble, p, c->comm, s));
            DDH_NCCL(g_rccl.Recv(recv + b * bs + (size_t)p * st, (size_t)count, ncclDouble, p, c->comm, s));
        }
    DDH_NCCL(g_rccl.GroupEnd());
    return 0;
}

// uneven blocks: rank p owns [p B, min((p + 1) B, n)) of an axis of length n, B = ceil(n / P) (Layout.local_chunks,
// core/distributor.py; Alltoallv transposes core/transposes.pyx:287-445)
static inline long blk_lo(long n, long B, int p) { return (p * B < n) ? p * B : n; }
static inline long blk_n(long n, long B, int p) {
    const long lo = blk_lo(n, B, p);
    return ((lo + B < n) ? lo + B : n) - lo;
}

// send block p = cnt_s[p] doubles at disp_s[p] of `send`; block q of `recv` = cnt_r[q] doubles at disp_r[q]
static int exchange_v(A2aPlan *pl, const std::vector<size_t> &cnt_s, const std::vector<size_t> &disp_s,
                      const std::vector<size_t> &cnt_r, const std::vector<size_t> &disp_r, hipStream_t s) {
    const Comm *c = pl->comm;
    const int me = c->rank;
    if (cnt_s[me] != cnt_r[me]) return fail("a2a: self block sizes differ");
    if (cnt_s[me])
        DDH_HIP(hipMemcpyAsync(pl->recv + disp_r[me], pl->send + disp_s[me], cnt_s[me] * sizeof(double),
                               hipMemcpyDeviceToDevice, s));
    if (c->nranks == 1) return 0;
    if (c->loopback) {
        for (int p = 0; p < c->nranks; ++p) {
            const size_t cnt = cnt_s[p] < cnt_r[p] ? cnt_s[p] : cnt_r[p];
            if (p != me && cnt)
                DDH_HIP(hipMemcpyAsync(pl->recv + disp_r[p], pl->send + disp_s[p], cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        return 0;
    }
    DDH_NCCL(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == me) continue;
        if (cnt_s[p]) DDH_NCCL(g_rccl.Send(pl->send + disp_s[p], cnt_s[p], ncclDouble, p, c->comm, s));
        if (cnt_r[p]) DDH_NCCL(g_rccl.Recv(pl->recv + disp_r[p], cnt_r[p], ncclDouble, p, c->comm, s));
    }
    DDH_NCCL(g_rccl.GroupEnd());
    return 0;
}

int ddh_a2a_plan(ddh_handle *plan, ddh_handle comm, long n0, long n1, long n2, long n3) {
    return ddh_a2a_plan_blocks(plan, comm, n0, n1, n2, n3, 0, 0);
}

int ddh_a2a_plan_blocks(ddh_handle *plan, ddh_handle comm, long n0, long n1, long n2, long n3, long block1, long block2) {
    Comm *c = (Comm *)lookup_handle(comm, H_COMM);
    if (!c) return -1;
    if (block1 < 0 || block2 < 0 || (block1 && block1 * c->nranks < n1) || (block2 && block2 * c->nranks < n2))
        return fail("ddh_a2a_plan_blocks: blocks do not cover the axis");
    if (!plan || n0 < 1 || n1 < 1 || n2 < 1 || n3 < 1) return fail("ddh_a2a_plan: bad shape");
    A2aPlan *pl = new A2aPlan();
    pl->kind = H_A2A;
    pl->comm = c;
    pl->n0 = n0; pl->n1 = n1; pl->n2 = n2; pl->n3 = n3;
    pl->B1 = block1 ? block1 : (n1 + c->nranks - 1) / c->nranks;
    pl->B2 = block2 ? block2 : (n2 + c->nranks - 1) / c->nranks;
    pl->even = (pl->B1 * c->nranks == n1) && (pl->B2 * c->nranks == n2);
    if (pl->even) {
        pl->local = (size_t)(n0 * n1 * n2 * n3) / (size_t)c->nranks;
    } else {
        // the larger of the rank's column-local and row-local arrays
        const size_t cl = (size_t)(n0 * n1 * blk_n(n2, pl->B2, c->rank) * n3);
        const size_t rl = (size_t)(n0 * blk_n(n1, pl->B1, c->rank) * n2 * n3);
        pl->local = cl > rl ? cl : rl;
        if (pl->local == 0) pl->local = 1;
    }
    if (check_hip(hipMalloc((void **)&pl->send, pl->local * sizeof(double) + 16), "hipMalloc") ||
        check_hip(hipMalloc((void **)&pl->recv, pl->local * sizeof(double) + 16), "hipMalloc")) {
        delete pl;
        return -2;
    }
    *plan = register_handle(pl);
    return 0;
}

// CL [N0][N1][N2/P][N3] -> RL [N0][N1/P][N2][N3]    (Transpose.decrement: towards coefficient space)
int ddh_a2a_localize_rows(ddh_handle plan, const double *cl, double *rl, void *stream) {
    A2aPlan *pl = (A2aPlan *)lookup_handle(plan, H_A2A);
    if (!pl) return -1;
    const int P = pl->comm->nranks;
    if (cl == rl) return fail("ddh_a2a_localize_rows: the two layouts must be different buffers");
    if (!pl->even) {
        // CL [N0][N1][n2(me)][N3] -> RL [N0][n1(me)][N2][N3] with uneven blocks
        const int me = pl->comm->rank;
        const long n2me = blk_n(pl->n2, pl->B2, me), n1me = blk_n(pl->n1, pl->B1, me);
        std::vector<size_t> cs(P), ds(P), cr(P), dr(P);
        for (int p = 0; p < P; ++p) {
            cs[p] = (size_t)(pl->n0 * blk_n(pl->n1, pl->B1, p) * n2me * pl->n3);
            ds[p] = (size_t)(pl->n0 * n2me * pl->n3 * blk_lo(pl->n1, pl->B1, p));
            cr[p] = (size_t)(pl->n0 * n1me * blk_n(pl->n2, pl->B2, p) * pl->n3);
            dr[p] = (size_t)(pl->n0 * n1me * pl->n3 * blk_lo(pl->n2, pl->B2, p));
        }
        if (n2me > 0)
            if (int s = ddh_a2av_pack_b(cl, pl->send, pl->n0, pl->n1, n2me * pl->n3, P, pl->B1, stream)) return s;
        if (int s = exchange_v(pl, cs, ds, cr, dr, as_stream(stream))) return s;
        if (n1me > 0) return ddh_a2av_unpack_b(pl->recv, rl, pl->n0 * n1me, pl->n2, pl->n3, P, pl->B2, stream);
        return 0;
    }
    // split N1 into P row blocks, exchange, gather the P column blocks along N2
    // (N0 == 1: the packed order [p][n1 / P][n2 / P][n3] IS the column-local array -- it is sent as it lies, no pack pass)
    const double *send = cl;
    if (pl->n0 > 1) {
        if (int s = ddh_a2a_pack(cl, pl->send, pl->n0, pl->n1, pl->n2 / P, pl->n3, P, stream)) return s;
        send = pl->send;
    }
    if (int s = exchange(pl, send, pl->recv, as_stream(stream))) return s;
    return ddh_a2a_unpack(pl->recv, rl, pl->n0, pl->n1 / P, pl->n2, pl->n3, P, stream);
}

// RL [N0][N1/P][N2][N3] -> CL [N0][N1][N2/P][N3]    (Transpose.increment: towards grid space)
int ddh_a2a_localize_columns(ddh_handle plan, const double *rl, double *cl, void *stream) {
    A2aPlan *pl = (A2aPlan *)lookup_handle(plan, H_A2A);
    if (!pl) return -1;
    const int P = pl->comm->nranks;
    if (cl == rl) return fail("ddh_a2a_localize_columns: the two layouts must be different buffers");
    if (!pl->even) {
        const int me = pl->comm->rank;
        const long n2me = blk_n(pl->n2, pl->B2, me), n1me = blk_n(pl->n1, pl->B1, me);
        std::vector<size_t> cs(P), ds(P), cr(P), dr(P);
        for (int p = 0; p < P; ++p) {
            cs[p] = (size_t)(pl->n0 * n1me * blk_n(pl->n2, pl->B2, p) * pl->n3);
            ds[p] = (size_t)(pl->n0 * n1me * pl->n3 * blk_lo(pl->n2, pl->B2, p));
            cr[p] = (size_t)(pl->n0 * blk_n(pl->n1, pl->B1, p) * n2me * pl->n3);
            dr[p] = (size_t)(pl->n0 * n2me * pl->n3 * blk_lo(pl->n1, pl->B1, p));
        }
        if (n1me > 0)
            if (int s = ddh_a2av_pack_b(rl, pl->send, pl->n0 * n1me, pl->n2, pl->n3, P, pl->B2, stream)) return s;
        if (int s = exchange_v(pl, cs, ds, cr, dr, as_stream(stream))) return s;
        if (n2me > 0) return ddh_a2av_unpack_b(pl->recv, cl, pl->n0, pl->n1, n2me * pl->n3, P, pl->B1, stream);
        return 0;
    }
    // split N2 into P column blocks ([N0 N1/P][N2][1][N3] view), exchange, gather the P row blocks along N1
    if (int s = ddh_a2a_pack(rl, pl->send, pl->n0 * (pl->n1 / P), pl->n2, 1, pl->n3, P, stream)) return s;
    // (N0 == 1: the received order [p][n1 / P][n2 / P][n3] IS the column-local array -- received in place, no unpack pass)
    if (pl->n0 == 1) return exchange(pl, pl->send, cl, as_stream(stream));
    if (int s = exchange(pl, pl->send, pl->recv, as_stream(stream))) return s;
    return ddh_a2a_unpack(pl->recv, cl, pl->n0, 1, pl->n1, (pl->n2 / P) * pl->n3, P, stream);
}

int ddh_a2a_forward(ddh_handle plan, const double *cl, double *rl, void *stream) {
    return ddh_a2a_localize_rows(plan, cl, rl, stream);
}
int ddh_a2a_backward(ddh_handle plan, const double *rl, double *cl, void *stream) {
    return ddh_a2a_localize_columns(plan, rl, cl, stream);
}

}  // extern "C"
