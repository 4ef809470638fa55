// Batched dense inversion on the device: (a M + b L)^-1 for every subproblem of the curvilinear solvers.
//
// The sphere's per-m and the shell's per-ell subproblem matrices (core/subsystems.py:497-596, built per subproblem
// by the reference and factorized with SuperLU whenever a0 / b0 or k H_ii change, core/timesteppers.py:172-181,
// 630-640, libraries/matsolvers.py:126-149) are a few hundred small dense systems here (n <= 768): a M + b L is formed
// and inverted in place on the device -- Gauss-Jordan with partial pivoting, one workgroup per system -- so a change
// of the timestep costs no host round trip.  Valid-mode filtering (core/subsystems.py:540-556): rows / columns that
// carry no mode for a system are paired into unit entries before the elimination and cleared afterwards, which
// yields exactly the inverse of the valid block embedded in zeros.
#include "ddh_common.h"

namespace ddh {

template <bool CX> struct DEl;
template <> struct DEl<false> {
    typedef double T;
    static __device__ __forceinline__ T zero() { return 0.0; }
    static __device__ __forceinline__ T one() { return 1.0; }
    static __device__ __forceinline__ double abs2(T a) { return a * a; }
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T inv(T a) { return 1.0 / a; }
    static __device__ __forceinline__ T neg(T a) { return -a; }
    static __device__ __forceinline__ T fms(T c, T a, T b) { return c - a * b; }
    static __device__ __forceinline__ T comb(T m, T l, double a, double b) { return a * m + b * l; }
    static __device__ __forceinline__ T lane_value(T x, int l);
};
template <> struct DEl<true> {
    typedef double2 T;
    static __device__ __forceinline__ T zero() { return make_double2(0.0, 0.0); }
    static __device__ __forceinline__ T one() { return make_double2(1.0, 0.0); }
    static __device__ __forceinline__ double abs2(T a) { return a.x * a.x + a.y * a.y; }
    static __device__ __forceinline__ T mul(T a, T b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
    static __device__ __forceinline__ T inv(T a) {
        const double d = 1.0 / (a.x * a.x + a.y * a.y);
        return make_double2(a.x * d, -a.y * d);
    }
    static __device__ __forceinline__ T neg(T a) { return make_double2(-a.x, -a.y); }
    static __device__ __forceinline__ T fms(T c, T a, T b) {
        return make_double2(c.x - (a.x * b.x - a.y * b.y), c.y - (a.x * b.y + a.y * b.x));
    }
    static __device__ __forceinline__ T comb(T m, T l, double a, double b) {
        return make_double2(a * m.x + b * l.x, a * m.y + b * l.y);
    }
    static __device__ __forceinline__ T lane_value(T x, int l);
};

// the value a given lane of the wavefront holds, through the scalar unit (l uniform)
__device__ __forceinline__ double lane_double(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

__device__ __forceinline__ double DEl<false>::lane_value(double x, int l) { return lane_double(x, l); }
__device__ __forceinline__ double2 DEl<true>::lane_value(double2 x, int l) {
    return make_double2(lane_double(x.x, l), lane_double(x.y, l));
}

constexpr int DI_T = 1024;       // threads per system
constexpr int DI_NMAX = 1024;    // largest system

struct DenseSys {
    long off;        // element offset of the n x n row-major matrix in M, L and the output
    int n;
    int pair0, npair;   // slice of the (invalid row, invalid column) pair lists
};

// out = (a M + b L)^-1 on the valid block, zeros elsewhere.  One workgroup per system.
//
// Blocked in-place Gauss-Jordan inversion with partial pivoting.  One elimination step k replaces column k by the unit
// vector e_k and applies a rank-1 update  A <- A - u r^T  (r = row k / pivot with r[k] = 1 / pivot, u = column k with
// u[k] = pivot - 1).  BS steps are collected before the matrix is touched: inside a block only the current pivot column
// and pivot row are brought up to date (n BS work each) from the stale matrix and the pending (u, r) pairs kept in LDS,
// then ONE pass over the matrix applies the rank-BS update -- the matrix streams through memory n / BS times instead
// of n times (the elimination is bound by that stream: 127 shell systems of 768^2 do not fit the caches).
template <bool CX, int BS>
__global__ void __launch_bounds__(DI_T)
dense_inverse_kernel(const DenseSys *__restrict__ sys, const void *__restrict__ Mv, const void *__restrict__ Lv,
                     double a, double b, const unsigned char *__restrict__ row_valid,
                     const unsigned char *__restrict__ col_valid, const long *__restrict__ valid_off,
                     const int *__restrict__ pair_r, const int *__restrict__ pair_c, void *__restrict__ outv,
                     int *__restrict__ flags) {
    typedef typename DEl<CX>::T E;
    typedef DEl<CX> O;
    const DenseSys S = sys[blockIdx.x];
    const int n = S.n, tid = threadIdx.x;
    if (n <= 0 || n > DI_NMAX) return;           // larger systems: the multi-launch path (big_* kernels below)
    const E *M = (const E *)Mv + S.off, *L = (const E *)Lv + S.off;
    E *A = (E *)outv + S.off;
    const unsigned char *rv = row_valid + valid_off[blockIdx.x], *cv = col_valid + valid_off[blockIdx.x];
    extern __shared__ double s_dyn[];
    E *sU = (E *)s_dyn;                      // [BS][n]
    E *sR = sU + (size_t)BS * n;             // [BS][n]
    E *s_col = sR + (size_t)BS * n;          // [n] current pivot column
    __shared__ int s_piv[DI_NMAX];
    __shared__ double s_best[DI_T / 64];
    __shared__ int s_arg[DI_T / 64];
    __shared__ int s_p;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    const long nn = (long)n * n;
    // ---- assemble: a M + b L on valid rows x valid columns, zero elsewhere, unit entries pairing the invalid ones
    for (long e = tid; e < nn; e += DI_T) {
        const int i = (int)(e / n), j = (int)(e - (long)i * n);
        A[e] = (rv[i] && cv[j]) ? O::comb(M[e], L[e], a, b) : O::zero();
    }
    __syncthreads();
    for (int k = tid; k < S.npair; k += DI_T) A[(long)pair_r[S.pair0 + k] * n + pair_c[S.pair0 + k]] = O::one();
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += BS) {
        const int nb = (n - k0 < BS) ? n - k0 : BS;
        for (int s = 0; s < nb; ++s) {
            const int k = k0 + s;
            // (1) pivot column k, up to date: stale column minus the pending updates
            for (int i = tid; i < n; i += DI_T) {
                E v = A[(long)i * n + k];
                for (int t = 0; t < s; ++t) v = O::fms(v, sU[(size_t)t * n + i], sR[(size_t)t * n + k]);
                s_col[i] = v;
            }
            __syncthreads();
            // (2) pivot search among rows >= k
            double best = -1.0;
            int arg = k;
            for (int i = k + tid; i < n; i += DI_T) {
                const double m = O::abs2(s_col[i]);
                if (m > best) { best = m; arg = i; }
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                const double ob = __shfl_xor(best, sft, 64);
                const int oa = __shfl_xor(arg, sft, 64);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_arg[tid >> 6] = arg; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < DI_T / 64; ++w)
                    if (s_best[w] > best || (s_best[w] == best && s_arg[w] < arg)) { best = s_best[w]; arg = s_arg[w]; }
                s_p = arg;
                s_piv[k] = arg;
                if (!(best > 0.0)) s_bad = 1;
            }
            __syncthreads();
            const int p = s_p;
            // (3) row interchange k <-> p: the stale matrix, the pending u vectors and the current column
            if (p != k) {
                for (int j = tid; j < n; j += DI_T) {
                    const E t = A[(long)k * n + j];
                    A[(long)k * n + j] = A[(long)p * n + j];
                    A[(long)p * n + j] = t;
                }
                if (tid < s) {
                    const E t = sU[(size_t)tid * n + k];
                    sU[(size_t)tid * n + k] = sU[(size_t)tid * n + p];
                    sU[(size_t)tid * n + p] = t;
                }
                if (tid == DI_T - 1) {
                    const E t = s_col[k];
                    s_col[k] = s_col[p];
                    s_col[p] = t;
                }
            }
            __syncthreads();
            // (4) pivot row k, up to date; columns replaced earlier in this block start from the unit vector's 0
            E piv = s_col[k];
            if (!(O::abs2(piv) > 0.0)) piv = O::one();
            const E ip = O::inv(piv);
            for (int j = tid; j < n; j += DI_T) {
                const bool repl = (j >= k0 && j < k);
                E v = repl ? O::zero() : A[(long)k * n + j];
                for (int t = repl ? j - k0 : 0; t < s; ++t) v = O::fms(v, sU[(size_t)t * n + k], sR[(size_t)t * n + j]);
                sR[(size_t)s * n + j] = (j == k) ? ip : O::mul(v, ip);
                E u = s_col[j];                                   // (n entries as well: same loop, index = row here)
                if (j == k) { u = piv; u = O::fms(u, O::one(), O::one()); }      // u[k] = pivot - 1
                sU[(size_t)s * n + j] = u;
            }
            __syncthreads();
        }
        if constexpr (CX) {
            // ---- one pass over the matrix: rank-nb update; thread = column (n <= DI_T), pending r values in registers.
            // One workgroup streams its matrix n / BS times and does n^2 (row, pending update) steps per thread: this loop is
            // the cost of a system.  The u value of a step is the same for every thread; read from LDS per step it was
            // the bottleneck of the COMPLEX systems (765 x 765: 194 ms, 136 ms with 64-bit instead of 128-bit reads, 93 ms now;
            // real systems read 64-bit values, for them the LDS broadcast is the cheaper way: 11 vs 24 ms at 640 x 640).  Each lane
            // reads the u values of ONE row of a 64-row chunk instead (conflict free, one read per 64 rows) and the value of a
            // row is broadcast from that lane's register through the scalar unit (v_readlane).  Rows go in batches of RB whose
            // loads are issued before the first store of the batch (a store followed by the next row's load may alias for all
            // the compiler knows: one load in flight per thread otherwise).
            if ((tid & ~63) < n) {                  // wavefronts without a column leave; the others keep all lanes (readlane)
                const int j = tid;
                const bool act = j < n;
                const int jc = act ? j : n - 1;     // lanes beyond the last column shadow it and never store
                const bool repl = (jc >= k0 && jc < k0 + nb);
                const int t0 = repl ? jc - k0 : 0;
                E r[BS];
    #pragma unroll
                for (int t = 0; t < BS; ++t) r[t] = (t >= t0 && t < nb) ? sR[(size_t)t * n + jc] : O::zero();
                constexpr int RB = 8;
                const int lane = tid & 63;
                for (int c0 = 0; c0 < n; c0 += 64) {
                    E ul[BS];
    #pragma unroll
                    for (int t = 0; t < BS; ++t) ul[t] = (c0 + lane < n) ? sU[(size_t)t * n + c0 + lane] : O::zero();
                    const int rows = (n - c0 < 64) ? n - c0 : 64;
                    int q0 = 0;
                    for (; q0 + RB <= rows; q0 += RB) {
                        E v[RB];
    #pragma unroll
                        for (int q = 0; q < RB; ++q) v[q] = A[(long)(c0 + q0 + q) * n + jc];
    #pragma unroll
                        for (int q = 0; q < RB; ++q) {
                            if (repl) v[q] = (c0 + q0 + q == jc) ? O::one() : O::zero();
    #pragma unroll
                            for (int t = 0; t < BS; ++t) v[q] = O::fms(v[q], O::lane_value(ul[t], q0 + q), r[t]);
                        }
                        if (act) {
    #pragma unroll
                            for (int q = 0; q < RB; ++q) A[(long)(c0 + q0 + q) * n + jc] = v[q];
                        }
                    }
                    for (; q0 < rows; ++q0) {
                        E v = repl ? ((c0 + q0 == jc) ? O::one() : O::zero()) : A[(long)(c0 + q0) * n + jc];
    #pragma unroll
                        for (int t = 0; t < BS; ++t) v = O::fms(v, O::lane_value(ul[t], q0), r[t]);
                        if (act) A[(long)(c0 + q0) * n + jc] = v;
                    }
                }
            }
        } else {
            // ---- one pass over the matrix: rank-nb update; thread = column, pending r values in registers
            for (int j = tid; j < n; j += DI_T) {
                const bool repl = (j >= k0 && j < k0 + nb);
                const int t0 = repl ? j - k0 : 0;
                E r[BS];
    #pragma unroll
                for (int t = 0; t < BS; ++t) r[t] = (t >= t0 && t < nb) ? sR[(size_t)t * n + j] : O::zero();
                // Rows in batches of RB, software-pipelined over two register sets: the loads of batch b + 1 are issued BEFORE
                // batch b is updated and stored.  (Row by row a thread had one load in flight -- a store followed by the next
                // row's load may alias for all the compiler knows; and loads wait in order behind earlier stores: with the
                // next batch's loads issued first, waiting for a batch never waits for a store acknowledgement.)  One
                // workgroup streams its matrix n / BS times: this loop is the cost of a system.
                constexpr int RB = 8;
                auto fetch = [&](int i0, E *v) {          // unconditional loads (straight-line code: exact vmcnt waits) ...
    #pragma unroll
                    for (int q = 0; q < RB; ++q) v[q] = A[(long)(i0 + q) * n + j];
                };
                auto finish = [&](int i0, E *v) {         // ... the replaced columns start from the unit vector instead
    #pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        if (repl) v[q] = (i0 + q == j) ? O::one() : O::zero();
    #pragma unroll
                        for (int t = 0; t < BS; ++t) v[q] = O::fms(v[q], sU[(size_t)t * n + i0 + q], r[t]);
                    }
    #pragma unroll
                    for (int q = 0; q < RB; ++q) A[(long)(i0 + q) * n + j] = v[q];
                };
                E va[RB], vb[RB];
                int i = 0;
                const int nfull = n / RB;            // full batches
                if (nfull > 0) fetch(0, va);
                int bidx = 0;
                for (; bidx + 1 < nfull; bidx += 2) {
                    fetch((bidx + 1) * RB, vb);
                    finish(bidx * RB, va);
                    if (bidx + 2 < nfull) fetch((bidx + 2) * RB, va);
                    finish((bidx + 1) * RB, vb);
                }
                if (bidx < nfull) finish(bidx * RB, va);
                i = nfull * RB;
                for (; i < n; ++i) {
                    E v = repl ? ((i == j) ? O::one() : O::zero()) : A[(long)i * n + j];
    #pragma unroll
                    for (int t = 0; t < BS; ++t) v = O::fms(v, sU[(size_t)t * n + i], r[t]);
                    A[(long)i * n + j] = v;
                }
            }
        }
        __syncthreads();
    }
    // ---- undo the row interchanges as column interchanges, in reverse order
    for (int k = n - 1; k >= 0; --k) {
        const int p = s_piv[k];
        if (p != k) {
            for (int i = tid; i < n; i += DI_T) {
                const E t = A[(long)i * n + k];
                A[(long)i * n + k] = A[(long)i * n + p];
                A[(long)i * n + p] = t;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- the paired unit entries are not part of the inverse
    for (int k = tid; k < S.npair; k += DI_T) A[(long)pair_c[S.pair0 + k] * n + pair_r[S.pair0 + k]] = O::zero();
    if (tid == 0 && s_bad) flags[blockIdx.x] = 1;
}


// ------------------------------------------------------------------------------------------------
// Systems larger than one workgroup's reach (n > DI_NMAX; the mean-mode pencil of the 3-D Rayleigh-Benard problem is
// 1289 x 1289): the same blocked Gauss-Jordan, one system at a time, split over launches -- a single-workgroup PANEL
// kernel brings the BS pivot columns / rows of a block up to date (steps (1)-(4) above, pending vectors in global
// memory) and a chip-wide UPDATE kernel applies the rank-BS update to the whole matrix.  The matrix lives in a scratch
// buffer; a final gather undoes the row interchanges (as a column permutation) into the output slot.
// ------------------------------------------------------------------------------------------------
constexpr int BIG_BS = 16;
constexpr int BIG_RW = 64;      // rows per workgroup of the update kernel

template <bool CX>
__global__ void __launch_bounds__(256)
big_assemble_kernel(const void *__restrict__ Mv, const void *__restrict__ Lv, double a, double b,
                    const unsigned char *__restrict__ rv, const unsigned char *__restrict__ cv, void *__restrict__ Av, int n) {
    typedef typename DEl<CX>::T E;
    typedef DEl<CX> O;
    const E *M = (const E *)Mv, *L = (const E *)Lv;
    E *A = (E *)Av;
    const long nn = (long)n * n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nn; e += (long)gridDim.x * 256) {
        const int i = (int)(e / n), j = (int)(e - (long)i * n);
        A[e] = (rv[i] && cv[j]) ? O::comb(M[e], L[e], a, b) : O::zero();
    }
}

template <bool CX>
__global__ void __launch_bounds__(256)
big_pairs_kernel(void *__restrict__ Av, int n, const int *__restrict__ pr, const int *__restrict__ pc, int npair, int transpose_zero) {
    typedef typename DEl<CX>::T E;
    typedef DEl<CX> O;
    E *A = (E *)Av;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < npair; k += gridDim.x * 256) {
        if (transpose_zero) A[(long)pc[k] * n + pr[k]] = O::zero();      // after the inversion: not part of the inverse
        else A[(long)pr[k] * n + pc[k]] = O::one();
    }
}

template <bool CX, int BS>
__global__ void __launch_bounds__(DI_T)
big_panel_kernel(void *__restrict__ Av, int n, int k0, void *__restrict__ sUv, void *__restrict__ sRv, void *__restrict__ scv,
                 int *__restrict__ s_piv, int *__restrict__ flag) {
    typedef typename DEl<CX>::T E;
    typedef DEl<CX> O;
    E *A = (E *)Av, *sU = (E *)sUv, *sR = (E *)sRv, *s_col = (E *)scv;
    const int tid = threadIdx.x;
    __shared__ double s_best[DI_T / 64];
    __shared__ int s_arg[DI_T / 64];
    __shared__ int s_p;
    const int nb = (n - k0 < BS) ? n - k0 : BS;
    for (int s = 0; s < nb; ++s) {
        const int k = k0 + s;
        for (int i = tid; i < n; i += DI_T) {
            E v = A[(long)i * n + k];
            for (int t = 0; t < s; ++t) v = O::fms(v, sU[(size_t)t * n + i], sR[(size_t)t * n + k]);
            s_col[i] = v;
        }
        __syncthreads();
        double best = -1.0;
        int arg = k;
        for (int i = k + tid; i < n; i += DI_T) {
            const double m = O::abs2(s_col[i]);
            if (m > best) { best = m; arg = i; }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const double ob = __shfl_xor(best, sft, 64);
            const int oa = __shfl_xor(arg, sft, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_arg[tid >> 6] = arg; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < DI_T / 64; ++w)
                if (s_best[w] > best || (s_best[w] == best && s_arg[w] < arg)) { best = s_best[w]; arg = s_arg[w]; }
            s_p = arg;
            s_piv[k] = arg;
            if (!(best > 0.0)) *flag = 1;
        }
        __syncthreads();
        const int p = s_p;
        if (p != k) {
            for (int j = tid; j < n; j += DI_T) {
                const E t = A[(long)k * n + j];
                A[(long)k * n + j] = A[(long)p * n + j];
                A[(long)p * n + j] = t;
            }
            if (tid < s) {
                const E t = sU[(size_t)tid * n + k];
                sU[(size_t)tid * n + k] = sU[(size_t)tid * n + p];
                sU[(size_t)tid * n + p] = t;
            }
            if (tid == DI_T - 1) {
                const E t = s_col[k];
                s_col[k] = s_col[p];
                s_col[p] = t;
            }
        }
        __syncthreads();
        E piv = s_col[k];
        if (!(O::abs2(piv) > 0.0)) piv = O::one();
        const E ip = O::inv(piv);
        for (int j = tid; j < n; j += DI_T) {
            const bool repl = (j >= k0 && j < k);
            E v = repl ? O::zero() : A[(long)k * n + j];
            for (int t = repl ? j - k0 : 0; t < s; ++t) v = O::fms(v, sU[(size_t)t * n + k], sR[(size_t)t * n + j]);
            sR[(size_t)s * n + j] = (j == k) ? ip : O::mul(v, ip);
            E u = s_col[j];
            if (j == k) { u = piv; u = O::fms(u, O::one(), O::one()); }
            sU[(size_t)s * n + j] = u;
        }
        __syncthreads();
    }
}

template <bool CX, int BS>
__global__ void __launch_bounds__(256)
big_update_kernel(void *__restrict__ Av, int n, int k0, const void *__restrict__ sUv, const void *__restrict__ sRv) {
    typedef typename DEl<CX>::T E;
    typedef DEl<CX> O;
    E *A = (E *)Av;
    const E *sU = (const E *)sUv, *sR = (const E *)sRv;
    __shared__ E su[BS][BIG_RW];
    const int nb = (n - k0 < BS) ? n - k0 : BS;
    const int i0 = blockIdx.y * BIG_RW;
    for (int w = threadIdx.x; w < BS * BIG_RW; w += 256) {
        const int t = w / BIG_RW, q = w - t * BIG_RW;
        su[t][q] = (t < nb && i0 + q < n) ? sU[(size_t)t * n + i0 + q] : O::zero();
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const bool repl = (j >= k0 && j < k0 + nb);
    const int t0 = repl ? j - k0 : 0;
    E r[BS];
#pragma unroll
    for (int t = 0; t < BS; ++t) r[t] = (t >= t0 && t < nb) ? sR[(size_t)t * n + j] : O::zero();
    const int rows = (n - i0 < BIG_RW) ? n - i0 : BIG_RW;
    for (int q = 0; q < rows; ++q) {
        const int i = i0 + q;
        E v = repl ? ((i == j) ? O::one() : O::zero()) : A[(long)i * n + j];
#pragma unroll
        for (int t = 0; t < BS; ++t) v = O::fms(v, su[t][q], r[t]);
        A[(long)i * n + j] = v;
    }
}

// out[i][j] = A[i][src[j]] with src = the column permutation that undoes the row interchanges in reverse order
template <bool CX>
__global__ void __launch_bounds__(256)
big_unpivot_kernel(const void *__restrict__ Av, void *__restrict__ outv, int n, const int *__restrict__ piv) {
    typedef typename DEl<CX>::T E;
    const E *A = (const E *)Av;
    E *out = (E *)outv;
    extern __shared__ int s_src[];
    for (int j = threadIdx.x; j < n; j += 256) s_src[j] = j;
    __syncthreads();
    if (threadIdx.x == 0) {
        // the column swaps (k, piv[k]), k = n - 1 .. 0, applied to A in turn: column j of the result is column s_src[j] of A
        for (int k = n - 1; k >= 0; --k) {
            const int p = piv[k];
            if (p != k) {
                const int t = s_src[k];
                s_src[k] = s_src[p];
                s_src[p] = t;
            }
        }
    }
    __syncthreads();
    const int i0 = blockIdx.x * BIG_RW;
    for (int q = 0; q < BIG_RW && i0 + q < n; ++q) {
        const long row = (long)(i0 + q) * n;
        for (int j = threadIdx.x; j < n; j += 256) out[row + j] = A[row + s_src[j]];
    }
}

struct DenseInverse : HandleBase {
    int nsys = 0, cx = 0;
    long total = 0;               // elements over all systems
    void *d_sys = nullptr, *d_M = nullptr, *d_L = nullptr, *d_rv = nullptr, *d_cv = nullptr, *d_voff = nullptr;
    void *d_pr = nullptr, *d_pc = nullptr, *d_flags = nullptr;
    std::vector<int> n_host;
    std::vector<DenseSys> sys_host;
    std::vector<long> voff_host;
    void *d_big = nullptr, *d_bU = nullptr, *d_bR = nullptr, *d_bcol = nullptr, *d_bpiv = nullptr;   // n > DI_NMAX systems
    ~DenseInverse() override {
        (void)hipFree(d_big); (void)hipFree(d_bU); (void)hipFree(d_bR); (void)hipFree(d_bcol); (void)hipFree(d_bpiv);
        (void)hipFree(d_sys); (void)hipFree(d_M); (void)hipFree(d_L); (void)hipFree(d_rv); (void)hipFree(d_cv);
        (void)hipFree(d_voff); (void)hipFree(d_pr); (void)hipFree(d_pc); (void)hipFree(d_flags);
    }
};

// [nl][RN][RN] row-major inverses -> EllTerms dense storage [t = co * R + ci][ell][n_in][n_out] (transposed blocks)
__global__ void __launch_bounds__(256)
ell_blocks_kernel(const double *__restrict__ inv, double *__restrict__ mats, int nl, int R, int nr) {
    const long RN = (long)R * nr;
    const long total = (long)nl * RN * RN;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        // destination-ordered: e = ((t * nl + l) * nr + ni) * nr + no
        const int no = (int)(e % nr);
        const int ni = (int)((e / nr) % nr);
        const int l = (int)((e / ((long)nr * nr)) % nl);
        const int t = (int)(e / ((long)nr * nr * nl));
        const int co = t / R, ci = t - co * R;
        mats[e] = inv[((long)l * RN + (long)co * nr + no) * RN + (long)ci * nr + ni];
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_dense_inverse_create(ddh_handle *h, int nsys, const int *n_h, int is_complex, const double *M_h,
                             const double *L_h, const unsigned char *row_valid_h, const unsigned char *col_valid_h) {
    if (!h || nsys < 1) return fail("dense_inverse_create: bad arguments");
    DenseInverse *p = new DenseInverse();
    p->kind = H_DENSEINV;
    p->nsys = nsys;
    p->cx = is_complex ? 1 : 0;
    std::vector<DenseSys> sys(nsys);
    std::vector<long> voff(nsys);
    std::vector<int> pr, pc;
    long off = 0, vo = 0;
    for (int s = 0; s < nsys; ++s) {
        const int n = n_h[s];
        if (n < 0 || n > 16384) { delete p; return fail("dense_inverse_create: system size out of range (0..16384)"); }
        sys[s].off = off;
        sys[s].n = n;
        sys[s].pair0 = (int)pr.size();
        voff[s] = vo;
        std::vector<int> br, bc;
        for (int i = 0; i < n; ++i) {
            if (!row_valid_h[vo + i]) br.push_back(i);
            if (!col_valid_h[vo + i]) bc.push_back(i);
        }
        if (br.size() != bc.size()) { delete p; return fail("dense_inverse_create: valid rows and columns do not balance"); }
        for (size_t k = 0; k < br.size(); ++k) { pr.push_back(br[k]); pc.push_back(bc[k]); }
        sys[s].npair = (int)br.size();
        off += (long)n * n;
        vo += n;
    }
    p->total = off;
    p->n_host.assign(n_h, n_h + nsys);
    p->sys_host = sys;
    p->voff_host = voff;
    const size_t eb = (p->cx ? 2 : 1) * sizeof(double);
    int st = 0;
    auto up = [&](void **d, const void *src, size_t bytes) {
        if (st) return;
        st = check_hip(hipMalloc(d, bytes + 16), "hipMalloc");
        if (!st && bytes) st = check_hip(hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice), "hipMemcpy");
    };
    up(&p->d_sys, sys.data(), sys.size() * sizeof(DenseSys));
    up(&p->d_M, M_h, (size_t)off * eb);
    up(&p->d_L, L_h, (size_t)off * eb);
    up(&p->d_rv, row_valid_h, (size_t)vo);
    up(&p->d_cv, col_valid_h, (size_t)vo);
    up(&p->d_voff, voff.data(), voff.size() * sizeof(long));
    up(&p->d_pr, pr.data(), pr.size() * sizeof(int));
    up(&p->d_pc, pc.data(), pc.size() * sizeof(int));
    if (!st) st = check_hip(hipMalloc(&p->d_flags, (size_t)nsys * sizeof(int)), "hipMalloc");
    if (st) { delete p; return st; }
    *h = register_handle(p);
    return 0;
}

int ddh_dense_inverse_elements(ddh_handle h, long *count) {
    DenseInverse *p = (DenseInverse *)lookup_handle(h, H_DENSEINV);
    if (!p) return -1;
    *count = p->total * (p->cx ? 2 : 1);
    return 0;
}

int ddh_dense_inverse_compute(ddh_handle h, double a, double b, double *out_d, int *nsingular_h, void *stream) {
    DenseInverse *p = (DenseInverse *)lookup_handle(h, H_DENSEINV);
    if (!p) return -1;
    hipStream_t s = as_stream(stream);
    DDH_HIP(hipMemsetAsync(p->d_flags, 0, (size_t)p->nsys * sizeof(int), s));
    int nmax = 0;
    for (int v : p->n_host) nmax = v > nmax ? v : nmax;
    // block size: as many pending rank-1 updates as the LDS holds ((2 BS + 1) n elements) -- the matrix streams through
    // memory n / BS times, and that stream is what a system costs
#define DDH_DI_LAUNCH(CXV, BSV, ELT)                                                                                   \
    {                                                                                                                  \
        const size_t lds = (size_t)(2 * BSV + 1) * nmax * sizeof(ELT);                                                 \
        auto kern = dense_inverse_kernel<CXV, BSV>;                                                                    \
        DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kern, dim3((unsigned)p->nsys), dim3(DI_T), lds, s, (const DenseSys *)p->d_sys, p->d_M,      \
                           p->d_L, a, b, (const unsigned char *)p->d_rv, (const unsigned char *)p->d_cv,               \
                           (const long *)p->d_voff, (const int *)p->d_pr, (const int *)p->d_pc, (void *)out_d,         \
                           (int *)p->d_flags);                                                                         \
    }
    const size_t budget = 150 * 1024;      // of the CU's 160 KiB (the kernel keeps ~5 KiB of static arrays)
    static const int bs_env = getenv("DDH_DENSE_BS") ? atoi(getenv("DDH_DENSE_BS")) : 0;
    int nbig = 0, nmax_big = 0;
    nmax = 0;
    for (int v : p->n_host) {
        if (v > DI_NMAX) {
            ++nbig;
            nmax_big = v > nmax_big ? v : nmax_big;
        } else {
            nmax = v > nmax ? v : nmax;
        }
    }
    if (nmax > 0) {
        if (p->cx) {
            // (5 instead of 4 pending updates fit for n <= 870 and were measured on the sphere's 256 systems: no change,
            // 195 ms -- there the largest systems, one workgroup each, set the time)
            DDH_DI_LAUNCH(true, 4, double2)
        } else {
            int bs = 8;
            if ((size_t)(2 * 12 + 1) * nmax * sizeof(double) <= budget) bs = 12;
            if (bs_env) bs = bs_env;
            if (bs == 12) DDH_DI_LAUNCH(false, 12, double) else DDH_DI_LAUNCH(false, 8, double)
        }
    }
#undef DDH_DI_LAUNCH
    DDH_HIP(hipGetLastError());
    if (nbig) {
        const size_t eb = (p->cx ? 2 : 1) * sizeof(double);
        if (!p->d_big) {
            DDH_HIP(hipMalloc(&p->d_big, (size_t)nmax_big * nmax_big * eb));
            DDH_HIP(hipMalloc(&p->d_bU, (size_t)BIG_BS * nmax_big * eb));
            DDH_HIP(hipMalloc(&p->d_bR, (size_t)BIG_BS * nmax_big * eb));
            DDH_HIP(hipMalloc(&p->d_bcol, (size_t)nmax_big * eb));
            DDH_HIP(hipMalloc(&p->d_bpiv, (size_t)nmax_big * sizeof(int)));
        }
        for (int sidx = 0; sidx < p->nsys; ++sidx) {
            const DenseSys &S = p->sys_host[sidx];
            const int n = S.n;
            if (n <= DI_NMAX) continue;
            const char *Mp = (const char *)p->d_M + (size_t)S.off * eb, *Lp = (const char *)p->d_L + (size_t)S.off * eb;
            const unsigned char *rvp = (const unsigned char *)p->d_rv + p->voff_host[sidx];
            const unsigned char *cvp = (const unsigned char *)p->d_cv + p->voff_host[sidx];
            const int *prp = (const int *)p->d_pr + S.pair0, *pcp = (const int *)p->d_pc + S.pair0;
            char *outp = (char *)out_d + (size_t)S.off * eb;
            int *flagp = (int *)p->d_flags + sidx;
            const dim3 ugrid((unsigned)((n + 255) / 256), (unsigned)((n + BIG_RW - 1) / BIG_RW));
#define DDH_BIG(CXV)                                                                                                   \
    {                                                                                                                  \
        hipLaunchKernelGGL(big_assemble_kernel<CXV>, dim3(2048), dim3(256), 0, s, Mp, Lp, a, b, rvp, cvp, p->d_big, n); \
        if (S.npair)                                                                                                   \
            hipLaunchKernelGGL(big_pairs_kernel<CXV>, dim3((unsigned)((S.npair + 255) / 256)), dim3(256), 0, s,         \
                               p->d_big, n, prp, pcp, S.npair, 0);                                                     \
        for (int k0 = 0; k0 < n; k0 += BIG_BS) {                                                                       \
            hipLaunchKernelGGL((big_panel_kernel<CXV, BIG_BS>), dim3(1), dim3(DI_T), 0, s, p->d_big, n, k0, p->d_bU,    \
                               p->d_bR, p->d_bcol, (int *)p->d_bpiv, flagp);                                           \
            hipLaunchKernelGGL((big_update_kernel<CXV, BIG_BS>), ugrid, dim3(256), 0, s, p->d_big, n, k0, p->d_bU,      \
                               p->d_bR);                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL(big_unpivot_kernel<CXV>, dim3((unsigned)((n + BIG_RW - 1) / BIG_RW)), dim3(256),             \
                           (size_t)n * sizeof(int), s, p->d_big, (void *)outp, n, (const int *)p->d_bpiv);             \
        if (S.npair)                                                                                                   \
            hipLaunchKernelGGL(big_pairs_kernel<CXV>, dim3((unsigned)((S.npair + 255) / 256)), dim3(256), 0, s,         \
                               (void *)outp, n, prp, pcp, S.npair, 1);                                                 \
    }
            if (p->cx) DDH_BIG(true) else DDH_BIG(false)
#undef DDH_BIG
        }
        DDH_HIP(hipGetLastError());
    }
    if (nsingular_h) {
        std::vector<int> f(p->nsys);
        DDH_HIP(hipMemcpyAsync(f.data(), p->d_flags, (size_t)p->nsys * sizeof(int), hipMemcpyDeviceToHost, s));
        DDH_HIP(hipStreamSynchronize(s));
        int c = 0;
        for (int v : f) c += v ? 1 : 0;
        *nsingular_h = c;
    }
    return 0;
}

int ddh_ell_blocks_from_dense(const double *inv_d, double *mats_d, int nl, int ncomp, int nr, void *stream) {
    if (nl < 1 || ncomp < 1 || nr < 1) return fail("ell_blocks_from_dense: bad sizes");
    hipLaunchKernelGGL(ell_blocks_kernel, dim3(4096), dim3(256), 0, as_stream(stream), inv_d, mats_d, nl, ncomp, nr);
    DDH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
