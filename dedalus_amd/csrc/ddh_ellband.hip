// Band LU of the per-group LHS systems of the curvilinear solvers (the shell's per-ell systems).
//
// The reference factorizes every subproblem matrix a0 M + b0 L with a sparse LU whenever the timestep changes and
// solves every (m, part) column against it (core/timesteppers.py:172-181, 630-640, libraries/matsolvers.py:129-160).
// core/ellband.py finds once, on the host, the permutation + column recombination that makes these matrices narrow
// bands; here
//   * ddh_ellband_factor forms a M + b L in band storage and runs a partial-pivoting band LU (gbtrf-shaped: row swaps
//     applied to the trailing columns only, multipliers kept per column) -- one workgroup per group, the kl + 1 active
//     rows live in LDS and slide down the band;
//   * ddh_ellband_solve sweeps all right-hand sides: the factors of a group are the same for every (m, part) slot, so a
//     lane owns a slot, the factor rows are fetched once per wavefront and broadcast through v_readlane, and the sliding windows (kl + 1 right-hand side
//     rows forward, kl + ku solved unknowns backward) stay in registers.  The forward sweep gathers the permuted rows
//     straight from the solver's [component][slot][group][n] right-hand side and applies the boundary-row combination
//     T; the backward sweep undoes the column recombination (X = P Y, a short upper band in the permuted order) and
//     scatters into the state layout.
// Algorithmic cost per solve: 2 (kl + W + mp) n flops and 16 n bytes per (group, slot) -- against 2 n^2 for the dense
// inverse -- and the factorization is O(n kl W) per group instead of O(n^3).
#include "ddh_common.h"

namespace ddh {

constexpr int EB_NBC = 8;        // boundary rows per group
constexpr int EB_MP = 16;        // super diagonals of the recombination in the permuted order

struct EllBandLu {
    double *Lm = nullptr;        // [nl][nmax][64]       column j: pivot row offset (as a double), then the multipliers of
                                 //                      rows j+1 .., zero padded
    double *U = nullptr;         // [nl][nmax][nv * 64]  row i: 1 / diagonal, the kl + ku super diagonals zero padded to wt,
                                 //                      then the recombination's super diagonals of that row
};

struct EllBand : HandleBase {
    int nl = 0, nmax = 0, kl = 0, ku = 0, mp = 0, nbc = 0, nslots = 0;
    int nw = 0, wt = 0;          // compiled window sizes (forward rows, backward unknowns)
    int np = 0, nslots_pad = 0;  // rows per group incl. padding; slots rounded up to whole wavefronts
    double *dump_d = nullptr;
    long slot_stride = 0;
    int *n_d = nullptr, *nbc_d = nullptr, *slot_limit_d = nullptr, *flag_d = nullptr;
    long *rowoff_d = nullptr, *coloff_d = nullptr;
    double *T_d = nullptr, *P_d = nullptr, *MB_d = nullptr, *LB_d = nullptr, *work_d = nullptr;
    std::vector<EllBandLu> lus;
    std::vector<int> n_h;
    ~EllBand() override {
        for (void *p : {(void *)n_d, (void *)nbc_d, (void *)slot_limit_d, (void *)flag_d, (void *)rowoff_d, (void *)coloff_d,
                        (void *)T_d, (void *)P_d, (void *)MB_d, (void *)LB_d, (void *)work_d, (void *)dump_d})
            if (p) (void)hipFree(p);
        for (auto &l : lus) {
            if (l.Lm) (void)hipFree(l.Lm);
            if (l.U) (void)hipFree(l.U);
        }
    }
};

// ---- factorization ---------------------------------------------------------------------------------------------------
// Active window: rows j .. j + kl, columns j .. j + kl + ku (WC = kl + ku + 1 of them: a row that was swapped up from
// j + kl reaches that far).  Row r sits in LDS row r % (kl + 1), column c in slot c % WC; the pivot row leaves through
// U and the row that enters (j + kl + 1) takes its place, slot by slot.
constexpr int EB_FT = 256;

__global__ void __launch_bounds__(EB_FT)
ellband_factor_kernel(const int *__restrict__ n_d, const double *__restrict__ MB, const double *__restrict__ LB, double a,
                      double b, double *__restrict__ Lm, double *__restrict__ U, int *__restrict__ flag, int nmax, int np,
                      int kl, int ku, int lrow, int nmul, int wt, int urow) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const int n = n_d[g];
    if (n == 0) return;
    const int WC = kl + ku + 1, NR = kl + 1;
    extern __shared__ double eb_lds[];
    double *win = eb_lds;                 // [NR][WC]
    double *lmul = win + NR * WC;         // [kl]
    __shared__ int s_p;
    const double *Mg = MB + (size_t)g * nmax * WC, *Lg = LB + (size_t)g * nmax * WC;
    double *Lmg = Lm + (size_t)g * np * lrow, *Ug = U + (size_t)g * np * urow;
    // rows 0 .. kl: entry d of row i is column i - kl + d (columns < 0 hold zeros and alias columns that are zero too)
    for (int e = tid; e < NR * WC; e += EB_FT) {
        const int i = e / WC, d = e % WC;
        double v = 0.0;
        if (i < n) v = a * Mg[(size_t)i * WC + d] + b * Lg[(size_t)i * WC + d];
        int c = (i - kl + d) % WC;
        if (c < 0) c += WC;
        win[i * WC + c] = v;
    }
    int jr = 0, jc = 0;                   // j % NR, j % WC
    for (int j = 0; j < n; ++j) {
        const int nrows = min(kl, n - 1 - j);
        __syncthreads();
        if (tid < 64) {                   // pivot: first largest |entry| of column j among rows j .. j + nrows
            double v = -1.0;
            int r = jr + tid;
            if (r >= NR) r -= NR;
            if (tid <= nrows) v = fabs(win[r * WC + jc]);
            int idx = tid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double v2 = __shfl_xor(v, o);
                const int i2 = __shfl_xor(idx, o);
                if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
            }
            if (tid == 0) {
                s_p = idx;
                Lmg[(size_t)j * lrow] = (double)idx;
                if (!(v > 0.0)) atomicAdd(flag, 1);
            }
        }
        __syncthreads();
        const int dp = s_p;
        if (dp != 0) {
            int pr = jr + dp;
            if (pr >= NR) pr -= NR;
            for (int c = tid; c < WC; c += EB_FT) {
                const double t = win[jr * WC + c];
                win[jr * WC + c] = win[pr * WC + c];
                win[pr * WC + c] = t;
            }
            __syncthreads();
        }
        const double pv = win[jr * WC + jc];
        const double ipv = pv != 0.0 ? 1.0 / pv : 0.0;
        if (tid < nmul) {
            double l = 0.0;
            if (tid < nrows) {
                int r = jr + 1 + tid;
                if (r >= NR) r -= NR;
                l = win[r * WC + jc] * ipv;
                win[r * WC + jc] = 0.0;
                lmul[tid] = l;
            }
            Lmg[(size_t)j * lrow + 1 + tid] = l;
        }
        __syncthreads();
        // trailing update: rows j+1 .. j+nrows, columns j+1 .. j+WC-1
        for (int e = tid; e < nrows * (WC - 1); e += EB_FT) {
            const int t = e / (WC - 1), c1 = e % (WC - 1) + 1;
            int r = jr + 1 + t;
            if (r >= NR) r -= NR;
            int c = jc + c1;
            if (c >= WC) c -= WC;
            win[r * WC + c] -= lmul[t] * win[jr * WC + c];
        }
        __syncthreads();
        // pivot row -> U; row j + kl + 1 enters in its place (its columns j+1 .. j+WC map onto the same slots)
        const int inew = j + kl + 1;
        for (int c1 = tid; c1 <= wt; c1 += EB_FT) {
            if (c1 < WC) {
                int c = jc + c1;
                if (c >= WC) c -= WC;
                const double u = win[jr * WC + c];
                Ug[(size_t)j * urow + c1] = c1 == 0 ? ipv : u;
                const int d = c1 == 0 ? WC - 1 : c1 - 1;
                double v = 0.0;
                if (inew < n) v = a * Mg[(size_t)inew * WC + d] + b * Lg[(size_t)inew * WC + d];
                win[jr * WC + c] = v;
            } else {
                Ug[(size_t)j * urow + c1] = 0.0;
            }
        }
        if (++jr == NR) jr = 0;
        if (++jc == WC) jc = 0;
    }
}

// ---- sweeps --------------------------------------------------------------------------------------------------------
// One wavefront per (group, 64 slots).  The factor row of the step (pivot + multipliers forward, U row + recombination
// backward) is the same for all lanes: it is fetched with ONE coalesced vector load per 64 entries (lane l holds entry
// l) EB_D rows ahead of its use, parked in a two-row LDS ring one row ahead, and read back as broadcast ds_read_b128
// (two coefficients per LDS instruction, every lane the same address).  A single wave per workgroup is in flight, so
// the sweep time is its instruction count: the scalar unit's loads would serialize a round trip per row, v_readlane
// costs two VALU slots per coefficient.
// rows of factor data in flight: a divisor of the window (compile-time ring slots), as deep as the window allows
constexpr int eb_depth(int) { return 4; }
constexpr int EB_FLW = 64;       // doubles per forward factor row: [0] pivot offset, [1 .. nw-1] multipliers

// Row tables and factor rows are padded with EB_PAD zero rows per group, so the unrolled row loops run whole blocks
// without per-row branches (a branch per row would end the scheduling region: every row would then pay its own LDS and
// memory latencies): rows past the end multiply zeros and land in padding / a dump word.
constexpr int EB_PAD = 112;

// NW: rows of the sliding right-hand-side window (kl + 1 <= NW, a multiple of EB_D)
template <int NW>
__global__ void __launch_bounds__(64)
ellband_forward_kernel(const int *__restrict__ n_d, const int *__restrict__ nbc_d, const int *__restrict__ slot_limit,
                       const long *__restrict__ rowoff, const double *__restrict__ T, const double *__restrict__ FL,
                       const double *__restrict__ rhs, double *__restrict__ work, int np, int nslots, int nslots_pad,
                       long slot_stride, int nbcmax) {
    constexpr int EB_D = eb_depth(NW);
    static_assert(NW % EB_D == 0 && NW % 2 == 0 && NW + 2 <= EB_FLW, "window: a multiple of the prefetch depth");
    __shared__ double2 ring[EB_FLW / 2];
    const int g = blockIdx.y;
    const int n = n_d[g];
    const int lim = slot_limit[g];
    if (n == 0 || (int)blockIdx.x * 64 >= lim) return;
    const int lane = threadIdx.x;
    const int s = blockIdx.x * 64 + lane;
    const bool live = s < lim;                                  // slots past the limit carry zeros
    const double *src = rhs + (size_t)(s < nslots ? s : 0) * slot_stride;
    const long *ro = rowoff + (size_t)g * np;
    const double *Fg = FL + (size_t)g * np * EB_FLW + lane;
    double *wk = work + (size_t)g * np * nslots_pad + s;
    double *ring_w = reinterpret_cast<double *>(&ring[0]) + lane;
    double bw[NW], fr[EB_D];
    double2 cf[2][NW / 2 + 1];                                  // coefficients of the current / the next row (+ row offset)
    ring_w[0] = Fg[0];
#pragma unroll
    for (int q = 0; q < EB_D; ++q) fr[q] = Fg[(size_t)(q + 1) * EB_FLW];                 // rows 1 .. EB_D
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const double v = src[ro[q]];
        bw[q] = (q < n && live) ? v : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NW / 2 + 1; ++q) cf[0][q] = ring[q];
    // boundary rows (the first nbc of the permuted order): the combinations T that pick one unknown each
    const int nbc = nbc_d[g];
    if (nbc > 0) {
        const double *Tg = T + (size_t)g * nbcmax * nbcmax;
        double tb[EB_NBC];
#pragma unroll
        for (int q = 0; q < EB_NBC; ++q) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < EB_NBC; ++p)
                if (q < nbc && p < nbc) acc += Tg[q * nbcmax + p] * bw[p];
            tb[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < EB_NBC; ++q)
            if (q < nbc) bw[q] = tb[q];
    }
    for (int j0 = 0; j0 < n; j0 += NW) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = j0 + u;
            const double2 *c = cf[u & 1];
            const int d = __builtin_amdgcn_readfirstlane((int)c[0].x);
            // Row interchange bw[u] <-> bw[(u + d) % NW], wave-uniform d: pivot rows are the rule in these systems (~80 % of
            // the columns, offsets up to ~12), so the offset is resolved in two levels of scalar branches (groups of four)
            // instead of a chain of NW; the empty asm keeps them branches -- as selects they cost 4 NW VALU slots per row.
            if (d != 0) {
                asm volatile("");
#pragma unroll
                for (int gq = 0; gq < (NW + 3) / 4; ++gq)
                    if ((d >> 2) == gq) {
                        asm volatile("");
#pragma unroll
                        for (int r = 4 * gq; r < 4 * gq + 4; ++r)
                            if (r >= 1 && r < NW && d == r) {
                                asm volatile("");
                                const double tmp = bw[u];
                                bw[u] = bw[(u + r) % NW];
                                bw[(u + r) % NW] = tmp;
                            }
                    }
            }
            const double y = bw[u];
            // row j + 1: memory -> LDS -> the other coefficient set, while this row's multiply-adds run
            ring_w[0] = fr[u % EB_D];
            fr[u % EB_D] = Fg[(size_t)(j + 1 + EB_D) * EB_FLW];
#pragma unroll
            for (int q = 0; q < NW / 2 + 1; ++q) cf[(u + 1) & 1][q] = ring[q];
            wk[(size_t)j * nslots_pad] = y;
            bw[(u + 1) % NW] -= c[0].y * y;
#pragma unroll
            for (int r = 2; r < NW; r += 2) {
                bw[(u + r) % NW] -= c[r / 2].x * y;
                if (r + 1 < NW) bw[(u + r + 1) % NW] -= c[r / 2].y * y;
            }
            // the row that enters the window: its offset rides in entry NW of this factor row (no scalar load in the loop)
            const long off = __double_as_longlong(c[NW / 2].x);
            const double v = src[off];
            bw[u] = (j + NW < n && live) ? v : 0.0;
        }
    }
}

// WT: solved unknowns kept (kl + ku <= WT, even); a backward factor row is WT + 1 entries of U
// (1 / diagonal first) followed by the EB_MP super diagonals of the recombination, padded to NV * 64 doubles.
// The sweep starts `pad` rows past the end (zero factor rows, results to the dump word) so that it ends on row 0
// with whole blocks.
template <int WT>
__global__ void __launch_bounds__(64)
ellband_backward_kernel(const int *__restrict__ n_d, const int *__restrict__ slot_limit, const long *__restrict__ coloff,
                        const double *__restrict__ FU, const double *__restrict__ work, double *__restrict__ x,
                        double *__restrict__ dump, int np, int nslots, int nslots_pad, long slot_stride) {
    constexpr int EB_D = eb_depth(WT);
    static_assert(WT % EB_D == 0 && WT % 2 == 0, "window: a multiple of the prefetch depth");
    constexpr int NV = (WT + 1 + EB_MP + 2 + 63) / 64;
    constexpr int RW = NV * 64;
    constexpr int NC = (WT + 1 + EB_MP + 1) / 2 + 1;          // double2 pairs of a row in use: U, recombination, column offset
    __shared__ double2 ring[2][RW / 2];
    const int g = blockIdx.y;
    const int n = n_d[g];
    const int lim = slot_limit[g];
    if (n == 0 || (int)blockIdx.x * 64 >= lim) return;
    const int lane = threadIdx.x;
    const int s = blockIdx.x * 64 + lane;
    const bool own = s < nslots;                               // lanes past the last slot write to the dump word
    const unsigned long long dmp_a = reinterpret_cast<unsigned long long>(dump + lane);
    const unsigned long long dst_a = own ? reinterpret_cast<unsigned long long>(x + (size_t)s * slot_stride) : dmp_a;
    const long ownm = own ? -1L : 0L;
    const double *Fg = FU + (size_t)g * np * RW + lane;
    const double *wk = work + (size_t)g * np * nslots_pad + s;
    double *ring_w = reinterpret_cast<double *>(&ring[0][0]) + lane;
    const int nblk = (n + WT - 1) / WT;
    const int top = nblk * WT - 1;                              // first row of the sweep (>= n - 1: padding rows)
    double yw[WT];                        // y of the rows below: y_(i+s1) sits at slot (t - s1) mod WT, t = top - i
    double fr[EB_D][NV], wr[EB_D];
#pragma unroll
    for (int q = 0; q < WT; ++q) yw[q] = 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v) ring_w[v * 64] = Fg[(size_t)top * RW + v * 64];             // t = 0
#pragma unroll
    for (int q = 0; q < EB_D; ++q) {
#pragma unroll
        for (int v = 0; v < NV; ++v) fr[q][v] = Fg[(size_t)max(top - 1 - q, 0) * RW + v * 64];      // t = 1 .. EB_D
        wr[q] = wk[(size_t)max(top - q, 0) * nslots_pad];                                    // t = 0 .. EB_D - 1
    }
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const int i = top - (b * WT + u);
            // row t + 1 -> the other half of the ring; row t + 1 + EB_D and the right-hand side of t + EB_D leave memory
            const double2 *row = ring[u & 1];
#pragma unroll
            for (int v = 0; v < NV; ++v) ring_w[((u + 1) & 1) * RW + v * 64] = fr[u % EB_D][v];
            const double w = wr[u % EB_D];
            {
                const int ip = max(i - 1 - EB_D, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) fr[u % EB_D][v] = Fg[(size_t)ip * RW + v * 64];
                wr[u % EB_D] = wk[(size_t)max(i - EB_D, 0) * nslots_pad];
            }
            // oldest unknowns first: only the last multiply-add of a row waits for the row before it
            double a0 = i < n ? w : 0.0, a1 = 0.0;
#pragma unroll
            for (int s1 = WT; s1 >= 2; s1 -= 2) {
                const double2 c = row[s1 / 2];
                if (s1 + 1 <= WT) a0 -= c.y * yw[(u - s1 - 1 + 2 * WT) % WT];
                a1 -= c.x * yw[(u - s1 + 2 * WT) % WT];
            }
            const double2 c0 = row[0];                  // (1 / diagonal, first super diagonal)
            a0 -= c0.y * yw[(u - 1 + 2 * WT) % WT];
            const double y = (a0 + a1) * c0.x;
            // recombination: entries WT + 1 .. WT + EB_MP of the row (WT even: the first one is the .y of a pair)
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int s1 = (EB_MP < WT ? EB_MP : WT) & ~1; s1 >= 2; s1 -= 2) {
                const double2 c = row[(WT + s1) / 2];
                if (s1 + 1 <= EB_MP && s1 + 1 <= WT) z0 += c.y * yw[(u - s1 - 1 + 2 * WT) % WT];
                z1 += c.x * yw[(u - s1 + 2 * WT) % WT];
            }
            z0 += row[WT / 2].y * yw[(u - 1 + 2 * WT) % WT];
            // column offset: the entry after the recombination band (no scalar load in the loop); integer selects, a
            // conditional store would put a branch into every row
            const long off = __double_as_longlong(row[NC - 1].x);
            const long keep = -(long)(i < n);
            const unsigned long long to = dmp_a + ((dst_a - dmp_a + (unsigned long long)((off & ownm) << 3)) & (unsigned long long)keep);
            *reinterpret_cast<double *>(to) = (z0 + z1) + y;
            yw[u] = y;
        }
    }
}

// Once per factorization storage: the recombination band and the column offset of row i behind its U entries; the
// offset of the row that enters the forward window (j + nw) behind the multipliers of column j.
__global__ void ellband_fill_rows_kernel(const double *__restrict__ P, const long *__restrict__ rowoff,
                                         const long *__restrict__ coloff, double *__restrict__ FL, double *__restrict__ FU,
                                         int nl, int np, int nw, int wt, int nv, int nc) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)nl * np) return;
    const int i = (int)(e % np);
    for (int sd = 0; sd < EB_MP; ++sd) FU[e * (nv * 64) + wt + 1 + sd] = P[e * EB_MP + sd];
    FU[e * (nv * 64) + 2 * (nc - 1)] = __longlong_as_double(coloff[e]);
    FL[e * EB_FLW + nw] = __longlong_as_double(i + nw < np ? rowoff[e + nw] : 0L);
}

struct EbVariant { int nw, wt; };
static const EbVariant eb_variants[] = {{12, 24}, {20, 40}, {28, 56}, {36, 64}, {36, 96}};

static inline int eb_nv(int wt) { return (wt + 1 + EB_MP + 2 + 63) / 64; }
static inline int eb_nc(int wt) { return (wt + 1 + EB_MP + 1) / 2 + 1; }

#ifdef DDH_EB_ABLATE      // measurement builds: DDH_EB_ABL bit 0 / 1 = coalesced (wrong) addressing of the gather / scatter
static int eb_abl() { const char *e = getenv("DDH_EB_ABL"); return e ? atoi(e) : 0; }
#else
static int eb_abl() { return 0; }
#endif
template <int NW>
static void launch_forward(EllBand *p, const EllBandLu &lu, const double *rhs, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL(ellband_forward_kernel<NW>, grid, dim3(64), 0, st, p->n_d, p->nbc_d, p->slot_limit_d, p->rowoff_d,
                       p->T_d, lu.Lm, rhs, p->work_d, p->np, p->nslots, p->nslots_pad, (eb_abl() & 1) ? 1L : p->slot_stride,
                       max(p->nbc, 1));
}
template <int WT>
static void launch_backward(EllBand *p, const EllBandLu &lu, double *x, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL(ellband_backward_kernel<WT>, grid, dim3(64), 0, st, p->n_d, p->slot_limit_d, p->coloff_d, lu.U,
                       p->work_d, x, p->dump_d, p->np, p->nslots, p->nslots_pad, (eb_abl() & 2) ? 1L : p->slot_stride);
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_ellband_create(ddh_handle *h, int nl, int nmax, int kl, int ku, int mp, int nbc, int nslots, long slot_stride,
                       const int *n_h, const int *nbc_h, const int *slot_limit_h, const long *rowoff_h,
                       const long *coloff_h, const double *T_h, const double *P_h, const double *MB_h,
                       const double *LB_h) {
    if (!h || nl < 1 || nmax < 1 || kl < 0 || ku < 0 || nslots < 1) return fail("ellband_create: bad arguments");
    if (nbc > EB_NBC) return fail("ellband_create: more than 8 boundary rows per group");
    if (mp > EB_MP) return fail("ellband_create: recombination band wider than 16");
    const EbVariant *v = nullptr;
    for (const auto &c : eb_variants)
        if (kl + 1 <= c.nw && kl + ku <= c.wt && nbc <= c.nw) { v = &c; break; }
    if (!v) return fail("ellband_create: band wider than the compiled windows (kl <= 35, kl + ku <= 96)");
    EllBand *p = new EllBand();
    p->kind = H_ELLBAND;
    p->nl = nl; p->nmax = nmax; p->kl = kl; p->ku = ku; p->mp = mp; p->nbc = nbc; p->nslots = nslots;
    p->slot_stride = slot_stride; p->nw = v->nw; p->wt = v->wt;
    p->n_h.assign(n_h, n_h + nl);
    const size_t W = kl + ku + 1, nb = max(nbc, 1);
    auto up = [&](void **d, const void *src, size_t bytes) -> int {
        DDH_HIP(hipMalloc(d, bytes));
        DDH_HIP(hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    int st = 0;
    st |= up((void **)&p->n_d, n_h, sizeof(int) * nl);
    st |= up((void **)&p->nbc_d, nbc_h, sizeof(int) * nl);
    st |= up((void **)&p->slot_limit_d, slot_limit_h, sizeof(int) * nl);
    p->np = nmax + EB_PAD;
    p->nslots_pad = (nslots + 63) / 64 * 64;
    const size_t np = p->np;
    std::vector<long> ro((size_t)nl * np, 0), co((size_t)nl * np, 0);
    std::vector<double> Pp((size_t)nl * np * EB_MP, 0.0);       // recombination band padded to the compiled width
    const int mpw = max(mp, 1);
    for (int g = 0; g < nl; ++g)
        for (int i = 0; i < n_h[g]; ++i) {
            ro[g * np + i] = rowoff_h[(size_t)g * nmax + i];
            co[g * np + i] = coloff_h[(size_t)g * nmax + i];
            for (int sd = 0; sd < mp; ++sd) Pp[(g * np + i) * EB_MP + sd] = P_h[((size_t)g * nmax + i) * mpw + sd];
        }
    st |= up((void **)&p->rowoff_d, ro.data(), sizeof(long) * ro.size());
    st |= up((void **)&p->coloff_d, co.data(), sizeof(long) * co.size());
    st |= up((void **)&p->T_d, T_h, sizeof(double) * nl * nb * nb);
    st |= up((void **)&p->MB_d, MB_h, sizeof(double) * nl * nmax * W);
    st |= up((void **)&p->LB_d, LB_h, sizeof(double) * nl * nmax * W);
    st |= up((void **)&p->P_d, Pp.data(), sizeof(double) * Pp.size());
    if (!st) st = check_hip(hipMalloc((void **)&p->work_d, sizeof(double) * (size_t)nl * np * p->nslots_pad), "ellband work");
    if (!st) st = check_hip(hipMemset(p->work_d, 0, sizeof(double) * (size_t)nl * np * p->nslots_pad), "ellband work");
    if (!st) st = check_hip(hipMalloc((void **)&p->dump_d, sizeof(double) * 64), "ellband dump");
    if (!st) st = check_hip(hipMalloc((void **)&p->flag_d, sizeof(int)), "ellband flag");
    if (st) { delete p; return st; }
    *h = register_handle(p);
    return 0;
}

// a M + b L of every group -> band LU number `index` (a new one when index == number of factorizations so far)
int ddh_ellband_factor(ddh_handle h, int index, double a, double b, int *nsingular_h, void *stream) {
    EllBand *p = (EllBand *)lookup_handle(h, H_ELLBAND);
    if (!p) return fail("ellband_factor: bad handle");
    if (index < 0 || index > (int)p->lus.size()) return fail("ellband_factor: bad factorization index");
    if (index == (int)p->lus.size()) {
        EllBandLu lu;
        const size_t rows = (size_t)p->nl * p->np;
        DDH_HIP(hipMalloc((void **)&lu.Lm, sizeof(double) * rows * EB_FLW));
        DDH_HIP(hipMalloc((void **)&lu.U, sizeof(double) * rows * eb_nv(p->wt) * 64));
        DDH_HIP(hipMemsetAsync(lu.Lm, 0, sizeof(double) * rows * EB_FLW, as_stream(stream)));
        DDH_HIP(hipMemsetAsync(lu.U, 0, sizeof(double) * rows * eb_nv(p->wt) * 64, as_stream(stream)));
        hipLaunchKernelGGL(ellband_fill_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream),
                           p->P_d, p->rowoff_d, p->coloff_d, lu.Lm, lu.U, p->nl, p->np, p->nw, p->wt, eb_nv(p->wt),
                           eb_nc(p->wt));
        p->lus.push_back(lu);
    }
    const EllBandLu &lu = p->lus[index];
    hipStream_t st = as_stream(stream);
    DDH_HIP(hipMemsetAsync(p->flag_d, 0, sizeof(int), st));
    const size_t lds = sizeof(double) * ((size_t)(p->kl + 1) * (p->kl + p->ku + 1) + p->kl + 1);
    hipLaunchKernelGGL(ellband_factor_kernel, dim3(p->nl), dim3(EB_FT), lds, st, p->n_d, p->MB_d, p->LB_d, a, b, lu.Lm, lu.U,
                       p->flag_d, p->nmax, p->np, p->kl, p->ku, EB_FLW, p->nw - 1, p->wt, eb_nv(p->wt) * 64);
    DDH_HIP(hipGetLastError());
    if (nsingular_h) {
        DDH_HIP(hipMemcpyAsync(nsingular_h, p->flag_d, sizeof(int), hipMemcpyDeviceToHost, st));
        DDH_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

// x (valid modes of the banded groups only; the caller clears x first) = (a M + b L)^-1 rhs, both [comp][slot][group][n]
int ddh_ellband_solve(ddh_handle h, int index, const double *rhs_d, double *x_d, void *stream) {
    EllBand *p = (EllBand *)lookup_handle(h, H_ELLBAND);
    if (!p) return fail("ellband_solve: bad handle");
    if (index < 0 || index >= (int)p->lus.size()) return fail("ellband_solve: no such factorization");
    if (rhs_d == x_d) return fail("ellband_solve: in-place solve is not supported");
    const EllBandLu &lu = p->lus[index];
    hipStream_t st = as_stream(stream);
    dim3 grid((p->nslots + 63) / 64, p->nl);
    switch (p->nw) {
        case 12: launch_forward<12>(p, lu, rhs_d, grid, st); break;
        case 20: launch_forward<20>(p, lu, rhs_d, grid, st); break;
        case 28: launch_forward<28>(p, lu, rhs_d, grid, st); break;
        default: launch_forward<36>(p, lu, rhs_d, grid, st); break;
    }
    switch (p->wt) {
        case 24: launch_backward<24>(p, lu, x_d, grid, st); break;
        case 40: launch_backward<40>(p, lu, x_d, grid, st); break;
        case 56: launch_backward<56>(p, lu, x_d, grid, st); break;
        case 64: launch_backward<64>(p, lu, x_d, grid, st); break;
        default: launch_backward<96>(p, lu, x_d, grid, st); break;
    }
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_ellband_info(ddh_handle h, int *nw, int *wt, long *factor_bytes) {
    EllBand *p = (EllBand *)lookup_handle(h, H_ELLBAND);
    if (!p) return fail("ellband_info: bad handle");
    if (nw) *nw = p->nw;
    if (wt) *wt = p->wt;
    if (factor_bytes) {
        long rows = 0;
        for (int v : p->n_h) rows += v;
        *factor_bytes = rows * (long)(sizeof(double) * (p->kl + p->kl + p->ku + 1) + sizeof(int));
    }
    return 0;
}

}  // extern "C"
