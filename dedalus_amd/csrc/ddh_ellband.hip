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
//     lane owns a slot, the factor row of a step is fetched once per wavefront, parked in LDS and read back as broadcast
//     loads, and the sliding windows (kl + 1 right-hand-side rows forward, kl + ku solved unknowns backward) stay in
//     registers.  The forward sweep gathers the permuted rows straight from the solver's [component][slot][group][n]
//     right-hand side and applies the boundary-row combination T; the backward sweep (16 slots per wave, the dot product
//     of a row split over the four lane quads) undoes the column recombination (X = P Y, a short upper band in the
//     permuted order) and scatters into the state layout.
// The sphere uses the same kernels to FORM its per-m dense inverses: unit right-hand sides against the band LU of the
// real-form transposed systems, then ddh_ellband_gather_complex_inverse (core/sphere.py::_inverse_batch).
// Algorithmic cost per solve: 2 (2 kl + ku + mp) n flops and 16 n bytes per (group, slot) -- against 2 n^2 for the dense
// inverse -- and the factorization is O(n kl (kl + ku)) per group instead of O(n^3).
#include "ddh_common.h"
#include <algorithm>

namespace ddh {

constexpr int EB_NBC = 8;        // boundary rows per group
constexpr int EB_MP = 16;        // super diagonals of the recombination in the permuted order
// backward factor rows (layout: see ellband_backward_kernel)
constexpr int EB_RING = 128;     // doubles per LDS ring slot of the backward sweep (two 64-lane stores per row)
__host__ __device__ constexpr int eb_qw(int wt) { return wt / 4 + EB_MP / 4; }
// doubles per backward factor row in memory: the four quads' entries, 1 / diagonal, column offset -- rounded to 128 B
__host__ __device__ constexpr int eb_rw(int wt) { return (4 * eb_qw(wt) + 2 + 15) / 16 * 16; }
__host__ __device__ constexpr int eb_u_index(int wt, int s1) { return s1 == 0 ? 4 * eb_qw(wt) : ((s1 - 1) % 4) * eb_qw(wt) + (s1 - 1) / 4; }

struct EllBandLu {
    double *Lm = nullptr;        // [nl][np][eb_flw(nw)]  column j: pivot row offset (as a double), the multipliers of rows
                                 //                       j+1 .., zero padded to nw - 1, the offset of the row entering the window
    double *U = nullptr;         // [nl][np][eb_rw(wt)]   row i in the quad layout of ellband_backward_kernel: super diagonals,
                                 //                       recombination, 1 / diagonal, column offset
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
        for (int c1 = tid; c1 < WC; c1 += EB_FT) {          // (entries past the band stay zero from the allocation)
            int c = jc + c1;
            if (c >= WC) c -= WC;
            const double u = win[jr * WC + c];
            Ug[(size_t)j * urow + eb_u_index(wt, c1)] = c1 == 0 ? ipv : u;
            const int d = c1 == 0 ? WC - 1 : c1 - 1;
            double v = 0.0;
            if (inew < n) v = a * Mg[(size_t)inew * WC + d] + b * Lg[(size_t)inew * WC + d];
            win[jr * WC + c] = v;
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
constexpr int eb_depth(int w) { return w % 8 == 0 ? 8 : 4; }
constexpr int EB_FLW = 64;       // doubles per LDS ring slot of the forward sweep (one 64-lane store per row)
// doubles per forward factor row in memory: [0] pivot offset, [1 .. nw-1] multipliers, [nw] offset of the row that enters
// the window, rounded to 128 B
__host__ __device__ constexpr int eb_flw(int nw) { return (nw + 1 + 15) / 16 * 16; }

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
    __shared__ double2 ring[2][EB_FLW / 2];
    const int g = blockIdx.y;
    const int n = n_d[g];
    const int lim = slot_limit[g];
    if (n == 0 || (int)blockIdx.x * 64 >= lim) return;
    const int lane = threadIdx.x;
    const int s = blockIdx.x * 64 + lane;
    const bool live = s < lim;                                  // slots past the limit carry zeros
    const double *src = rhs + (size_t)(s < nslots ? s : 0) * slot_stride;
    const long *ro = rowoff + (size_t)g * np;
    constexpr int FW = eb_flw(NW);
    const double *Fg = FL + (size_t)g * np * FW + min(lane, FW - 1);
    double *wk = work + (size_t)g * np * nslots_pad + s;
    double *ring_w = reinterpret_cast<double *>(&ring[0][0]) + lane;
    double bw[NW], fr[EB_D];
    double2 cf[2][NW / 2 + 1];                                  // coefficients of the current / the next row (+ row offset)
    ring_w[0] = Fg[0];                                          // rows 0, 1 -> the ring
    ring_w[EB_FLW] = Fg[FW];
#pragma unroll
    for (int q = 0; q < EB_D; ++q) fr[q] = Fg[(size_t)(q + 2) * FW];                 // rows 2 .. EB_D + 1
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const double v = src[ro[q]];
        bw[q] = (q < n && live) ? v : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NW / 2 + 1; ++q) cf[0][q] = ring[0][q];
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
            // coefficients of row j + 1 (in the ring since row j - 1) -> the other register set: a whole row of work ahead of
            // their use, the lone wave of the workgroup has nothing else to hide the LDS latency behind
#pragma unroll
            for (int q = 0; q < NW / 2 + 1; ++q) cf[(u + 1) & 1][q] = ring[(u + 1) & 1][q];
            // Row interchange, wave-uniform offset d = 4 a + b: y = bw[u + d], and the row it displaces takes that place.
            // Pivot rows are the rule in these systems (~80 % of the columns, offsets up to ~12): one scalar branch per
            // group of four window rows (a), selects inside (b) -- a chain of NW branches or of NW selects costs several
            // times the multiply-adds of the row.
            if (d != 0) {
                asm volatile("");
                const int ga = d >> 2, gb = d & 3;
#pragma unroll
                for (int a = 0; a < NW / 4; ++a)
                    if (ga == a) {
                        asm volatile("");
                        const double old = bw[u];
                        double ysel = bw[(u + 4 * a) % NW];
#pragma unroll
                        for (int bb = 1; bb < 4; ++bb) ysel = (gb == bb) ? bw[(u + 4 * a + bb) % NW] : ysel;
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb)
                            if (4 * a + bb >= 1) bw[(u + 4 * a + bb) % NW] = (gb == bb) ? old : bw[(u + 4 * a + bb) % NW];
                        bw[u] = ysel;
                    }
            }
            const double y = bw[u];
            // row j + 2: memory -> the ring slot row j has left; row j + 2 + EB_D leaves memory
            ring_w[(u & 1) * EB_FLW] = fr[u % EB_D];
            fr[u % EB_D] = Fg[(size_t)(j + 2 + EB_D) * FW];
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

// Sum over the four 16-lane quads of a wave, result in every lane: gfx950's v_permlane16_swap / v_permlane32_swap
// exchange 16- / 32-lane halves between two registers in the VALU (a ds_bpermute round trip per step would sit on the
// row-to-row dependency chain of the sweep).
__device__ __forceinline__ double eb_quad_sum(double a) {
    unsigned lo = __double2loint(a), hi = __double2hiint(a);
    const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __hiloint2double(h16[0], l16[0]) + __hiloint2double(h16[1], l16[1]);
    lo = __double2loint(x);
    hi = __double2hiint(x);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(h32[0], l32[0]) + __hiloint2double(h32[1], l32[1]);
}

// WT: solved unknowns kept (kl + ku <= WT, a multiple of 4).  Backward sweep: one wavefront per (group, 16 slots); the
// four 16-lane quads of the wave split the dot product of a row -- quad q takes the super diagonals s1 = 4k + q + 1 --
// and an xor all-reduce over the quads (two steps) completes it.  A lone wave per SIMD is bound by its instruction
// count: this cuts the multiply-adds per lane to a quarter and makes four times as many waves (the machine has ~3x more
// SIMDs than a 64-slot split has waves).  Every lane keeps the whole window, rotated by its quad: the solved y of step
// t sits at slot (t + q + 1) mod WT, so the unknown quad q needs for its k-th coefficient, step t - 4k - q - 1, is at
// slot (t - 4k) mod WT -- a compile-time register for every quad.
// A factor row (eb_rw(WT) doubles, 80 for WT = 56): quad q's [q QW, (q + 1) QW): WT / 4 entries of U then EB_MP / 4 of the recombination;
// [4 QW] = 1 / diagonal, [4 QW + 1] = column offset of the row.  The sweep starts past the end (zero rows, results to
// the dump word) so that it ends on row 0 with whole blocks.

template <int WT>
__global__ void __launch_bounds__(64)
ellband_backward_kernel(const int *__restrict__ n_d, const int *__restrict__ slot_limit, const double *__restrict__ FU,
                        const double *__restrict__ work, double *__restrict__ x, double *__restrict__ dump, int np,
                        int nslots, int nslots_pad, long slot_stride, int abl) {
    constexpr int EB_D = eb_depth(WT);
    constexpr int KU = WT / 4, KP = EB_MP / 4, QW = KU + KP, RW = eb_rw(WT), RG = EB_RING;
    static_assert(WT % 4 == 0 && WT % EB_D == 0 && QW % 2 == 0 && KU % 2 == 0 && RW <= RG, "row layout");
    __shared__ double2 ring[2][RG / 2];
    const int g = blockIdx.y;
    const int n = n_d[g];
    const int lim = slot_limit[g];
    if (n == 0 || (int)blockIdx.x * 16 >= lim) return;
    const int lane = threadIdx.x, q = lane >> 4;
    const int s = blockIdx.x * 16 + (lane & 15);
    const bool own = q == 0 && s < nslots;                     // the quad that stores; the others write to the dump word
    const unsigned long long dmp_a = reinterpret_cast<unsigned long long>(dump + lane);
    const unsigned long long dst_a = own ? reinterpret_cast<unsigned long long>(x + (size_t)s * slot_stride) : dmp_a;
    const long ownm = own ? -1L : 0L;
    const double *Fg = FU + (size_t)g * np * RW;
    const int fl[2] = {min(lane, RW - 1), min(64 + lane, RW - 1)};      // the row's entries this lane fetches
    const double *wk = work + (size_t)g * np * nslots_pad + (s < nslots_pad ? s : 0);
    double *ring_w = reinterpret_cast<double *>(&ring[0][0]) + lane;
#ifdef DDH_EB_ABLATE      // measurement builds only: abl bit 2 = results to the dump word, 3 = one right-hand-side row, 4 = one factor row
    const long keepm = (abl & 4) ? 0L : -1L;
    const size_t wrow = (abl & 8) ? 0 : (size_t)nslots_pad;
    const size_t frow = (abl & 16) ? 0 : (size_t)RW;
#else
    constexpr long keepm = -1L;
    const size_t wrow = (size_t)nslots_pad;
    constexpr size_t frow = (size_t)RW;
#endif
    const int nblk = (n + WT - 1) / WT;
    const int top = nblk * WT - 1;                              // first row of the sweep (>= n - 1: padding rows)
    double yw[WT];
    double fr[EB_D][2], wr[EB_D];
#pragma unroll
    for (int k = 0; k < WT; ++k) yw[k] = 0.0;
#pragma unroll
    for (int v = 0; v < 2; ++v) ring_w[v * 64] = Fg[(size_t)top * frow + fl[v]];              // t = 0
#pragma unroll
    for (int k = 0; k < EB_D; ++k) {
#pragma unroll
        for (int v = 0; v < 2; ++v) fr[k][v] = Fg[(size_t)max(top - 1 - k, 0) * frow + fl[v]];       // t = 1 .. EB_D
        wr[k] = wk[(size_t)max(top - k, 0) * wrow];                                    // t = 0 .. EB_D - 1
    }
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            const int i = top - (b * WT + u);
            const double2 *rq = ring[u & 1] + q * (QW / 2);          // this quad's coefficients
            const double2 sh = ring[u & 1][2 * QW];                  // (1 / diagonal, column offset)
            // row t + 1 -> the other half of the ring; row t + 1 + EB_D and the right-hand side of t + EB_D leave memory
#pragma unroll
            for (int v = 0; v < 2; ++v) ring_w[((u + 1) & 1) * RG + v * 64] = fr[u % EB_D][v];
            const double w = wr[u % EB_D];
            {
                const int ip = max(i - 1 - EB_D, 0);
#pragma unroll
                for (int v = 0; v < 2; ++v) fr[u % EB_D][v] = Fg[(size_t)ip * frow + fl[v]];
                wr[u % EB_D] = wk[(size_t)max(i - EB_D, 0) * wrow];
            }
            // oldest unknowns first: only the last multiply-add of a row waits for the row before it
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = KU - 2; k >= 0; k -= 2) {
                const double2 c = rq[k / 2];
                a1 -= c.y * yw[(u - 4 * (k + 1) + 2 * WT) % WT];
                a0 -= c.x * yw[(u - 4 * k + 2 * WT) % WT];
            }
            double z0 = 0.0;
#pragma unroll
            for (int k = KP - 2; k >= 0; k -= 2) {
                const double2 c = rq[(KU + k) / 2];
                z0 += c.y * yw[(u - 4 * (k + 1) + 2 * WT) % WT];
                z0 += c.x * yw[(u - 4 * k + 2 * WT) % WT];
            }
            const double a = eb_quad_sum(a0 + a1);
            z0 = eb_quad_sum(z0);
            const double y = ((i < n ? w : 0.0) + a) * sh.x;
            // integer selects: a conditional store would put a branch into every row
            const long off = __double_as_longlong(sh.y);
            const long keep = -(long)(i < n) & keepm;
            const unsigned long long to = dmp_a + ((dst_a - dmp_a + (unsigned long long)((off & ownm) << 3)) & (unsigned long long)keep);
            *reinterpret_cast<__attribute__((address_space(1))) double *>(to) = z0 + y;      // (global, not flat: keeps lgkmcnt for the LDS)
            // y of this step -> slot (u + 1 + q) mod WT
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) yw[(u + 1 + qq) % WT] = (q == qq) ? y : yw[(u + 1 + qq) % WT];
        }
    }
}

// Once per factorization storage: the recombination band and the column offset of row i behind its U entries; the
// offset of the row that enters the forward window (j + nw) behind the multipliers of column j.
__global__ void ellband_fill_rows_kernel(const double *__restrict__ P, const long *__restrict__ rowoff,
                                         const long *__restrict__ coloff, double *__restrict__ FL, double *__restrict__ FU,
                                         int nl, int np, int nw, int wt) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)nl * np) return;
    const int i = (int)(e % np);
    const int qw = eb_qw(wt);
    for (int sd = 0; sd < EB_MP; ++sd) FU[e * eb_rw(wt) + (sd % 4) * qw + wt / 4 + sd / 4] = P[e * EB_MP + sd];
    FU[e * eb_rw(wt) + 4 * qw + 1] = __longlong_as_double(coloff[e]);
    FL[e * eb_flw(nw) + nw] = __longlong_as_double(i + nw < np ? rowoff[e + nw] : 0L);
}

// Complex inverses from unit solves of the real-form, transposed systems (the sphere's per-m systems, which have ONE
// right-hand side per step and therefore keep their dense-inverse GEMV -- but no longer an O(n^3) inversion): system
// vectors x [2 R][slot][m][ell], components 2c / 2c + 1 = real / imaginary part of component c, slot
// s(c, ell) = (nl - 1 - ell) R + c holding  (a M + b L)_m^-T e_(c, ell)  = row (c, ell) of the inverse.  out: per m the
// row-major (R (nl - m))^2 complex matrix, unknown j = c (nl - m) + (ell - m), at complex offset off[m]
// (the layout of ddh_dense_inverse_compute / ddh_cgemv_batch_mats).
__global__ void __launch_bounds__(256)
ellband_gather_cinv_kernel(const double *__restrict__ x, double2 *__restrict__ out, const long *__restrict__ off, int R,
                           int nl, int nm, int nslots) {
    const int m = blockIdx.y;
    const int ne = nl - m;
    if (ne <= 0) return;
    const long n = (long)R * ne;
    const size_t plane = (size_t)nslots * nm * nl;
    double2 *o = out + off[m];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n * n; e += (long)gridDim.x * 256) {
        const int j = (int)(e / n), jp = (int)(e - (long)j * n);
        const int c = j / ne, el = j - c * ne, cp = jp / ne, elp = jp - cp * ne;
        const int sl = (nl - 1 - (m + el)) * R + c;
        const size_t base = ((size_t)(2 * cp) * nslots + sl) * nm * nl + (size_t)m * nl + (m + elp);
        o[e] = make_double2(x[base], x[base + plane]);
    }
}

struct EbVariant { int nw, wt; };
static const EbVariant eb_variants[] = {{12, 24}, {20, 40}, {28, 56}, {36, 64}, {36, 96}};


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
    hipLaunchKernelGGL(ellband_backward_kernel<WT>, dim3((p->nslots + 15) / 16, grid.y), dim3(64), 0, st, p->n_d,
                       p->slot_limit_d, lu.U, p->work_d, x, p->dump_d, p->np, p->nslots, p->nslots_pad,
                       (eb_abl() & 2) ? 1L : p->slot_stride, eb_abl());
}


// ---- inverse of a bordered system whose band block has one zero column -------------------------------------------------
// The k = 0 pencil of a Cartesian problem with a pressure gauge (core/subsystems.py: the subproblem that holds tau_p and
// "integ(p) = 0"):  A = [[B, c], [r^T, e]] in the solver's permuted order, B the n x n band block, singular ONLY because
// the column of the constant pressure mode (j0) vanishes, c the gauge variable's column, r the gauge row.  With the two
// columns exchanged, A2 = [[B2, 0], [w^T, d]] (B2 = B with column j0 replaced by c: a band again, w = r with e at j0,
// d = r[j0]) is block triangular, so A^-1 follows from X = B2^-1 (n unit solves of the band LU) and one weighted
// column sum:   rows j != j0 of A^-1 = (X[j][:], 0);  row n (the gauge variable) = (X[j0][:], 0);
//               row j0 = (-(w^T X) / d, 1 / d).        out: (n + 1) x (n + 1), row-major.
__global__ void __launch_bounds__(256)
bordered_inverse_kernel(const double *__restrict__ X, int n, int j0, const double *__restrict__ wM,
                        const double *__restrict__ wL, double a, double b, double d, double *__restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;              // column of the inverse = equation (right-hand side) index
    const int N = n + 1;
    if (s >= N) return;
    if (blockIdx.y == 0) {                                      // row j0: the weighted column sum
        double v = 0.0;
        if (s < n)
            for (int i = 0; i < n; ++i) v += (a * wM[i] + b * wL[i]) * X[(size_t)i * n + s];
        out[(size_t)j0 * N + s] = s < n ? -v / d : 1.0 / d;
        return;
    }
    for (int j = blockIdx.y - 1; j < N; j += gridDim.y - 1) {   // the other rows: copies
        if (j == j0) continue;
        const int src = j < n ? j : j0;
        out[(size_t)j * N + s] = s < n ? X[(size_t)src * n + s] : 0.0;
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_ellband_bordered_inverse(const double *X_d, int n, int j0, const double *wM_d, const double *wL_d, double dM,
                                 double dL, double a, double b, double *out_d, void *stream) {
    if (!X_d || !wM_d || !wL_d || !out_d || n < 1 || j0 < 0 || j0 >= n) return fail("ellband_bordered_inverse: bad arguments");
    const double d = a * dM + b * dL;
    if (d == 0.0) return fail("ellband_bordered_inverse: the gauge row does not see the free mode");
    hipLaunchKernelGGL(bordered_inverse_kernel, dim3((unsigned)((n + 1 + 255) / 256), 65), dim3(256), 0, as_stream(stream), X_d, n,
                       j0, wM_d, wL_d, a, b, d, out_d);
    DDH_HIP(hipGetLastError());
    return 0;
}

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
        DDH_HIP(hipMalloc((void **)&lu.Lm, sizeof(double) * rows * eb_flw(p->nw)));
        DDH_HIP(hipMalloc((void **)&lu.U, sizeof(double) * rows * eb_rw(p->wt)));
        DDH_HIP(hipMemsetAsync(lu.Lm, 0, sizeof(double) * rows * eb_flw(p->nw), as_stream(stream)));
        DDH_HIP(hipMemsetAsync(lu.U, 0, sizeof(double) * rows * eb_rw(p->wt), as_stream(stream)));
        hipLaunchKernelGGL(ellband_fill_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream),
                           p->P_d, p->rowoff_d, p->coloff_d, lu.Lm, lu.U, p->nl, p->np, p->nw, p->wt);
        p->lus.push_back(lu);
    }
    const EllBandLu &lu = p->lus[index];
    hipStream_t st = as_stream(stream);
    DDH_HIP(hipMemsetAsync(p->flag_d, 0, sizeof(int), st));
    const size_t lds = sizeof(double) * ((size_t)(p->kl + 1) * (p->kl + p->ku + 1) + p->kl + 1);
    hipLaunchKernelGGL(ellband_factor_kernel, dim3(p->nl), dim3(EB_FT), lds, st, p->n_d, p->MB_d, p->LB_d, a, b, lu.Lm, lu.U,
                       p->flag_d, p->nmax, p->np, p->kl, p->ku, eb_flw(p->nw), p->nw - 1, p->wt, eb_rw(p->wt));
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

int ddh_ellband_gather_complex_inverse(const double *x_d, double *out_d, const long *off_d, int ncomp, int nl, int nm,
                                       int nslots, void *stream) {
    if (!x_d || !out_d || !off_d || ncomp < 1 || nl < 1 || nm < 1 || nslots < ncomp * nl)
        return fail("ellband_gather_complex_inverse: bad arguments");
    const long nmax = (long)ncomp * nl;
    const unsigned bx = (unsigned)std::min<long>((nmax * nmax + 255) / 256, 4096);
    hipLaunchKernelGGL(ellband_gather_cinv_kernel, dim3(bx, nm), dim3(256), 0, as_stream(stream), x_d, (double2 *)out_d, off_d,
                       ncomp, nl, nm, nslots);
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
