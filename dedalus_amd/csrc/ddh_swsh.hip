// Grouped dense transforms: the spin-weighted spherical harmonic colatitude transform.
// One launch applies, for every local azimuthal wavenumber m, that m's (Lmax+1-|m|) x Ntheta matrix to its
// slice of the data -- the reference does this in a Python loop of small matmuls
// (SWSHColatitudeTransform.forward_reduced / backward_reduced, core/transforms.py:1258-1288).
//
// The work is bound by streaming the matrices (S: sum_m (255-m) x 384 doubles = 101 MB per spin against
// 1.5 MB of data), so:
//   * few right-hand-side columns (sphere fields: 2 = cos/msin parts): one wave per output row, lanes
//     stride along the contraction index (coalesced matrix rows), wave reduction -- a batched GEMV;
//   * many columns (shell fields carry the radial axis behind theta): LDS-tiled FP64 GEMM.
#include "ddh_common.h"

namespace ddh {

struct GroupDev {
    int mat_rows;            // rows (n_ell) of the matrices, -1: no matrix (|m| > Lmax)
    long off_f, off_b;       // offsets of the forward / backward matrix in the packed arrays
    int g_start, c_start, count, ell_start, ell_step, n_ell;
    // second set of right-hand sides served by the SAME matrices (GEMV path): pair_mode 0 none, 1 plain,
    // 2 mirrored: the spin -s transform from the spin +s matrices, F_{-s}[l, j] = (-1)^(l + m) F_{+s}[l, N-1-j]
    int pair_g, pair_c, pair_mode, parity;
};

struct GmmtPlan : HandleBase {
    std::vector<GroupDev> hgroups;
    int n_grid = 0, ngroups = 0, max_ell = 0, max_count = 0, paired = 0;
    long max_g_end = 0, max_c_end = 0, max_l_end = 0;
    GroupDev *d_groups = nullptr;
    double *d_fwd = nullptr, *d_bwd = nullptr;
    ~GmmtPlan() override {
        (void)hipFree(d_groups);
        (void)hipFree(d_fwd);
        (void)hipFree(d_bwd);
    }
};

struct GmmtDims {
    long n0, n1g, n1c, n2c, n3;
    int n_grid;
};

// address helpers (C order)
__device__ __forceinline__ long g_index(const GmmtDims &d, long o, long i1, long t, long x) {
    return ((o * d.n1g + i1) * d.n_grid + t) * d.n3 + x;
}
__device__ __forceinline__ long c_index(const GmmtDims &d, long o, long i1, long l, long x) {
    return ((o * d.n1c + i1) * d.n2c + l) * d.n3 + x;
}

constexpr int GV_COLS = 8;     // GEMV path: at most this many right-hand-side columns

// ---- batched GEMV: block = 4 waves, each wave owns GV_ROWS output rows of one group ----------------------
// The right-hand-side columns (cos / msin parts, leading and trailing axes) are few, so the kernel is bound
// by streaming the matrices: every lane keeps GV_ROWS matrix loads in flight per step and the (cached)
// right-hand-side values are loaded once for all the rows of the wave; all index arithmetic is hoisted.
constexpr int GV_ROWS = 4;

template <bool FWD, int NCOL, bool PAIRS>
__global__ void __launch_bounds__(256)
grouped_gemv_kernel(const GroupDev *__restrict__ groups, const double *__restrict__ mats, const double *__restrict__ in,
                    double *__restrict__ out, GmmtDims d, int ncols) {
    const GroupDev gr = groups[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave) * GV_ROWS;
    const int nrows = FWD ? gr.n_ell : d.n_grid;       // output rows
    // per column: input base / stride along the contraction index, output base / stride along the row index
    long ibase[NCOL], obase[NCOL];
    bool cok[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const long x = c % d.n3, jc = (c / d.n3) % gr.count, o = c / (d.n3 * gr.count);
        cok[c] = c < ncols && o < d.n0;
        ibase[c] = FWD ? g_index(d, o, gr.g_start + jc, 0, x) : c_index(d, o, gr.c_start + jc, gr.ell_start, x);
        obase[c] = FWD ? c_index(d, o, gr.c_start + jc, gr.ell_start, x) : g_index(d, o, gr.g_start + jc, 0, x);
    }
    const long istride = FWD ? d.n3 : (long)gr.ell_step * d.n3;
    const long ostride = FWD ? (long)gr.ell_step * d.n3 : d.n3;
    // paired right-hand sides (same matrices, read once): bases of the partner slices
    const bool paired = PAIRS && gr.pair_mode != 0, mirror = PAIRS && gr.pair_mode == 2;
    long ibase2[NCOL], obase2[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const long x = c % d.n3, jc = (c / d.n3) % gr.count, o = c / (d.n3 * gr.count);
        ibase2[c] = FWD ? g_index(d, o, gr.pair_g + jc, 0, x) : c_index(d, o, gr.pair_c + jc, gr.ell_start, x);
        obase2[c] = FWD ? c_index(d, o, gr.pair_c + jc, gr.ell_start, x) : g_index(d, o, gr.pair_g + jc, 0, x);
    }
    if (gr.mat_rows < 0) {                              // |m| > Lmax: forward skips, backward writes zeros
        if (!FWD && lane == 0) {
            for (int r = 0; r < GV_ROWS; ++r) {
                if (row0 + r >= d.n_grid) break;
#pragma unroll
                for (int c = 0; c < NCOL; ++c) {
                    if (cok[c]) out[obase[c] + (long)(row0 + r) * ostride] = 0.0;
                    if (cok[c] && paired) out[obase2[c] + (long)(row0 + r) * ostride] = 0.0;
                }
            }
        }
        return;
    }
    if (row0 >= nrows) return;
    const int K = FWD ? d.n_grid : gr.n_ell;            // contraction length
    const double *A = mats + (FWD ? gr.off_f : gr.off_b) + (long)row0 * K;
    double acc[GV_ROWS][NCOL], acc2[GV_ROWS][NCOL];
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) acc[r][c] = acc2[r][c] = 0.0;
    bool rok[GV_ROWS];
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) rok[r] = row0 + r < nrows;
    for (int k = lane; k < K; k += 64) {
        double a[GV_ROWS], xv[NCOL], xw[NCOL];
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) a[r] = rok[r] ? A[(long)r * K + k] : 0.0;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) xv[c] = cok[c] ? in[ibase[c] + (long)k * istride] : 0.0;
        if (PAIRS && paired) {
            // forward: the partner's grid values in reversed colatitude order; backward: its coefficients with the
            // sign (-1)^(l + m) of the contraction index l = ell_start + k
            const long kk = (FWD && mirror) ? (long)(K - 1 - k) : (long)k;
            const double sg = (!FWD && mirror && ((gr.ell_start + k + gr.parity) & 1)) ? -1.0 : 1.0;
#pragma unroll
            for (int c = 0; c < NCOL; ++c) xw[c] = cok[c] ? sg * in[ibase2[c] + kk * istride] : 0.0;
        } else {
#pragma unroll
            for (int c = 0; c < NCOL; ++c) xw[c] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                acc[r][c] += a[r] * xv[c];
                if (PAIRS) acc2[r][c] += a[r] * xw[c];
            }
    }
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) {
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            double v = acc[r][c], w = acc2[r][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                v += __shfl_down(v, off, 64);
                if (PAIRS) w += __shfl_down(w, off, 64);
            }
            if (lane == 0 && rok[r] && cok[c]) {
                out[obase[c] + (long)(row0 + r) * ostride] = v;
                if (paired) {
                    // forward: sign of the output row l = ell_start + row; backward: reversed output row
                    const int row = row0 + r;
                    const double so = (FWD && mirror && ((gr.ell_start + row + gr.parity) & 1)) ? -1.0 : 1.0;
                    const long orow = (!FWD && mirror) ? (long)(nrows - 1 - row) : (long)row;
                    out[obase2[c] + orow * ostride] = so * w;
                }
            }
        }
    }
}

// ---- LDS-tiled GEMM: tile = 32 output rows x 64 columns, contraction in steps of 16 --------------------
constexpr int GT_I = 32, GT_X = 64, GT_J = 16;

template <bool FWD>
__global__ void __launch_bounds__(256)
grouped_gemm_kernel(const GroupDev *__restrict__ groups, const double *__restrict__ mats, const double *__restrict__ in,
                    double *__restrict__ out, GmmtDims d, long ncols) {
    __shared__ double sA[GT_I][GT_J + 1];
    __shared__ double sB[GT_J][GT_X];
    const GroupDev gr = groups[blockIdx.z];
    const int nrows = FWD ? gr.n_ell : d.n_grid;
    const int i0 = blockIdx.y * GT_I;
    if (i0 >= nrows) return;
    const long x0 = (long)blockIdx.x * GT_X;
    const int tx = threadIdx.x % GT_X, ty = threadIdx.x / GT_X;
    // this thread's output column and, for the B tile loads, the column it fetches
    const long col = x0 + tx;
    const bool col_ok = col < ncols && col / (d.n3 * gr.count) < d.n0;
    const long cx = col_ok ? col % d.n3 : 0, cj = col_ok ? (col / d.n3) % gr.count : 0,
               co = col_ok ? col / (d.n3 * gr.count) : 0;
    if (gr.mat_rows < 0) {
        if (!FWD && col_ok) {
            for (int r = ty; r < GT_I; r += 4)
                if (i0 + r < d.n_grid) out[g_index(d, co, gr.g_start + cj, i0 + r, cx)] = 0.0;
        }
        return;
    }
    const int K = FWD ? d.n_grid : gr.n_ell;
    const double *A = mats + (FWD ? gr.off_f : gr.off_b);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j0 = 0; j0 < K; j0 += GT_J) {
        for (int w = threadIdx.x; w < GT_I * GT_J; w += 256) {
            const int jj = w % GT_J, ii = w / GT_J;
            sA[ii][jj] = (i0 + ii < nrows && j0 + jj < K) ? A[(long)(i0 + ii) * K + j0 + jj] : 0.0;
        }
        for (int jj = ty; jj < GT_J; jj += 4) {
            double v = 0.0;
            const int k = j0 + jj;
            if (col_ok && k < K)
                v = in[FWD ? g_index(d, co, gr.g_start + cj, k, cx)
                           : c_index(d, co, gr.c_start + cj, gr.ell_start + (long)k * gr.ell_step, cx)];
            sB[jj][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < GT_J; ++jj) {
            const double bv = sB[jj][tx];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += sA[ty * 8 + r][jj] * bv;
        }
        __syncthreads();
    }
    if (col_ok) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = i0 + ty * 8 + r;
            if (i < nrows) {
                const long idx = FWD ? c_index(d, co, gr.c_start + cj, gr.ell_start + (long)i * gr.ell_step, cx)
                                     : g_index(d, co, gr.g_start + cj, i, cx);
                out[idx] = acc[r];
            }
        }
    }
}


// ---- FP64 MFMA GEMM for many right-hand-side columns (shell fields: the radial axis rides behind theta) ------------
// Workgroup = 4 waves, tile = 64 output rows (16 per wave) x 64 columns of ONE (leading index, m part) slice, so the
// columns are contiguous radial points; contraction in chunks of 32 staged through LDS for the data operand, the
// matrix operand comes straight from L2 (each element feeds 4 MFMAs; a group's matrix is shared by all its slices'
// workgroups).  v_mfma_f64_16x16x4: A one f64 per lane (row = lane & 15, k = lane >> 4), B (k = lane >> 4,
// col = lane & 15), C/D (col = lane & 15, row = (lane >> 4) + 4 r).
constexpr int GM_M = 64, GM_N = 64, GM_K = 32, GM_LD = GM_N + 1;
typedef double gm_d4 __attribute__((ext_vector_type(4)));

template <bool FWD>
__global__ void __launch_bounds__(256)
grouped_gemm_mfma_kernel(const GroupDev *__restrict__ groups, const double *__restrict__ mats,
                         const double *__restrict__ in, double *__restrict__ out, GmmtDims d, int xtiles) {
    __shared__ double sB[GM_K * GM_LD];
    const GroupDev gr = groups[blockIdx.z];
    const int nrows = FWD ? gr.n_ell : d.n_grid;
    const int i0 = blockIdx.y * GM_M;
    const int slice = blockIdx.x / xtiles, xt = blockIdx.x - slice * xtiles;
    const long co = slice / gr.count, cj = slice - co * gr.count;      // (leading index, part) of this slice
    if (co >= d.n0) return;
    const int x0 = xt * GM_N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (gr.mat_rows < 0) {                               // |m| > Lmax: forward skips, backward writes zeros
        if (!FWD) {
            for (int w = tid; w < GM_M * GM_N; w += 256) {
                const int r = w / GM_N, x = w - r * GM_N;
                if (i0 + r < d.n_grid && x0 + x < d.n3) out[g_index(d, co, gr.g_start + cj, i0 + r, x0 + x)] = 0.0;
            }
        }
        return;
    }
    if (i0 >= nrows) return;
    const int K = FWD ? d.n_grid : gr.n_ell;
    const double *A = mats + (FWD ? gr.off_f : gr.off_b);
    const int arow = i0 + 16 * wave + (lane & 15);
    const bool arow_ok = arow < nrows;
    const double *Arow = A + (long)(arow_ok ? arow : 0) * K;
    gm_d4 acc[GM_N / 16];
#pragma unroll
    for (int jt = 0; jt < GM_N / 16; ++jt) acc[jt] = (gm_d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += GM_K) {
        __syncthreads();
        // stage the data operand: rows k0 .. k0 + 31 of this slice, 64 contiguous columns each
        for (int w = tid; w < GM_K * GM_N; w += 256) {
            const int kk = w / GM_N, x = w - kk * GM_N, k = k0 + kk;
            double v = 0.0;
            if (k < K && x0 + x < d.n3)
                v = in[FWD ? g_index(d, co, gr.g_start + cj, k, x0 + x)
                           : c_index(d, co, gr.c_start + cj, gr.ell_start + (long)k * gr.ell_step, x0 + x)];
            sB[kk * GM_LD + x] = v;
        }
        __syncthreads();
        if (i0 + 16 * wave >= nrows) continue;           // (wave-uniform) no output rows in this wave's 16-row strip
#pragma unroll
        for (int k4 = 0; k4 < GM_K; k4 += 4) {
            const int k = k0 + k4 + (lane >> 4);
            const double a = (arow_ok && k < K) ? Arow[k] : 0.0;
            const double *br = sB + (k4 + (lane >> 4)) * GM_LD + (lane & 15);
#pragma unroll
            for (int jt = 0; jt < GM_N / 16; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br[jt * 16], acc[jt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jt = 0; jt < GM_N / 16; ++jt) {
        const int x = x0 + jt * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * wave + (lane >> 4) + 4 * r;
            if (i < nrows && x < d.n3) {
                const long idx = FWD ? c_index(d, co, gr.c_start + cj, gr.ell_start + (long)i * gr.ell_step, x)
                                     : g_index(d, co, gr.g_start + cj, i, x);
                out[idx] = acc[jt][r];
            }
        }
    }
}

template <bool FWD>
static int launch_grouped(GmmtPlan *pl, const double *in, double *out, long n0, long n1g, long n1c, long n2c, long n3,
                          void *stream) {
    if (n0 <= 0 || n3 <= 0 || pl->ngroups == 0) return 0;
    {
        const size_t ng = (size_t)n0 * n1g * pl->n_grid * n3, nc = (size_t)n0 * n1c * n2c * n3;
        if (int st0 = resolve_alias(&in, out, FWD ? ng : nc, FWD ? nc : ng, as_stream(stream))) return st0;
    }
    if (pl->max_g_end > n1g || pl->max_c_end > n1c || pl->max_l_end > n2c)
        return fail("grouped_mmt: a group's slices exceed the array extents");
    GmmtDims d{n0, n1g, n1c, n2c, n3, pl->n_grid};
    const long ncols = n0 * (long)pl->max_count * n3;      // columns of the widest group
    hipStream_t st = as_stream(stream);
    const int max_rows = FWD ? pl->max_ell : pl->n_grid;
    const double *mats = FWD ? pl->d_fwd : pl->d_bwd;
    if (pl->paired && ncols > GV_COLS)
        return fail("grouped_mmt: paired groups are implemented for the GEMV path (few columns) only");
    static const bool no_mfma = getenv("DDH_SWSH_NO_MFMA") != nullptr;
    if (ncols <= GV_COLS) {
        dim3 grid((unsigned)((max_rows + 4 * GV_ROWS - 1) / (4 * GV_ROWS)), (unsigned)pl->ngroups);
#define DDH_GEMV(NC)                                                                                               \
    {                                                                                                              \
        if (pl->paired)                                                                                            \
            hipLaunchKernelGGL((grouped_gemv_kernel<FWD, NC, true>), grid, dim3(256), 0, st, pl->d_groups, mats, in, out, d, (int)ncols); \
        else                                                                                                       \
            hipLaunchKernelGGL((grouped_gemv_kernel<FWD, NC, false>), grid, dim3(256), 0, st, pl->d_groups, mats, in, out, d, (int)ncols); \
    }
        if (ncols <= 2) DDH_GEMV(2) else if (ncols <= 4) DDH_GEMV(4) else DDH_GEMV(8)
#undef DDH_GEMV
    } else if (n3 >= 16 && !no_mfma) {
        const int xtiles = (int)((n3 + GM_N - 1) / GM_N);
        dim3 grid((unsigned)(xtiles * n0 * pl->max_count), (unsigned)((max_rows + GM_M - 1) / GM_M), (unsigned)pl->ngroups);
        if (grid.z > 65535 || grid.y > 65535) return fail("grouped_mmt: too many groups / rows for one launch");
        hipLaunchKernelGGL(grouped_gemm_mfma_kernel<FWD>, grid, dim3(256), 0, st, pl->d_groups, mats, in, out, d, xtiles);
    } else {
        dim3 grid((unsigned)((ncols + GT_X - 1) / GT_X), (unsigned)((max_rows + GT_I - 1) / GT_I), (unsigned)pl->ngroups);
        if (grid.z > 65535 || grid.y > 65535) return fail("grouped_mmt: too many groups / rows for one launch");
        hipLaunchKernelGGL(grouped_gemm_kernel<FWD>, grid, dim3(256), 0, st, pl->d_groups, mats, in, out, d, ncols);
    }
    DDH_HIP(hipGetLastError());
    return 0;
}

// ---- regularity recombination (row a13) ----------------------------------------------------------------
// data[ncomp][n1][n2][n3] in place: one workgroup per (i1, i2) = (m slot, ell slot); the 3^rank x 3^rank
// matrix Q(ell) sits in LDS, threads run along the trailing (radial) axis.  The radial factor (dR/r)^k
// of ShellBasis.forward/backward_transform_radius multiplies every point and commutes with Q: fused.
template <int NC>
__global__ void __launch_bounds__(256)
regularity_kernel(double *__restrict__ data, const int *__restrict__ ell_map, const double *__restrict__ q,
                  const double *__restrict__ fac, long n12, long n3, int nell, int forward) {
    __shared__ double sQ[NC * NC];
    const long i12 = blockIdx.x;
    const int ell = ell_map ? ell_map[i12] : -1;
    const bool mix = (NC > 1) && ell >= 0 && ell < nell;
    if (mix) {
        for (int w = threadIdx.x; w < NC * NC; w += blockDim.x) {
            const int r = w / NC, c = w % NC;
            sQ[w] = q[((long)ell * NC + r) * NC + c];
            (void)c;
            (void)forward;
        }
    }
    __syncthreads();
    const long cstride = n12 * n3;
    for (long x = threadIdx.x; x < n3; x += blockDim.x) {
        double v[NC], o[NC];
        double *p = data + i12 * n3 + x;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = p[c * cstride];
        if (mix) {
#pragma unroll
            for (int r = 0; r < NC; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < NC; ++c) acc += sQ[r * NC + c] * v[c];
                o[r] = acc;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) o[c] = v[c];
        }
        const double f = fac ? fac[x] : 1.0;
        if (mix || fac) {
#pragma unroll
            for (int c = 0; c < NC; ++c) p[c * cstride] = f * o[c];
        }
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_plan_grouped_mmt(ddh_handle *plan, int n_grid, int ngroups, const ddh_mmt_group *groups_h, int nmats,
                         const int *mat_rows_h, const double *const *fwd_h, const double *const *bwd_h) {
    if (n_grid < 1 || ngroups < 0 || nmats < 0) return fail("plan_grouped_mmt: bad sizes");
    std::vector<long> off(nmats + 1, 0);
    for (int i = 0; i < nmats; ++i) {
        if (mat_rows_h[i] < 0) return fail("plan_grouped_mmt: negative matrix size");
        off[i + 1] = off[i] + (long)mat_rows_h[i] * n_grid;
    }
    GmmtPlan *pl = new GmmtPlan();
    pl->kind = H_GMMT;
    pl->n_grid = n_grid;
    pl->ngroups = ngroups;
    std::vector<GroupDev> gd(ngroups > 0 ? ngroups : 1);
    for (int g = 0; g < ngroups; ++g) {
        const ddh_mmt_group &h = groups_h[g];
        GroupDev &o = gd[g];
        if (h.mat >= nmats || h.count < 1 || (h.ell_step != 1 && h.ell_step != -1) || h.n_ell < 0) {
            delete pl;
            return fail("plan_grouped_mmt: malformed group");
        }
        if (h.mat >= 0 && mat_rows_h[h.mat] != h.n_ell) {
            delete pl;
            return fail("plan_grouped_mmt: group row count differs from its matrix");
        }
        o.mat_rows = h.mat >= 0 ? h.n_ell : -1;
        o.off_f = o.off_b = h.mat >= 0 ? off[h.mat] : 0;
        o.g_start = h.g_start; o.c_start = h.c_start; o.count = h.count;
        o.ell_start = h.ell_start; o.ell_step = h.ell_step; o.n_ell = h.mat >= 0 ? h.n_ell : 0;
        o.pair_g = o.pair_c = 0; o.pair_mode = 0; o.parity = 0;
        const long l_lo = h.ell_step > 0 ? h.ell_start : h.ell_start - (long)(o.n_ell > 0 ? o.n_ell - 1 : 0);
        const long l_hi = h.ell_step > 0 ? h.ell_start + (long)(o.n_ell > 0 ? o.n_ell - 1 : 0) : h.ell_start;
        if (h.g_start < 0 || h.c_start < 0 || (o.n_ell > 0 && l_lo < 0)) {
            delete pl;
            return fail("plan_grouped_mmt: negative slice start");
        }
        if (o.n_ell > pl->max_ell) pl->max_ell = o.n_ell;
        if (h.count > pl->max_count) pl->max_count = h.count;
        if (h.g_start + h.count > pl->max_g_end) pl->max_g_end = h.g_start + h.count;
        if (h.mat >= 0 && h.c_start + h.count > pl->max_c_end) pl->max_c_end = h.c_start + h.count;
        if (o.n_ell > 0 && l_hi + 1 > pl->max_l_end) pl->max_l_end = l_hi + 1;
    }
    const size_t nb = (size_t)(off[nmats] > 0 ? off[nmats] : 1) * sizeof(double);
    std::vector<double> pf((size_t)off[nmats] + 1, 0.0), pb((size_t)off[nmats] + 1, 0.0);
    for (int i = 0; i < nmats; ++i) {
        const size_t n = (size_t)mat_rows_h[i] * n_grid;
        if (n) {
            memcpy(pf.data() + off[i], fwd_h[i], n * sizeof(double));
            memcpy(pb.data() + off[i], bwd_h[i], n * sizeof(double));
        }
    }
    if (check_hip(hipMalloc((void **)&pl->d_groups, gd.size() * sizeof(GroupDev)), "hipMalloc") ||
        check_hip(hipMemcpy(pl->d_groups, gd.data(), gd.size() * sizeof(GroupDev), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMalloc((void **)&pl->d_fwd, nb), "hipMalloc") ||
        check_hip(hipMalloc((void **)&pl->d_bwd, nb), "hipMalloc") ||
        check_hip(hipMemcpy(pl->d_fwd, pf.data(), nb, hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(pl->d_bwd, pb.data(), nb, hipMemcpyHostToDevice), "hipMemcpy")) {
        delete pl;
        return -2;
    }
    pl->hgroups = gd;
    *plan = register_handle(pl);
    return 0;
}

// Second right-hand-side set per group, served by the group's own matrices (GEMV path only, ell_step = +1):
// pair_mode[g] = 0 none, 1 plain (another component with the same spin weight), 2 mirrored (the component of
// opposite spin weight: F_{-s}[l, j] = (-1)^(l + m) F_{+s}[l, N-1-j], parity[g] = m & 1).  pair_g / pair_c are the
// first grid / coefficient slice of the partner (same count as the group).
int ddh_grouped_mmt_set_pairs(ddh_handle plan, int ngroups, const int *pair_g_h, const int *pair_c_h,
                              const int *pair_mode_h, const int *parity_h) {
    GmmtPlan *pl = (GmmtPlan *)lookup_handle(plan, H_GMMT);
    if (!pl) return -1;
    if (ngroups != pl->ngroups) return fail("grouped_mmt_set_pairs: group count differs from the plan");
    for (int g = 0; g < ngroups; ++g) {
        GroupDev &o = pl->hgroups[g];
        if (pair_mode_h[g] < 0 || pair_mode_h[g] > 2) return fail("grouped_mmt_set_pairs: bad pair mode");
        if (pair_mode_h[g] && (o.ell_step != 1 || pair_g_h[g] < 0 || pair_c_h[g] < 0))
            return fail("grouped_mmt_set_pairs: pairs need ell_step = 1 and valid slices");
        o.pair_g = pair_g_h[g]; o.pair_c = pair_c_h[g]; o.pair_mode = pair_mode_h[g]; o.parity = parity_h[g] & 1;
        if (o.pair_mode) {
            if (o.pair_g + o.count > pl->max_g_end) pl->max_g_end = o.pair_g + o.count;
            if (o.mat_rows >= 0 && o.pair_c + o.count > pl->max_c_end) pl->max_c_end = o.pair_c + o.count;
        }
    }
    DDH_HIP(hipMemcpy(pl->d_groups, pl->hgroups.data(), pl->hgroups.size() * sizeof(GroupDev), hipMemcpyHostToDevice));
    pl->paired = 1;
    return 0;
}

int ddh_regularity_recombine(double *data, int ncomp, long n1, long n2, long n3, const int *slot_map_d, int nmats,
                             const double *mats_d, const double *radial_factor_d, void *stream) {
    const int *ell_map_d = slot_map_d;
    const double *q_d = mats_d;
    const int nell = nmats, forward = 0;
    if (n1 <= 0 || n2 <= 0 || n3 <= 0) return 0;
    if (ncomp != 1 && ncomp != 3 && ncomp != 9) return fail("ddh_regularity_recombine: tensor rank 0, 1 or 2 (1, 3, 9 components)");
    if (ncomp > 1 && (!ell_map_d || !q_d)) return fail("ddh_regularity_recombine: ell map and Q table required");
    const long n12 = n1 * n2;
    if (n12 > 0x7fffffffL) return fail("ddh_regularity_recombine: too many (m, ell) slots");
    const int T = n3 >= 256 ? 256 : (n3 > 64 ? 128 : 64);
    hipStream_t st = as_stream(stream);
    if (ncomp == 1)
        hipLaunchKernelGGL(regularity_kernel<1>, dim3((unsigned)n12), dim3(T), 0, st, data, ell_map_d, q_d,
                           radial_factor_d, n12, n3, nell, forward);
    else if (ncomp == 3)
        hipLaunchKernelGGL(regularity_kernel<3>, dim3((unsigned)n12), dim3(T), 0, st, data, ell_map_d, q_d,
                           radial_factor_d, n12, n3, nell, forward);
    else
        hipLaunchKernelGGL(regularity_kernel<9>, dim3((unsigned)n12), dim3(T), 0, st, data, ell_map_d, q_d,
                           radial_factor_d, n12, n3, nell, forward);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_grouped_mmt_forward(ddh_handle plan, const double *g, double *c, long n0, long n1g, long n1c, long n2c,
                            long n3, void *stream) {
    GmmtPlan *pl = (GmmtPlan *)lookup_handle(plan, H_GMMT);
    if (!pl) return -1;
    return launch_grouped<true>(pl, g, c, n0, n1g, n1c, n2c, n3, stream);
}

int ddh_grouped_mmt_backward(ddh_handle plan, const double *c, double *g, long n0, long n1g, long n1c, long n2c,
                             long n3, void *stream) {
    GmmtPlan *pl = (GmmtPlan *)lookup_handle(plan, H_GMMT);
    if (!pl) return -1;
    return launch_grouped<false>(pl, c, g, n0, n1g, n1c, n2c, n3, stream);
}

}  // extern "C"
