// Coefficient-space kernels of the sphere path (BASELINE config 4: viscous
// shallow water on S2), gfx950.
//
// Layout: a tensor field's coefficients are real arrays [component][2 m + part][ell], part 0/1 = cos/msin of
// the azimuthal mode m (one complex number z = cos + i msin per (component, m, ell)), ell contiguous;
// the colatitude-grid stage is [component][2 m + part][theta].
//
//  * ddh_spin_recombine: coordinate <-> spin components (SpinRecombinationBasis.forward/
//    backward_spin_recombination, core/basis.py:1595-1663, libraries/spin_recombination.pyx:9-56): a fixed
//    small real matrix on the (component, part) pairs of every (m, theta) point.
//  * ddh_sphere_terms_*: every linear sphere operator of the reference is local in ell -- SeparableSphere
//    operators multiply (m, ell) groups by a symbol (core/operators.py:2725-2866: grad, div, lap, average,
//    convert), MulCosine couples ell to ell +- 1 (core/operators.py:2995-3046), SpinSkew multiplies by +-i
//    (:2125-2147) -- so operators and their compositions are "term lists"
//        y[co][m][ell] += coef_t[m][ell] * x[ci][m][ell + d_t]          (complex, per term t)
//    applied by one streaming kernel over all (m, ell); this replaces the reference's per-m Python loop of
//    scipy CSR products (Subproblem matrices, core/subsystems.py:497-596, timesteppers.py:588-591).
//  * ddh_cgemv_batch_*: the per-m implicit solve.  The per-m systems (all variables x ell >= m) are small
//    (<= 3 x 255 unknowns at Lmax = 254); their LHS inverses are formed once per timestep size on the host
//    and applied as one batched complex GEMV (one wavefront per row), replacing the per-m SuperLU solves
//    (libraries/matsolvers.py:126-149, timesteppers.py:630-643).
#include "ddh_common.h"

#include <cstdlib>

namespace ddh {


// ------------------------------------------------------------------------------------------------
struct SpinMat {
    double r[324];    // [2 nc][2 nc] row major, nc <= 9 (rank 2 in spherical coordinates)
};

template <int NC>
__global__ void __launch_bounds__(256)
spin_recombine_kernel(const double *__restrict__ in, double *__restrict__ out, SpinMat M, long npairs, long inner) {
    const long total = npairs * inner;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long q = i / inner, x = i - q * inner;
    const long cstride = 2 * npairs * inner;
    const long base = (2 * q) * inner + x;
    double v[2 * NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        v[2 * c] = in[c * cstride + base];
        v[2 * c + 1] = in[c * cstride + base + inner];
    }
#pragma unroll
    for (int r = 0; r < 2 * NC; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 2 * NC; ++c) acc += M.r[r * 2 * NC + c] * v[c];
        out[(r >> 1) * cstride + base + (r & 1) * inner] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
struct SphereTerms : HandleBase {
    int nm = 0, nl = 0, ncomp_out = 0, nterms = 0;
    int *d_meta = nullptr;       // [nterms][3]: co, ci, d (sorted by co)
    int *d_first = nullptr;      // [ncomp_out + 1] term ranges per output component
    double2 *d_coef = nullptr;   // [nterms][nm][nl]
    ~SphereTerms() override {
        (void)hipFree(d_meta);
        (void)hipFree(d_first);
        (void)hipFree(d_coef);
    }
};

__global__ void __launch_bounds__(256)
sphere_terms_kernel(const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ meta,
                    const int *__restrict__ first, const double2 *__restrict__ coef, int nm, int nl, int ncomp_out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (l >= nl) return;
    const long row = (long)nl;
    const long cstride = 2L * nm * row;
    for (int co = 0; co < ncomp_out; ++co) {
        double2 acc = make_double2(0.0, 0.0);
        for (int t = first[co]; t < first[co + 1]; ++t) {
            const int ci = meta[3 * t + 1], d = meta[3 * t + 2];
            const int ls = l + d;
            if (ls < 0 || ls >= nl) continue;
            const double2 c = coef[((long)t * nm + m) * row + l];
            const double xr = x[ci * cstride + (2L * m) * row + ls];
            const double xi = x[ci * cstride + (2L * m + 1) * row + ls];
            acc.x += c.x * xr - c.y * xi;
            acc.y += c.x * xi + c.y * xr;
        }
        y[co * cstride + (2L * m) * row + l] = acc.x;
        y[co * cstride + (2L * m + 1) * row + l] = acc.y;
    }
}

// ------------------------------------------------------------------------------------------------
struct CgemvBatch : HandleBase {
    int nm = 0, nl = 0, ncomp = 0, max_n = 0;
    long *d_off = nullptr;       // [nm + 1] matrix offsets (complex elements)
    double2 *d_mats = nullptr;
    size_t bytes = 0;
    ~CgemvBatch() override {
        (void)hipFree(d_off);
        (void)hipFree(d_mats);
    }
};

// one wavefront per row of one m; unknown j = comp * (nl - m) + (ell - m)
__global__ void __launch_bounds__(256)
cgemv_batch_kernel(const double *__restrict__ x, double *__restrict__ y, const long *__restrict__ off,
                   const double2 *__restrict__ mats, int nm, int nl, int ncomp) {
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nell = nl - m;
    const int n = ncomp * (nell > 0 ? nell : 0);
    const long row = (long)nl;
    const long cstride = 2L * nm * row;
    if (nell <= 0) {                                  // no mode at this m: zero its slots
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < ncomp * nl; idx += gridDim.x * 256) {
            const int c = idx / nl, l = idx - c * nl;
            y[c * cstride + (2L * m) * row + l] = 0.0;
            y[c * cstride + (2L * m + 1) * row + l] = 0.0;
        }
        return;
    }
    if (j >= ncomp * nl) return;
    if (j >= n) {
        // slots below the diagonal (ell < m) of this m carry no mode: zero them once per (comp, ell < m)
        const int jj = j - n;                      // 0 .. ncomp * m - 1
        if (m > 0 && jj < ncomp * m && lane == 0) {
            const int c = jj / m, l = jj - c * m;
            if (l < nl) {
                y[c * cstride + (2L * m) * row + l] = 0.0;
                y[c * cstride + (2L * m + 1) * row + l] = 0.0;
            }
        }
        return;
    }
    const double2 *A = mats + off[m] + (long)j * n;
    double2 acc = make_double2(0.0, 0.0);
    for (int k = lane; k < n; k += 64) {
        const int c = k / nell, l = m + (k - c * nell);
        const double2 a = A[k];
        const double xr = x[c * cstride + (2L * m) * row + l];
        const double xi = x[c * cstride + (2L * m + 1) * row + l];
        acc.x += a.x * xr - a.y * xi;
        acc.y += a.x * xi + a.y * xr;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        acc.x += __shfl_xor(acc.x, s, 64);
        acc.y += __shfl_xor(acc.y, s, 64);
    }
    if (lane == 0) {
        const int c = j / nell, l = m + (j - c * nell);
        y[c * cstride + (2L * m) * row + l] = acc.x;
        y[c * cstride + (2L * m + 1) * row + l] = acc.y;
    }
}

// ------------------------------------------------------------------------------------------------
// ell-dependent radial operators of shell fields (csrc of core/shell.py): fields are
// [component][2 m + part][ell][n]; a term maps component ci to co through a real matrix A_t[ell] (n_out x n_in),
// the same for every m and part:  y[co][i1][ell][:] = sum_t A_t[ell] x[ci_t][i1][ell][:].
// This is SphericalEllOperator.operate / subproblem_matrix (core/operators.py:3108-3222: per (m, ell)
// apply_matrix of a radial matrix in a Python loop over ell_maps) for all slots in one launch, and -- with
// the per-ell LHS inverses as the matrices -- the per-ell solve of the shell's subproblems.

struct EllTerms : HandleBase {
    int nm = 0, nl = 0, nr = 0, ncomp_out = 0, nterms = 0, nmat = 0;
    int *d_meta = nullptr;       // [nterms][2]: co, ci (sorted by co)
    int *d_first = nullptr;      // [ncomp_out + 1]
    int *d_slot = nullptr;       // [2 nm][nl]: matrix index of the slot, -1: no mode
    short *d_band = nullptr;     // [nterms][nmat][nr][2]: first / one-past-last non-zero column of every row
    short *d_rows = nullptr;     // [nterms][nmat][2]: first / one-past-last non-zero row
    double *d_mats = nullptr;    // [nterms][nmat][n_in][n_out]  (transposed: threads run along n_out)
    int *d_tmap = nullptr;       // dense path: [ncomp_out][ncomp_in] -> term index or -1
    int dense = 0;               // 1: mostly full blocks, default slot map -> per-ell FP64 MFMA GEMM
    int ncomp_in = 0;
    ~EllTerms() override {
        (void)hipFree(d_meta);
        (void)hipFree(d_first);
        (void)hipFree(d_slot);
        (void)hipFree(d_band);
        (void)hipFree(d_rows);
        (void)hipFree(d_mats);
        (void)hipFree(d_tmap);
    }
};

// Dense blocks (the per-ell LHS inverses): C[(co, no)][slot] = sum_(ci, ni) A_ell[(co, no)][(ci, ni)] X_ell[(ci, ni)][slot] is
// a per-ell GEMM with up to 2 (ell + 1) right-hand-side columns.  Workgroup tile: 64 rows (4 waves x 16) x 128 slots,
// FP64 MFMA 16x16x4 (A: one f64 per lane, row = lane & 15, k = lane >> 4; B: k = lane >> 4, col = lane & 15;
// C/D: col = lane & 15, row = (lane >> 4) + 4 r); X is staged through LDS in chunks of 64 k, every matrix element
// loaded from L2/HBM feeds 8 MFMAs.  Results leave through LDS so the stores run along the contiguous radial index.
constexpr int EG_M = 64, EG_N = 128, EG_K = 64, EG_LD = EG_N + 1;
typedef double d4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
ell_gemm_kernel(const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tmap,
                const double *__restrict__ mats, int nm, int nl, int nr, int nmat, int ncomp_in, int ncomp_out,
                int accumulate) {
    extern __shared__ double sm[];                    // [EG_K][EG_LD] (X chunk), reused as [EG_N][EG_M] for the output
    const int l = blockIdx.y;
    const int i1_0 = blockIdx.z * EG_N;
    const int m0 = blockIdx.x * EG_M;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long cstride = 2L * nm * nl * nr;
    const int mrow = m0 + 16 * wave;                  // first output row of this wave
    const int co = mrow / nr, no0 = mrow - co * nr;
    const bool any_live = (i1_0 >> 1) <= l;           // slot i1 is live iff i1 / 2 <= ell
    if (!any_live) {
        if (accumulate) return;
        for (int w = tid; w < EG_N * EG_M; w += 256) {
            const int j = w / EG_M, r = w - j * EG_M, i1 = i1_0 + j;
            const int mm = m0 + r, c2 = mm / nr, n2 = mm - c2 * nr;
            if (i1 < 2 * nm) y[c2 * cstride + ((long)i1 * nl + l) * nr + n2] = 0.0;
        }
        return;
    }
    if (accumulate) {                                 // nothing to add for an output component without dense blocks
        bool any = false;                             // (the 64-row tile lies inside one component: nr % 64 == 0)
        for (int ci = 0; ci < ncomp_in; ++ci) any = any || tmap[co * ncomp_in + ci] >= 0;
        if (!any) return;
    }
    d4v acc[EG_N / 16];
#pragma unroll
    for (int jt = 0; jt < EG_N / 16; ++jt) acc[jt] = (d4v){0.0, 0.0, 0.0, 0.0};
    for (int ci = 0; ci < ncomp_in; ++ci) {
        const int t = tmap[co * ncomp_in + ci];       // wave-uniform (a 16-row tile lies inside one component)
        for (int k0 = 0; k0 < nr; k0 += EG_K) {
            __syncthreads();
            // stage X[(ci, k0 + k)][slot j]: one slot per wave instruction, 64 consecutive radial modes
            for (int j = wave; j < EG_N; j += 4) {
                const int i1 = i1_0 + j;
                double v = 0.0;
                if (i1 < 2 * nm && (i1 >> 1) <= l) v = x[ci * cstride + ((long)i1 * nl + l) * nr + k0 + lane];
                sm[lane * EG_LD + j] = v;
            }
            __syncthreads();
            if (t < 0) continue;
            const double *A = mats + (((long)t * nmat + (nmat == 1 ? 0 : l)) * nr + k0) * nr + no0;
#pragma unroll 4
            for (int k4 = 0; k4 < EG_K; k4 += 4) {
                const double a = A[(long)(k4 + (lane >> 4)) * nr + (lane & 15)];
                const double *xr = sm + (k4 + (lane >> 4)) * EG_LD + (lane & 15);
#pragma unroll
                for (int jt = 0; jt < EG_N / 16; ++jt)
                    acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xr[jt * 16], acc[jt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // C/D -> LDS [slot j][row r of the 64-row tile]
#pragma unroll
    for (int jt = 0; jt < EG_N / 16; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            sm[(jt * 16 + (lane & 15)) * EG_M + 16 * wave + (lane >> 4) + 4 * r] = acc[jt][r];
    __syncthreads();
    for (int w = tid; w < EG_N * EG_M; w += 256) {
        const int j = w / EG_M, r = w - j * EG_M, i1 = i1_0 + j;
        const int mm = m0 + r, c2 = mm / nr, n2 = mm - c2 * nr;
        if (i1 < 2 * nm) {
            const long yi = c2 * cstride + ((long)i1 * nl + l) * nr + n2;
            y[yi] = accumulate ? y[yi] + sm[w] : sm[w];
        }
    }
}

// One workgroup = ELL_S consecutive (m, part) slots of one ell, threads run along the output radial index.  The
// input rows of all components are staged once in LDS ([component][n][slot], so a thread reads its ELL_S
// right-hand sides with one wide LDS load per n); every matrix element fetched from L2/HBM is used ELL_S times.
// Per term and matrix the band offsets (kl, ku) of the non-zeros bound the inner loop: differential operators are
// banded in n, the LHS inverses are dense.
constexpr int ELL_S = 8;     // slots per workgroup
constexpr int ELL_CO = 4;    // output components processed concurrently (threadIdx.y)

__global__ void __launch_bounds__(1024)
ell_terms_kernel(const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ meta,
                 const int *__restrict__ first, const int *__restrict__ slot_map, const short *__restrict__ band,
                 const short *__restrict__ rows, const double *__restrict__ mats, int nm, int nl, int nr, int nmat,
                 int ncomp_in, int ncomp_out) {
    extern __shared__ double sx[];                   // [ncomp_in][nr][ELL_S]
    const int l = blockIdx.y, i0 = blockIdx.x * ELL_S;   // groups of one ell are neighbours: they stream the same matrices
    const int tlin = threadIdx.y * blockDim.x + threadIdx.x, tall = blockDim.x * blockDim.y;
    const long cstride = 2L * nm * nl * nr;
    int mid[ELL_S];
    bool same = true;
    int mid0 = -1;
#pragma unroll
    for (int s = 0; s < ELL_S; ++s) {
        const int i1 = i0 + s;
        mid[s] = (i1 < 2 * nm) ? slot_map[i1 * nl + l] : -1;
        if (mid[s] >= 0) {
            if (mid0 < 0) mid0 = mid[s];
            else if (mid[s] != mid0) same = false;
        }
    }
    if (mid0 < 0) {                                   // no slot of the group carries a mode: zero them
        for (int co = threadIdx.y; co < ncomp_out; co += ELL_CO)
            for (int no = threadIdx.x; no < nr; no += blockDim.x)
#pragma unroll
                for (int s = 0; s < ELL_S; ++s)
                    if (i0 + s < 2 * nm) y[co * cstride + ((long)(i0 + s) * nl + l) * nr + no] = 0.0;
        return;
    }
    for (int w = tlin; w < ncomp_in * nr; w += tall) {
        const int ci = w / nr, ni = w - ci * nr;
#pragma unroll
        for (int s = 0; s < ELL_S; ++s) {
            const int i1 = i0 + s;
            sx[(long)w * ELL_S + s] = (mid[s] >= 0) ? x[ci * cstride + ((long)i1 * nl + l) * nr + ni] : 0.0;
        }
    }
    __syncthreads();
    for (int co = threadIdx.y; co < ncomp_out; co += ELL_CO) {
        for (int no = threadIdx.x; no < nr; no += blockDim.x) {
            double acc[ELL_S];
#pragma unroll
            for (int s = 0; s < ELL_S; ++s) acc[s] = 0.0;
            for (int t = first[co]; t < first[co + 1]; ++t) {
                const int ci = meta[2 * t + 1];
                const double *xs = sx + (long)ci * nr * ELL_S;
                if (same) {
                    const long ri = 2 * ((long)t * nmat + mid0);
                    if (no < rows[ri] || no >= rows[ri + 1]) continue;      // this row of the block is empty
                    const long bi = 2 * (((long)t * nmat + mid0) * nr + no);
                    const int n0 = band[bi], n1 = band[bi + 1];          // non-zero columns of this row
                    const double *A = mats + (((long)t * nmat + mid0) * nr) * nr + no;
                    for (int ni = n0; ni < n1; ++ni) {
                        const double a = A[(long)ni * nr];
                        const double4 xv = *reinterpret_cast<const double4 *>(xs + (long)ni * ELL_S);
                        const double4 xw = *reinterpret_cast<const double4 *>(xs + (long)ni * ELL_S + 4);
                        acc[0] += a * xv.x;
                        acc[1] += a * xv.y;
                        acc[2] += a * xv.z;
                        acc[3] += a * xv.w;
                        acc[4] += a * xw.x;
                        acc[5] += a * xw.y;
                        acc[6] += a * xw.z;
                        acc[7] += a * xw.w;
                    }
                } else {                              // slots of the group use different matrices (rare)
#pragma unroll
                    for (int s = 0; s < ELL_S; ++s) {
                        if (mid[s] < 0) continue;
                        const double *A = mats + (((long)t * nmat + mid[s]) * nr) * nr + no;
                        for (int ni = 0; ni < nr; ++ni) acc[s] += A[(long)ni * nr] * xs[(long)ni * ELL_S + s];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < ELL_S; ++s)
                if (i0 + s < 2 * nm) y[co * cstride + ((long)(i0 + s) * nl + l) * nr + no] = (mid[s] >= 0) ? acc[s] : 0.0;
        }
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

int ddh_spin_recombine(const double *in, double *out, int ncomp, long npairs, long inner, const double *mat_h,
                       void *stream) {
    if (npairs <= 0 || inner <= 0) return 0;
    if (int st0 = resolve_alias(&in, out, (size_t)(2 * ncomp) * npairs * inner, (size_t)(2 * ncomp) * npairs * inner,
                                as_stream(stream)))
        return st0;
    if (ncomp != 1 && ncomp != 2 && ncomp != 4 && ncomp != 3 && ncomp != 9)
        return fail("ddh_spin_recombine: 1, 2, 4 (rank 0-2 on S2) or 3, 9 (rank 1-2 in spherical coordinates) components");
    SpinMat M;
    memset(&M, 0, sizeof(M));
    for (int i = 0; i < 4 * ncomp * ncomp; ++i) M.r[i] = mat_h[i];
    const long total = npairs * inner;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (ncomp == 1) hipLaunchKernelGGL(spin_recombine_kernel<1>, dim3(blocks), dim3(256), 0, st, in, out, M, npairs, inner);
    else if (ncomp == 2) hipLaunchKernelGGL(spin_recombine_kernel<2>, dim3(blocks), dim3(256), 0, st, in, out, M, npairs, inner);
    else if (ncomp == 4) hipLaunchKernelGGL(spin_recombine_kernel<4>, dim3(blocks), dim3(256), 0, st, in, out, M, npairs, inner);
    else if (ncomp == 3) hipLaunchKernelGGL(spin_recombine_kernel<3>, dim3(blocks), dim3(256), 0, st, in, out, M, npairs, inner);
    else hipLaunchKernelGGL(spin_recombine_kernel<9>, dim3(blocks), dim3(256), 0, st, in, out, M, npairs, inner);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_sphere_terms_create(ddh_handle *h, int nm, int nl, int ncomp_out, int nterms, const int *co_h, const int *ci_h,
                            const int *d_h, const double *coef_h) {
    if (nm < 1 || nl < 1 || ncomp_out < 1 || nterms < 0) return fail("sphere_terms_create: bad sizes");
    SphereTerms *p = new SphereTerms();
    p->kind = H_STERMS;
    p->nm = nm; p->nl = nl; p->ncomp_out = ncomp_out; p->nterms = nterms;
    std::vector<int> meta(3 * (size_t)(nterms > 0 ? nterms : 1)), first(ncomp_out + 1, 0);
    for (int t = 0; t < nterms; ++t) {
        if (co_h[t] < 0 || co_h[t] >= ncomp_out || (t > 0 && co_h[t] < co_h[t - 1])) {
            delete p;
            return fail("sphere_terms_create: terms must be sorted by output component");
        }
        meta[3 * t] = co_h[t]; meta[3 * t + 1] = ci_h[t]; meta[3 * t + 2] = d_h[t];
        first[co_h[t] + 1] = t + 1;
    }
    for (int c = 0; c < ncomp_out; ++c)
        if (first[c + 1] < first[c]) first[c + 1] = first[c];
    const size_t cb = (size_t)(nterms > 0 ? nterms : 1) * nm * nl * sizeof(double2);
    if (check_hip(hipMalloc((void **)&p->d_meta, meta.size() * sizeof(int)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_first, first.size() * sizeof(int)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_coef, cb), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        (nterms > 0 && check_hip(hipMemcpy(p->d_coef, coef_h, cb, hipMemcpyHostToDevice), "hipMemcpy"))) {
        delete p;
        return -2;
    }
    *h = register_handle(p);
    return 0;
}

int ddh_sphere_terms_apply(ddh_handle h, const double *x, double *y, void *stream) {
    SphereTerms *p = (SphereTerms *)lookup_handle(h, H_STERMS);
    if (!p) return -1;
    if (x == y) return fail("sphere_terms_apply: in-place unsupported");
    const dim3 grid((unsigned)((p->nl + 255) / 256), (unsigned)p->nm), block(256);
    hipLaunchKernelGGL(sphere_terms_kernel, grid, block, 0, as_stream(stream), x, y, p->d_meta, p->d_first, p->d_coef,
                       p->nm, p->nl, p->ncomp_out);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_ell_terms_create(ddh_handle *h, int nm, int nl, int nr, int ncomp_out, int nterms, const int *co_h,
                         const int *ci_h, int nmat, const double *mats_h, const int *slot_map_h) {
    if (nm < 1 || nl < 1 || nr < 1 || ncomp_out < 1 || nterms < 0 || nmat < 1) return fail("ell_terms_create: bad sizes");
    for (long i = 0; i < 2L * nm * nl; ++i)
        if (slot_map_h[i] >= nmat) return fail("ell_terms_create: slot map points past the matrices");
    EllTerms *p = new EllTerms();
    p->kind = H_ELLT;
    p->nm = nm; p->nl = nl; p->nr = nr; p->ncomp_out = ncomp_out; p->nterms = nterms; p->nmat = nmat;
    std::vector<int> meta(2 * (size_t)(nterms > 0 ? nterms : 1)), first(ncomp_out + 1, 0);
    for (int t = 0; t < nterms; ++t) {
        if (co_h[t] < 0 || co_h[t] >= ncomp_out || (t > 0 && co_h[t] < co_h[t - 1])) {
            delete p;
            return fail("ell_terms_create: terms must be sorted by output component");
        }
        meta[2 * t] = co_h[t]; meta[2 * t + 1] = ci_h[t];
        first[co_h[t] + 1] = t + 1;
        if (ci_h[t] + 1 > p->ncomp_in) p->ncomp_in = ci_h[t] + 1;
    }
    if (p->ncomp_in < 1) p->ncomp_in = 1;
    for (int c = 0; c < ncomp_out; ++c)
        if (first[c + 1] < first[c]) first[c + 1] = first[c];
    const size_t per = (size_t)nmat * nr * nr;
    const size_t mb = (size_t)(nterms > 0 ? nterms : 1) * per * sizeof(double);
    // transpose every matrix to [n_in][n_out]
    std::vector<double> tr((size_t)(nterms > 0 ? nterms : 1) * per, 0.0);
    std::vector<short> band(2 * (size_t)(nterms > 0 ? nterms : 1) * nmat * nr, 0);
    std::vector<short> rows(2 * (size_t)(nterms > 0 ? nterms : 1) * nmat, 0);
    for (size_t t = 0; t < (size_t)nterms; ++t)
        for (size_t l = 0; l < (size_t)nmat; ++l) {
            int r0 = nr, r1 = 0;
            for (int i = 0; i < nr; ++i) {
                int lo = nr, hi = 0;
                for (int j = 0; j < nr; ++j) {
                    const double v = mats_h[(t * nmat + l) * nr * nr + (size_t)i * nr + j];
                    tr[(t * nmat + l) * nr * nr + (size_t)j * nr + i] = v;
                    if (v != 0.0) {
                        if (j < lo) lo = j;
                        hi = j + 1;
                    }
                }
                if (hi == 0) lo = 0;
                else {
                    if (i < r0) r0 = i;
                    r1 = i + 1;
                }
                band[2 * ((t * nmat + l) * nr + i)] = (short)lo;
                band[2 * ((t * nmat + l) * nr + i) + 1] = (short)hi;
            }
            if (r1 == 0) r0 = 0;
            rows[2 * (t * nmat + l)] = (short)r0;
            rows[2 * (t * nmat + l) + 1] = (short)r1;
        }
    // dense path: default slot map (matrix index = ell), rows mostly full, sizes that tile
    {
        // (nmat == 1: one matrix shared by every ell -- the radial transforms as GEMMs -- with the default liveness)
        bool def_map = (nmat == nl || nmat == 1);
        for (int i1 = 0; def_map && i1 < 2 * nm; ++i1)
            for (int l = 0; l < nl; ++l) {
                const int want = ((i1 >> 1) <= l) ? (nmat == 1 ? 0 : l) : -1;
                if (slot_map_h[i1 * nl + l] != want) { def_map = false; break; }
            }
        double fill = 0.0;
        for (size_t i = 0; i < band.size() / 2; ++i) fill += band[2 * i + 1] - band[2 * i];
        fill /= (double)(band.size() / 2) * nr;
        p->dense = def_map && nterms > 0 && fill > 0.5 && nr % EG_K == 0 && (ncomp_out * nr) % EG_M == 0;
        std::vector<int> tmap((size_t)ncomp_out * p->ncomp_in, -1);
        for (int t = 0; t < nterms; ++t) tmap[(size_t)co_h[t] * p->ncomp_in + ci_h[t]] = t;
        if (check_hip(hipMalloc((void **)&p->d_tmap, tmap.size() * sizeof(int)), "hipMalloc") ||
            check_hip(hipMemcpy(p->d_tmap, tmap.data(), tmap.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy")) {
            delete p;
            return -2;
        }
    }
    if (check_hip(hipMalloc((void **)&p->d_meta, meta.size() * sizeof(int)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_first, first.size() * sizeof(int)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_mats, mb), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_slot, 2L * nm * nl * sizeof(int)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_band, band.size() * sizeof(short)), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_band, band.data(), band.size() * sizeof(short), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMalloc((void **)&p->d_rows, rows.size() * sizeof(short)), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_rows, rows.data(), rows.size() * sizeof(short), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_slot, slot_map_h, 2L * nm * nl * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMemcpy(p->d_mats, tr.data(), mb, hipMemcpyHostToDevice), "hipMemcpy")) {
        delete p;
        return -2;
    }
    *h = register_handle(p);
    return 0;
}

int ddh_ell_terms_apply(ddh_handle h, const double *x, double *y, void *stream) {
    return ddh_ell_terms_apply_acc(h, x, y, 0, stream);
}

int ddh_ell_terms_apply_acc(ddh_handle h, const double *x, double *y, int accumulate, void *stream) {
    EllTerms *p = (EllTerms *)lookup_handle(h, H_ELLT);
    if (!p) return -1;
    if (x == y) return fail("ell_terms_apply: in-place unsupported");
    static const bool no_gemm = getenv("DDH_ELL_NO_GEMM") != nullptr;
    if (p->dense && !no_gemm) {
        const dim3 grid((unsigned)(p->ncomp_out * p->nr / EG_M), (unsigned)p->nl, (unsigned)((2 * p->nm + EG_N - 1) / EG_N));
        const size_t lds = (size_t)(EG_K * EG_LD > EG_N * EG_M ? EG_K * EG_LD : EG_N * EG_M) * sizeof(double);
        DDH_HIP(hipFuncSetAttribute((const void *)ell_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ell_gemm_kernel, grid, dim3(256), lds, as_stream(stream), x, y, p->d_tmap, p->d_mats, p->nm, p->nl,
                           p->nr, p->nmat, p->ncomp_in, p->ncomp_out, accumulate);
        DDH_HIP(hipGetLastError());
        return 0;
    }
    if (accumulate) return fail("ell_terms_apply_acc: accumulation is implemented for the dense (GEMM) path only");
    const int T = p->nr >= 256 ? 256 : (p->nr > 64 ? 128 : 64);
    const dim3 grid((unsigned)((2 * p->nm + ELL_S - 1) / ELL_S), (unsigned)p->nl), block(T, ELL_CO);
    const size_t lds = (size_t)p->ncomp_in * p->nr * ELL_S * sizeof(double);
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return fail("ell_terms_apply: too many components x radial modes for the LDS staging");
        DDH_HIP(hipFuncSetAttribute((const void *)ell_terms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(ell_terms_kernel, grid, block, lds, as_stream(stream), x, y, p->d_meta, p->d_first, p->d_slot,
                       p->d_band, p->d_rows, p->d_mats, p->nm, p->nl, p->nr, p->nmat, p->ncomp_in, p->ncomp_out);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_cgemv_batch_create(ddh_handle *h, int nm, int nl, int ncomp, const double *mats_h) {
    if (nm < 1 || nl < 1 || ncomp < 1) return fail("cgemv_batch_create: bad sizes");
    CgemvBatch *p = new CgemvBatch();
    p->kind = H_CGEMV;
    p->nm = nm; p->nl = nl; p->ncomp = ncomp;
    std::vector<long> off(nm + 1, 0);
    for (int m = 0; m < nm; ++m) {
        const long n = (long)ncomp * (nl - m > 0 ? nl - m : 0);
        off[m + 1] = off[m] + n * n;
    }
    p->max_n = ncomp * nl;
    p->bytes = (size_t)off[nm] * sizeof(double2);
    if (check_hip(hipMalloc((void **)&p->d_off, off.size() * sizeof(long)), "hipMalloc") ||
        check_hip(hipMalloc((void **)&p->d_mats, p->bytes ? p->bytes : 16), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_off, off.data(), off.size() * sizeof(long), hipMemcpyHostToDevice), "hipMemcpy") ||
        (p->bytes && mats_h && check_hip(hipMemcpy(p->d_mats, mats_h, p->bytes, hipMemcpyHostToDevice), "hipMemcpy"))) {
        delete p;
        return -2;
    }
    *h = register_handle(p);
    return 0;
}

int ddh_cgemv_batch_mats(ddh_handle h, double **mats_d) {
    CgemvBatch *p = (CgemvBatch *)lookup_handle(h, H_CGEMV);
    if (!p) return -1;
    *mats_d = (double *)p->d_mats;
    return 0;
}

int ddh_ell_terms_create_dense(ddh_handle *h, int nm, int nl, int nr, int ncomp) {
    if (nm < 1 || nl < 1 || nr < 1 || ncomp < 1) return fail("ell_terms_create_dense: bad sizes");
    if (nr % EG_K || (ncomp * nr) % EG_M) return fail("ell_terms_create_dense: sizes do not tile the GEMM kernel");
    EllTerms *p = new EllTerms();
    p->kind = H_ELLT;
    p->nm = nm; p->nl = nl; p->nr = nr; p->ncomp_out = ncomp; p->ncomp_in = ncomp; p->nterms = ncomp * ncomp; p->nmat = nl;
    p->dense = 1;
    std::vector<int> tmap((size_t)ncomp * ncomp);
    for (int t = 0; t < ncomp * ncomp; ++t) tmap[t] = t;
    const size_t mb = (size_t)ncomp * ncomp * nl * nr * nr * sizeof(double);
    if (check_hip(hipMalloc((void **)&p->d_tmap, tmap.size() * sizeof(int)), "hipMalloc") ||
        check_hip(hipMemcpy(p->d_tmap, tmap.data(), tmap.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy") ||
        check_hip(hipMalloc((void **)&p->d_mats, mb), "hipMalloc") || check_hip(hipMemset(p->d_mats, 0, mb), "hipMemset")) {
        delete p;
        return -2;
    }
    *h = register_handle(p);
    return 0;
}

// tmap[t] = t when block t has a non-zero entry for some ell, else -1 (the GEMM skips it); one workgroup per block
__global__ void __launch_bounds__(256) ell_prune_kernel(const double *__restrict__ mats, int *__restrict__ tmap, long per) {
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    const double *b = mats + (long)blockIdx.x * per;
    int any = 0;
    for (long i = threadIdx.x; i < per; i += 256) any |= (b[i] != 0.0);
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) tmap[blockIdx.x] = s_any ? (int)blockIdx.x : -1;
}

int ddh_ell_terms_prune(ddh_handle h, void *stream) {
    EllTerms *p = (EllTerms *)lookup_handle(h, H_ELLT);
    if (!p) return -1;
    if (!p->dense || p->nterms != p->ncomp_out * p->ncomp_in) return fail("ell_terms_prune: all-blocks dense handles only");
    hipLaunchKernelGGL(ell_prune_kernel, dim3((unsigned)p->nterms), dim3(256), 0, as_stream(stream), p->d_mats, p->d_tmap,
                       (long)p->nmat * p->nr * p->nr);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_ell_terms_mats(ddh_handle h, double **mats_d) {
    EllTerms *p = (EllTerms *)lookup_handle(h, H_ELLT);
    if (!p) return -1;
    *mats_d = p->d_mats;
    return 0;
}

int ddh_cgemv_batch_apply(ddh_handle h, const double *x, double *y, void *stream) {
    CgemvBatch *p = (CgemvBatch *)lookup_handle(h, H_CGEMV);
    if (!p) return -1;
    if (x == y) return fail("cgemv_batch_apply: in-place unsupported");
    const dim3 grid((unsigned)((p->max_n + 3) / 4), (unsigned)p->nm), block(256);
    hipLaunchKernelGGL(cgemv_batch_kernel, grid, block, 0, as_stream(stream), x, y, p->d_off, p->d_mats, p->nm, p->nl,
                       p->ncomp);
    DDH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
