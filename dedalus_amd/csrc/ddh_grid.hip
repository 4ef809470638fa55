// Streaming (HBM-bound) vector and grid-space kernels, gfx950.
//  - ddh_lincomb:        RHS assembly  y = sum_t alpha_t x_t   (timesteppers.py:617-623, 156-166)
//  - ddh_grid_bilinear:  out[c] = sum_t coef_t a[ia_t] b[ib_t] (arithmetic.py:666-674, 708-728, 855-866)
//  - ddh_grid_cfl:       max_x sum_c |u_c| / dx_c              (operators.py:4342-4419, basis.py:6078-6113)
//  - ddh_a2a_pack/unpack: block re-ordering around the RCCL all-to-all (transposes.pyx:359-445)
// All are one read of each input and one write of the output with 16-byte accesses.
#include "ddh_common.h"

namespace ddh {

constexpr int MAX_TERMS = 16;
constexpr int MAX_BIL = 32;

struct LincombArgs {
    const double *x[MAX_TERMS];
    double alpha[MAX_TERMS];
    int nterms;
};

__global__ void __launch_bounds__(256) lincomb_kernel(double *y, LincombArgs a, long n2, long n) {
    // n2 = number of double2 elements; tail handled by the last thread block
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll 4
        for (int t = 0; t < a.nterms; ++t) {
            const double2 v = reinterpret_cast<const double2 *>(a.x[t])[i];
            acc.x += a.alpha[t] * v.x;
            acc.y += a.alpha[t] * v.y;
        }
        reinterpret_cast<double2 *>(y)[i] = acc;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double acc = 0.0;
        for (int t = 0; t < a.nterms; ++t) acc += a.alpha[t] * a.x[t][n - 1];
        y[n - 1] = acc;
    }
}

struct BilArgs {
    int nterms;
    int ncomp_out;
    int ic[MAX_BIL], ia[MAX_BIL], ib[MAX_BIL];
    double coef[MAX_BIL];
};

// Each operand component is loaded once per point (terms are sorted by (ia, ib) on the host); the
// NOUT accumulators live in registers.
template <int NOUT>
__global__ void __launch_bounds__(256)
bilinear_kernel(double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, long n,
                BilArgs args) {
    const long n2 = n >> 1;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 acc[NOUT];
#pragma unroll
        for (int c = 0; c < NOUT; ++c) acc[c] = make_double2(0.0, 0.0);
        int last_a = -1, last_b = -1;
        double2 av = make_double2(0.0, 0.0), bv = av;
        for (int t = 0; t < args.nterms; ++t) {
            if (args.ia[t] != last_a) {
                last_a = args.ia[t];
                av = reinterpret_cast<const double2 *>(a + (long)last_a * n)[i];
            }
            if (args.ib[t] != last_b) {
                last_b = args.ib[t];
                bv = reinterpret_cast<const double2 *>(b + (long)last_b * n)[i];
            }
            const double cf = args.coef[t];
            const double2 pr = make_double2(cf * av.x * bv.x, cf * av.y * bv.y);
            const int ic = args.ic[t];
#pragma unroll
            for (int c = 0; c < NOUT; ++c)
                if (ic == c) { acc[c].x += pr.x; acc[c].y += pr.y; }
        }
#pragma unroll
        for (int c = 0; c < NOUT; ++c)
            if (c < args.ncomp_out) reinterpret_cast<double2 *>(out + (long)c * n)[i] = acc[c];
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        for (int c = 0; c < args.ncomp_out; ++c) {
            double acc = 0.0;
            for (int t = 0; t < args.nterms; ++t)
                if (args.ic[t] == c) acc += args.coef[t] * a[(long)args.ia[t] * n + n - 1] * b[(long)args.ib[t] * n + n - 1];
            out[(long)c * n + n - 1] = acc;
        }
    }
}

constexpr int MAX_AXES = 3;
struct CflArgs {
    const double *inv[MAX_AXES];   // per velocity component: 1/spacing along its own axis
    long len[MAX_AXES];            // storage axis lengths
    int comp_axis[MAX_AXES];       // storage axis of each velocity component
    int naxes;
    int ncomp;
};

__device__ __forceinline__ void atomic_max_double(double *addr, double v) {
    // values are non-negative: integer ordering == floating ordering
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ void __launch_bounds__(256) cfl_kernel(double *result, const double *__restrict__ u, long n, CflArgs a) {
    __shared__ double red[256];
    double m = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // decompose i into axis indices (last axis fastest)
        long rem = i;
        long idx[MAX_AXES];
        for (int ax = a.naxes - 1; ax >= 0; --ax) {
            idx[ax] = rem % a.len[ax];
            rem /= a.len[ax];
        }
        double f = 0.0;
        for (int c = 0; c < a.ncomp; ++c) f += fabs(u[(long)c * n + i]) * a.inv[c][idx[a.comp_axis[c]]];
        m = fmax(m, f);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomic_max_double(result, red[0]);
}

// spherical advective CFL (Spherical3DAdvectiveCFL, core/basis.py:6183-6204 with S2AdvectiveCFL :6156-6180):
// max over the grid of sqrt(u_phi^2 + u_theta^2) * inv_h[r] + |u_r| * inv_dr[r],  u = [3][n_ang][nr]
__global__ void __launch_bounds__(256)
cfl_spherical_kernel(double *result, const double *__restrict__ u, long n_ang, int nr, const double *__restrict__ inv_h,
                     const double *__restrict__ inv_dr) {
    __shared__ double red[256];
    double m = 0.0;
    const long n = n_ang * nr;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int ir = (int)(i % nr);
        const double up = u[i], ut = u[n + i], ur = u[2 * n + i];
        m = fmax(m, sqrt(up * up + ut * ut) * inv_h[ir] + fabs(ur) * inv_dr[ir]);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomic_max_double(result, red[0]);
}

// Copy of one contiguous segment, shared by the pack / unpack kernels below.  The segments of a transpose are few and
// large (per-component exchange of the 512^2 x 256 problem on 8 ranks: EIGHT segments of 25 MB) or many and small: the grid
// is (chunks of a segment) x (segments), every thread moves four 16-byte words per round so that a wave has 4 KiB in
// flight.  (Round 5 launched one workgroup per segment: 57 - 400 GB/s at the per-rank shapes, profiles/r6_rank_emulation.txt)
__device__ __forceinline__ void copy_segment(const double *__restrict__ s, double *__restrict__ d, long cnt, bool vec) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if (vec) {
        const double2 *s2 = reinterpret_cast<const double2 *>(s);
        double2 *d2 = reinterpret_cast<double2 *>(d);
        const long n2 = cnt >> 1;
        long i = tid;
        for (; i + 3 * nth < n2; i += 4 * nth) {
            const double2 a = s2[i], b = s2[i + nth], c = s2[i + 2 * nth], e = s2[i + 3 * nth];
            d2[i] = a;
            d2[i + nth] = b;
            d2[i + 2 * nth] = c;
            d2[i + 3 * nth] = e;
        }
        for (; i < n2; i += nth) d2[i] = s2[i];
    } else {
        for (long i = tid; i < cnt; i += nth) d[i] = s[i];
    }
}
// grid of a segment copy: x = chunks of 1024 16-byte words of one segment (at most 1024), y = segments (at most 65535)
static inline dim3 seg_grid(long nseg, long seg_doubles) {
    long cx = (seg_doubles / 2 + 1023) / 1024;
    if (cx < 1) cx = 1;
    while (cx > 1 && cx * nseg > 16384) cx = (cx + 1) / 2;       // enough workgroups to fill the chip a few times over
    if (cx > 1024) cx = 1024;
    return dim3((unsigned)cx, (unsigned)(nseg < 65535 ? nseg : 65535));
}

// [outer][na][nb*inner]  ->  [P][outer][na/P][nb*inner]   (split axis a into P blocks)
__global__ void __launch_bounds__(256)
a2a_pack_kernel(const double *__restrict__ src, double *__restrict__ dst, long outer, long na, long row, int P) {
    // [o][p][il][row] -> [p][o][il][row]: for fixed (o, p) the (il, row) block is contiguous on both sides
    const long blk = na / P;
    const long seg = blk * row;
    const long nseg = outer * P;
    const bool vec = (seg & 1) == 0;     // even segments keep every 16-byte access aligned
    for (long r = blockIdx.y; r < nseg; r += gridDim.y) {
        const long o = r / P, p = r % P;
        copy_segment(src + r * seg, dst + (p * outer + o) * seg, seg, vec);
    }
}

__global__ void __launch_bounds__(256)
a2a_unpack_kernel(const double *__restrict__ src, double *__restrict__ dst, long outer, long na, long nb, long inner,
                  int P) {
    const long blk = nb / P;
    const long seg = blk * inner;  // contiguous doubles per (p, o, ia)
    const long nseg = (long)P * outer * na;
    const bool vec = (seg & 1) == 0 && ((nb * inner) & 1) == 0;
    for (long r = blockIdx.y; r < nseg; r += gridDim.y) {
        const long p = r / (outer * na);
        const long oi = r % (outer * na);
        copy_segment(src + r * seg, dst + (oi * nb + p * blk) * inner, seg, vec);
    }
}

// Uneven blocks (the reference's Alltoallv transposes, core/transposes.pyx:287-445): an axis of length n is dealt out in
// blocks of B = ceil(n / P) -- rank p owns [p B, min((p + 1) B, n)), trailing ranks may own nothing (Layout.local_chunks
// of core/distributor.py).  Block p of the packed buffer starts where the blocks before it end.
__global__ void __launch_bounds__(256)
a2av_pack_kernel(const double *__restrict__ src, double *__restrict__ dst, long outer, long na, long row, int P, long B) {
    const long nseg = outer * P;
    for (long r = blockIdx.y; r < nseg; r += gridDim.y) {
        const long o = r / P, p = r % P;
        const long lo = (p * B < na) ? p * B : na;
        const long hi = (lo + B < na) ? lo + B : na;
        const long cnt = (hi - lo) * row;
        const long so = (o * na + lo) * row, dof = outer * row * lo + o * cnt;
        copy_segment(src + so, dst + dof, cnt, ((cnt | so | dof) & 1) == 0);
    }
}

__global__ void __launch_bounds__(256)
a2av_unpack_kernel(const double *__restrict__ src, double *__restrict__ dst, long outer_na, long nb, long inner, int P, long B) {
    const long nseg = (long)P * outer_na;
    for (long r = blockIdx.y; r < nseg; r += gridDim.y) {
        const long p = r / outer_na, oi = r % outer_na;
        const long lo = (p * B < nb) ? p * B : nb;
        const long hi = (lo + B < nb) ? lo + B : nb;
        const long cnt = (hi - lo) * inner;
        const long so = outer_na * inner * lo + oi * cnt, dof = (oi * nb + lo) * inner;
        copy_segment(src + so, dst + dof, cnt, ((cnt | so | dof) & 1) == 0);
    }
}

// min / max / sum of n doubles: fixed-shape tree (block partials in a fixed order, then one block over the partials),
// so the result does not depend on scheduling.  out3 = {min, max, sum}.
constexpr int RED_BLOCKS = 1024;

// NaN-propagating min / max (np.min / np.max semantics: a blown-up field must show up in flow.max(), fmin / fmax drop it)
__device__ __forceinline__ double nan_min(double x, double y) { return (x != x) ? x : ((y != y) ? y : fmin(x, y)); }
__device__ __forceinline__ double nan_max(double x, double y) { return (x != x) ? x : ((y != y) ? y : fmax(x, y)); }

__device__ __forceinline__ void red3_combine(double &mn, double &mx, double &sm, double a, double b, double c) {
    mn = nan_min(mn, a);
    mx = nan_max(mx, b);
    sm += c;
}

__device__ __forceinline__ void red3_block(double mn, double mx, double sm, double *out3) {
    __shared__ double s_mn[4], s_mx[4], s_sm[4];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1)
        red3_combine(mn, mx, sm, __shfl_xor(mn, sft, 64), __shfl_xor(mx, sft, 64), __shfl_xor(sm, sft, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mn[w] = mn;
        s_mx[w] = mx;
        s_sm[w] = sm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) red3_combine(mn, mx, sm, s_mn[i], s_mx[i], s_sm[i]);
        out3[0] = mn;
        out3[1] = mx;
        out3[2] = sm;
    }
}

__global__ void __launch_bounds__(256) reduce3_partial_kernel(const double *__restrict__ x, long n, double *__restrict__ part) {
    // block b owns the contiguous chunk [b * per, (b + 1) * per): partial sums are independent of the launch shape
    const long per = (n + RED_BLOCKS - 1) / RED_BLOCKS;
    const long lo = (long)blockIdx.x * per;
    const long hi = lo + per < n ? lo + per : n;
    double mn = INFINITY, mx = -INFINITY, sm = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const double v = x[i];
        mn = nan_min(mn, v);
        mx = nan_max(mx, v);
        sm += v;
    }
    red3_block(mn, mx, sm, part + 3 * blockIdx.x);
}

__global__ void __launch_bounds__(256) reduce3_final_kernel(const double *__restrict__ part, double *__restrict__ out3) {
    double mn = INFINITY, mx = -INFINITY, sm = 0.0;
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) red3_combine(mn, mx, sm, part[3 * i], part[3 * i + 1], part[3 * i + 2]);
    red3_block(mn, mx, sm, out3);
}

static unsigned stream_grid(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace ddh

using namespace ddh;

__global__ void scatter_add_kernel(double *__restrict__ y, const long *__restrict__ idx,
                                   const double *__restrict__ vals, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[idx[i]] += vals[i];
}

__global__ void scatter_set_kernel(double *__restrict__ y, const long *__restrict__ idx,
                                   const double *__restrict__ vals, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[idx[i]] = vals[i];
}

// rows of [nx][ny] doubles <-> the tile-major layout [nx / 8][ny / 8][8][8] of the solver's system vectors (ddh_pencil.hip:
// tile_offset); one 16-byte word per thread, the tiled side accessed linearly
__global__ void __launch_bounds__(256)
tile_rows_kernel(const double *__restrict__ src, double *__restrict__ dst, long nrows, long nx, long ny, int to_tiled,
                 long band_rows) {
    const long plane2 = nx * ny / 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * plane2) return;
    const long row = i / plane2, o = 2 * (i - row * plane2);
    const long tile = o >> 6, in = o & 63, tpr = ny >> 3;
    const long nat = row * nx * ny + ((tile / tpr) * 8 + (in >> 3)) * ny + (tile % tpr) * 8 + (in & 7);
    // tiled side: rows one after the other, or (band_rows) the rows of a kx band together: [kx / 8][band_rows][ky / 8][8][8]
    const long til = band_rows ? ((tile / tpr) * band_rows + row) * (8 * ny) + ((tile % tpr) << 6) + in : row * nx * ny + o;
    if (to_tiled) *reinterpret_cast<double2 *>(dst + til) = *reinterpret_cast<const double2 *>(src + nat);
    else *reinterpret_cast<double2 *>(dst + nat) = *reinterpret_cast<const double2 *>(src + til);
}

extern "C" {

/* nrows rows of [nx][ny] doubles between the natural layout and the tile-major layout of the solver's system vectors
 * ([kx / 8][ky / 8][kx % 8][ky % 8] within a row; ddh_pencil_set_state_tiled, ddh_pencil_solve_recombined_tiled): to_tiled = 1
 * natural -> tiled, 0 tiled -> natural.  Out of place; nx and ny multiples of 8.  band_rows != 0: the tiled side is a block
 * of nrows rows of a kx-band-major vector of band_rows rows ([kx / 8][band_rows][ky / 8][8][8]; its pointer = that of the
 * block's first row in band 0), the natural side the nrows rows [nrows][nx][ny].  The host layer uses it where a state
 * field kept tile-major by the solver is read or written in the natural order (user access, output, generic operators). */
int ddh_tile_rows(const double *src, double *dst, long nrows, long nx, long ny, int to_tiled, long band_rows, void *stream) {
    if (nrows <= 0) return 0;
    if (band_rows < 0) return fail("ddh_tile_rows: band_rows >= 0");
    if (nx < 8 || ny < 8 || (nx & 7) || (ny & 7)) return fail("ddh_tile_rows: nx and ny multiples of 8");
    if (src == dst) return fail("ddh_tile_rows: in-place unsupported");
    const long n = nrows * nx * ny / 2;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), src, dst, nrows,
                       nx, ny, to_tiled, band_rows);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_scatter_set(double *y, const long *idx_d, const double *vals_d, long n, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scatter_set_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), y,
                       idx_d, vals_d, n);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_scatter_add(double *y, const long *idx_d, const double *vals_d, long n, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), y,
                       idx_d, vals_d, n);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_lincomb(double *y, int nterms, const double *const *xs_h, const double *alpha_h, long n, void *stream) {
    if (nterms < 1 || nterms > MAX_TERMS) return fail("ddh_lincomb: 1..16 terms supported");
    if (n <= 0) return 0;
    LincombArgs a;
    a.nterms = nterms;
    for (int t = 0; t < nterms; ++t) {
        a.x[t] = xs_h[t];
        a.alpha[t] = alpha_h[t];
        if (((uintptr_t)xs_h[t]) & 15) return fail("ddh_lincomb: operands must be 16-byte aligned");
    }
    if (((uintptr_t)y) & 15) return fail("ddh_lincomb: output must be 16-byte aligned");
    hipLaunchKernelGGL(lincomb_kernel, dim3(stream_grid(n / 2)), dim3(256), 0, as_stream(stream), y, a, n / 2, n);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_grid_bilinear(double *out, int ncomp_out, const double *a, const double *b, long n, int nterms,
                      const int *ic_h, const int *ia_h, const int *ib_h, const double *coef_h, void *stream) {
    if (nterms < 1 || nterms > MAX_BIL) return fail("ddh_grid_bilinear: 1..32 terms supported");
    if (n <= 0) return 0;
    if (n & 1) {
        // component offsets c*n must stay 16-byte aligned for the vector path
        if (ncomp_out > 1) return fail("ddh_grid_bilinear: odd point count with several components unsupported");
    }
    if (ncomp_out > 9) return fail("ddh_grid_bilinear: at most 9 output components");
    BilArgs args;
    args.nterms = nterms;
    args.ncomp_out = ncomp_out;
    // sort terms by (ia, ib) so that every operand component is read once per point
    int order[MAX_BIL];
    for (int t = 0; t < nterms; ++t) order[t] = t;
    for (int i = 1; i < nterms; ++i) {
        int key = order[i], j = i - 1;
        while (j >= 0 && (ia_h[order[j]] > ia_h[key] || (ia_h[order[j]] == ia_h[key] && ib_h[order[j]] > ib_h[key]))) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = key;
    }
    for (int k = 0; k < nterms; ++k) {
        const int t = order[k];
        args.ic[k] = ic_h[t];
        args.ia[k] = ia_h[t];
        args.ib[k] = ib_h[t];
        args.coef[k] = coef_h[t];
        if (ic_h[t] < 0 || ic_h[t] >= ncomp_out) return fail("ddh_grid_bilinear: output component out of range");
        if ((n & 1) && (ia_h[t] || ib_h[t])) return fail("ddh_grid_bilinear: odd point count with several components unsupported");
    }
    const dim3 grid(stream_grid(n / 2)), block(256);
    hipStream_t st = as_stream(stream);
    if (ncomp_out <= 1)
        hipLaunchKernelGGL(bilinear_kernel<1>, grid, block, 0, st, out, a, b, n, args);
    else if (ncomp_out <= 3)
        hipLaunchKernelGGL(bilinear_kernel<3>, grid, block, 0, st, out, a, b, n, args);
    else
        hipLaunchKernelGGL(bilinear_kernel<9>, grid, block, 0, st, out, a, b, n, args);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_grid_cfl(double *result_d, const double *u, int ncomp, long n, const double *const *inv_spacing_comp,
                 const int *comp_axis_h, const long *axis_len_h, int naxes, void *stream) {
    if (naxes < 1 || naxes > MAX_AXES || ncomp < 1 || ncomp > MAX_AXES) return fail("ddh_grid_cfl: 1..3 axes / components");
    CflArgs a;
    a.naxes = naxes;
    a.ncomp = ncomp;
    long tot = 1;
    for (int i = 0; i < naxes; ++i) {
        a.len[i] = axis_len_h[i];
        tot *= axis_len_h[i];
    }
    for (int c = 0; c < ncomp; ++c) {
        a.inv[c] = inv_spacing_comp[c];
        a.comp_axis[c] = comp_axis_h[c];
        if (comp_axis_h[c] < 0 || comp_axis_h[c] >= naxes) return fail("ddh_grid_cfl: component axis out of range");
    }
    if (tot != n) return fail("ddh_grid_cfl: axis lengths do not multiply to n");
    DDH_HIP(hipMemsetAsync(result_d, 0, sizeof(double), as_stream(stream)));
    hipLaunchKernelGGL(cfl_kernel, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), result_d, u, n, a);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_grid_cfl_spherical(double *result_d, const double *u, long n_ang, int nr, const double *inv_h_d,
                           const double *inv_dr_d, void *stream) {
    if (n_ang <= 0 || nr <= 0) return fail("ddh_grid_cfl_spherical: empty grid");
    DDH_HIP(hipMemsetAsync(result_d, 0, sizeof(double), as_stream(stream)));
    hipLaunchKernelGGL(cfl_spherical_kernel, dim3(stream_grid(n_ang * nr)), dim3(256), 0, as_stream(stream), result_d, u,
                       n_ang, nr, inv_h_d, inv_dr_d);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_grid_reduce(double *out3_d, const double *x, long n, double *work_d, void *stream) {
    if (n <= 0) return fail("ddh_grid_reduce: empty array");
    hipLaunchKernelGGL(reduce3_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, as_stream(stream), x, n, work_d);
    hipLaunchKernelGGL(reduce3_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const double *)work_d, out3_d);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_a2a_pack(const double *src, double *dst, long outer, long na, long nb, long inner, int nparts, void *stream) {
    if (nparts < 1 || na % nparts) return fail("ddh_a2a_pack: axis length must be divisible by nparts");
    const long row = nb * inner;
    const long nrows = outer * nparts;
    if (nrows <= 0 || na * row == 0) return 0;
    hipLaunchKernelGGL(a2a_pack_kernel, seg_grid(nrows, (na / nparts) * row), dim3(256), 0, as_stream(stream), src, dst, outer, na,
                       row, nparts);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_a2a_unpack(const double *src, double *dst, long outer, long na, long nb, long inner, int nparts, void *stream) {
    if (nparts < 1 || nb % nparts) return fail("ddh_a2a_unpack: axis length must be divisible by nparts");
    const long nseg = (long)nparts * outer * na;
    if (nseg <= 0 || nb * inner == 0) return 0;
    hipLaunchKernelGGL(a2a_unpack_kernel, seg_grid(nseg, (nb / nparts) * inner), dim3(256), 0, as_stream(stream), src, dst, outer,
                       na, nb, inner, nparts);
    DDH_HIP(hipGetLastError());
    return 0;
}

/* uneven blocks of B = ceil(n / P) along the split axis (Alltoallv transposes, core/transposes.pyx:287-445) */
int ddh_a2av_pack(const double *src, double *dst, long outer, long na, long row, int nparts, void *stream) {
    return ddh_a2av_pack_b(src, dst, outer, na, row, nparts, 0, stream);
}
int ddh_a2av_unpack(const double *src, double *dst, long outer_na, long nb, long inner, int nparts, void *stream) {
    return ddh_a2av_unpack_b(src, dst, outer_na, nb, inner, nparts, 0, stream);
}

int ddh_a2av_pack_b(const double *src, double *dst, long outer, long na, long row, int nparts, long block, void *stream) {
    if (nparts < 1 || na < 1) return fail("ddh_a2av_pack: bad arguments");
    const long B = block ? block : (na + nparts - 1) / nparts;
    const long nseg = outer * nparts;
    if (nseg <= 0) return 0;
    hipLaunchKernelGGL(a2av_pack_kernel, seg_grid(nseg, B * row), dim3(256), 0, as_stream(stream), src, dst, outer, na, row, nparts, B);
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_a2av_unpack_b(const double *src, double *dst, long outer_na, long nb, long inner, int nparts, long block, void *stream) {
    if (nparts < 1 || nb < 1) return fail("ddh_a2av_unpack: bad arguments");
    const long B = block ? block : (nb + nparts - 1) / nparts;
    const long nseg = (long)nparts * outer_na;
    if (nseg <= 0) return 0;
    hipLaunchKernelGGL(a2av_unpack_kernel, seg_grid(nseg, B * inner), dim3(256), 0, as_stream(stream), src, dst, outer_na, nb, inner, nparts, B);
    DDH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
