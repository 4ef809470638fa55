// Batched 1-D spectral transforms along one axis of [outer][n][inner] float64 arrays, gfx950.
//
// One workgroup transforms B "line pairs" held in LDS.  Two real lines are packed into one
// complex FFT (real part = line a, imaginary part = line b) so no flop or LDS byte is wasted on
// Hermitian redundancy.  When the transformed axis is strided (inner > 1) the two lines of a pair
// are neighbours along the contiguous inner axis, i.e. one 16-byte load; when the axis is the
// contiguous one (inner == 1) the two lines are neighbours along the outer axis.
//
// The FFT itself is a mixed-radix (2,3,4,5,7) Stockham autosort in LDS, in place with register
// staging (each thread holds <= 12 complex values per pass), twiddles from an exact table.
// All of the reference's separate pack / scale / pad / truncate NumPy passes
// (core/transforms.py:469-509, 726-746, 844-890) are fused into the load and store phases, so each
// transform is exactly one HBM read of its input and one HBM write of its output.
#include "ddh_common.h"
#include <algorithm>

#include <cmath>
#include <cstdlib>

#include "ddh_fft_dev.h"

namespace ddh {


// LDS index padding: one extra element after every 16 (= one 256-B bank row), so that the stride-R
// writes of the first Stockham pass (and the DCT permutation) spread over the banks.
__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }

// One Stockham pass of radix R over B lines of length N held in buf[line*ld + j].
// Twiddles come from a two-level table held in LDS (W^q = hi[q >> 5] * lo[q & 31], < 1 KiB per
// workgroup): the per-butterfly table reads from global memory competed with the streaming traffic for
// L1 and cost an L2 round trip per pass.  Powers w^t are built by binary products (depth <= 4).
__host__ __device__ __forceinline__ int tw_entries(int N, int direct) { return direct ? N : 32 + (N >> 5) + 1; }
template <typename P>
__device__ __forceinline__ void load_twiddles(const P &p, double2 *tw_lo, double2 *&tw_hi, int tid, int T) {
    if (p.twdirect) {
        for (int i = tid; i < p.N; i += T) tw_lo[i] = p.tw[i];
        tw_hi = nullptr;
        return;
    }
    for (int i = tid; i < 32 + (p.N >> 5) + 1; i += T) {
        if (i < 32) {
            tw_lo[i] = p.tw[i < p.N ? i : 0];
        } else {
            const int q = (i - 32) << 5;
            tw_hi[i - 32] = p.tw[q < p.N ? q : 0];
        }
    }
}
__device__ __forceinline__ double2 lds_twiddle(const double2 *tw_lo, const double2 *tw_hi, int q, int sign) {
    double2 w = tw_hi ? cmul(tw_hi[q >> 5], tw_lo[q & 31]) : tw_lo[q];
    if (sign > 0) w.y = -w.y;
    return w;
}

template <int R>
__device__ __forceinline__ void fft_pass(double2 *buf, int ld, int B, int N, int Ns, const FastDiv &fd_nb,
                                         const FastDiv &fd_ns, const double2 *tw_lo, const double2 *tw_hi, int sign,
                                         int tid, int T, int dbg = 0) {
    constexpr int MAXI = (12 + R - 1) / R;   // ceil(12 / R): at least 12 (at most 16) complex values staged per thread
    const int nb = N / R;
    const int total = nb * B;
    const int twstep = nb / Ns;  // N / (Ns*R)
    // The padded index lpad(i) = i + (i >> 4) is linear along a butterfly's reads when the read stride nb
    // is a multiple of 16, and along its writes when Ns is (or for the first radix-16 pass): one add per
    // LDS access instead of the add/shift/add chain.
    const bool lin_r = (nb & 15) == 0;
    const bool lin_w = ((Ns & 15) == 0) || (Ns == 1 && R == 16);
    const int rstride = nb + (nb >> 4);
    const int wstride = (Ns == 1) ? 1 : Ns + (Ns >> 4);
    double2 v[MAXI][R];
    int wl[MAXI], wj[MAXI];     // line * ld (-1: no butterfly) and lpad(j0) (linear writes) or j0
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int w = tid + it * T;
        wl[it] = -1;
        wj[it] = 0;
        if (w < total) {
            unsigned uline, uj, uq, uk;
            fd_nb.divmod((unsigned)w, uline, uj);
            fd_ns.divmod(uj, uq, uk);
            const int line = (int)uline, j = (int)uj, k = (int)uk;
            const double2 *x = buf + line * ld;
            if (lin_r) {
                const double2 *x0 = x + lpad(j);
#pragma unroll
                for (int t = 0; t < R; ++t) v[it][t] = x0[t * rstride];
            } else {
#pragma unroll
                for (int t = 0; t < R; ++t) v[it][t] = x[lpad(j + t * nb)];
            }
            if (Ns > 1 && !(dbg & 1)) {
                double2 wp[R];    // wp[t] = w^t
                wp[1] = lds_twiddle(tw_lo, tw_hi, k * twstep, sign);
#pragma unroll
                for (int t = 2; t < R; ++t) {
                    // binary products: t = hb + rest with hb the highest power of two <= t
                    const int hb = (t >= 8) ? 8 : (t >= 4) ? 4 : 2;
                    wp[t] = (t == hb) ? cmul(wp[hb / 2], wp[hb / 2]) : cmul(wp[hb], wp[t - hb]);
                }
#pragma unroll
                for (int t = 1; t < R; ++t) v[it][t] = cmul(v[it][t], wp[t]);
            }
            if (!(dbg & 1)) butterfly<R>(v[it], sign);
            const int j0 = (j - k) * R + k;
            wl[it] = line * ld;
            wj[it] = lin_w ? lpad(j0) : j0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        if (wl[it] >= 0) {
            double2 *x = buf + wl[it];
            if (lin_w) {
                double2 *x0 = x + wj[it];
#pragma unroll
                for (int u = 0; u < R; ++u) x0[u * wstride] = v[it][u];
            } else {
#pragma unroll
                for (int u = 0; u < R; ++u) x[lpad(wj[it] + u * Ns)] = v[it][u];
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void lds_fft(double2 *buf, const FftDev &p, const double2 *tw_lo, const double2 *tw_hi,
                                        int sign, int tid, int T, int nlines_wg = 0) {
    int Ns = 1;
    if (p.dbg & 2) return;
    const int B = nlines_wg > 0 ? nlines_wg : p.B;   // lines (pairs) in the workgroup's buffer
    for (int i = 0; i < p.nradix; ++i) {
        const int R = p.radix[i];
        switch (R) {
            case 2: fft_pass<2>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 3: fft_pass<3>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 4: fft_pass<4>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 5: fft_pass<5>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 6: fft_pass<6>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 8: fft_pass<8>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            case 16: fft_pass<16>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
            default: fft_pass<7>(buf, p.ld, B, p.N, Ns, p.fd_nb[i], p.fd_ns[i], tw_lo, tw_hi, sign, tid, T, p.dbg); break;
        }
        Ns *= R;
    }
}

// ------------------------------------------------------------------------------------------------
// Pair addressing.  INNER: pairs along the contiguous inner axis; otherwise (inner == 1) pairs of
// adjacent outer lines.
// ------------------------------------------------------------------------------------------------
template <bool INNER>
struct PairIO {
    long outer_idx;  // INNER only
    long inner;      // INNER only
    long nlines;     // !INNER: number of real lines (= outer)
    bool vec;        // INNER: inner even -> aligned double2 access

    __device__ __forceinline__ double2 load(const double *base, long n_axis, long j, long q) const {
        if (INNER) {
            const double *p = base + ((outer_idx * n_axis + j) * inner + 2 * q);
            if (vec) return *reinterpret_cast<const double2 *>(p);
            double a = p[0];
            double b = (2 * q + 1 < inner) ? p[1] : 0.0;
            return make_double2(a, b);
        } else {
            const double *p = base + (2 * q) * n_axis + j;
            double a = p[0];
            double b = (2 * q + 1 < nlines) ? p[n_axis] : 0.0;
            return make_double2(a, b);
        }
    }
    __device__ __forceinline__ void store(double *base, long n_axis, long j, long q, double2 v) const {
        if (INNER) {
            double *p = base + ((outer_idx * n_axis + j) * inner + 2 * q);
            if (vec) {
                *reinterpret_cast<double2 *>(p) = v;
            } else {
                p[0] = v.x;
                if (2 * q + 1 < inner) p[1] = v.y;
            }
        } else {
            double *p = base + (2 * q) * n_axis + j;
            p[0] = v.x;
            if (2 * q + 1 < nlines) p[n_axis] = v.y;
        }
    }
};

// item -> (axis index j, pair slot b) so that global accesses coalesce
template <bool INNER>
__device__ __forceinline__ void split_item(int w, const FastDiv &fdB, const FastDiv &fdn, int &j, int &b) {
    unsigned q, r;
    if (INNER) {
        fdB.divmod((unsigned)w, q, r);
        b = (int)r;
        j = (int)q;
    } else {
        fdn.divmod((unsigned)w, q, r);
        j = (int)r;
        b = (int)q;
    }
}

constexpr int LDU = 6;   // global loads staged per thread before the dependent LDS writes

__device__ __forceinline__ int dct_perm(int j, int N) { return (j & 1) ? (N - 1 - (j >> 1)) : (j >> 1); }

template <int MODE, bool INNER, int TMAX, int MINW>
__global__ void __launch_bounds__(TMAX, MINW)
fft_axis_kernel(FftDev p, const double *__restrict__ src, double *__restrict__ dst, long outer, long inner,
                long npairs, unsigned blocks_per_outer) {
    extern __shared__ double2 lds[];
    double2 *buf = lds;                 // [B][ld]
    double2 *tw_lo = lds + p.B * p.ld;  // [32]        W^q          (or the full table [N])
    double2 *tw_hi = tw_lo + 32;        // [N/32 + 1]  W^(32 q)
    load_twiddles(p, tw_lo, tw_hi, threadIdx.x, blockDim.x);
    // Chebyshev normalisation (Appendix A of SURVEY.md; transforms.py:720-724, 737-746, 823-826, 844-860)
    // computed on the fly: forward sgn*sqrt(pi/2)/N (k=0: sqrt(pi)/(2N)), backward sgn/(2 sqrt(pi/2)) (k=0: 1/sqrt(pi))
    const double kSqPi = 1.7724538509055160272981674833411, kSqPi2 = 1.2533141373155002512078826424055;
    const double fs0 = kSqPi / (2.0 * (double)p.N), fs1 = kSqPi2 / (double)p.N;
    const double bs0 = 1.0 / kSqPi, bs1 = 0.5 / kSqPi2;
    auto fscale_of = [&](int k) -> double { return k == 0 ? fs0 : ((k & 1) ? -fs1 : fs1); };
    auto bscale_of = [&](int k) -> double { return k == 0 ? bs0 : ((k & 1) ? -bs1 : bs1); };
    auto half_of = [&](int k) -> double2 { return p.half[k]; };   // small table, read once per element
    const int tid = threadIdx.x, T = blockDim.x;
    unsigned bid = xcd_swizzle(blockIdx.x, gridDim.x);
    if (INNER && p.spread_s > 1) {
        // Strided lines whose rows are a large power of two apart map a contiguous window of workgroups
        // to few HBM channels: spread each XCD's run of workgroups over its whole address range, in
        // chunks of spread_c neighbours (which share DRAM pages).
        const unsigned per = gridDim.x >> 3, S = (unsigned)p.spread_s, c = (unsigned)p.spread_c;
        if (per > 0 && bid < (per << 3) && per % (S * c) == 0) {
            const unsigned x = bid / per, l = bid - x * per;
            const unsigned chunk = l / c, r = l - chunk * c, nchunk = per / c;
            bid = x * per + ((chunk % S) * (nchunk / S) + chunk / S) * c + r;
        }
    }
    PairIO<INNER> io;
    long q0;
    if (INNER) {
        io.outer_idx = bid / blocks_per_outer;
        io.inner = inner;
        io.vec = (inner % 2 == 0);
        q0 = (long)(bid % blocks_per_outer) * p.B;
    } else {
        io.nlines = outer;
        q0 = (long)bid * p.B;
    }
    const int N = p.N, M = p.M, B = p.B, ld = p.ld;
    const double invN = 1.0 / (double)N;
    long long t_start = 0, t_loaded = 0, t_fft = 0;
    if (p.prof) t_start = clock64();

    // Dual output (RFFT_BWD only, ddh_rfft_backward_dual): a second pass transforms the SAME coefficient tile again with
    // another derivative scale into dst2 -- the field and its derivative along the axis from one HBM read of the
    // coefficients (the second pass re-reads the tile the workgroup has just read: L2 hits).
    // CHEB_BWD (ddh_cheb_backward_dual): pass 0 is the plain transform of the coefficients (no conversion solve), pass 1
    // transforms dvec[k] * c[k + 1] -- the z-derivative, a one-superdiagonal operator into the plan's (a0+1, b0+1) basis --
    // through the plan's conversion solve.
    const int npass = ((MODE == RFFT_BWD || MODE == CHEB_BWD) && p.dst2 != nullptr) ? 2 : 1;
    const bool cheb_plain = (p.nbands == 0);
    double *const dst_first = dst;
    for (int pass = 0; pass < npass; ++pass) {
    const double dsc = (pass == 0) ? p.dscale : p.dscale2;
    dst = (pass == 0) ? dst_first : p.dst2;
    if (pass) __syncthreads();          // the LDS tile of the first pass has been stored
    // ---------------------------------------------------------------- load + pre-process
    if (p.dbg & 4) {
        // timing ablation: no loads
    } else if (MODE == RFFT_FWD && !INNER && (N % 2 == 0)) {
        const int Nh = N / 2;
        for (int b = 0; b < B; ++b) {
            const bool has_a = (q0 + b < npairs);
            const bool has_b = has_a && (2 * (q0 + b) + 1 < io.nlines);
            const double *pa = src + (2 * (q0 + b)) * (long)N;
            for (int jh = tid; jh < Nh; jh += T) {
                double2 va = make_double2(0.0, 0.0), vb = va;
                if (has_a) va = *reinterpret_cast<const double2 *>(pa + 2 * jh);
                if (has_b) vb = *reinterpret_cast<const double2 *>(pa + N + 2 * jh);
                buf[b * ld + lpad(2 * jh)] = make_double2(va.x, vb.x);
                buf[b * ld + lpad(2 * jh + 1)] = make_double2(va.y, vb.y);
            }
        }
    } else if (MODE == RFFT_FWD) {
        // LDU items per thread in flight: all loads of a group are issued before the first LDS write
        for (int w0 = tid; w0 < N * B; w0 += LDU * T) {
            double2 v[LDU];
            int at[LDU];
#pragma unroll
            for (int u = 0; u < LDU; ++u) {
                const int w = w0 + u * T;
                at[u] = -1;
                v[u] = make_double2(0.0, 0.0);
                if (w < N * B) {
                    int j, b;
                    split_item<INNER>(w, p.fdB, p.fdN, j, b);
                    if (q0 + b < npairs) v[u] = io.load(src, N, j, q0 + b);
                    at[u] = b * ld + lpad(j);
                }
            }
#pragma unroll
            for (int u = 0; u < LDU; ++u)
                if (at[u] >= 0) buf[at[u]] = v[u];
        }
    } else if (MODE == RFFT_BWD) {
        const int K = p.K;
        // zero the dealiased band K < k < N-K
        const int nzero = N - 2 * K - 1;
        for (int w = tid; w < nzero * B; w += T) {
            unsigned q, r;
            p.fdB.divmod((unsigned)w, q, r);
            const int b = (int)r, j = (int)q;
            buf[b * ld + lpad(K + 1 + j)] = make_double2(0.0, 0.0);
        }
        if (INNER) {
            for (int w0 = tid; w0 < (K + 1) * B; w0 += LDU * T) {
                double2 cc[LDU], ss[LDU];
                int kk[LDU], bb[LDU];
#pragma unroll
                for (int u = 0; u < LDU; ++u) {
                    const int w = w0 + u * T;
                    kk[u] = -1;
                    bb[u] = 0;
                    cc[u] = ss[u] = make_double2(0.0, 0.0);
                    if (w < (K + 1) * B) {
                        unsigned q, r;
                        p.fdB.divmod((unsigned)w, q, r);
                        bb[u] = (int)r;
                        kk[u] = (int)q;
                        if (q0 + bb[u] < npairs) {
                            cc[u] = io.load(src, M, 2 * kk[u], q0 + bb[u]);
                            ss[u] = io.load(src, M, 2 * kk[u] + 1, q0 + bb[u]);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < LDU; ++u) {
                    if (kk[u] < 0) continue;
                    const int k = kk[u], b = bb[u];
                    double2 c = cc[u], s = ss[u];
                    if (dsc != 0.0) {   // d/dx: (cos, msin) -> (-kappa msin, kappa cos)
                        const double kap = dsc * (double)k;
                        const double2 c2 = make_double2(-kap * s.x, -kap * s.y);
                        s = make_double2(kap * c.x, kap * c.y);
                        c = c2;
                    }
                    if (k == 0) {
                        buf[b * ld] = c;  // a0 of line a + i a0 of line b
                    } else {
                        buf[b * ld + lpad(k)] = make_double2(0.5 * (c.x - s.y), 0.5 * (s.x + c.y));
                        buf[b * ld + lpad(N - k)] = make_double2(0.5 * (c.x + s.y), 0.5 * (c.y - s.x));
                    }
                }
            }
        } else {
            // contiguous lines: element 2k, 2k+1 of each line
            for (int w = tid; w < (K + 1) * B; w += T) {
                unsigned q, r;
                p.fdK1.divmod((unsigned)w, q, r);
                const int k = (int)r, b = (int)q;
                double2 c = make_double2(0.0, 0.0), s = c;
                if (q0 + b < npairs) {
                    // (cos, msin) of mode k are adjacent: one 16-byte load per line
                    const double *pa = src + (2 * (q0 + b)) * (long)M + 2 * k;
                    const double2 va = *reinterpret_cast<const double2 *>(pa);
                    double2 vb = make_double2(0.0, 0.0);
                    if (2 * (q0 + b) + 1 < io.nlines) vb = *reinterpret_cast<const double2 *>(pa + M);
                    c = make_double2(va.x, vb.x);
                    s = make_double2(va.y, vb.y);
                    if (dsc != 0.0) {
                        const double kap = dsc * (double)k;
                        const double2 c2 = make_double2(-kap * s.x, -kap * s.y);
                        s = make_double2(kap * c.x, kap * c.y);
                        c = c2;
                    }
                }
                if (k == 0) {
                    buf[b * ld] = c;
                } else {
                    buf[b * ld + lpad(k)] = make_double2(0.5 * (c.x - s.y), 0.5 * (s.x + c.y));
                    buf[b * ld + lpad(N - k)] = make_double2(0.5 * (c.x + s.y), 0.5 * (c.y - s.x));
                }
            }
        }
    } else if (MODE == CHEB_FWD) {
        for (int w0 = tid; w0 < N * B; w0 += LDU * T) {
            double2 v[LDU];
            int at[LDU];
#pragma unroll
            for (int u = 0; u < LDU; ++u) {
                const int w = w0 + u * T;
                at[u] = -1;
                v[u] = make_double2(0.0, 0.0);
                if (w < N * B) {
                    int j, b;
                    split_item<INNER>(w, p.fdB, p.fdN, j, b);
                    if (q0 + b < npairs) v[u] = io.load(src, N, j, q0 + b);
                    at[u] = b * ld + lpad(dct_perm(j, N));
                }
            }
#pragma unroll
            for (int u = 0; u < LDU; ++u)
                if (at[u] >= 0) buf[at[u]] = v[u];
        }
    } else if (MODE == CHEB_BWD && (cheb_plain || (npass == 2 && pass == 0))) {
        // grid basis == coefficient basis: build the DCT-III input straight from global memory
        // (each coefficient is read as k and as N-k; the second read hits L1/L2), no staging buffer
        const int Mk = (M < N) ? M : N;
        for (int w0 = tid; w0 < N * B; w0 += LDU * T) {
            double2 ce[LDU], cf[LDU], hh[LDU];
            int kk[LDU], at[LDU];
#pragma unroll
            for (int u = 0; u < LDU; ++u) {
                const int w = w0 + u * T;
                at[u] = -1;
                kk[u] = 0;
                ce[u] = cf[u] = hh[u] = make_double2(0.0, 0.0);
                if (w < N * B) {
                    int k, b;
                    split_item<INNER>(w, p.fdB, p.fdN, k, b);
                    kk[u] = k;
                    at[u] = b * ld + lpad(k);
                    hh[u] = half_of(k);
                    if (q0 + b < npairs) {
                        if (k < Mk) ce[u] = io.load(src, M, k, q0 + b);
                        const int kr = N - k;
                        if (k > 0 && kr < Mk) cf[u] = io.load(src, M, kr, q0 + b);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LDU; ++u) {
                if (at[u] < 0) continue;
                const int k = kk[u];
                const double se = bscale_of(k), sf = bscale_of(N - k);
                const double2 e = make_double2(se * ce[u].x, se * ce[u].y);
                const double2 f = make_double2(sf * cf[u].x, sf * cf[u].y);
                const double cr = hh[u].x, ci = -hh[u].y;
                const double var = e.x * cr + f.x * ci, vai = e.x * ci - f.x * cr;
                const double vbr = e.y * cr + f.y * ci, vbi = e.y * ci - f.y * cr;
                buf[at[u]] = make_double2(var - vbi, vai + vbr);
            }
        }
    } else if (MODE == CHEB_BWD) {
        // Coefficients in a basis (a, b) != (a0, b0): conversion solve first.  They are staged IN the transform buffer,
        // coefficient k at the padded slot of index k, solved in place and then turned into the DCT-III input pairwise
        // in place (slots k and N - k depend on coefficients k and N - k only): no staging buffer, the tile is as
        // large as the plain transform's.  Coefficients beyond the grid size are dropped BEFORE the conversion solve
        // (transforms.py:878-881); with a zero right-hand side there the solve leaves them zero, so it starts at Mk - 1.
        const int Mk = (M < N) ? M : N;
        for (int w0 = tid; w0 < Mk * B; w0 += LDU * T) {      // LDU loads in flight per thread
            double2 vv[LDU];
            double dk[LDU];
            int at[LDU];
#pragma unroll
            for (int u = 0; u < LDU; ++u) {
                const int w = w0 + u * T;
                at[u] = -1;
                vv[u] = make_double2(0.0, 0.0);
                dk[u] = 1.0;
                if (w < Mk * B) {
                    int k, b;
                    if (Mk == M) {
                        split_item<INNER>(w, p.fdB, p.fdM, k, b);
                    } else {
                        split_item<INNER>(w, p.fdB, p.fdN, k, b);
                    }
                    at[u] = b * ld + lpad(k);
                    if (npass == 2) {
                        if (q0 + b < npairs && k + 1 < M) {
                            vv[u] = io.load(src, M, k + 1, q0 + b);
                            dk[u] = p.dvec[k];
                        }
                    } else if (q0 + b < npairs) {
                        vv[u] = io.load(src, M, k, q0 + b);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LDU; ++u)
                if (at[u] >= 0) buf[at[u]] = make_double2(dk[u] * vv[u].x, dk[u] * vv[u].y);
        }
        __syncthreads();
        {
            // upper-banded back substitution (solve_upper_sparse, transforms.py:876-890): independent chains
            // k == r (mod gcd_off), both lines of a pair at once.
            const int g = p.gcd_off;
            const int nchain = g * B;
            const int L = (Mk + g - 1) / g;                 // longest chain
            const int E = (L + 63) >> 6;                    // chain elements per lane of a wavefront
            if (p.bsub != nullptr && p.bsub_order == 1 && E <= 8) {
                // First-order chains x_i = beta_i - alpha_i x_{i-1}: one wavefront per chain, affine-map prefix scan
                // over the lanes (each lane composes its E consecutive elements, 6 shuffle steps combine the lanes).
                const int lane = tid & 63, wv = tid >> 6;
                const double *t0 = p.bsub, *t1 = p.bsub + M;
                for (int ch = wv; ch < nchain; ch += T / 64) {
                    const int r = ch % g, b = ch / g;
                    double2 *c = buf + b * ld;
                    const int kstart = Mk - 1 - ((Mk - 1 - r) % g);
                    double2 beta[8];
                    double alpha[8];
                    double A = 1.0;
                    double2 Bv = make_double2(0.0, 0.0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        beta[e] = make_double2(0.0, 0.0);
                        alpha[e] = 0.0;
                        const int k = kstart - (lane * E + e) * g;
                        if (e < E && k >= 0) {
                            const double2 v = c[lpad(k)];
                            const double iv = t0[k];
                            beta[e] = make_double2(v.x * iv, v.y * iv);
                            alpha[e] = t1[k];
                        }
                        if (e < E) {        // elements past the chain end are the map x -> 0 - 0 x: harmless, never stored
                            Bv = make_double2(beta[e].x - alpha[e] * Bv.x, beta[e].y - alpha[e] * Bv.y);
                            A = -alpha[e] * A;
                        }
                    }
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const double Ap = __shfl_up(A, off, 64);
                        const double Bx = __shfl_up(Bv.x, off, 64), By = __shfl_up(Bv.y, off, 64);
                        if (lane >= off) {
                            Bv = make_double2(Bv.x + A * Bx, Bv.y + A * By);
                            A = A * Ap;
                        }
                    }
                    double xx = __shfl_up(Bv.x, 1, 64), xy = __shfl_up(Bv.y, 1, 64);
                    if (lane == 0) xx = xy = 0.0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = kstart - (lane * E + e) * g;
                        if (e < E && k >= 0) {
                            xx = beta[e].x - alpha[e] * xx;
                            xy = beta[e].y - alpha[e] * xy;
                            c[lpad(k)] = make_double2(xx, xy);
                        }
                    }
                }
            } else {
                for (int w = tid; w < nchain; w += T) {
                    const int r = w % g, b = w / g;
                    double2 *c = buf + b * ld;
                    const int kstart = Mk - 1 - ((Mk - 1 - r) % g);  // largest k < Mk with k % g == r
                    if (p.bsub != nullptr) {
                        // the two previous unknowns of the chain stay in registers: the loads of c and of the table do
                        // not depend on the chain, which is then one fused multiply-add pair per step
                        const double *t0 = p.bsub, *t1 = p.bsub + M, *t2 = p.bsub + 2 * M;
                        double2 x1 = make_double2(0.0, 0.0), x2 = x1;
#pragma unroll 4
                        for (int k = kstart; k >= 0; k -= g) {
                            const double2 v = c[lpad(k)];
                            const double iv = t0[k], a1 = t1[k], a2 = t2[k];
                            double2 x;
                            x.x = v.x * iv - a1 * x1.x - a2 * x2.x;
                            x.y = v.y * iv - a1 * x1.y - a2 * x2.y;
                            c[lpad(k)] = x;
                            x2 = x1;
                            x1 = x;
                        }
                        continue;
                    }
                    for (int k = kstart; k >= 0; k -= g) {
                        double2 acc = c[lpad(k)];
                        for (int d = 1; d < p.nbands; ++d) {
                            const int kk = k + p.boff[d];
                            if (kk < Mk) {
                                const double a = p.bands[d * M + k];
                                acc.x -= a * c[lpad(kk)].x;
                                acc.y -= a * c[lpad(kk)].y;
                            }
                        }
                        const double inv = 1.0 / p.bands[k];
                        c[lpad(k)] = make_double2(acc.x * inv, acc.y * inv);
                    }
                }
            }
        }
        __syncthreads();
        const int KH = N / 2 + 1;
        for (int w = tid; w < KH * B; w += T) {
            const int b = w / KH, k = w - b * KH;
            double2 *line = buf + b * ld;
            const int kr = N - k;
            double2 e = make_double2(0.0, 0.0), f = e;
            if (k < Mk) {
                const double s = bscale_of(k);
                const double2 v = line[lpad(k)];
                e = make_double2(s * v.x, s * v.y);
            }
            if (k > 0 && kr < Mk) {
                const double s = bscale_of(kr);
                const double2 v = line[lpad(kr)];
                f = make_double2(s * v.x, s * v.y);
            }
            {
                const double2 h = half_of(k);  // exp(-i pi k / 2N); we need its conjugate
                const double cr = h.x, ci = -h.y;
                // V^a = (e.x - i f.x)(cr + i ci), V^b likewise with .y ; Z = V^a + i V^b
                const double var = e.x * cr + f.x * ci, vai = e.x * ci - f.x * cr;
                const double vbr = e.y * cr + f.y * ci, vbi = e.y * ci - f.y * cr;
                line[lpad(k)] = make_double2(var - vbi, vai + vbr);
            }
            if (k > 0 && kr != k) {            // the partner slot: the roles of e and f swap
                const double2 h = half_of(kr);
                const double cr = h.x, ci = -h.y;
                const double var = f.x * cr + e.x * ci, vai = f.x * ci - e.x * cr;
                const double vbr = f.y * cr + e.y * ci, vbi = f.y * ci - e.y * cr;
                line[lpad(kr)] = make_double2(var - vbi, vai + vbr);
            }
        }
    } else if (MODE == CFFT_FWD) {
        // complex lines, no pairing: "pair" slot = one complex line; load() returns (re, im)
        for (int w = tid; w < N * B; w += T) {
            int j, b;
            split_item<INNER>(w, p.fdB, p.fdN, j, b);
            double2 v = make_double2(0.0, 0.0);
            if (q0 + b < npairs) {
                const long line = q0 + b;
                const double *ptr = INNER ? src + 2 * ((io.outer_idx * N + j) * inner + line)
                                          : src + 2 * (line * N + j);
                v = *reinterpret_cast<const double2 *>(ptr);
            }
            buf[b * ld + lpad(j)] = v;
        }
    } else if (MODE == CFFT_BWD) {
        const int K = p.K;
        for (int w = tid; w < N * B; w += T) {
            int j, b;
            split_item<INNER>(w, p.fdB, p.fdN, j, b);
            // fft position j -> wavenumber k in (-N/2, N/2]; keep |k| <= K
            int k = (j <= N / 2) ? j : j - N;
            double2 v = make_double2(0.0, 0.0);
            if (k >= -K && k <= K && q0 + b < npairs) {
                const long line = q0 + b;
                const int m = (k >= 0) ? k : M + k;
                const double *ptr = INNER ? src + 2 * ((io.outer_idx * M + m) * inner + line)
                                          : src + 2 * (line * M + m);
                v = *reinterpret_cast<const double2 *>(ptr);
            }
            buf[b * ld + lpad(j)] = v;
        }
    }
    __syncthreads();
    if (p.prof) t_loaded = clock64();

    // ---------------------------------------------------------------- FFT in LDS
    const int sign = (MODE == RFFT_FWD || MODE == CHEB_FWD || MODE == CFFT_FWD) ? -1 : +1;
    lds_fft(buf, p, tw_lo, tw_hi, sign, tid, T);
    if (p.prof) t_fft = clock64();

    // ---------------------------------------------------------------- post-process + store
    if (p.dbg & 16) {
        // timing ablation: no stores
    } else if (MODE == RFFT_BWD && !INNER && (N % 2 == 0)) {
        // contiguous lines: each thread stores two consecutive grid points of line a and of line b (16 B each)
        const int Nh = N / 2;
        for (int b = 0; b < B; ++b) {
            if (q0 + b >= npairs) break;
            double *pa = dst + (2 * (q0 + b)) * (long)N;
            const bool has_b = (2 * (q0 + b) + 1 < io.nlines);
            for (int jh = tid; jh < Nh; jh += T) {
                const double2 z0 = buf[b * ld + lpad(2 * jh)], z1 = buf[b * ld + lpad(2 * jh + 1)];
                *reinterpret_cast<double2 *>(pa + 2 * jh) = make_double2(z0.x, z1.x);
                if (has_b) *reinterpret_cast<double2 *>(pa + N + 2 * jh) = make_double2(z0.y, z1.y);
            }
        }
    } else if (MODE == RFFT_BWD || MODE == CHEB_BWD) {
        for (int w = tid; w < N * B; w += T) {
            int j, b;
            split_item<INNER>(w, p.fdB, p.fdN, j, b);
            if (q0 + b < npairs) {
                const int pos = (MODE == CHEB_BWD) ? dct_perm(j, N) : j;
                io.store(dst, N, j, q0 + b, buf[b * ld + lpad(pos)]);
            }
        }
    } else if (MODE == RFFT_FWD) {
        const int K = p.K;
        const int Mh = M / 2;
        for (int w = tid; w < Mh * B; w += T) {
            int k, b;
            split_item<INNER>(w, p.fdB, p.fdMh, k, b);
            if (q0 + b >= npairs) continue;
            double2 c = make_double2(0.0, 0.0), s = c;
            if (k == 0) {
                const double2 z = buf[b * ld];
                c = make_double2(z.x * invN, z.y * invN);
            } else if (k <= K) {
                const double2 z1 = buf[b * ld + lpad(k)], z2 = buf[b * ld + lpad(N - k)];
                c = make_double2((z1.x + z2.x) * invN, (z1.y + z2.y) * invN);
                s = make_double2((z1.y - z2.y) * invN, (z2.x - z1.x) * invN);
            }
            if (!INNER) {
                double *pa = dst + (2 * (q0 + b)) * (long)M + 2 * k;
                *reinterpret_cast<double2 *>(pa) = make_double2(c.x, s.x);
                if (2 * (q0 + b) + 1 < io.nlines) *reinterpret_cast<double2 *>(pa + M) = make_double2(c.y, s.y);
            } else {
                io.store(dst, M, 2 * k, q0 + b, c);
                io.store(dst, M, 2 * k + 1, q0 + b, s);
            }
        }
    } else if (MODE == CHEB_FWD) {
        const int Mk = (M < N) ? M : N;
        if (p.nbands == 0) {
            for (int w = tid; w < M * B; w += T) {
                int k, b;
                split_item<INNER>(w, p.fdB, p.fdM, k, b);
                if (q0 + b >= npairs) continue;
                double2 c = make_double2(0.0, 0.0);
                if (k < Mk) {
                    const double2 z1 = buf[b * ld + lpad(k)], z2 = buf[b * ld + lpad(((k == 0) ? 0 : N - k))];
                    const double2 h = half_of(k);
                    const double s = fscale_of(k);
                    c.x = s * ((z1.x + z2.x) * h.x - (z1.y - z2.y) * h.y);
                    c.y = s * ((z1.y + z2.y) * h.x - (z2.x - z1.x) * h.y);
                }
                io.store(dst, M, k, q0 + b, c);
            }
        } else {
            // forward_conversion apply (transforms.py:862-874): c'_k = sum_d C[k, k + off_d] c_(k + off_d), with every
            // coefficient c_j of the family's own basis formed on the fly from the FFT outputs j and N - j (a few more
            // LDS reads per output instead of a staging pass over the tile and a barrier)
            for (int w = tid; w < M * B; w += T) {
                int k, b;
                split_item<INNER>(w, p.fdB, p.fdM, k, b);
                if (q0 + b >= npairs) continue;
                const double2 *line = buf + b * ld;
                double2 acc = make_double2(0.0, 0.0);
                for (int d = 0; d < p.nbands; ++d) {
                    const int kk = k + p.boff[d];
                    if (kk < Mk) {
                        const double a = p.bands[d * M + k];
                        const double2 z1 = line[lpad(kk)], z2 = line[lpad((kk == 0) ? 0 : N - kk)];
                        const double2 h = half_of(kk);
                        const double sc = a * fscale_of(kk);
                        acc.x += sc * ((z1.x + z2.x) * h.x - (z1.y - z2.y) * h.y);
                        acc.y += sc * ((z1.y + z2.y) * h.x - (z2.x - z1.x) * h.y);
                    }
                }
                io.store(dst, M, k, q0 + b, acc);
            }
        }
    } else if (MODE == CFFT_FWD) {
        const int K = p.K;
        for (int w = tid; w < M * B; w += T) {
            int m, b;
            split_item<INNER>(w, p.fdB, p.fdM, m, b);
            if (q0 + b >= npairs) continue;
            // coefficient slot m -> wavenumber (transforms.py:201-208): k = m for m <= KM else m - M
            const int KM = (M - 1) / 2;
            const int k = (m <= KM) ? m : m - M;
            double2 v = make_double2(0.0, 0.0);
            if (k >= -K && k <= K && !(2 * m == M)) {
                const double2 z = buf[b * ld + lpad(((k >= 0) ? k : N + k))];
                v = make_double2(z.x * invN, z.y * invN);
            }
            const long line = q0 + b;
            double *ptr = INNER ? dst + 2 * ((io.outer_idx * M + m) * inner + line) : dst + 2 * (line * M + m);
            *reinterpret_cast<double2 *>(ptr) = v;
        }
    } else if (MODE == CFFT_BWD) {
        for (int w = tid; w < N * B; w += T) {
            int j, b;
            split_item<INNER>(w, p.fdB, p.fdN, j, b);
            if (q0 + b >= npairs) continue;
            const long line = q0 + b;
            double *ptr = INNER ? dst + 2 * ((io.outer_idx * N + j) * inner + line) : dst + 2 * (line * N + j);
            *reinterpret_cast<double2 *>(ptr) = buf[b * ld + lpad(j)];
        }
    }
    }   // pass
    if (p.prof) {
        __syncthreads();
        if (tid == 0) {
            const long long t_end = clock64();
            atomicAdd(&p.prof[0], (unsigned long long)(t_loaded - t_start));
            atomicAdd(&p.prof[1], (unsigned long long)(t_fft - t_loaded));
            atomicAdd(&p.prof[2], (unsigned long long)(t_end - t_fft));
            atomicAdd(&p.prof[3], 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused "grid stage" along the contiguous real-Fourier axis:
//     out[ic] = forward_rfft( sum_t coef_t * backward_rfft(a[ia_t]) * backward_rfft(b[ib_t]) )
// The dealiased grid values of the last axis never leave the CU: the `a` operands (e.g. the velocity) and
// the accumulators live in registers, one LDS buffer serves all FFTs.  This replaces three HBM passes of
// the reference sequence (backward transform of every operand, DotProduct/MultiplyFields.operate,
// forward transform: core/transforms.py:537-565, core/arithmetic.py:666-674, 855-866) by one read of
// the operands' coefficient lines and one write of the results' coefficient lines.
// ------------------------------------------------------------------------------------------------

// coefficient lines (2q, 2q+1) of one operand -> Hermitian-packed spectrum of the line pair in buf
// (contiguous-axis RFFT_BWD pre-step), optionally differentiated:
// (cos, msin) -> (-kappa msin, kappa cos), kappa = dscale * k
__device__ __forceinline__ void fused_load_lines(double2 *buf, const FftDev &p, const double *src, double dscale,
                                                 long q, long nlines, int tid, int T) {
    const int N = p.N, M = p.M, K = p.K;
    for (int w = K + 1 + tid; w < N - K; w += T) buf[lpad(w)] = make_double2(0.0, 0.0);
    const bool second = 2 * q + 1 < nlines;
    const double *pa = src + (2 * q) * (long)M;
    for (int k = tid; k <= K; k += T) {
        const double2 va = *reinterpret_cast<const double2 *>(pa + 2 * k);
        double2 vb = make_double2(0.0, 0.0);
        if (second) vb = *reinterpret_cast<const double2 *>(pa + M + 2 * k);
        double2 c = make_double2(va.x, vb.x);
        double2 s = make_double2(va.y, vb.y);
        if (dscale != 0.0) {
            const double kap = dscale * (double)k;
            const double2 c2 = make_double2(-kap * s.x, -kap * s.y);
            s = make_double2(kap * c.x, kap * c.y);
            c = c2;
        }
        if (k == 0) {
            buf[0] = c;
        } else {
            buf[lpad(k)] = make_double2(0.5 * (c.x - s.y), 0.5 * (s.x + c.y));
            buf[lpad(N - k)] = make_double2(0.5 * (c.x + s.y), 0.5 * (c.y - s.x));
        }
    }
}

// spectrum in buf -> interleaved (cos, msin) coefficient lines (contiguous-axis RFFT_FWD post-step)
__device__ __forceinline__ void fused_store_lines(const double2 *buf, const FftDev &p, double *dst, long q,
                                                  long nlines, int tid, int T) {
    const int N = p.N, M = p.M, K = p.K;
    const double invN = 1.0 / (double)N;
    const bool second = 2 * q + 1 < nlines;
    double *pa = dst + (2 * q) * (long)M;
    for (int k = tid; k < M / 2; k += T) {
        double2 c = make_double2(0.0, 0.0), s = c;
        if (k == 0) {
            const double2 z = buf[0];
            c = make_double2(z.x * invN, z.y * invN);
        } else if (k <= K) {
            const double2 z1 = buf[lpad(k)], z2 = buf[lpad(N - k)];
            c = make_double2((z1.x + z2.x) * invN, (z1.y + z2.y) * invN);
            s = make_double2((z1.y - z2.y) * invN, (z2.x - z1.x) * invN);
        }
        *reinterpret_cast<double2 *>(pa + 2 * k) = make_double2(c.x, s.x);
        if (second) *reinterpret_cast<double2 *>(pa + M + 2 * k) = make_double2(c.y, s.y);
    }
}

// One workgroup = one pair of lines.  PTS grid points per thread (N <= PTS * 256); up to G operands are
// transformed together so that the barriers and LDS round trips of an FFT are shared by G transforms.
template <int PTS, int G>
__global__ void __launch_bounds__(FUSED_T, 2)
fused_rfft_bilinear_kernel(FftDev p, FusedArgs f, long nlines, long npairs) {
    extern __shared__ double2 lds[];
    double2 *buf = lds;                                // [G][ld]   FFT work buffers
    double2 *tw_lo = lds + G * p.ld;
    double2 *tw_hi = tw_lo + 32;
    const int tid = threadIdx.x, T = FUSED_T;
    load_twiddles(p, tw_lo, tw_hi, tid, T);
    // kernel arguments indexed at run time go through LDS (static indices here keep them out of scratch)
    __shared__ const double *s_src[FUSED_LOADS];
    __shared__ double s_dscale[FUSED_LOADS];
    __shared__ double *s_out[FUSED_NC];
    __shared__ double s_coef[FUSED_TERMS];
    __shared__ short s_tbeg[FUSED_LOADS + 1], s_bbeg[FUSED_LOADS + 1];
    __shared__ signed char s_flush[FUSED_LOADS], s_ia[FUSED_TERMS];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < FUSED_LOADS; ++i) {
            s_src[i] = f.src[i];
            s_dscale[i] = f.dscale[i];
            s_tbeg[i] = f.tbeg[i];
            s_bbeg[i] = f.bbeg[i];
            s_flush[i] = f.flush[i];
        }
        s_tbeg[FUSED_LOADS] = f.tbeg[FUSED_LOADS];
        s_bbeg[FUSED_LOADS] = f.bbeg[FUSED_LOADS];
#pragma unroll
        for (int i = 0; i < FUSED_TERMS; ++i) {
            s_coef[i] = f.coef[i];
            s_ia[i] = f.ia[i];
        }
#pragma unroll
        for (int i = 0; i < FUSED_NC; ++i) s_out[i] = f.out[i];
    }
    __syncthreads();                                      // argument tables and twiddles are visible
    const long q = (p.rot & 2) ? (long)blockIdx.x : (long)xcd_swizzle(blockIdx.x, gridDim.x);   // line pair of this workgroup
    const int N = p.N, ld = p.ld;
    // LDS address of this thread's i-th grid point
    int addr[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int e = tid + i * T;
        addr[i] = (e < N) ? lpad(e) : -1;
    }
    double2 areg[FUSED_NA][PTS];
    double2 acc[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        acc[i] = make_double2(0.0, 0.0);
#pragma unroll
        for (int ia = 0; ia < FUSED_NA; ++ia) areg[ia][i] = make_double2(0.0, 0.0);
    }
    const int na = f.na, nbatch = f.nbatch;
    long long pt[4] = {0, 0, 0, 0}, tc = 0;   // debug phase clocks: wait, load, fft, rest
    if (p.prof) tc = clock64();
#define DDH_TICK(slot)                      \
    if (p.prof) {                           \
        const long long now = clock64();    \
        pt[slot] += now - tc;               \
        tc = now;                           \
    }
#pragma unroll 1
    for (int ib = 0; ib < nbatch; ++ib) {
        const int l0 = s_bbeg[ib], cnt = s_bbeg[ib + 1] - l0;
        DDH_TICK(3)
        __syncthreads();                     // buf is free
        DDH_TICK(0)
        if (!(p.dbg & 4)) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (g < cnt) fused_load_lines(buf + g * ld, p, s_src[l0 + g], s_dscale[l0 + g], q, nlines, tid, T);
        }
        __syncthreads();
        DDH_TICK(1)
        lds_fft(buf, p, tw_lo, tw_hi, +1, tid, T, cnt);
        DDH_TICK(2)
        if (l0 < na) {
            // `a` operands stay in registers (run-time index, static register selection)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < cnt) {
#pragma unroll
                    for (int i = 0; i < PTS; ++i) {
                        const double2 v = (addr[i] >= 0) ? buf[g * ld + addr[i]] : make_double2(0.0, 0.0);
#pragma unroll
                        for (int ia = 0; ia < FUSED_NA; ++ia)
                            if (l0 + g == ia) areg[ia][i] = v;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g < cnt) {
                double2 wv[PTS];
#pragma unroll
                for (int i = 0; i < PTS; ++i) wv[i] = (addr[i] >= 0) ? buf[g * ld + addr[i]] : make_double2(0.0, 0.0);
                const int t1 = (p.dbg & 8) ? 0 : s_tbeg[l0 + g + 1];
#pragma unroll 1
                for (int t = s_tbeg[l0 + g]; t < t1; ++t) {
                    const double cf = s_coef[t];
                    const int tia = s_ia[t];
#pragma unroll
                    for (int i = 0; i < PTS; ++i) {
                        double2 av = areg[0][i];
#pragma unroll
                        for (int ia = 1; ia < FUSED_NA; ++ia)
                            if (tia == ia) av = areg[ia][i];
                        // two packed lines: real parts multiply real parts, imaginary parts imaginary parts
                        acc[i].x += cf * av.x * wv[i].x;
                        acc[i].y += cf * av.y * wv[i].y;
                    }
                }
            }
        }
        const int oc = s_flush[ib];
        if (oc >= 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PTS; ++i) {
                if (addr[i] >= 0) buf[addr[i]] = acc[i];
                acc[i] = make_double2(0.0, 0.0);
            }
            __syncthreads();
            DDH_TICK(3)
            lds_fft(buf, p, tw_lo, tw_hi, -1, tid, T, 1);
            DDH_TICK(2)
            fused_store_lines(buf, p, s_out[oc], q, nlines, tid, T);
        }
    }
    DDH_TICK(3)
#undef DDH_TICK
    if (p.prof && tid == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&p.prof[i], (unsigned long long)pt[i]);
}

// ddh_fftwave.hip: wave-per-four-pairs transforms along a strided axis; 0 = launched, 1 = shape not covered, < 0 error
int wave_axis_try(int mode, const FftDev &d, const double *src, double *dst, long outer, long inner, double *dst2,
                  const double *dvec, double dscale, double dscale2, hipStream_t st);
// ddh_gridwave.hip: wave-per-line variant of the fused grid stage for N = 128*C
// ddh_fftwave.hip: the same lane code along a CONTIGUOUS axis (Chebyshev, N = 192: the shell's radial transforms)
int wave_contig_try(int mode, const FftDev &d, const double *src, double *dst, long outer, double *dst2, hipStream_t st);
bool gridwave_supported(const FftDev &d);
int launch_gridwave(const FftDev &d, const FusedArgs &f, long nlines, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool factorize(int n, int *radix, int &nradix) {
    nradix = 0;
    while (n % 16 == 0) { radix[nradix++] = 16; n /= 16; }
    while (n % 8 == 0) { radix[nradix++] = 8; n /= 8; }
    while (n % 4 == 0) { radix[nradix++] = 4; n /= 4; }
    while (n % 2 == 0) { radix[nradix++] = 2; n /= 2; }
    while (n % 3 == 0) { radix[nradix++] = 3; n /= 3; }
    while (n % 5 == 0) { radix[nradix++] = 5; n /= 5; }
    while (n % 7 == 0) { radix[nradix++] = 7; n /= 7; }
    if (nradix == 0 && n == 1) return true;  // N == 1
    if (n != 1 || nradix > MAX_RADIX_PASSES) return false;
    // radix 6 (round 6): 2 x 3 -> 6 and 4 x 3 x 3 -> 6 x 6 -- one LDS pass (two workgroup barriers) fewer for the 3/2-padded
    // sizes 96, 576, 1536, ... (16 16 2 3 -> 16 16 6; 16 4 3 3 -> 16 6 6).  DDH_FFT_NO_R6=1 keeps the old schedules (A/B).
    static const bool no6 = getenv("DDH_FFT_NO_R6") != nullptr;
    auto count = [&](int r) { int c = 0; for (int i = 0; i < nradix; ++i) c += radix[i] == r; return c; };
    auto drop = [&](int r) { for (int i = 0; i < nradix; ++i) if (radix[i] == r) { for (int j = i; j + 1 < nradix; ++j) radix[j] = radix[j + 1]; --nradix; return; } };
    while (!no6 && count(2) >= 1 && count(3) >= 1) { drop(2); drop(3); radix[nradix++] = 6; }
    while (!no6 && count(4) >= 1 && count(3) >= 2) { drop(4); drop(3); drop(3); radix[nradix++] = 6; radix[nradix++] = 6; }
    return true;
}

template <typename T>
static int upload(void **dptr, const std::vector<T> &v) {
    DDH_HIP(hipMalloc(dptr, v.size() * sizeof(T) + 16));
    DDH_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int make_plan(ddh_handle *out, int kind, int n_grid, int n_coeff, int nbands, const int *boff,
                     const double *bands) {
    if (n_grid < 1 || n_coeff < 1) return fail("plan: sizes must be positive");
    FftPlan *pl = new FftPlan();
    pl->kind = H_FFT;
    pl->tkind = kind;
    FftDev &d = pl->dev;
    memset(&d, 0, sizeof(d));
    d.N = n_grid;
    d.M = n_coeff;
    if (!factorize(n_grid, d.radix, d.nradix)) {
        delete pl;
        return fail("plan: grid size must factor into 2,3,5,7 (got " + std::to_string(n_grid) + ")");
    }
    if (kind == K_RFFT) {
        if (n_coeff % 2) {
            delete pl;
            return fail("plan_rfft: n_coeff must be even");
        }
        int KN = (n_grid - 1) / 2, KM = (n_coeff - 1) / 2;
        d.K = KN < KM ? KN : KM;
    } else if (kind == K_CFFT) {
        int KN = (n_grid - 1) / 2, KM = (n_coeff - 1) / 2;
        d.K = KN < KM ? KN : KM;
    }
    const int N = n_grid, M = n_coeff;
    std::vector<double2> tw(N), half(N);
    for (int q = 0; q < N; ++q) {
        long double a = -2.0L * M_PIl * (long double)q / (long double)N;
        tw[q] = make_double2((double)cosl(a), (double)sinl(a));
        long double h = -M_PIl * (long double)q / (2.0L * (long double)N);
        half[q] = make_double2((double)cosl(h), (double)sinl(h));
    }
    int st = upload(&pl->d_tw, tw);
    if (!st) st = upload(&pl->d_half, half);
    d.tw = (const double2 *)pl->d_tw;
    d.half = (const double2 *)pl->d_half;
    if (!st && kind == K_CHEB) {
        // Appendix A of SURVEY.md / transforms.py:720-724, 737-746, 823-826, 844-860
        const int L = N > M ? N : M;
        std::vector<double> fs(L, 0.0), bs(L, 0.0);
        const long double sqpi = sqrtl(M_PIl), sqpi2 = sqrtl(M_PIl / 2.0L);
        for (int k = 0; k < L; ++k) {
            long double sgn = (k % 2) ? -1.0L : 1.0L;
            if (k == 0) {
                fs[k] = (double)(sqpi / (2.0L * N));
                bs[k] = (double)(1.0L / sqpi);
            } else {
                fs[k] = (double)(sgn * sqpi2 / (long double)N);
                bs[k] = (double)(sgn / (2.0L * sqpi2));
            }
        }
        st = upload(&pl->d_fscale, fs);
        if (!st) st = upload(&pl->d_bscale, bs);
        d.fscale = (const double *)pl->d_fscale;
        d.bscale = (const double *)pl->d_bscale;
        if (nbands > MAX_BANDS) {
            delete pl;
            return fail("plan_cheb: at most 4 conversion bands supported");
        }
        d.nbands = nbands;
        d.gcd_off = 1;
        if (!st && nbands > 0) {
            if (boff[0] != 0) {
                delete pl;
                return fail("plan_cheb: first band must be the diagonal");
            }
            int g = 0;
            for (int i = 0; i < nbands; ++i) {
                d.boff[i] = boff[i];
                int a = boff[i], b = g;
                while (b) { int t = a % b; a = b; b = t; }
                g = a;
            }
            d.gcd_off = g > 0 ? g : 1;
            std::vector<double> bv(bands, bands + (size_t)nbands * M);
            // Back-substitution table for the register-carried solve (kernel, CHEB_BWD): along a chain k, k+g, k+2g, ...
            //   x[k] = bsub[0][k] c[k] - bsub[1][k] x[k+g] - bsub[2][k] x[k+2g]
            // (bsub[0] = 1/diag, bsub[j] = band with offset j g over diag); built when no band reaches beyond 2 g.
            bool ring_ok = true;
            for (int i = 1; i < nbands; ++i)
                if (boff[i] / d.gcd_off > 2) ring_ok = false;
            if (ring_ok) {
                const size_t base = bv.size();
                bv.resize(base + (size_t)3 * M, 0.0);
                for (int k = 0; k < M; ++k) {
                    const double inv = 1.0 / bands[k];
                    bv[base + k] = inv;
                    for (int i = 1; i < nbands; ++i)
                        if (k + boff[i] < M) bv[base + (size_t)(boff[i] / d.gcd_off) * M + k] += bands[(size_t)i * M + k] * inv;
                }
            }
            st = upload(&pl->d_bands, bv);
            d.bands = (const double *)pl->d_bands;
            d.bsub = ring_ok ? d.bands + (size_t)nbands * M : nullptr;
            d.bsub_order = 0;
            if (ring_ok) {
                const size_t base = (size_t)nbands * M;
                d.bsub_order = 1;
                for (int k = 0; k < M; ++k)
                    if (bv[base + (size_t)2 * M + k] != 0.0) d.bsub_order = 2;
            }
        }
    }
    if (st) {
        delete pl;
        return st;
    }
    d.ld = N + (N >> 4) + 1;
    d.rot = getenv("DDH_FUSED_ROT") ? atoi(getenv("DDH_FUSED_ROT")) : 0;
    d.dbg = getenv("DDH_FFT_DBG") ? atoi(getenv("DDH_FFT_DBG")) : 0;
    d.spread_s = 0;
    d.spread_c = 1;
    if (const char *sp = getenv("DDH_FFT_SPREAD")) {
        d.spread_s = atoi(sp);
        const char *cm = strchr(sp, ',');
        d.spread_c = cm ? atoi(cm + 1) : 1;
        if (d.spread_c < 1) d.spread_c = 1;
    }
    d.twdirect = (getenv("DDH_FFT_TWDIRECT") && atoi(getenv("DDH_FFT_TWDIRECT"))) ? 1 : 0;
    if (const char *rs = getenv("DDH_FFT_RADIX")) {   // tuning aid: "16,16,3" replaces the schedule when it fits N
        int r[MAX_RADIX_PASSES], n = 0, prod = 1;
        for (const char *c = rs; *c && n < MAX_RADIX_PASSES;) {
            r[n] = atoi(c);
            prod *= r[n] > 0 ? r[n] : 1;
            ++n;
            while (*c && *c != ',') ++c;
            if (*c == ',') ++c;
        }
        bool ok = prod == N;
        for (int i = 0; i < n; ++i)
            ok = ok && (r[i] == 2 || r[i] == 3 || r[i] == 4 || r[i] == 5 || r[i] == 6 || r[i] == 7 || r[i] == 8 || r[i] == 16);
        if (ok) {
            d.nradix = n;
            for (int i = 0; i < n; ++i) d.radix[i] = r[i];
        }
    }
    d.prof = nullptr;
    if (getenv("DDH_FFT_PROF") && atoi(getenv("DDH_FFT_PROF"))) {
        void *pm = nullptr;
        if (hipMalloc(&pm, 4 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(pm, 0, 4 * sizeof(unsigned long long));
            d.prof = (unsigned long long *)pm;
        }
    }
    d.fdN.set((unsigned)N);
    d.fdM.set((unsigned)M);
    d.fdMh.set((unsigned)(M / 2 > 0 ? M / 2 : 1));
    d.fdK1.set((unsigned)(d.K + 1));
    {
        int Ns = 1;
        for (int i = 0; i < d.nradix; ++i) {
            d.fd_nb[i].set((unsigned)(N / d.radix[i]));
            d.fd_ns[i].set((unsigned)Ns);
            Ns *= d.radix[i];
        }
    }
    *out = register_handle(pl);
    return 0;
}

static long g_wave_launches = 0;      // ddh_fft_wave_launches

template <int MODE>
static int launch(FftPlan *pl, const double *src, double *dst, long outer, long inner, void *stream,
                  double dscale = 0.0, double *dst2 = nullptr, double dscale2 = 0.0, const double *dvec = nullptr) {
    if (outer <= 0 || inner <= 0) return 0;
    FftDev d = pl->dev;
    d.dscale = dscale;
    d.dst2 = dst2;
    d.dscale2 = dscale2;
    d.dvec = dvec;
    const bool is_cfft = (MODE == CFFT_FWD || MODE == CFFT_BWD);
    const bool inner_mode = inner > 1;
    if (!is_cfft && inner_mode) {
        // strided axis at an instantiated size: one wavefront per four line pairs (ddh_fftwave.hip)
        const int wst = wave_axis_try(MODE, d, src, dst, outer, inner, dst2, dvec, dscale, dscale2, as_stream(stream));
        if (wst == 0) ++g_wave_launches;
        if (wst <= 0) return wst;
    }
    if (!is_cfft && !inner_mode) {
        const int wst = wave_contig_try(MODE, d, src, dst, outer, dst2, as_stream(stream));
        if (wst == 0) ++g_wave_launches;
        if (wst <= 0) return wst;
    }
    if (d.xb && inner_mode) return fail("x-blocked stage layout (ddh_fft_set_stage_layout): only the strided-axis wave kernels at their "
                          "instantiated sizes read / write it -- this transform would have used another kernel");
    if (d.ctile_nseg) return fail("tile-major coefficient rows (ddh_cheb_forward_tiled, ddh_fft_set_coeff_tiled): only the strided-axis "
                                  "Chebyshev wave kernels at their instantiated sizes read / write that layout");
    long npairs;
    if (is_cfft)
        npairs = inner_mode ? inner : outer;
    else
        npairs = inner_mode ? (inner + 1) / 2 : (outer + 1) / 2;
    // lines per workgroup: 64 B of contiguous data per row when strided; bounded by LDS (<= 64 KiB
    // so that at least two workgroups share a CU) and by 12 staged values per thread.
    const int N = d.N;
    const size_t per_line = (size_t)d.ld * sizeof(double2);
    static const int envB = getenv("DDH_FFT_B") ? atoi(getenv("DDH_FFT_B")) : 0;
    static const long lds_cap = getenv("DDH_FFT_LDSCAP") ? atol(getenv("DDH_FFT_LDSCAP")) : 64 * 1024;
    int B = inner_mode ? 8 : 4;      // strided: 128-byte contiguous segments per row when LDS allows
    if (!inner_mode) {
        // contiguous SHORT lines (the shell's radial transforms: 192 <- 128): 4 line pairs are a tile of a few KiB, the
        // workgroup's fixed costs (twiddle tables, barriers) dominate -- up to 16 pairs while a thread keeps <= 12 items
        while (B < 16 && (long)N * (2 * B) <= 12L * 256) B *= 2;
    }
    if (envB > 0) B = envB;
    while (B > 1 && (long)(per_line * B) > lds_cap) B /= 2;
    if ((long)B > npairs) B = (int)npairs;
    if (per_line * B > 160 * 1024) return fail("transform: axis too long for the LDS kernel");
    int T = 256;
    while ((long)N * B > 12L * T && T < 1024) T *= 2;
    if ((long)N * B > 12L * T) return fail("transform: axis too long for the LDS kernel (registers)");
    d.B = B;
    d.fdB.set((unsigned)B);
    const unsigned bpo = (unsigned)((npairs + B - 1) / B);
    const unsigned long nblocks = inner_mode ? (unsigned long)bpo * (unsigned long)outer : bpo;
    if (nblocks > 0x7fffffffUL) return fail("transform: grid too large");
    const size_t lds = per_line * B + (size_t)tw_entries(N, d.twdirect) * sizeof(double2);
    hipStream_t s = as_stream(stream);
#define DDH_FFT_LAUNCH(INNERV, TMAXV, MINWV)                                                                \
    {                                                                                                       \
        auto kern = fft_axis_kernel<MODE, INNERV, TMAXV, MINWV>;                                            \
        if (lds > 64 * 1024)                                                                                \
            DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                        (int)lds));                                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(T), lds, s, d, src, dst, outer, inner,       \
                           npairs, bpo);                                                                    \
    }
    // register budget follows the real block size (a 1024-thread bound would cap at 64 VGPRs and spill)
    static const int occ4 = (getenv("DDH_FFT_OCC4") && atoi(getenv("DDH_FFT_OCC4"))) ? 1 : 0;
    if (inner_mode) {
        if (T <= 256) { if (occ4) DDH_FFT_LAUNCH(true, 256, 4) else DDH_FFT_LAUNCH(true, 256, 1) }
        else DDH_FFT_LAUNCH(true, 1024, 1)
    } else {
        if (T <= 256) { if (occ4) DDH_FFT_LAUNCH(false, 256, 4) else DDH_FFT_LAUNCH(false, 256, 1) }
        else DDH_FFT_LAUNCH(false, 1024, 1)
    }
#undef DDH_FFT_LAUNCH
    DDH_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// dense matrix transform along an axis: out[o,i,x] = sum_j mat[i,j] in[o,j,x]
// ------------------------------------------------------------------------------------------------
struct MmtPlan : HandleBase {
    int n_out, n_in;
    double *d_mat = nullptr;
    ~MmtPlan() override { (void)hipFree(d_mat); }
};

constexpr int MT_I = 32, MT_X = 64, MT_J = 16;

__global__ void __launch_bounds__(256)
mmt_kernel(const double *__restrict__ mat, const double *__restrict__ in, double *__restrict__ out, int n_out,
           int n_in, long inner) {
    // tile: MT_I output rows x MT_X inner columns; threads: 256 = 64 (x) * 4 (i groups of 8)
    __shared__ double sA[MT_I][MT_J + 1];
    __shared__ double sB[MT_J][MT_X];
    const long o = blockIdx.z;
    const int i0 = blockIdx.y * MT_I;
    const long x0 = (long)blockIdx.x * MT_X;
    const int tx = threadIdx.x % MT_X, ty = threadIdx.x / MT_X;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double *inb = in + o * (long)n_in * inner;
    for (int j0 = 0; j0 < n_in; j0 += MT_J) {
        for (int w = threadIdx.x; w < MT_I * MT_J; w += 256) {
            int jj = w % MT_J, ii = w / MT_J;
            sA[ii][jj] = (i0 + ii < n_out && j0 + jj < n_in) ? mat[(long)(i0 + ii) * n_in + j0 + jj] : 0.0;
        }
        for (int w = threadIdx.x; w < MT_J * MT_X; w += 256) {
            int xx = w % MT_X, jj = w / MT_X;
            sB[jj][xx] = (j0 + jj < n_in && x0 + xx < inner) ? inb[(long)(j0 + jj) * inner + x0 + xx] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < MT_J; ++jj) {
            const double bv = sB[jj][tx];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] += sA[ty * 8 + r][jj] * bv;
        }
        __syncthreads();
    }
    if (x0 + tx < inner) {
        double *ob = out + o * (long)n_out * inner;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = i0 + ty * 8 + r;
            if (i < n_out) ob[(long)i * inner + x0 + tx] = acc[r];
        }
    }
}

// The same product on the matrix cores: v_mfma_f64_16x16x4 (A: row = lane & 15, k = lane >> 4; B: k = lane >> 4,
// col = lane & 15; C/D: col = lane & 15, row = (lane >> 4) + 4 r).  Workgroup = 4 waves, tile = 64 output rows x 64
// inner columns, contraction in chunks of 32 through LDS for the data operand, the matrix operand from L2.
constexpr int MM_M = 64, MM_N = 64, MM_K = 32, MM_LD = MM_N + 1;
typedef double mm_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
mmt_mfma_kernel(const double *__restrict__ mat, const double *__restrict__ in, double *__restrict__ out, int n_out,
                int n_in, long inner) {
    __shared__ double sB[MM_K * MM_LD];
    const long o = blockIdx.z;
    const int i0 = blockIdx.y * MM_M;
    const long x0 = (long)blockIdx.x * MM_N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double *inb = in + o * (long)n_in * inner;
    double *ob = out + o * (long)n_out * inner;
    const int arow = i0 + 16 * wave + (lane & 15);
    const bool arow_ok = arow < n_out;
    const double *Arow = mat + (long)(arow_ok ? arow : 0) * n_in;
    mm_d4 acc[MM_N / 16];
#pragma unroll
    for (int jt = 0; jt < MM_N / 16; ++jt) acc[jt] = (mm_d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < n_in; k0 += MM_K) {
        __syncthreads();
        for (int w = tid; w < MM_K * MM_N; w += 256) {
            const int kk = w / MM_N, x = w - kk * MM_N;
            sB[kk * MM_LD + x] = (k0 + kk < n_in && x0 + x < inner) ? inb[(long)(k0 + kk) * inner + x0 + x] : 0.0;
        }
        __syncthreads();
        if (i0 + 16 * wave >= n_out) continue;
#pragma unroll
        for (int k4 = 0; k4 < MM_K; k4 += 4) {
            const int k = k0 + k4 + (lane >> 4);
            const double a = (arow_ok && k < n_in) ? Arow[k] : 0.0;
            const double *br = sB + (k4 + (lane >> 4)) * MM_LD + (lane & 15);
#pragma unroll
            for (int jt = 0; jt < MM_N / 16; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br[jt * 16], acc[jt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jt = 0; jt < MM_N / 16; ++jt) {
        const long x = x0 + jt * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * wave + (lane >> 4) + 4 * r;
            if (i < n_out && x < inner) ob[(long)i * inner + x] = acc[jt][r];
        }
    }
}

}  // namespace ddh

using namespace ddh;

extern "C" {

/* debug only (not part of the documented ABI): average cycles per workgroup in the three phases */
int ddh_debug_fft_prof(ddh_handle plan, double *out4) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl || !pl->dev.prof) return fail("no profiling buffer (set DDH_FFT_PROF=1 before planning)");
    unsigned long long h[4];
    DDH_HIP(hipMemcpy(h, pl->dev.prof, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; ++i) out4[i] = h[3] ? (double)h[i] / (double)h[3] : 0.0;
    out4[3] = (double)h[3];
    (void)hipMemset(pl->dev.prof, 0, sizeof(h));
    return 0;
}

int ddh_plan_rfft(ddh_handle *plan, int n_grid, int n_coeff) {
    return make_plan(plan, K_RFFT, n_grid, n_coeff, 0, nullptr, nullptr);
}
int ddh_plan_cfft(ddh_handle *plan, int n_grid, int n_coeff) {
    return make_plan(plan, K_CFFT, n_grid, n_coeff, 0, nullptr, nullptr);
}
int ddh_plan_cheb(ddh_handle *plan, int n_grid, int n_coeff, int nbands, const int *band_offsets_h,
                  const double *bands_h) {
    return make_plan(plan, K_CHEB, n_grid, n_coeff, nbands, band_offsets_h, bands_h);
}

#define DDH_FFT_ENTRY(name, KIND, MODE)                                                               \
    int name(ddh_handle plan, const double *a, double *b, long outer, long inner, void *stream) {     \
        FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);                                          \
        if (!pl) return -1;                                                                           \
        if (pl->tkind != KIND) return fail(#name ": plan is of a different transform kind");           \
        if (outer <= 0 || inner <= 0) return 0;                                                       \
        {                                                                                             \
            const size_t cx = (KIND == K_CFFT) ? 2 : 1;                                               \
            const bool fwd = (MODE == RFFT_FWD || MODE == CHEB_FWD || MODE == CFFT_FWD);              \
            const size_t ng = (size_t)pl->dev.N * outer * inner * cx, nc = (size_t)pl->dev.M * outer * inner * cx; \
            if (int st0 = resolve_alias(&a, b, fwd ? ng : nc, fwd ? nc : ng, as_stream(stream))) return st0; \
        }                                                                                             \
        return launch<MODE>(pl, a, b, outer, inner, stream);                                          \
    }

DDH_FFT_ENTRY(ddh_rfft_forward, K_RFFT, RFFT_FWD)
DDH_FFT_ENTRY(ddh_rfft_backward, K_RFFT, RFFT_BWD)
int ddh_rfft_backward_deriv(ddh_handle plan, const double *c, double *g, long outer, long inner, double dscale,
                            void *stream) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_RFFT) return fail("ddh_rfft_backward_deriv: plan is of a different transform kind");
    if (outer <= 0 || inner <= 0) return 0;
    if (int st0 = resolve_alias(&c, g, (size_t)pl->dev.M * outer * inner, (size_t)pl->dev.N * outer * inner,
                                as_stream(stream)))
        return st0;
    return launch<RFFT_BWD>(pl, c, g, outer, inner, stream, dscale);
}
int ddh_rfft_backward_dual(ddh_handle plan, const double *c, double *g, double *g_deriv, long outer, long inner,
                           double dscale, void *stream) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_RFFT) return fail("ddh_rfft_backward_dual: plan is of a different transform kind");
    if (outer <= 0 || inner <= 0) return 0;
    if (!g || !g_deriv || g == g_deriv) return fail("ddh_rfft_backward_dual: two distinct outputs required");
    const size_t nc = (size_t)pl->dev.M * outer * inner, ng = (size_t)pl->dev.N * outer * inner;
    const char *c0 = (const char *)c, *c1 = c0 + nc * sizeof(double);
    for (const double *o : {(const double *)g, (const double *)g_deriv}) {
        const char *o0 = (const char *)o, *o1 = o0 + ng * sizeof(double);
        if (!(c1 <= o0 || o1 <= c0)) return fail("ddh_rfft_backward_dual: the outputs must not overlap the input");
    }
    return launch<RFFT_BWD>(pl, c, g, outer, inner, stream, 0.0, g_deriv, dscale);
}
int ddh_cheb_backward_dual(ddh_handle plan, const double *c, double *g, double *g_deriv, const double *dvec, long outer,
                           long inner, void *stream) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_CHEB) return fail("ddh_cheb_backward_dual: plan is of a different transform kind");
    if (pl->dev.nbands == 0)
        return fail("ddh_cheb_backward_dual: the plan must be that of the derivative's basis (with conversion bands)");
    if (outer <= 0 || inner <= 0) return 0;
    if (!g || !g_deriv || g == g_deriv || !dvec) return fail("ddh_cheb_backward_dual: two distinct outputs and dvec required");
    const size_t nc = (size_t)pl->dev.M * outer * inner, ng = (size_t)pl->dev.N * outer * inner;
    const char *c0 = (const char *)c, *c1 = c0 + nc * sizeof(double);
    for (const double *o : {(const double *)g, (const double *)g_deriv}) {
        const char *o0 = (const char *)o, *o1 = o0 + ng * sizeof(double);
        if (!(c1 <= o0 || o1 <= c0)) return fail("ddh_cheb_backward_dual: the outputs must not overlap the input");
    }
    return launch<CHEB_BWD>(pl, c, g, outer, inner, stream, 0.0, g_deriv, 0.0, dvec);
}
DDH_FFT_ENTRY(ddh_cheb_forward, K_CHEB, CHEB_FWD)
int ddh_fft_wave_launches(long *count) {
    if (!count) return fail("ddh_fft_wave_launches: null pointer");
    *count = g_wave_launches;
    return 0;
}
int ddh_fft_set_stage_layout(ddh_handle plan, long value) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_CHEB && pl->tkind != K_RFFT) return fail("ddh_fft_set_stage_layout: Chebyshev or real-Fourier plans");
    if (value < 0 || value > 0x7fffffffL) return fail("ddh_fft_set_stage_layout: bad value");
    pl->dev.xb = (unsigned)value;
    return 0;
}

int ddh_fft_set_stage_block(ddh_handle plan, int rows) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_RFFT) return fail("ddh_fft_set_stage_block: real-Fourier plans");
    if (!(rows == 0 || rows == 64 || rows == 128 || rows == 256)) return fail("ddh_fft_set_stage_block: 64, 128 or 256 rows (0 = 64)");
    pl->dev.xbB = (unsigned)rows;
    return 0;
}

/* The coefficient-side array of this Chebyshev plan's following STRIDED-axis transforms (both directions) has rows
 * [nx][ny = row_len] stored tile-major ([kx / 8][ky / 8][kx % 8][ky % 8]: the state vector of a pack with
 * ddh_pencil_set_state_tiled); row_len = 0: natural again.  band_rows != 0: the array is a block of rows of a kx-band-major
 * state vector of band_rows rows, [kx / 8][band_rows][ky / 8][8][8], and the pointer handed to the transform is that of the
 * block's first row in band 0 (state + row0 * 8 * row_len).  Like the stage layout it is a property of the next launches, set
 * by the caller before each of them; transforms that cannot take the wave kernels fail instead of reading another layout. */
int ddh_fft_set_coeff_tiled(ddh_handle plan, long row_len, long band_rows) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_CHEB) return fail("ddh_fft_set_coeff_tiled: Chebyshev plans");
    if (row_len < 0 || (row_len & 7) || row_len > 0x7fffffffL) return fail("ddh_fft_set_coeff_tiled: row_len a multiple of 8");
    if (band_rows < 0 || (band_rows && !row_len)) return fail("ddh_fft_set_coeff_tiled: band_rows needs row_len");
    pl->dev.ctile_nseg = (unsigned)(row_len / 8);
    pl->dev.cband = (unsigned long)band_rows * 8UL * (unsigned long)row_len;
    return 0;
}

/* Real-Fourier plan with the blocked stage layout (ddh_fft_set_stage_layout / _block): the next strided transforms cover
 * planes z0 .. z0 + nplanes of every component only -- the array on the other side holds nplanes planes per component,
 * outer = components x nplanes.  nplanes = 0: all planes again.  The grid stage of a sharded run in windows of z planes,
 * pipelined against windowed exchanges (ddh_comm_alltoall_part). */
int ddh_fft_set_stage_window(ddh_handle plan, int z0, int nplanes) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_RFFT) return fail("ddh_fft_set_stage_window: real-Fourier plans");
    if (z0 < 0 || nplanes < 0) return fail("ddh_fft_set_stage_window: bad window");
    pl->dev.xbw0 = (unsigned)z0;
    pl->dev.xbwn = (unsigned)nplanes;
    return 0;
}

int ddh_cheb_forward_tiled(ddh_handle plan, const double *g, double *c, long outer, long inner, long row_len, void *stream) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_CHEB) return fail("ddh_cheb_forward_tiled: plan is of a different transform kind");
    if (row_len < 8 || (row_len & 7) || inner % row_len || (inner / row_len) % 8)
        return fail("ddh_cheb_forward_tiled: inner = nx * ny with nx and ny = row_len multiples of 8");
    if (g == c) return fail("ddh_cheb_forward_tiled: in-place unsupported");
    if (outer <= 0 || inner <= 0) return 0;
    const unsigned saved = pl->dev.ctile_nseg;
    const unsigned long saved_band = pl->dev.cband;
    pl->dev.ctile_nseg = (unsigned)(row_len / 8);
    pl->dev.cband = 0;
    const int st = launch<CHEB_FWD>(pl, g, c, outer, inner, stream);
    pl->dev.ctile_nseg = saved;
    pl->dev.cband = saved_band;
    return st;
}
DDH_FFT_ENTRY(ddh_cheb_backward, K_CHEB, CHEB_BWD)
DDH_FFT_ENTRY(ddh_cfft_forward, K_CFFT, CFFT_FWD)
DDH_FFT_ENTRY(ddh_cfft_backward, K_CFFT, CFFT_BWD)

int ddh_rfft_bilinear_fused(ddh_handle plan, int na, const double *const *a_h, const double *a_dscale_h, int nb,
                            const double *const *b_h, const double *b_dscale_h, int nc, double *const *out_h,
                            long nlines, int nterms, const int *ic_h, const int *ia_h, const int *ib_h,
                            const double *coef_h, void *stream) {
    FftPlan *pl = (FftPlan *)lookup_handle(plan, H_FFT);
    if (!pl) return -1;
    if (pl->tkind != K_RFFT) return fail("rfft_bilinear_fused: needs an rfft plan");
    if (na < 1 || na > FUSED_NA || nb < 1 || nb > FUSED_NB || nc < 1 || nc > FUSED_NC || nterms < 1 ||
        nterms > FUSED_TERMS)
        return fail("rfft_bilinear_fused: operand counts out of range (na<=3, nb<=12, nc<=4, terms<=32)");
    if (nlines <= 0) return 0;
    FftDev d = pl->dev;
    const int T = FUSED_T;
    // A WIDER INTERNAL GRID where this grid size has no wave kernel (round 6).  The stage maps coefficient lines to
    // coefficient lines; its grid never reaches memory.  Bilinear products of lines with modes <= K are free of aliasing in
    // the retained modes on ANY grid of N' > 3 K points, so the stage may run on the smallest N' >= N the wave kernels are
    // instantiated for (N' = 128 C) and returns the same coefficients up to round-off -- e.g. 192-point lines (128 modes) on
    // 256 points, 576-point lines (384 modes) on 768 -- instead of the workgroup-per-line-pair kernel below (0.05-0.12 of the
    // HBM rate, profiles/r6_offsize_configs.txt).  DDH_FUSED_WIDER=0 keeps the plan's own size.
    static const bool wider = !(getenv("DDH_FUSED_WIDER") && atoi(getenv("DDH_FUSED_WIDER")) == 0);
    if (wider && !gridwave_supported(d)) {
        if (!pl->fused_alt) {
            static const int sizes[] = {256, 384, 512, 768, 1024};
            for (int Np : sizes) {
                FftDev t = d;
                t.N = Np;
                if (Np >= d.N && Np > 3 * d.K && gridwave_supported(t)) {
                    ddh_handle h = 0;
                    if (int st = make_plan(&h, K_RFFT, Np, d.M, 0, nullptr, nullptr)) return st;
                    pl->fused_alt = h;
                    break;
                }
            }
        }
        if (pl->fused_alt) {
            FftPlan *alt = (FftPlan *)lookup_handle(pl->fused_alt, H_FFT);
            if (!alt) return -1;
            d = alt->dev;
        }
    }
    if ((long)d.N > 6L * T) return fail("rfft_bilinear_fused: axis too long for the fused kernel");
    // operands transformed together: 3 when the points fit 3 per thread and 12 staged values per thread
    static const int envG = getenv("DDH_FUSED_G") ? atoi(getenv("DDH_FUSED_G")) : 0;
    int G = ((long)d.N <= 3L * T && 3L * d.N <= 12L * T) ? 3 : 1;
    if (envG == 1) G = 1;
    const bool wave_path = gridwave_supported(d);     // one wavefront per line (ddh_gridwave.hip)
    if (wave_path) G = 1;
    d.B = 1;
    d.fdB.set(1u);
    int order[FUSED_TERMS];
    for (int t = 0; t < nterms; ++t) {
        if (ia_h[t] < 0 || ia_h[t] >= na || ib_h[t] < 0 || ib_h[t] >= nb || ic_h[t] < 0 || ic_h[t] >= nc)
            return fail("rfft_bilinear_fused: term index out of range");
        order[t] = t;
    }
    std::stable_sort(order, order + nterms, [&](int x, int y) {
        return ic_h[x] != ic_h[y] ? ic_h[x] < ic_h[y] : ib_h[x] < ib_h[y];
    });
    // load sequence: the `a` operands, then per result its `b` operands (one load per (ic, ib) group);
    // batches of up to G consecutive loads that belong to the same result share one FFT
    FusedArgs f;
    memset(&f, 0, sizeof(f));
    f.na = na;
    int nl = 0, nbt = 0;
    int load_ic[FUSED_LOADS];
    for (int i = 0; i < na; ++i) {
        f.src[nl] = a_h[i];
        f.dscale[nl] = a_dscale_h ? a_dscale_h[i] : 0.0;
        f.tbeg[nl] = 0;
        load_ic[nl] = -1;
        ++nl;
    }
    bool has_terms[FUSED_NC] = {false, false, false, false};
    for (int t = 0; t < nterms; ++t) {
        const int o = order[t];
        const bool fresh = (t == 0) || ic_h[o] != ic_h[order[t - 1]] || ib_h[o] != ib_h[order[t - 1]];
        if (fresh) {
            f.src[nl] = b_h[ib_h[o]];
            f.dscale[nl] = b_dscale_h ? b_dscale_h[ib_h[o]] : 0.0;
            f.tbeg[nl] = (short)t;
            load_ic[nl] = ic_h[o];
            ++nl;
        }
        f.coef[t] = coef_h[o];
        f.ia[t] = (signed char)ia_h[o];
        has_terms[ic_h[o]] = true;
    }
    for (int l = nl; l <= FUSED_LOADS; ++l) f.tbeg[l] = (short)nterms;
    for (int l = 0; l < nl;) {
        int e = l + 1;
        while (e < nl && e - l < G && load_ic[e] == load_ic[l]) ++e;
        f.bbeg[nbt] = (short)l;
        const bool last_of_result = load_ic[l] >= 0 && (e == nl || load_ic[e] != load_ic[l]);
        f.flush[nbt] = (signed char)(last_of_result ? load_ic[l] : -1);
        ++nbt;
        l = e;
    }
    for (int i = nbt; i <= FUSED_LOADS; ++i) f.bbeg[i] = (short)nl;
    f.nbatch = nbt;
    for (int i = 0; i < nc; ++i) {
        f.out[i] = out_h[i];
        if (!has_terms[i])
            DDH_HIP(hipMemsetAsync(out_h[i], 0, (size_t)nlines * d.M * sizeof(double), as_stream(stream)));
    }
    if (wave_path) return launch_gridwave(d, f, nlines, as_stream(stream));
    const long npairs = (nlines + 1) / 2;
    if ((unsigned long)npairs > 0x7fffffffUL) return fail("rfft_bilinear_fused: grid too large");
    const size_t lds = ((size_t)d.ld * G + (size_t)tw_entries(d.N, d.twdirect)) * sizeof(double2);
    const dim3 grid((unsigned)npairs), block(T);
    hipStream_t st = as_stream(stream);
    // (four workgroups per CU instead of two were tried in round 6: 128 registers spill 350-550 bytes per lane and the
    //  kernel runs 1.6-2.9 x slower)
    if ((long)d.N <= 3L * T) {
        if (G == 3) hipLaunchKernelGGL((fused_rfft_bilinear_kernel<3, 3>), grid, block, lds, st, d, f, nlines, npairs);
        else hipLaunchKernelGGL((fused_rfft_bilinear_kernel<3, 1>), grid, block, lds, st, d, f, nlines, npairs);
    } else {
        hipLaunchKernelGGL((fused_rfft_bilinear_kernel<6, 1>), grid, block, lds, st, d, f, nlines, npairs);
    }
    DDH_HIP(hipGetLastError());
    return 0;
}

int ddh_plan_mmt(ddh_handle *plan, int n_out, int n_in, const double *mat_h) {
    if (n_out < 1 || n_in < 1) return fail("plan_mmt: sizes must be positive");
    MmtPlan *pl = new MmtPlan();
    pl->kind = H_MMT;
    pl->n_out = n_out;
    pl->n_in = n_in;
    size_t bytes = (size_t)n_out * n_in * sizeof(double);
    if (check_hip(hipMalloc((void **)&pl->d_mat, bytes), "hipMalloc") ||
        check_hip(hipMemcpy(pl->d_mat, mat_h, bytes, hipMemcpyHostToDevice), "hipMemcpy")) {
        delete pl;
        return -2;
    }
    *plan = register_handle(pl);
    return 0;
}

int ddh_mmt_apply(ddh_handle plan, const double *in, double *out, long outer, long inner, void *stream) {
    MmtPlan *pl = (MmtPlan *)lookup_handle(plan, H_MMT);
    if (!pl) return -1;
    if (outer <= 0 || inner <= 0) return 0;
    if (int st0 = resolve_alias(&in, out, (size_t)pl->n_in * outer * inner, (size_t)pl->n_out * outer * inner,
                                as_stream(stream)))
        return st0;
    if (outer > 65535) return fail("ddh_mmt_apply: outer too large");
    static const bool no_mfma = getenv("DDH_MMT_NO_MFMA") != nullptr;
    if (inner >= 16 && !no_mfma) {      // enough contiguous columns per row for 16-wide MFMA column tiles
        dim3 g2((unsigned)((inner + MM_N - 1) / MM_N), (unsigned)((pl->n_out + MM_M - 1) / MM_M), (unsigned)outer);
        hipLaunchKernelGGL(mmt_mfma_kernel, g2, dim3(256), 0, as_stream(stream), pl->d_mat, in, out, pl->n_out, pl->n_in,
                           inner);
        DDH_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid((unsigned)((inner + MT_X - 1) / MT_X), (unsigned)((pl->n_out + MT_I - 1) / MT_I), (unsigned)outer);
    hipLaunchKernelGGL(mmt_kernel, grid, dim3(256), 0, as_stream(stream), pl->d_mat, in, out, pl->n_out, pl->n_in,
                       inner);
    DDH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
