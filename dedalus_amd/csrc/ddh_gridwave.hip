// Wave-per-line fused grid stage along the contiguous real-Fourier axis (gfx950).
//
//     out[ic] = forward_rfft( sum_t coef_t * backward_rfft(a[ia_t]) * backward_rfft(b[ib_t]) )
//
// Same contract as fused_rfft_bilinear_kernel in ddh_fft.hip (which stays the general path), replacing
// the reference's backward transforms + DotProduct/MultiplyFields + forward transform along the last
// axis (core/transforms.py:469-565, core/arithmetic.py:666-674, 855-866).
//
// Design for CDNA4: ONE 64-lane wavefront owns ONE real line, so there is no workgroup barrier anywhere
// in the transform loop and waves run completely decoupled (LDS operations of one wave execute in
// order, which is all the synchronisation the exchanges need).  A real line of N = 128*C grid points is
// transformed as a complex FFT of half length H = 64*C (even/odd samples packed as re/im):
//   backward:  spectrum pre-processing at the global load -> radix-8 -> LDS exchange -> radix-8 -> LDS
//              exchange -> radix-C, leaving C complex (= 2C real) grid values per lane in registers;
//   forward:   the mirror image, then Hermitian post-processing from LDS and coalesced 16-B stores.
// The grid-point <-> (lane, register) map is the same for every operand, which is all the point-wise
// products need, so no pass ever sorts.  Twiddles come from an LDS copy of the plan's exp(-2 pi i q/N).
// Exchange layouts keep every ds_read_b128 lane-contiguous (conflict free) and pad the strided side
// (the writes, whose cost is the data transfer to the LDS, tolerate the residual 2-way conflicts).
#include "ddh_fft_dev.h"

#include <algorithm>
#include <cstdlib>

namespace ddh {

namespace gw {


constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int C>
struct GW {
    static constexpr int H = 64 * C;        // complex FFT length
    static constexpr int N = 128 * C;       // real grid points
    static constexpr int NB = 8 * C;        // butterflies (active lanes) of the radix-8 passes
    static constexpr int S1 = NB + (C % 8); // exchange 1 stride: >= NB and == C (mod 8)
    static constexpr int S2 = 65;           // backward exchange 2: [c][lane]
    static constexpr int S2F = NB + 1;      // forward exchange 2: [n2][n1*C + c], odd stride
    static constexpr int LDW = cmax(cmax(8 * S1, C * S2), cmax(8 * S2F, H + 1));   // double2 per wave
    // twiddle tables in LDS (forward sign; every lookup is lane base + compile-time offset):
    static constexpr int T_N = 0;           // [k]          exp(-2 pi i k / N),                   k < H
    // [n2][n1*C+c] exp(-2 pi i ((n1 + 8 n2) c) / H), row stride STR: the backward reads are lane-contiguous,
    // STR keeps the forward reads (lane = n2 + 8 n1) at most 2-way conflicting
    static constexpr int STR = C == 2 ? 20 : C == 3 ? 28 : C == 4 ? 33 : C == 6 ? 52 : 65;
    static constexpr int T_2 = H;
    static constexpr int T_1 = H + 8 * STR; // [n1][b]      exp(-2 pi i n1 b / 64) (symmetric)
    static constexpr int TW = T_1 + 64;     // table entries
};

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in program order; only the compiler has to be kept from
    // moving accesses across the exchange.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double2 conj2(double2 a) { return make_double2(a.x, -a.y); }

// (cos, msin) coefficient pair of wavenumber k -> TWICE the spectrum value X[k] of the unnormalised c2r
// transform (X[k] = (cos + i msin) / 2 for k > 0, X[0] = cos): the factor 1/2 of every operand is folded into
// the term coefficients on the host.  DERIV: differentiated, X' = i kappa X with kappa = dscale * k.
template <bool DERIV, bool MAYBE_ZERO>
__device__ __forceinline__ double2 coef_to_spec2(double2 cs, int k, double dscale) {
    if (DERIV) {
        const double kap = dscale * (double)k;
        return make_double2(-kap * cs.y, kap * cs.x);
    }
    if (MAYBE_ZERO && k == 0) return make_double2(2.0 * cs.x, 0.0);
    return cs;
}

// staged coefficient pairs of one line: pair k = lane + 64 t
template <int NT>
struct LineLoads {
    double2 x[NT];
};

typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) d2v *gptr;

__device__ __forceinline__ void gstore16(double *p, double2 v) {
    d2v r;
    r.x = v.x;
    r.y = v.y;
    *(gptr)(p) = r;
}

// Buffer descriptor of one coefficient line, valid for the retained wavenumbers k <= K only: loads of
// anything beyond return zero from the hardware range check, and every load address is one shared lane
// offset plus an instruction immediate.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t line_rsrc(const double *line, int K) {
    const unsigned long long a = (unsigned long long)line;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *base = (void *)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (K + 1) * 16, 0x00020000);
}
__device__ __forceinline__ double2 bload16(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u4v q = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    double2 v;
    v.x = __hiloint2double((int)q.y, (int)q.x);
    v.y = __hiloint2double((int)q.w, (int)q.z);
    return v;
}

// issue the loads of one operand line: 1 KiB per wave instruction, fully coalesced
template <int NT>
__device__ __forceinline__ void issue_loads(LineLoads<NT> &ld, const double *line, int lane, int K) {
    const __amdgpu_buffer_rsrc_t r = line_rsrc(line, K);
#pragma unroll
    for (int t = 0; t < NT; ++t) ld.x[t] = bload16(r, 16 * lane + 1024 * t);
}

// spectrum pre-processing for radix-8 butterfly j = lane (< NB): its inputs Z[j + NB a] (times 2),
//   Z = (X[k] + conj X[H-k]) + i w^k (X[k] - conj X[H-k]),  w = exp(+2 pi i / N)
template <int C, int NT, bool DERIV>
__device__ __forceinline__ void build_z_from_lds(double dscale, const double2 *wb, const double2 *tw, int lane,
                                                 double2 *v) {
    using G = GW<C>;
    constexpr int KM = 64 * NT - 1;                                   // largest loaded wavenumber
    constexpr int ND = (KM / G::NB + 1) < 8 ? (KM / G::NB + 1) : 8;   // direct inputs: a < ND
    constexpr int AM0n = G::H - KM - (G::NB - 1);                     // mirror inputs: a >= AM0
    constexpr int AM0 = AM0n <= 0 ? 0 : (AM0n + G::NB - 1) / G::NB;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int k = lane + G::NB * a;
        double2 xd = make_double2(0.0, 0.0), xm = xd;
        if (a < ND) {
            if (a == 0) xd = coef_to_spec2<DERIV, true>(wb[k], k, dscale);
            else xd = coef_to_spec2<DERIV, false>(wb[k], k, dscale);
        }
        if (a >= AM0) {
            xm = coef_to_spec2<DERIV, false>(wb[G::H - k], G::H - k, dscale);
            if (a == 0 && k == 0) xm = make_double2(0.0, 0.0);        // X[H] is not stored (and is zero)
        }
        const double2 w = conj2(tw[G::T_N + k]);
        if (a < ND && a >= AM0) {
            const double2 A = make_double2(xd.x + xm.x, xd.y - xm.y);
            const double2 B = make_double2(xd.x - xm.x, xd.y + xm.y);
            const double2 wB = cmul(w, B);
            v[a] = make_double2(A.x - wB.y, A.y + wB.x);
        } else if (a < ND) {                                           // Z = X + i w X
            const double2 wB = cmul(w, xd);
            v[a] = make_double2(xd.x - wB.y, xd.y + wB.x);
        } else if (a >= AM0) {                                         // Z = conj Xm - i w conj Xm
            const double2 cm = make_double2(xm.x, -xm.y);
            const double2 wB = cmul(w, cm);
            v[a] = make_double2(cm.x + wB.y, cm.y - wB.x);
        } else {
            v[a] = make_double2(0.0, 0.0);
        }
    }
}

// Staged pairs -> LDS in natural order (zeros beyond the loaded range), then the pre-processing
template <int C, int NT>
__device__ __forceinline__ void build_z(const LineLoads<NT> &ld, double dscale, double2 *wb, const double2 *tw,
                                        int lane, double2 *v, int dbg = 0) {
    using G = GW<C>;
    // (blocks beyond the largest index the pruned pre-processing reads are not written: an LDS store costs 13 cycles)
    constexpr int KM = 64 * NT - 1;
    constexpr int NDc = (KM / G::NB + 1) < 8 ? (KM / G::NB + 1) : 8;
    constexpr int AM0n = G::H - KM - (G::NB - 1);
    constexpr int AM0c = AM0n <= 0 ? 0 : (AM0n + G::NB - 1) / G::NB;
    constexpr int RMAX = cmax(NDc * G::NB - 1, G::H - AM0c * G::NB);     // largest staged index read below
    constexpr int TWR = (RMAX / 64 + 1) < C ? (RMAX / 64 + 1) : C;
#pragma unroll
    for (int t = 0; t < TWR; ++t)
        if (!(dbg & 64)) wb[lane + 64 * t] = (t < NT) ? ld.x[t] : make_double2(0.0, 0.0);
    wave_sync();
    if (dbg & 128) {                                       // timing ablation: no LDS reads
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = make_double2(ld.x[a % NT].x + a, ld.x[a % NT].y * dscale);
        return;
    }
    if (lane < G::NB) {
        if (dscale != 0.0) build_z_from_lds<C, NT, true>(dscale, wb, tw, lane, v);     // wave-uniform branch
        else build_z_from_lds<C, NT, false>(dscale, wb, tw, lane, v);
    }
}

// backward transform from the pre-processed spectrum v: result g[n3] = x[2n] + i x[2n+1],
// n = (n1 + 8 n2) + 64 n3, lane = n2 + 8 n1
// t1r / t2r (TWREG): the lane's 7 + 8 twiddles of the two radix-8 passes held in registers for the whole kernel instead of
// being read from the LDS tables in every transform (15 of the 53 LDS reads of a backward transform; the kernel is bound
// by the LDS pipe and runs at 2 waves per SIMD whatever it does with the 60 registers)
template <int C, bool TWREG = false>
__device__ __forceinline__ void backward_line(double2 *v, double2 *wb, const double2 *tw, int lane, double2 *g,
                                              int dbg, const double2 *t1r = nullptr, const double2 *t2r = nullptr) {
    using G = GW<C>;
    const int b1 = lane / C, c1 = lane - b1 * C;     // pass 1: butterfly j = lane = b*C + c; pass 2: lr = n1*C + c
    if (lane < G::NB) {
        if (!(dbg & 1)) butterfly<8>(v, +1);
        // twiddle W64^(n1 b), then exchange 1: [b][n1*C + c]
        if (!(dbg & 64)) wb[b1 * G::S1 + c1] = v[0];
#pragma unroll
        for (int n1 = 1; n1 < 8; ++n1) {
            const double2 w = conj2(TWREG ? t1r[n1 - 1] : tw[G::T_1 + n1 * 8 + b1]);
            if (dbg & 64) v[n1] = cmul(v[n1], w); else
            wb[b1 * G::S1 + n1 * C + c1] = cmul(v[n1], w);
        }
    }
    wave_sync();
    if (lane < G::NB) {
#pragma unroll
        for (int b = 0; b < 8; ++b) if (!(dbg & 128)) v[b] = wb[b * G::S1 + lane];
        if (!(dbg & 1)) butterfly<8>(v, +1);
        // twiddle omega^((n1 + 8 n2) c), omega = exp(2 pi i / H); exchange 2: [c][n2 + 8 n1]
        const int n1 = b1, c = c1;
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const double2 w = conj2(TWREG ? t2r[n2] : tw[G::T_2 + n2 * G::STR + lane]);
            if (dbg & 64) v[n2] = cmul(v[n2], w); else
            wb[c * G::S2 + n2 + 8 * n1] = cmul(v[n2], w);
        }
    }
    wave_sync();
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = (dbg & 128) ? v[c] : wb[c * G::S2 + lane];
    if (!(dbg & 1)) butterfly<C>(g, +1);
}

// forward transform of the grid values g (same lane/register map) and store of the coefficient line
template <int C, bool TWREG = false>
__device__ __forceinline__ void forward_line(double2 *g, double2 *wb, const double2 *tw, int lane, double *dst,
                                             int M, int K, int dbg, const double2 *t1r = nullptr) {
    using G = GW<C>;
    const int n2L = lane & 7, n1L = lane >> 3;
    if (!(dbg & 1)) butterfly<C>(g, -1);
    {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double2 w = tw[G::T_2 + n2L * G::STR + n1L * C + c];
            if (dbg & 64) g[c] = (c == 0) ? g[0] : cmul(g[c], w); else
            wb[n2L * G::S2F + n1L * C + c] = (c == 0) ? g[0] : cmul(g[c], w);
        }
    }
    wave_sync();
    const int n1 = lane / C, c1 = lane - n1 * C;
    double2 v[8];
    if (lane < G::NB) {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[n2] = (dbg & 128) ? g[n2 % C] : wb[n2 * G::S2F + lane];
        if (!(dbg & 1)) butterfly<8>(v, -1);
        // twiddle conj W64^(n1 b); exchange 1: [n1][b*C + c]
        if (!(dbg & 64)) wb[n1 * G::S1 + c1] = v[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) {
            const double2 w = TWREG ? t1r[b - 1] : tw[G::T_1 + b * 8 + n1];       // (the table is symmetric)
            if (dbg & 64) v[b] = cmul(v[b], w); else
            wb[n1 * G::S1 + b * C + c1] = cmul(v[b], w);
        }
    }
    wave_sync();
    if (lane < G::NB) {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (!(dbg & 128)) v[q] = wb[q * G::S1 + lane];
        if (!(dbg & 1)) butterfly<8>(v, -1);
#pragma unroll
        for (int a = 0; a < 8; ++a) if (!(dbg & 64)) wb[lane + G::NB * a] = v[a];     // natural order Zf[k]
    }
    wave_sync();
    // Y[k] = E + exp(-2 pi i k / N) O,  E = (Zf[k] + conj Zf[H-k]) / 2,  O = -i (Zf[k] - conj Zf[H-k]) / 2
    const double invN = 1.0 / (double)G::N;
    const int Mh = M >> 1;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int k = lane + 64 * t;
        if (64 * t < Mh && k < Mh) {
            double2 out = make_double2(0.0, 0.0);
            if (k == 0) {
                const double2 z = wb[0];
                out.x = (z.x + z.y) * invN;
            } else if (k <= K) {
                const double2 z = (dbg & 128) ? v[t] : wb[k], zm = (dbg & 128) ? v[7 - t] : wb[G::H - k];
                const double2 E = make_double2(0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y));
                const double2 D = make_double2(0.5 * (z.x - zm.x), 0.5 * (z.y + zm.y));
                const double2 O = make_double2(D.y, -D.x);
                const double2 wO = cmul(tw[G::T_N + k], O);
                out.x = 2.0 * invN * (E.x + wO.x);
                out.y = 2.0 * invN * (E.y + wO.y);
            }
            if (!(dbg & 16)) gstore16(dst + 2 * k, out);
        }
    }
    wave_sync();
}

// WAVES lines (wavefronts) per workgroup; DBG compiles the timing-ablation switches in (FftDev::dbg bits:
// 1 no butterfly math, 4 no global loads, 16 no global stores, 32 no forward transforms)
template <int C, int NT, int WAVES, bool DBG, int OCC, bool TWREG = false>
__global__ void __launch_bounds__(64 * WAVES, OCC)
gridwave_bilinear_kernel(FftDev p, FusedArgs f, long nlines) {
    constexpr int GW_T = 64 * WAVES, GW_WAVES = WAVES;
    const int dbg = DBG ? p.dbg : 0;
    using G = GW<C>;
    extern __shared__ double2 lds[];
    double2 *tw = lds;                                   // twiddle tables (GW<C>::T_*)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2 *wb = lds + G::TW + wave * G::LDW;          // this wave's exchange buffer
    for (int i = tid; i < G::TW; i += GW_T) {
        int q;                                           // index into the plan's exp(-2 pi i q / N)
        if (i < G::H) {
            q = i;
        } else if (i < G::T_1) {
            const int r = i - G::T_2, n2 = r / G::STR, lr = r - n2 * G::STR, n1 = lr / C, c = lr - n1 * C;
            q = (lr < G::NB) ? 2 * (((n1 + 8 * n2) * c) % G::H) : 0;
        } else {
            const int r = i - G::T_1;
            q = (((r >> 3) * (r & 7)) & 63) * (2 * C);
        }
        tw[i] = p.tw[q];
    }
    // kernel arguments indexed at run time go through LDS
    __shared__ const double *s_src[FUSED_LOADS];
    __shared__ double s_dscale[FUSED_LOADS];
    __shared__ double *s_out[FUSED_NC];
    __shared__ double s_coef[FUSED_TERMS];
    __shared__ short s_tbeg[FUSED_LOADS + 1];
    __shared__ signed char s_flush[FUSED_LOADS], s_ia[FUSED_TERMS];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < FUSED_LOADS; ++i) {
            s_src[i] = f.src[i];
            s_dscale[i] = f.dscale[i];
            s_tbeg[i] = f.tbeg[i];
            s_flush[i] = f.flush[i];
        }
        s_tbeg[FUSED_LOADS] = f.tbeg[FUSED_LOADS];
#pragma unroll
        for (int i = 0; i < FUSED_TERMS; ++i) {
            s_coef[i] = f.coef[i];
            s_ia[i] = f.ia[i];
        }
#pragma unroll
        for (int i = 0; i < FUSED_NC; ++i) s_out[i] = f.out[i];
    }
    __syncthreads();                                     // the only workgroup barrier
    const int M = p.M, K = p.K;
    const int na = f.na, nloads = f.nbatch;              // host builds one load per batch for this kernel
    double2 t1r[7], t2r[8];
    if (TWREG) {
        const int lc = lane < G::NB ? lane : 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) t1r[i] = tw[G::T_1 + (i + 1) * 8 + lc / C];
#pragma unroll
        for (int i = 0; i < 8; ++i) t2r[i] = tw[G::T_2 + i * G::STR + lc];
    }
    // a wave takes several lines (grid = lines / (waves x DDH_GW_LPW)): the table fill, the argument staging and the
    // register twiddles are paid once per 8 lines instead of once per line (6.98 -> 6.85 ms at 768 x 384 lines of 768)
#pragma unroll 1
  for (long line = (long)blockIdx.x * GW_WAVES + wave; line < nlines; line += (long)gridDim.x * GW_WAVES) {
    const long off = line * (long)M;

    LineLoads<NT> ld;
    if (!(dbg & 4)) issue_loads<NT>(ld, s_src[0] + off, lane, K);
    else
#pragma unroll
        for (int t = 0; t < NT; ++t) ld.x[t] = make_double2(1.0, 0.5);
    // One backward transform: operand l (already staged in ld) -> grid values g; prefetches operand l + 1.
    // Lane-derived addresses and constants are re-derived every time (a handful of integer operations)
    // instead of being hoisted out of the loops, where they would pin ~60 registers.
    auto backward = [&](int l, double2 *g) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        double2 v[8];
        build_z<C, NT>(ld, s_dscale[l], wb, tw, ln, v, dbg);
        if (l + 1 < nloads && !(dbg & 4)) issue_loads<NT>(ld, s_src[l + 1] + off, ln, K);
        backward_line<C, TWREG>(v, wb, tw, ln, g, dbg, t1r, t2r);
    };
    // the `a` operands stay in registers (twice their grid values, like every transformed operand)
    double2 areg[FUSED_NA][C];
#pragma unroll
    for (int ia = 0; ia < FUSED_NA; ++ia) {
        if (ia < na) {
            backward(ia, areg[ia]);
        } else {
#pragma unroll
            for (int i = 0; i < C; ++i) areg[ia][i] = make_double2(0.0, 0.0);
        }
    }
    double2 acc[C];
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int l = na; l < nloads; ++l) {
        double2 g[C];
        backward(l, g);
        const int t1 = s_tbeg[l + 1];
#pragma unroll 1
        for (int t = s_tbeg[l]; t < t1; ++t) {
            const double cf = s_coef[t];                 // includes the 1/4 of the two doubled operands
            const int tia = s_ia[t];                     // wave-uniform
#pragma unroll
            for (int ia = 0; ia < FUSED_NA; ++ia) {
                if (tia == ia) {
#pragma unroll
                    for (int i = 0; i < C; ++i) {
                        // packed even/odd samples: real parts multiply real parts, imaginary parts imaginary parts
                        acc[i].x += (cf * g[i].x) * areg[ia][i].x;
                        acc[i].y += (cf * g[i].y) * areg[ia][i].y;
                    }
                }
            }
        }
        const int oc = s_flush[l];
        if (oc >= 0 && !(dbg & 32)) {
            int lf = lane;
            asm volatile("" : "+v"(lf));
            forward_line<C, TWREG>(acc, wb, tw, lf, s_out[oc] + off, M, K, dbg, t1r);
#pragma unroll
            for (int i = 0; i < C; ++i) acc[i] = make_double2(0.0, 0.0);
        }
    }
  }
}

template <int C, int WAVES, int OCC>
int launch_cw(const FftDev &d, const FusedArgs &f_in, long nlines, hipStream_t st) {
    using G = GW<C>;
    constexpr int NT32 = (2 * C + 2) / 3;                  // 64-pair blocks that hold M/2 = N/3 pairs (3/2 dealiasing)
    // lines per wave: 8 when that still leaves >= 2048 workgroups (8 per CU), fewer for small problems (2-D: a few
    // hundred lines in all)
    static const int lpw_env = getenv("DDH_GW_LPW") ? std::max(1, atoi(getenv("DDH_GW_LPW"))) : 0;
    const long lpw = lpw_env ? lpw_env : std::min<long>(8, std::max<long>(1, nlines / ((long)WAVES * 2048)));
    const long nwg = (nlines + (long)WAVES * lpw - 1) / ((long)WAVES * lpw);
    if ((unsigned long)nwg > 0x7fffffffUL) return fail("rfft_bilinear_fused: grid too large");
    const size_t lds = ((size_t)G::TW + (size_t)WAVES * G::LDW) * sizeof(double2);
    const dim3 grid((unsigned)nwg), block(64 * WAVES);
    const bool narrow = d.K + 1 <= 64 * NT32;
    static const bool twreg = getenv("DDH_GW_TWREG") ? atoi(getenv("DDH_GW_TWREG")) != 0 : true;
    FusedArgs f = f_in;
    for (int t = 0; t < FUSED_TERMS; ++t) f.coef[t] *= 0.25;     // both factors of a term arrive doubled
    if (C == 6 && d.dbg && narrow)                         // timing ablations (tools/bench_fused.py)
        hipLaunchKernelGGL((gridwave_bilinear_kernel<C, NT32, WAVES, C == 6, OCC>), grid, block, lds, st, d, f, nlines);
    else if (narrow && twreg && C == 6 && OCC == 2)
        hipLaunchKernelGGL((gridwave_bilinear_kernel<C, NT32, WAVES, false, OCC, (C == 6 && OCC == 2)>), grid, block, lds, st, d, f,
                           nlines);
    else if (narrow)
        hipLaunchKernelGGL((gridwave_bilinear_kernel<C, NT32, WAVES, false, OCC>), grid, block, lds, st, d, f, nlines);
    else
        hipLaunchKernelGGL((gridwave_bilinear_kernel<C, C, WAVES, false, OCC>), grid, block, lds, st, d, f, nlines);
    DDH_HIP(hipGetLastError());
    return 0;
}

template <int C>
int launch_c(const FftDev &d, const FusedArgs &f, long nlines, hipStream_t st) {
    static const int waves = getenv("DDH_GW_WAVES") ? atoi(getenv("DDH_GW_WAVES")) : 4;
    static const int occ = getenv("DDH_GW_OCC") ? atoi(getenv("DDH_GW_OCC")) : 2;
    if (C == 6 && occ == 3) return launch_cw<C, 4, (C == 6 ? 3 : 2)>(d, f, nlines, st);
    if (waves == 8) return launch_cw<C, 8, 2>(d, f, nlines, st);
    return launch_cw<C, 4, 2>(d, f, nlines, st);
}

}  // namespace gw

using namespace gw;

bool gridwave_supported(const FftDev &d) {
    static const bool off = getenv("DDH_FUSED_OLD") != nullptr;
    if (off) return false;
    if (d.N % 128 != 0 || (d.M & 1) || d.M < 2 || d.M > d.N) return false;
    const int C = d.N / 128;
    return C == 2 || C == 3 || C == 4 || C == 6 || C == 8;
}

bool gridwave2_supported(const FftDev &d);                                       // ddh_gridwave2.hip
int launch_gridwave2(const FftDev &d, const FusedArgs &f, long nlines, hipStream_t st);

// f must have been built with one load per batch
int launch_gridwave(const FftDev &d, const FusedArgs &f, long nlines, hipStream_t st) {
    if (!d.dbg && gridwave2_supported(d)) return launch_gridwave2(d, f, nlines, st);
    switch (d.N / 128) {
        case 2: return launch_c<2>(d, f, nlines, st);
        case 3: return launch_c<3>(d, f, nlines, st);
        case 4: return launch_c<4>(d, f, nlines, st);
        case 6: return launch_c<6>(d, f, nlines, st);
        case 8: return launch_c<8>(d, f, nlines, st);
    }
    return fail("gridwave: unsupported size");
}

}  // namespace ddh
