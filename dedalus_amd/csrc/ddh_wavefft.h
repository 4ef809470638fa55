// Wave-level transforms along a STRIDED axis (gfx950): one 64-lane wavefront owns four line pairs.
//
// A strided line pair is 16 contiguous bytes per row (two neighbouring real lines, packed re/im); four pairs are one
// 64-byte row segment.  Lane l works on pair p = l & 3 and is member q = l >> 2 of the 16 lanes that share that pair's
// N = 16 R point complex FFT, R values per lane:
//
//     k = q + 16 t  ->  [R-point DFT over t, in registers, compile-time twiddles]  ->  twiddle W^(b q)
//       ->  exchange across the four 16-lane rows (q1 = q >> 2)  ->  radix 4  ->  twiddle W16^(a1 q0)
//       ->  exchange among the four lanes q0 = q & 3 of a row   ->  radix 4
//       ->  n = R (q0 + 4 a0) + (R/4) q1 + i,   a0 < 4, i < R/4
//
// Every global access of the wave is a 64-byte row segment per four lanes, both at the load (decimated rows q + 16 t)
// and at the store (the rows of the output map above); the FFT itself touches LDS twice (two in-order exchanges), there
// is no workgroup barrier and the waves of a workgroup share nothing but read-only tables.
//
// The file is plain C++ over four macros, so tests/host_emu compiles the SAME lane code with g++ and runs the 64 lanes
// as threads (WF_SYNC = barrier): index maps and arithmetic are checked on the CPU against the numpy oracle.
//
// Replaces, for the strided z axis, the reference's scipy DCT + separate scale / pad / truncate / conversion passes
// (core/transforms.py:715-902).
#pragma once
#include "ddh_butterfly.h"

#ifdef DDH_HOST_EMU
namespace ddh {
namespace wf {
void emu_barrier();
double emu_shfl_up(double v, int delta, int lane);
}  // namespace wf
}  // namespace ddh
#define WF_SYNC() ::ddh::wf::emu_barrier()
#define WF_SHFL_UP(v, delta, lane) ::ddh::wf::emu_shfl_up((v), (delta), (lane))
#define WF_SCHED_FENCE() \
    do {                 \
    } while (0)
#define WF_OPAQUE_U32(x) \
    do {                 \
    } while (0)
#define WF_OPAQUE_LANE(x) \
    do {                  \
    } while (0)
#else
// LDS operations of one wave execute in program order; only the compiler has to be kept from moving accesses.
#define WF_SYNC()                                              \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
#define WF_SHFL_UP(v, delta, lane) __shfl_up((v), (delta), 64)
// keeps the instruction scheduler from hoisting a whole phase's LDS reads over the previous phase (register pressure)
#define WF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// a wave-uniform value the optimiser must treat as unknown: address arithmetic derived from it stays inside the tile
// loop instead of being hoisted out of it and spilled
#define WF_OPAQUE_U32(x) asm volatile("" : "+s"(x))
// the same for the lane index: everything derived from it (LDS addresses, predicates) is re-derived per tile -- a
// handful of integer operations -- instead of living in ~60 registers across the whole tile loop
#define WF_OPAQUE_LANE(x) asm volatile("" : "+v"(x))
#endif

namespace ddh {
namespace wf {

struct Lane {
    int lane, p, q, q0, q1, l16;
};
DDH_DEV Lane make_lane(int lane) {
    Lane L;
    L.lane = lane;
    L.p = lane & 3;
    L.q = lane >> 2;
    L.q0 = L.q & 3;
    L.q1 = L.q >> 2;
    L.l16 = lane & 15;
    return L;
}

DDH_DEV double2 cscale(double s, double2 a) { return make_double2(s * a.x, s * a.y); }
DDH_DEV double2 conj2(double2 a) { return make_double2(a.x, -a.y); }

// table entry exp(-2 pi i m / N) -> exp(SIGN 2 pi i m / N)
template <int SIGN>
DDH_DEV double2 twid(const double2 *tw, int m) {
    double2 w = tw[m];
    if (SIGN > 0) w.y = -w.y;
    return w;
}

// a * exp(sign 2 pi i m / D) for compile-time m, D in {12, 24}
template <int D>
DDH_DEV double2 mul_root(double2 a, int m, int sign) {
    // cos / sin (2 pi j / 24), j = 0..6 (first quadrant)
    const double c[7] = {1.0, 0.96592582628906828674974319972889737, 0.86602540378443864676372317075293618,
                         0.70710678118654752440084436210484904, 0.5, 0.25881904510252076234889883762404833, 0.0};
    const int j = m * (24 / D) % 24;       // as a 24th root
    if (j == 0) return a;
    if (j == 12) return make_double2(-a.x, -a.y);
    if (j == 6) return muli(a, sign);
    if (j == 18) return muli(a, -sign);
    const int quad = j / 6, r = j % 6;     // angle = quad * 90 deg + r * 15 deg
    double wr = c[r], wi = c[6 - r];
    if (quad == 1) { const double t = wr; wr = -wi; wi = t; }
    else if (quad == 2) { wr = -wr; wi = -wi; }
    else if (quad == 3) { const double t = wr; wr = wi; wi = -t; }
    if (sign < 0) wi = -wi;
    return make_double2(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
}

// R-point DFT of the lane's own values, natural order in and out: v[b] = sum_t v[t] exp(sign 2 pi i b t / R)
template <int R>
DDH_DEV void dft_inlane(double2 *v, int sign);
template <>
DDH_DEV void dft_inlane<4>(double2 *v, int sign) { butterfly<4>(v, sign); }
template <>
DDH_DEV void dft_inlane<8>(double2 *v, int sign) { butterfly<8>(v, sign); }
template <>
DDH_DEV void dft_inlane<16>(double2 *v, int sign) { butterfly<16>(v, sign); }
// R = 3 A: t = t0 + 3 t1, b = A b0 + b1:  W_R^(b t) = W_3^(b0 t0) W_R^(b1 t0) W_A^(b1 t1)
template <int A>
DDH_DEV void dft_inlane_3A(double2 *v, int sign) {
    constexpr int R = 3 * A;
    double2 g[3][A];
#pragma unroll
    for (int t0 = 0; t0 < 3; ++t0) {
#pragma unroll
        for (int t1 = 0; t1 < A; ++t1) g[t0][t1] = v[t0 + 3 * t1];
        butterfly<A>(g[t0], sign);
    }
#pragma unroll
    for (int b1 = 0; b1 < A; ++b1) {
        double2 h[3];
        h[0] = g[0][b1];
        h[1] = mul_root<R>(g[1][b1], b1, sign);
        h[2] = mul_root<R>(g[2][b1], 2 * b1, sign);
        butterfly<3>(h, sign);
#pragma unroll
        for (int b0 = 0; b0 < 3; ++b0) v[A * b0 + b1] = h[b0];
    }
}
template <>
DDH_DEV void dft_inlane<24>(double2 *v, int sign) { dft_inlane_3A<8>(v, sign); }
template <>
DDH_DEV void dft_inlane<12>(double2 *v, int sign) { dft_inlane_3A<4>(v, sign); }

// default number of exchange chunks of wfft<R>: two where R / 4 is even, else three (R = 12) or one (R = 4)
template <int R>
struct WaveCH {
    static constexpr int RQ = R / 4;
    static constexpr int ch = (RQ % 2 == 0) ? 2 : ((RQ % 3 == 0) ? 3 : 1);
};

// LDS elements (double2) of the exchange buffer wfft<R, ., CH> needs
template <int R, int CH>
struct WfftBuf {
    static constexpr int IC = (R / 4) / CH;
    static constexpr int size = IC * 4 * 68;       // >= IC * 256 (first exchange), second exchange padded to 68
};

// The N = 16 R point FFT of the four pairs of a wave.
//   in:  v[t]            = x[q + 16 t]
//   out: v[a0 (R/4) + i] = X[R (q0 + 4 a0) + (R/4) q1 + i],   X[n] = sum_k x[k] exp(SIGN 2 pi i n k / N)
// tw[m] = exp(-2 pi i m / N) (m < N), xb = this wave's exchange buffer (WfftBuf<R, CH>::size elements), exchanged in
// CH chunks so that the buffer stays small.
// TWS: stride of the twiddle table (tw has TWS * 16 R entries: a sub-transform of a longer FFT uses the long table).
template <int R, int SIGN, int CH, int TWS = 1>
DDH_DEV void wfft(double2 (&v)[R], double2 *xb, const double2 *tw, const Lane &L) {
    constexpr int RQ = R / 4, IC = RQ / CH;
    static_assert(R % 4 == 0 && RQ % CH == 0, "R must split into 4 x CH chunks");
    dft_inlane<R>(v, SIGN);
#pragma unroll
    for (int b = 1; b < R; ++b) {
        v[b] = cmul(v[b], twid<SIGN>(tw, TWS * b * L.q));            // (R - 1) * 15 < 16 R
        if ((b & 3) == 3) WF_SCHED_FENCE();                           // at most four table reads in flight
    }
    // exchange 1: lane (q1, q0, p) keeps b in [q1 RQ, (q1 + 1) RQ) and collects it from the four rows
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        WF_SYNC();
#pragma unroll
        for (int il = 0; il < IC; ++il)
#pragma unroll
            for (int q1w = 0; q1w < 4; ++q1w) xb[((il * 4 + q1w) * 4 + L.q1) * 16 + L.l16] = v[q1w * RQ + c * IC + il];
        WF_SYNC();
#pragma unroll
        for (int il = 0; il < IC; ++il)
#pragma unroll
            for (int q1s = 0; q1s < 4; ++q1s) v[q1s * RQ + c * IC + il] = xb[((il * 4 + L.q1) * 4 + q1s) * 16 + L.l16];
    }
#pragma unroll
    for (int i = 0; i < RQ; ++i) dft4_inplace(v[i], v[RQ + i], v[2 * RQ + i], v[3 * RQ + i], SIGN);
#pragma unroll
    for (int a1 = 1; a1 < 4; ++a1) {
        const double2 w = twid<SIGN>(tw, TWS * R * a1 * L.q0);                       // W16^(a1 q0), 9 R < 16 R
#pragma unroll
        for (int i = 0; i < RQ; ++i) v[a1 * RQ + i] = cmul(v[a1 * RQ + i], w);
    }
    // exchange 2: lane (q1, q0, p) keeps a1 = q0 and collects it from the four lanes q0' of its row
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        WF_SYNC();
#pragma unroll
        for (int il = 0; il < IC; ++il)
#pragma unroll
            for (int a1 = 0; a1 < 4; ++a1) xb[(il * 4 + L.q0) * 68 + L.q1 * 16 + a1 * 4 + L.p] = v[a1 * RQ + c * IC + il];
        WF_SYNC();
#pragma unroll
        for (int il = 0; il < IC; ++il)
#pragma unroll
            for (int q0s = 0; q0s < 4; ++q0s) v[q0s * RQ + c * IC + il] = xb[(il * 4 + q0s) * 68 + L.q1 * 16 + L.q0 * 4 + L.p];
    }
#pragma unroll
    for (int i = 0; i < RQ; ++i) dft4_inplace(v[i], v[RQ + i], v[2 * RQ + i], v[3 * RQ + i], SIGN);
}

// ------------------------------------------------------------------------------------------------
// Chebyshev (DCT-II / DCT-III through one complex FFT of the permuted pair, core/transforms.py:715-902)
// ------------------------------------------------------------------------------------------------
struct ChebTabs {
    const double2 *tw;      // [N]  exp(-2 pi i m / N)
    const double2 *half;    // [N]  exp(-i pi k / 2N)
    const double *bands;    // [nbands][M] conversion bands (forward apply)
    const double *bsub;     // [2][M] back-substitution table: 1 / diag, next-band / diag (backward solve)
    const double *dvec;     // [M] derivative superdiagonal (dual backward)
    int M, Mk, nbands, gcd_off;
    int boff1, boff2, boff3;    // offsets of the conversion bands 1..3 (band 0 is the diagonal)
    double fs0, fs1, bs0, bs1;     // normalisations (Appendix A of SURVEY.md; transforms.py:720-746, 823-860)
};

// grid row of FFT position n (Makhoul permutation, N even): n < N/2 -> 2 n, else 2 (N - 1 - n) + 1
DDH_DEV int cheb_row_of(int n, int N) { return (2 * n < N) ? 2 * n : 2 * (N - 1 - n) + 1; }

// Global addressing of a tile: a wave-uniform base pointer plus a 32-bit byte offset per lane (row * rsb + 16 p),
// rsb = bytes between rows.  The host guarantees that a tile's rows span less than 4 GiB.
DDH_DEV double2 gload(const double *base, unsigned off) {
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + off);
}
DDH_DEV void gstore(double *base, unsigned off, double2 v) {
    *reinterpret_cast<double2 *>(reinterpret_cast<char *>(base) + off) = v;
}

// Contiguous axis (CONTIG variants below): the lines of the wave's four pairs are lines 2 p, 2 p + 1 of an array of lines
// lsb bytes apart (global memory, or the wave's LDS staging area), row k of a line is its k-th double.
DDH_DEV double2 cload(const double *base, unsigned off, unsigned lsb) {
    const char *b = reinterpret_cast<const char *>(base) + off;
    return make_double2(*reinterpret_cast<const double *>(b), *reinterpret_cast<const double *>(b + lsb));
}
DDH_DEV void cstore(double *base, unsigned off, unsigned lsb, double2 v) {
    char *b = reinterpret_cast<char *>(base) + off;
    *reinterpret_cast<double *>(b) = v.x;
    *reinterpret_cast<double *>(b + lsb) = v.y;
}
template <int NLC>
DDH_DEV void cheb_bwd_load_contig(double2 (&c)[NLC], const double *src_t, unsigned lsb, bool pvalid, const Lane &L) {
    WF_OPAQUE_U32(lsb);
    const unsigned o0 = (pvalid ? (unsigned)(2 * L.p) * lsb : 0u) + 8u * (unsigned)L.q;      // (a missing pair re-reads pair 0)
#pragma unroll
    for (int t = 0; t < NLC; ++t) c[t] = cload(src_t, o0 + (unsigned)(128 * t), lsb);
}

// loads of one backward tile: c[t] = coefficient row q + 16 t of the lane's pair (zero for a pair beyond the array)
template <int NLC>
DDH_DEV void cheb_bwd_load(double2 (&c)[NLC], const double *src_t, unsigned rsb, bool pvalid, const Lane &L) {
    WF_OPAQUE_U32(rsb);
    // a pair beyond the array re-reads pair 0 of the tile (always present); its results are never stored
    const unsigned o0 = (unsigned)L.q * rsb + (pvalid ? 16u * (unsigned)L.p : 0u);
#pragma unroll
    for (int t = 0; t < NLC; ++t) c[t] = gload(src_t, o0 + (unsigned)(16 * t) * rsb);
}

// DCT-III input Z[k] = conj(h_k) (bs_k c[k] - i bs_(N-k) c[N-k]) for the lane's slots k = q + 16 t.
// FROM_REGS: the lane's own coefficients come from c (and have just been staged in S for the partners);
// otherwise both come from S (after the conversion solve).
template <int R, int NLC, bool FROM_REGS>
DDH_DEV void cheb_bwd_build(double2 (&v)[R], const double2 (&c)[NLC], const double2 *S, const ChebTabs &T, const Lane &L) {
    constexpr int N = 16 * R, Mk = 16 * NLC;
    const double s1 = (L.q & 1) ? -T.bs1 : T.bs1;                   // k, N - k and q have one parity
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int k = L.q + 16 * t, kr = N - k;
        double2 e = make_double2(0.0, 0.0), f = e;
        if (t < NLC) {
            const double se = (t == 0 && L.q == 0) ? T.bs0 : s1;
            const double2 x = FROM_REGS ? c[t < NLC ? t : 0] : S[k * 4 + L.p];
            e = cscale(se, x);
        }
        if (16 * t + 15 > N - Mk) {                                  // some lane's partner row exists
            const bool has = (k > 0) && (kr < Mk);
            const double2 x = S[(has ? kr : 0) * 4 + L.p];
            f = cscale(has ? s1 : 0.0, x);
        }
        const double2 w = make_double2(e.x + f.y, e.y - f.x);       // e - i f
        v[t] = cmul(conj2(T.half[k]), w);
        if ((t & 3) == 3) WF_SCHED_FENCE();                          // at most four slots' LDS reads in flight
    }
}

// first-order back substitution x[k] = bsub0[k] d[k] - bsub1[k] x[k + g] in S, g = gcd_off in {1, 2}: the wave holds
// 4 g chains (pair, residue), 64 / (4 g) lanes per chain with NLC consecutive chain elements each; affine-map scan.
template <int NLC>
DDH_DEV void cheb_solve_chains(double2 *S, const ChebTabs &T, const Lane &L) {
    constexpr int Mk = 16 * NLC;
    const int g = T.gcd_off, G4 = 4 * g;
    const int r = (L.lane >> 2) % g, seg = L.lane / G4, nseg = 64 / G4;
    const int kstart = Mk - 1 - ((Mk - 1 - r) % g);
    const double *t0 = T.bsub, *t1 = T.bsub + T.M;
    // two sweeps over the lane's elements (the second re-reads them: LDS reads are cheaper than 96 live registers)
    double A = 1.0;
    double2 Bv = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int e = 0; e < NLC; ++e) {
        const int k = kstart - (seg * NLC + e) * g;                 // >= 0: Mk = 16 NLC elements in all
        const double2 be = cscale(t0[k], S[k * 4 + L.p]);
        const double al = t1[k];
        Bv = make_double2(be.x - al * Bv.x, be.y - al * Bv.y);
        A = -al * A;
    }
    for (int off = 1; off < nseg; off <<= 1) {
        const double Ap = WF_SHFL_UP(A, off * G4, L.lane);
        const double Bx = WF_SHFL_UP(Bv.x, off * G4, L.lane), By = WF_SHFL_UP(Bv.y, off * G4, L.lane);
        if (seg >= off) {
            Bv = make_double2(Bv.x + A * Bx, Bv.y + A * By);
            A = A * Ap;
        }
    }
    double xx = WF_SHFL_UP(Bv.x, G4, L.lane), xy = WF_SHFL_UP(Bv.y, G4, L.lane);
    if (seg == 0) xx = xy = 0.0;
#pragma unroll 4
    for (int e = 0; e < NLC; ++e) {
        const int k = kstart - (seg * NLC + e) * g;
        const double2 be = cscale(t0[k], S[k * 4 + L.p]);
        const double al = t1[k];
        xx = be.x - al * xx;
        xy = be.y - al * xy;
        S[k * 4 + L.p] = make_double2(xx, xy);
    }
}

// FFT result (wfft output map) -> grid rows of the lane's pair
template <int R>
DDH_DEV void cheb_bwd_store(const double2 (&v)[R], double *dst_t, unsigned rsb, bool pvalid, const Lane &L) {
    constexpr int N = 16 * R, RQ = R / 4;
    WF_OPAQUE_U32(rsb);                                              // (outside the divergent region: a scalar register)
    if (!pvalid) return;
    const int nl = R * L.q0 + RQ * L.q1;                             // n = nl + 4 R a0 + i
    const unsigned lo = (unsigned)(2 * nl) * rsb + 16u * (unsigned)L.p;                  // row 2 n          (n < N / 2)
    const unsigned hi = (unsigned)(2 * (N - 1 - nl) + 1) * rsb + 16u * (unsigned)L.p;    // row 2 (N - 1 - n) + 1
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
            const unsigned d = (unsigned)(2 * (4 * R * a0 + i)) * rsb;
            gstore(dst_t, (a0 < 2) ? lo + d : hi - d, v[a0 * RQ + i]);                   // a0 < 2  <=>  n < N / 2
        }
}

template <int R>
DDH_DEV void cheb_bwd_store_contig(const double2 (&v)[R], double *dst_t, unsigned lsb, bool pvalid, const Lane &L) {
    constexpr int N = 16 * R, RQ = R / 4;
    WF_OPAQUE_U32(lsb);
    if (!pvalid) return;
    const int nl = R * L.q0 + RQ * L.q1;                             // n = nl + 4 R a0 + i
    const unsigned po = (unsigned)(2 * L.p) * lsb;
    const unsigned lo = po + 8u * (unsigned)(2 * nl);                                    // row 2 n          (n < N / 2)
    const unsigned hi = po + 8u * (unsigned)(2 * (N - 1 - nl) + 1);                      // row 2 (N - 1 - n) + 1
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
            const unsigned d = 8u * (unsigned)(2 * (4 * R * a0 + i));
            cstore(dst_t, (a0 < 2) ? lo + d : hi - d, lsb, v[a0 * RQ + i]);
        }
}

// One backward pass on staged data.  mode 0: plain (Z from the coefficients in c); 1: derivative pass of the dual
// transform (d[j] = dvec[j] c[j + 1], conversion solve, Z from LDS); 2: conversion solve of c itself.
// S: this wave's LDS region of max(16 NLC * 4, WfftBuf<R, CH>::size) elements.
// As soon as c has been consumed, the coefficient rows of the wave's NEXT tile (next_src_t) are requested into the
// same registers; they are in flight during the FFT and the stores of this pass.  The request is unconditional (a
// conditional one makes the compiler keep both register sets and spill): after its last tile a wave passes
// next_rsb = 0, i.e. sixteen reads of one 64-byte segment it already has in cache.
// CONTIG: contiguous axis -- rsb / next_rsb are the bytes between LINES of dst_t / next_src_t (cload / cstore addressing).
template <int R, int NLC, int CH, int mode, bool PREFETCH, bool CONTIG = false>
DDH_DEV void cheb_bwd_pass(double2 (&c)[NLC], double2 *S, const ChebTabs &T, double *dst_t, unsigned rsb,
                           bool pvalid, int lane, const double *next_src_t, unsigned next_rsb, bool next_valid) {
    constexpr int Mk = 16 * NLC;
    WF_OPAQUE_LANE(lane);
    const Lane L = make_lane(lane);
    double2 v[R];
    WF_SYNC();                                                       // the previous pass has drained its exchanges
    if (mode == 1) {
#pragma unroll
        for (int t = 0; t < NLC; ++t) {
            const int k = L.q + 16 * t;
            if (k >= 1) S[(k - 1) * 4 + L.p] = cscale(T.dvec[k - 1], c[t]);
            else S[(Mk - 1) * 4 + L.p] = make_double2(0.0, 0.0);     // c[M] does not exist
        }
    } else {
#pragma unroll
        for (int t = 0; t < NLC; ++t) S[(L.q + 16 * t) * 4 + L.p] = c[t];
    }
    WF_SYNC();
    if (mode == 0) {
        cheb_bwd_build<R, NLC, true>(v, c, S, T, L);
        WF_SCHED_FENCE();            // c is dead from here on: its registers take the prefetch
        if (PREFETCH) {
            if constexpr (CONTIG) cheb_bwd_load_contig<NLC>(c, next_src_t, next_rsb, next_valid, L);
            else cheb_bwd_load<NLC>(c, next_src_t, next_rsb, next_valid, L);
        }
        WF_SCHED_FENCE();
    } else {
        WF_SCHED_FENCE();
        if (PREFETCH) {
            if constexpr (CONTIG) cheb_bwd_load_contig<NLC>(c, next_src_t, next_rsb, next_valid, L);
            else cheb_bwd_load<NLC>(c, next_src_t, next_rsb, next_valid, L);
        }
        WF_SCHED_FENCE();
        cheb_solve_chains<NLC>(S, T, L);
        WF_SYNC();
        cheb_bwd_build<R, NLC, false>(v, c, S, T, L);
    }
    wfft<R, +1, CH>(v, S, T.tw, L);
    if constexpr (CONTIG) cheb_bwd_store_contig<R>(v, dst_t, rsb, pvalid, L);
    else cheb_bwd_store<R>(v, dst_t, rsb, pvalid, L);
}

// Forward: grid rows -> coefficients k = q + 16 t, t < NST (Mk = 16 NST), optional conversion bands.
// S: max(8 R * 4, 16 NST * 4, WfftBuf size) elements.
template <int R, int NST, int CH, bool CONTIG = false>
DDH_DEV void cheb_fwd_tile(const double *src_t, double *dst_t, unsigned rsb, unsigned rsbd, bool pvalid, double2 *S,
                           const ChebTabs &T, int lane) {      // rsb: bytes between grid rows (src), rsbd: coefficient rows (dst)
                                                                // (CONTIG: bytes between the LINES of src / dst)
    constexpr int N = 16 * R, RQ = R / 4, Mk = 16 * NST, H8 = 8 * R;
    WF_OPAQUE_LANE(lane);
    const Lane L = make_lane(lane);
    double2 v[R];
    {
        WF_OPAQUE_U32(rsb);
        // FFT position n = q + 16 t holds grid row 2 n (t < R / 2) or 2 (N - 1 - n) + 1
        if constexpr (CONTIG) {
            const unsigned po = pvalid ? (unsigned)(2 * L.p) * rsb : 0u;
            const unsigned lo = 8u * (unsigned)(2 * L.q) + po;
            const unsigned hi = 8u * (unsigned)(2 * N - 1 - 2 * L.q) + po;
#pragma unroll
            for (int t = 0; t < R; ++t)
                v[t] = cload(src_t, (2 * t < R) ? lo + 8u * (unsigned)(32 * t) : hi - 8u * (unsigned)(32 * t), rsb);
        } else {
        const unsigned po = pvalid ? 16u * (unsigned)L.p : 0u;       // a pair beyond the array re-reads pair 0
        const unsigned lo = (unsigned)(2 * L.q) * rsb + po;
        const unsigned hi = (unsigned)(2 * N - 1 - 2 * L.q) * rsb + po;
#pragma unroll
        for (int t = 0; t < R; ++t)
            v[t] = gload(src_t, (2 * t < R) ? lo + (unsigned)(32 * t) * rsb : hi - (unsigned)(32 * t) * rsb);
        }
    }
    wfft<R, -1, CH>(v, S, T.tw, L);
    // c~[k] = fs_k (X[k] h_k + X[N - k] conj h_k)  (X[N] = X[0]); natural-order exchange in two halves of N / 2
    double2 acc[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) acc[t] = make_double2(0.0, 0.0);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int base = round * H8;
        WF_SYNC();
#pragma unroll
        for (int a0 = 2 * round; a0 < 2 * round + 2; ++a0)
#pragma unroll
            for (int i = 0; i < RQ; ++i) S[(R * (L.q0 + 4 * a0) + RQ * L.q1 + i - base) * 4 + L.p] = v[a0 * RQ + i];
        WF_SYNC();
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            const int k = L.q + 16 * t;
            const double2 h = T.half[k];
            if ((2 * t < R) == (round == 0)) {                       // X[k] lives in this half
                const double2 x = S[(k - base) * 4 + L.p];
                const double2 y = cmul(x, h);
                acc[t].x += y.x;
                acc[t].y += y.y;
            }
            // X[N - k]: k in the lower half -> upper half (except k = 0 -> X[0]); k in the upper half -> lower half
            // (except k = N / 2 -> itself)
            const bool maybe = (round == 0) ? (2 * t >= R || t == 0) : (2 * t <= R);
            if (maybe) {
                const int kr = (k == 0) ? 0 : N - k;
                const bool in = (kr >= base) && (kr < base + H8);
                const double2 x = S[(in ? kr - base : 0) * 4 + L.p];
                const double2 y = cmul(x, conj2(h));
                if (in) {
                    acc[t].x += y.x;
                    acc[t].y += y.y;
                }
            }
        }
    }
    const double f1 = (L.q & 1) ? -T.fs1 : T.fs1;
#pragma unroll
    for (int t = 0; t < NST; ++t) acc[t] = cscale((t == 0 && L.q == 0) ? T.fs0 : f1, acc[t]);
    if (T.nbands > 0) {
        // conversion apply: out[k] = sum_d bands[d][k] c~[k + boff[d]]  (boff[0] = 0)
        WF_SYNC();
#pragma unroll
        for (int t = 0; t < NST; ++t) S[(L.q + 16 * t) * 4 + L.p] = acc[t];
        WF_SYNC();
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            const int k = L.q + 16 * t;
            double2 o = cscale(T.bands[k], acc[t]);
#pragma unroll
            for (int d = 1; d < 4; ++d) {
                const int off = (d == 1) ? T.boff1 : (d == 2) ? T.boff2 : T.boff3;
                const int kk = k + off;
                if (d < T.nbands && kk < Mk) {
                    const double a = T.bands[d * T.M + k];
                    const double2 x = S[kk * 4 + L.p];
                    o.x += a * x.x;
                    o.y += a * x.y;
                }
            }
            acc[t] = o;
        }
    }
    WF_OPAQUE_U32(rsbd);
    if (pvalid) {
        if constexpr (CONTIG) {
            const unsigned o0 = (unsigned)(2 * L.p) * rsbd + 8u * (unsigned)L.q;
#pragma unroll
            for (int t = 0; t < NST; ++t) cstore(dst_t, o0 + (unsigned)(128 * t), rsbd, acc[t]);
        } else {
        const unsigned o0 = (unsigned)L.q * rsbd + 16u * (unsigned)L.p;
#pragma unroll
        for (int t = 0; t < NST; ++t) gstore(dst_t, o0 + (unsigned)(16 * t) * rsbd, acc[t]);
        }
    }
}

template <int R, int NL, int CH>
struct ChebWaveLds {
    static constexpr int a = 16 * NL * 4, b = WfftBuf<R, CH>::size, c = 8 * R * 4;
    static constexpr int size = (a > b ? (a > c ? a : c) : (b > c ? b : c));       // double2 elements per wave
};


// ------------------------------------------------------------------------------------------------
// Real Fourier along a strided axis with 3/2 dealiasing: N = 3 * 16 R grid points, M = 2 * 16 R coefficient rows
// (cos, msin interleaved; K = 16 R - 1), two real lines per pair (core/transforms.py:469-565,
// libraries/fftw/fftw_wrappers.pyx:61-214).  A third of the length-N spectrum is the dealiasing gap, so the length-N
// transform is three length-N/3 transforms of pre-combined inputs (backward: grid rows 3 m + r from
// u_r[k] = w^(r k) (Z[k] + W3^(2 r) Z[k + 2N/3])) and, forward, three transforms F_r of the rows 3 m + r accumulated
// into X[k] = sum_r w^(-r k) F_r[k], X[N - k] = sum_r w^(r k) F_r[N/3 - k].
// ------------------------------------------------------------------------------------------------
template <int R>
struct RfftWaveLds {
    static constexpr int a = 16 * R * 4;                 // natural-order exchange of N/3 values x 4 pairs (forward); half of the
                                                         // coefficient rows (backward staging: 8 R modes x 2 rows x 4 pairs)
    static constexpr int b = WfftBuf<R, WaveCH<R>::ch>::size;
    static constexpr int park = (R >= 16) ? 128 : 0;     // two more parking slots per lane for the backward tile (see there):
                                                         // 8 waves x 18 KiB + the 12 KiB table = 156 of the CU's 160 KiB
    static constexpr int size = (a > b ? a : b) + park;
};

// Backward.  tw[m] = exp(-2 pi i m / N).  BK: 0 plain transform into dst_t; 1 differentiated (spectrum times i kappa,
// kappa = dsc * k) into dst_t; 2 both from one read of the coefficients: plain into dst_t, differentiated into dst2_t.
template <int R, int BK>
// Coefficient rows k of a tile sit at (k / 64) * rsb64 + (k % 64) * rsb bytes: rsb64 = 64 rsb is the natural layout
// [kx][ky]; the x-blocked stage layout [kx / 64][z][kx % 64][ky] (ddh_fft_set_stage_layout) has the z planes of a block
// of 64 rows in between.  The lane's rows are 2 q + 32 t (+ 1), q < 16: block t / 2, row 32 (t % 2) + 2 q (+ 1) inside it.
DDH_DEV void rfft_bwd_tile(const double *src_t, double *dst_t, double *dst2_t, unsigned rsb, unsigned rsb64, bool pvalid,
                           double dsc, double2 *S, const double2 *tw, int lane, int bsh = 1) {
    // bsh: the blocks of the blocked coefficient layout hold 32 << bsh rows (rsb64 = bytes between blocks): 64 (bsh = 1) on
    // one rank, nx / P in a sharded run where the received layout [p][z][nx / P][ky] is read as it arrives
    constexpr int H = 16 * R;                            // N / 3 = modes per pair incl. k = 0
    WF_OPAQUE_LANE(lane);
    const Lane L = make_lane(lane);
    double2 A[R], Zm[R];                                 // Z[k], Z[k + 2H] (= Z[N - (H - k)]) of the lane's modes k = q + 16 t
    {
        double2 c[R], s[R];
        WF_OPAQUE_U32(rsb);
        WF_OPAQUE_U32(rsb64);
        const unsigned o0 = (unsigned)(2 * L.q) * rsb + (pvalid ? 16u * (unsigned)L.p : 0u);
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const unsigned ob = (unsigned)(t >> bsh) * rsb64 + (unsigned)(32 * (t & ((1 << bsh) - 1))) * rsb;
            c[t] = gload(src_t, o0 + ob);
            s[t] = gload(src_t, o0 + ob + rsb);
        }
        // the partner mode H - k sits in another lane: two half exchanges through LDS, [cos | msin][mode][pair]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            WF_SYNC();
#pragma unroll
            for (int t = half * (R / 2); t < (half + 1) * (R / 2); ++t) {
                const int ml = L.q + 16 * t - half * (H / 2);
                S[ml * 4 + L.p] = c[t];
                S[(H / 2 + ml) * 4 + L.p] = s[t];
            }
            WF_SYNC();
            // staged: modes [half H/2, (half + 1) H/2); readers: the lanes' slots of the OTHER half (k = H/2 pairs with
            // itself and is served by the second half)
#pragma unroll
            for (int t = (1 - half) * (R / 2); t < (2 - half) * (R / 2) + half; ++t) {
                const int k = L.q + 16 * t, km = H - k;
                const bool in = (k > 0) && (km >= half * (H / 2)) && (km < (half + 1) * (H / 2));
                const int ml = in ? km - half * (H / 2) : 0;
                const double2 cm = S[ml * 4 + L.p], sm = S[(H / 2 + ml) * 4 + L.p];
                const double2 z = make_double2(0.5 * (cm.x + sm.y), 0.5 * (cm.y - sm.x));
                if (half == 1 && t == R / 2) {           // only lane q = 0 (k = H/2) is served here; the others keep theirs
                    Zm[t] = make_double2(in ? z.x : Zm[t].x, in ? z.y : Zm[t].y);
                } else {
                    Zm[t] = make_double2(in ? z.x : 0.0, in ? z.y : 0.0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const bool k0 = (t == 0 && L.q == 0);        // k = 0: the msin row is no mode
            A[t] = make_double2(k0 ? c[t].x : 0.5 * (c[t].x - s[t].y), k0 ? c[t].y : 0.5 * (s[t].x + c[t].y));
        }
    }
    const double s3 = 0.86602540378443864676372317075293618;
    // Register relief: the staging area beyond the FFT's exchange buffer is free from here on, and the last ZL of the
    // lane's Z[N - k'] values are parked there (lane-contiguous slots, read back once per transform).  With everything
    // in registers the R = 16 kernels spill (plain 16, differentiated 72, dual 160-196 bytes of scratch per lane).
    // (the differentiated and the dual transform exchange in four chunks instead of two: half the exchange buffer = four
    //  more parking slots per lane; with 9 slots the dual kernel still spilled 60 bytes per lane at 256 registers -- 1.16 x
    //  its algorithmic HBM traffic --, with 13 none: 1.38 -> 1.29 ms at 512^2 x 384)
    constexpr int CHX = (R >= 16 && BK != 0) ? 4 : WaveCH<R>::ch;
    constexpr int XB = WfftBuf<R, CHX>::size;
    constexpr int ZLmax = (RfftWaveLds<R>::size - XB) / 64;
    constexpr int ZLwant = (R >= 16) ? (BK == 2 ? 13 : (BK == 1 ? 12 : 4)) : 0;
    constexpr int ZL = ZLwant < ZLmax ? ZLwant : ZLmax;
    double2 *ZS = S + XB + lane;
    if (ZL > 0) {
        WF_SYNC();                                       // the staged coefficient rows have been read by every lane
#pragma unroll
        for (int t = R - ZL; t < R; ++t) ZS[(t - (R - ZL)) * 64] = Zm[t];
    }
    // differentiated spectrum: Z[k] <- i kappa Z[k], Z[N - k'] <- -i kappa' Z[N - k'] (kappa = dsc * k): in place for the
    // register-held values (BK == 2: between the two passes, so that both run the same pass body), when read for the
    // parked ones.
    auto differentiate = [&]() {
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int k = L.q + 16 * t;
            const double ka = dsc * (double)k, km = dsc * (double)(H - k);
            A[t] = make_double2(-ka * A[t].y, ka * A[t].x);
            if (t < R - ZL) Zm[t] = make_double2(km * Zm[t].y, -km * Zm[t].x);
        }
    };
    if (BK == 1) differentiate();
#pragma unroll
    for (int pass = 0; pass < (BK == 2 ? 2 : 1); ++pass) {
        double *out_t = dst_t;
        const bool deriv = (BK == 1) || (BK == 2 && pass == 1);       // compile time (the passes are unrolled)
        if (BK == 2 && pass == 1) {
            differentiate();
            out_t = dst2_t;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double2 v[R];
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const int k = L.q + 16 * t;
                const double2 a = A[t];
                double2 zm;
                if (t < R - ZL) {
                    zm = Zm[t];
                } else {
                    zm = ZS[(t - (R - ZL)) * 64];
                    if (deriv) {
                        const double km = dsc * (double)(H - k);
                        zm = make_double2(km * zm.y, -km * zm.x);
                    }
                }
                double2 wz = zm;                         // W3^(2 r) zm
                if (r == 1) wz = make_double2(-0.5 * zm.x + s3 * zm.y, -0.5 * zm.y - s3 * zm.x);
                if (r == 2) wz = make_double2(-0.5 * zm.x - s3 * zm.y, -0.5 * zm.y + s3 * zm.x);
                const double2 u = make_double2(a.x + wz.x, a.y + wz.y);
                v[t] = (r == 0) ? u : cmul(u, twid<+1>(tw, r * k));
                if ((t & 3) == 3) WF_SCHED_FENCE();
            }
            wfft<R, +1, CHX, 3>(v, S, tw, L);
            WF_OPAQUE_U32(rsb);
            if (pvalid) {
                const unsigned o0 = (unsigned)(3 * (R * L.q0 + (R / 4) * L.q1) + r) * rsb + 16u * (unsigned)L.p;
#pragma unroll
                for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
                    for (int i = 0; i < R / 4; ++i)
                        gstore(out_t, o0 + (unsigned)(3 * (4 * R * a0 + i)) * rsb, v[a0 * (R / 4) + i]);
            }
        }
    }
}

// Forward: grid rows -> (cos, msin) rows 2 k, 2 k + 1 of the lane's modes k = q + 16 t.
// With T_r[j] = exp(-2 pi i r j / N) F_r[j] (F_r = length-H transform of the rows 3 m + r):
//     X[j] = sum_r T_r[j],     X[N - (H - j)] = sum_r W3^r T_r[j],   W3 = exp(2 pi i / 3),
// so both halves of the spectrum accumulate in the lane that owns F_r[j] (the wfft output slot) and one natural-order
// exchange at the end hands X[k], X[N - k] to the lane that stores mode k.
template <int R>
DDH_DEV void rfft_fwd_tile(const double *src_t, double *dst_t, unsigned rsb, unsigned rsb64, bool pvalid, double2 *S,
                           const double2 *tw, int lane, int bsh = 1) {      // rsb64, bsh: see rfft_bwd_tile (coefficient side = dst here)
    constexpr int H = 16 * R, N = 3 * H, RQ = R / 4;
    const double s3 = 0.86602540378443864676372317075293618;
    double2 P[R], Q[R];                                  // X[j], X[N - (H - j)] at the wfft output slots j
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double2 v[R];
        WF_SCHED_FENCE();                                // the rows of r are requested after r - 1 has been accumulated
        int ln = lane;
        WF_OPAQUE_LANE(ln);                              // addresses are re-derived per r, not kept across the three
        const Lane Lr = make_lane(ln);
        {
            WF_OPAQUE_U32(rsb);
            const unsigned o0 = (unsigned)(3 * Lr.q + r) * rsb + (pvalid ? 16u * (unsigned)Lr.p : 0u);
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = gload(src_t, o0 + (unsigned)(48 * t) * rsb);
        }
        WF_SCHED_FENCE();
        wfft<R, -1, WaveCH<R>::ch, 3>(v, S, tw, Lr);
        const double2 *twr = tw + r * (R * Lr.q0 + RQ * Lr.q1);       // exp(-2 pi i r j / N), j = R (q0 + 4 a0) + RQ q1 + i
#pragma unroll
        for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                const int j = a0 * RQ + i;
                if (r == 0) {
                    P[j] = v[j];
                    Q[j] = v[j];
                } else {
                    const double2 T = cmul(v[j], twr[r * (4 * R * a0 + i)]);
                    P[j] = make_double2(P[j].x + T.x, P[j].y + T.y);
                    if (r == 1) Q[j] = make_double2(Q[j].x - 0.5 * T.x - s3 * T.y, Q[j].y - 0.5 * T.y + s3 * T.x);
                    else Q[j] = make_double2(Q[j].x - 0.5 * T.x + s3 * T.y, Q[j].y - 0.5 * T.y - s3 * T.x);
                }
                if ((j & 3) == 3) WF_SCHED_FENCE();
            }
    }
    WF_OPAQUE_LANE(lane);
    const Lane L = make_lane(lane);
    double2 z1[R];
    // X[k] = P[k]
    WF_SYNC();
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
        for (int i = 0; i < RQ; ++i) S[(R * (L.q0 + 4 * a0) + RQ * L.q1 + i) * 4 + L.p] = P[a0 * RQ + i];
    WF_SYNC();
#pragma unroll
    for (int t = 0; t < R; ++t) z1[t] = S[(L.q + 16 * t) * 4 + L.p];
    // X[N - k] = Q[H - k]:  H - k = (16 - q) + 16 (R - 1 - t)  (k = 0 is not used)
    WF_SYNC();
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0)
#pragma unroll
        for (int i = 0; i < RQ; ++i) S[(R * (L.q0 + 4 * a0) + RQ * L.q1 + i) * 4 + L.p] = Q[a0 * RQ + i];
    WF_SYNC();
    WF_OPAQUE_U32(rsb);
    WF_OPAQUE_U32(rsb64);
    const double invN = 1.0 / (double)N;
    const unsigned o0 = (unsigned)(2 * L.q) * rsb + 16u * (unsigned)L.p;
    const double2 *Sm = S + (16 - L.q) * 4 + L.p;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const double2 z2 = (t == 0) ? S[(L.q == 0 ? 0 : H - L.q) * 4 + L.p] : Sm[16 * (R - 1 - t) * 4];
        double2 c = make_double2((z1[t].x + z2.x) * invN, (z1[t].y + z2.y) * invN);
        double2 s = make_double2((z1[t].y - z2.y) * invN, (z2.x - z1[t].x) * invN);
        if (t == 0 && L.q == 0) {                        // k = 0
            c = make_double2(z1[t].x * invN, z1[t].y * invN);
            s = make_double2(0.0, 0.0);
        }
        if (pvalid) {
            const unsigned ob = (unsigned)(t >> bsh) * rsb64 + (unsigned)(32 * (t & ((1 << bsh) - 1))) * rsb;
            gstore(dst_t, o0 + ob, c);
            gstore(dst_t, o0 + ob + rsb, s);
        }
    }
}

}  // namespace wf
}  // namespace ddh
