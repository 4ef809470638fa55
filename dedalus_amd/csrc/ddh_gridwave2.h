// Lane code of the second-generation wave-per-line grid stage (ddh_gridwave2.hip): the real line of N = 128 C grid
// points as a complex FFT of half length H = 64 C, decomposed H = C x 8 x 8 with the radix-C pass on the COEFFICIENT
// side and both radix-8 passes on the GRID side:
//
//   backward:  64 lanes x C spectrum values Z[k0 + 64 c] (k0 = lane), built straight from two coalesced global loads
//              per 64 pairs -- the pairs k0 + 64 t and their mirror images H - k, lane-reversed -- with no staging
//              round trip through the LDS  ->  radix C over c  ->  twiddle  ->  exchange  ->  radix 8  ->  twiddle  ->
//              exchange  ->  radix 8,  leaving 8 C butterfly lanes x 8 complex grid values
//                  g[q] = x[2 m] + i x[2 m + 1],   m = mc + C (p + 8 q),   lane = 8 mc + p;
//   forward:   the mirror image from the same (lane, register) map, then the Hermitian post-processing with one small
//              exchange for the mirror values and coalesced 16-byte stores.
//
// Against the first generation (ddh_gridwave.hip: radix 8 x 8 x C with the radix-C pass on the grid side) a backward
// transform issues 14 LDS stores and ~27 LDS loads instead of 20 and 38, a forward transform 20 / 23 instead of 22 / 40:
// ds_write_b128 costs ~13 LDS cycles per wave instruction on gfx950 against 4 for ds_read_b128, and the LDS pipe was the
// largest single share of that kernel.  The pre- and post-processing run on all 64 lanes (C values each) instead of on
// 8 C lanes (8 values each).
//
// Plain C++ over the WF_* macros of ddh_wavefft.h: tests/host_emu/emu_gridwave2.cpp compiles the SAME lane code with g++
// (64 threads = 64 lanes) and tests/test_host_emu_gridwave2.py checks it against numpy without a GPU.
//
// Replaces the reference's backward transforms + DotProduct / MultiplyFields + forward transform along the last axis
// (core/transforms.py:469-565, core/arithmetic.py:666-674, 855-866).
#pragma once
#include "ddh_wavefft.h"

// Between the loads and the stores of one wave to the SAME exchange buffer nothing is needed on the device (the LDS
// operations of a wave execute in program order); the emulator's lanes are free-running threads and need a barrier.
#ifdef DDH_HOST_EMU
#define GW2_EMU_SYNC() WF_SYNC()
#else
#define GW2_EMU_SYNC() \
    do {               \
    } while (0)
#endif

namespace ddh {
namespace gw2 {

constexpr int cmax2(int a, int b) { return a > b ? a : b; }

template <int C>
struct G2 {
    static constexpr int H = 64 * C;         // complex FFT length
    static constexpr int N = 128 * C;        // real grid points
    static constexpr int NB = 8 * C;         // butterfly lanes of the radix-8 passes
    static constexpr int S1 = 72;            // exchange 1: [mc][k0], row stride == 8 (mod 16): 16-lane read groups conflict free
    static constexpr int S2 = 72;            // exchange 2: [mc][9 r + s] (r, s < 8): both sides conflict free
    static constexpr int LDW = cmax2(C * S1, C * 64 + 1);      // double2 per wave
    // twiddle tables in LDS (forward sign):
    static constexpr int T_N = 0;            // [k]        exp(-2 pi i k / N),        k < H
    static constexpr int T_P = H;            // [mc][k0]   exp(-2 pi i k0 mc / H),    row stride S1
    static constexpr int T_64 = H + C * S1;  // [j]        exp(-2 pi i j / 64)
    static constexpr int TW = T_64 + 64;
    // plan table index (exp(-2 pi i q / N)) of LDS table entry i
    static DDH_DEV int table_q(int i) {
        if (i < T_P) return i;
        if (i < T_64) {
            const int r = i - T_P, mc = r / S1, k0 = r - mc * S1;
            return (k0 < 64) ? 2 * ((k0 * mc) % H) : 0;
        }
        return ((i - T_64) & 63) * (2 * C);
    }
};

DDH_DEV double2 conj2(double2 a) { return make_double2(a.x, -a.y); }

// (cos, msin) coefficient pair of wavenumber k -> TWICE the spectrum value X[k] of the unnormalised c2r transform
// (X[k] = (cos + i msin) / 2 for k > 0, X[0] = cos; the factor 1/2 of every operand is folded into the term coefficients on
// the host).  DERIV: differentiated, X' = i kappa X, kappa = dscale * k.
template <bool DERIV>
DDH_DEV double2 coef_to_spec2(double2 cs, double kf, bool is_zero, double dscale) {
    if (DERIV) {
        const double kap = dscale * kf;
        return make_double2(-kap * cs.y, kap * cs.x);
    }
    if (is_zero) return make_double2(2.0 * cs.x, 0.0);
    return cs;
}

// the pairs a lane holds of one coefficient line: d[t] = pair lane + 64 t, m[t] = pair (64 - lane) + 64 t
template <int NT>
struct Loads {
    double2 d[NT], m[NT];
};

// Spectrum pre-processing: z[c] = 2 Z[k], k = lane + 64 c,
//     Z[k] = (X[k] + conj X[H-k]) + i w^k (X[k] - conj X[H-k]),   w = exp(+2 pi i / N)
// Only pairs k < 64 NT are stored (3/2 padding: the rest of the spectrum is zero).
template <int C, int NT, bool DERIV>
DDH_DEV void build_z(const Loads<NT> &ld, double dscale, const double2 *tw, int lane, double2 *z) {
    using G = G2<C>;
    const double lf = (double)lane;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const bool has_d = c < NT;                     // k = lane + 64 c < 64 NT
        const int tm = C - 1 - c;                      // mirror H - k = (64 - lane) + 64 tm: the lane-reversed load tm
        const bool has_m = tm < NT;                    // (lane 0: pair 64 (tm + 1), zero from the range check when tm = NT - 1)
        double2 xd = make_double2(0.0, 0.0), xm = xd;
        if (has_d) xd = coef_to_spec2<DERIV>(ld.d[c], lf + 64.0 * c, c == 0 && lane == 0, dscale);
        if (has_m) xm = coef_to_spec2<DERIV>(ld.m[tm < NT ? tm : 0], (double)(G::H - 64 * c) - lf, false, dscale);
        const double2 w = conj2(tw[G::T_N + lane + 64 * c]);
        if (has_d && has_m) {
            const double2 A = make_double2(xd.x + xm.x, xd.y - xm.y);
            const double2 B = make_double2(xd.x - xm.x, xd.y + xm.y);
            const double2 wB = cmul(w, B);
            z[c] = make_double2(A.x - wB.y, A.y + wB.x);
        } else if (has_d) {                            // Z = X + i w X
            const double2 wB = cmul(w, xd);
            z[c] = make_double2(xd.x - wB.y, xd.y + wB.x);
        } else if (has_m) {                            // Z = conj Xm - i w conj Xm
            const double2 cm = make_double2(xm.x, -xm.y);
            const double2 wB = cmul(w, cm);
            z[c] = make_double2(cm.x + wB.y, cm.y - wB.x);
        } else {
            z[c] = make_double2(0.0, 0.0);
        }
    }
}

// The same from an LDS staging area filled by LDS-DMA loads (buffer_load ... lds: no destination registers while the
// pairs of the next operands are in flight): st[k] = pair k for k < 64 NT, st[64 NT] = 0.  The mirror pairs are the same
// blocks read lane-reversed: pair (64 - lane) + 64 t = st[64 t + 64 - lane] (lane 0: the first pair of the next block).
template <int NT>
struct Staging {
    static constexpr int SIZE = 64 * NT + 8;       // double2 per buffer (the pad keeps the second buffer 128-byte aligned)
};
template <int C, int NT, bool DERIV>
DDH_DEV void build_z_staged(const double2 *st, double dscale, const double2 *tw, int lane, double2 *z) {
    Loads<NT> ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ld.d[t] = st[t * 64 + lane];
        ld.m[t] = st[t * 64 + 64 - lane];
    }
    build_z<C, NT, DERIV>(ld, dscale, tw, lane, z);
}

// Backward transform of the pre-processed spectrum z[c] = 2 Z[lane + 64 c]:
//     g[q] = 2 (x[2 m] + i x[2 m + 1]),  m = mc + C (p + 8 q),  lane = 8 mc + p < 8 C   (other lanes: copies of mc = 0)
// t64r: exp(-2 pi i (lane % 8) (i + 1) / 64), i < 7, in registers (TWREG) or null (LDS table)
template <int C, bool TWREG>
DDH_DEV void backward_line(double2 *z, double2 *wb, const double2 *tw, int lane, double2 *g, const double2 *t64r) {
    using G = G2<C>;
    // radix C over c -> mc, twiddle exp(+2 pi i k0 mc / H), exchange 1: [mc][k0]
    butterfly<C>(z, +1);
    wb[lane] = z[0];
#pragma unroll
    for (int mc = 1; mc < C; ++mc) wb[mc * G::S1 + lane] = cmul(z[mc], conj2(tw[G::T_P + mc * G::S1 + lane]));
    WF_SYNC();
    // lanes beyond the 8 C butterflies run the same instructions on the data of mc = 0 (no exec-mask changes, every value
    // defined) and only skip the stores
    const bool act = lane < G::NB;
    const int hi = act ? lane >> 3 : 0, lo = lane & 7;
    double2 v[8];
    // (mc, a) = (hi, lo): inputs k0 = a + 8 b; radix 8 over b -> p, twiddle exp(+2 pi i a p / 64); exchange 2: [mc][9 p + a]
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = wb[hi * G::S1 + lo + 8 * b];
    GW2_EMU_SYNC();
    butterfly<8>(v, +1);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], conj2(TWREG ? t64r[p - 1] : tw[G::T_64 + lo * p]));
    if (act) {
#pragma unroll
        for (int p = 0; p < 8; ++p) wb[hi * G::S2 + 9 * p + lo] = v[p];
    }
    WF_SYNC();
    // (mc, p) = (hi, lo): inputs a = 0 .. 7; radix 8 over a -> q
#pragma unroll
    for (int a = 0; a < 8; ++a) g[a] = wb[hi * G::S2 + 9 * lo + a];
    butterfly<8>(g, +1);
}

// Forward transform of the grid values y[q] (same map) and store of the coefficient line:
//     pair k = (2 / N) sum_n y_n exp(-2 pi i k n / N)   (k = 0: 1 / N, real),  k <= K;  zeros up to M / 2
// STORE(k, value): the caller's 16-byte store of pair k
template <int C, int NT, bool TWREG, class Store>
DDH_DEV void forward_line(double2 *y, double2 *wb, const double2 *tw, int lane, int M, int K, const double2 *t64r,
                          Store store) {
    using G = G2<C>;
    const bool act = lane < G::NB;
    const int hi = act ? lane >> 3 : 0, lo = lane & 7;
    // (mc, p) = (hi, lo): radix 8 over q -> a, twiddle exp(-2 pi i a p / 64), exchange: [mc][9 a + p]
    butterfly<8>(y, -1);
#pragma unroll
    for (int a = 1; a < 8; ++a) y[a] = cmul(y[a], TWREG ? t64r[a - 1] : tw[G::T_64 + lo * a]);
    if (act) {
#pragma unroll
        for (int a = 0; a < 8; ++a) wb[hi * G::S2 + 9 * a + lo] = y[a];
    }
    WF_SYNC();
    // (mc, a) = (hi, lo): inputs p = 0 .. 7; radix 8 over p -> b; exchange: [mc][a + 8 b]
    double2 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = wb[hi * G::S2 + 9 * lo + p];
    GW2_EMU_SYNC();
    butterfly<8>(v, -1);
    if (act) {
#pragma unroll
        for (int b = 0; b < 8; ++b) wb[hi * G::S1 + lo + 8 * b] = v[b];
    }
    WF_SYNC();
    // lane = k0: twiddle exp(-2 pi i k0 mc / H), radix C over mc -> c:  zf[c] = Zf[k0 + 64 c]
    double2 zf[C];
    zf[0] = wb[lane];
#pragma unroll
    for (int mc = 1; mc < C; ++mc) zf[mc] = cmul(wb[mc * G::S1 + lane], tw[G::T_P + mc * G::S1 + lane]);
    butterfly<C>(zf, -1);
    // mirror values Zf[H - k], k = lane + 64 t: slot C - 1 - t of lane 64 - lane (lane 0: its own slot C - t)
    WF_SYNC();
#pragma unroll
    for (int t = 0; t < NT; ++t) wb[t * 64 + lane] = zf[C - 1 - t];
    WF_SYNC();
    // Y[k] = E + exp(-2 pi i k / N) O,  E = (Zf[k] + conj Zf[H-k]) / 2,  O = -i (Zf[k] - conj Zf[H-k]) / 2
    const double invN = 1.0 / (double)G::N;
    const int Mh = M >> 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = lane + 64 * t;
        if (64 * t < Mh && k < Mh) {
            double2 out = make_double2(0.0, 0.0);
            if (k == 0) {
                out.x = (zf[0].x + zf[0].y) * invN;
            } else if (k <= K) {
                const double2 z = zf[t];
                double2 zm = wb[t * 64 + ((64 - lane) & 63)];
                if (lane == 0) zm = zf[(C - t) < C ? (C - t) : 0];         // (t >= 1 here)
                const double2 E = make_double2(0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y));
                const double2 D = make_double2(0.5 * (z.x - zm.x), 0.5 * (z.y + zm.y));
                const double2 O = make_double2(D.y, -D.x);
                const double2 wO = cmul(tw[G::T_N + k], O);
                out.x = 2.0 * invN * (E.x + wO.x);
                out.y = 2.0 * invN * (E.y + wO.y);
            }
            store(k, out);
        }
    }
    WF_SYNC();
}

}  // namespace gw2
}  // namespace ddh
