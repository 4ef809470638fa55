// Host emulation of the wave-level strided transforms (dedalus_amd/csrc/ddh_wavefft.h): the SAME lane code the GPU
// kernels run, compiled with g++; the 64 lanes of a wavefront are 64 threads, WF_SYNC is a barrier and the LDS is a
// shared array.  Test infrastructure only (tests/test_host_emu_wavefft.py): checks index maps and arithmetic of the
// kernels against the numpy oracle without a GPU.
#define DDH_HOST_EMU
#include <pthread.h>

#include <cstring>
#include <thread>
#include <vector>

#include "../../dedalus_amd/csrc/ddh_wavefft.h"

static pthread_barrier_t g_bar;
static double g_shfl[64];

namespace ddh {
namespace wf {
void emu_barrier() { pthread_barrier_wait(&g_bar); }
double emu_shfl_up(double v, int delta, int lane) {
    g_shfl[lane] = v;
    pthread_barrier_wait(&g_bar);
    const double r = (lane - delta >= 0) ? g_shfl[lane - delta] : v;
    pthread_barrier_wait(&g_bar);
    return r;
}
}  // namespace wf
}  // namespace ddh

using namespace ddh;
using namespace ddh::wf;

struct EmuTabs {
    const double *tw, *half;      // [N][2]
    const double *bands;          // [nbands][M]
    const double *bsub;           // [2][M]
    const double *dvec;           // [M]
    int M, nbands, gcd_off;
    int boff[4];
};

static ChebTabs make_tabs(const EmuTabs &e, int N, int Mk) {
    ChebTabs T;
    T.tw = reinterpret_cast<const double2 *>(e.tw);
    T.half = reinterpret_cast<const double2 *>(e.half);
    T.bands = e.bands;
    T.bsub = e.bsub;
    T.dvec = e.dvec;
    T.M = e.M;
    T.Mk = Mk;
    T.nbands = e.nbands;
    T.gcd_off = e.gcd_off;
    T.boff1 = e.boff[1]; T.boff2 = e.boff[2]; T.boff3 = e.boff[3];
    const double kSqPi = 1.7724538509055160272981674833411, kSqPi2 = 1.2533141373155002512078826424055;
    T.fs0 = kSqPi / (2.0 * (double)N);
    T.fs1 = kSqPi2 / (double)N;
    T.bs0 = 1.0 / kSqPi;
    T.bs1 = 0.5 / kSqPi2;
    return T;
}

template <typename F>
static void run_wave(F body) {
    pthread_barrier_init(&g_bar, nullptr, 64);
    std::vector<std::thread> th;
    for (int lane = 0; lane < 64; ++lane) th.emplace_back([=]() { body(lane); });
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&g_bar);
}

template <int R, int NL, int CH>
static void cheb_bwd(const EmuTabs &e, int kind, const double *src, double *dst, double *dst2, long outer, long inner) {
    constexpr int N = 16 * R;
    const ChebTabs T = make_tabs(e, N, 16 * NL);
    const long npairs = inner / 2, tpo = (npairs + 3) / 4, ntiles = tpo * outer;
    std::vector<double2> S(ChebWaveLds<R, NL, CH>::size);
    run_wave([&](int lane) {
        const Lane L = make_lane(lane);
        double2 c[NL];
        auto locate = [&](long tile, long &oc, long &og, bool &valid) {
            const long o = tile / tpo, tb = tile % tpo;
            oc = (o * e.M) * inner + 8 * tb;
            og = (o * N) * inner + 8 * tb;
            valid = 4 * tb + L.p < npairs;
        };
        long oc, og;
        bool valid;
        locate(0, oc, og, valid);
        cheb_bwd_load<NL>(c, src + oc, (unsigned)(inner * 8), valid, L);
        for (long tile = 0; tile < ntiles; ++tile) {
            const bool more = tile + 1 < ntiles;
            long ocn = 0, ogn = 0;
            bool validn = false;
            if (more) locate(tile + 1, ocn, ogn, validn);
            const unsigned rsb = (unsigned)(inner * 8);
            if (!more) { ocn = oc; ogn = og; validn = valid; }
            const unsigned rsbn = more ? rsb : 0u;
            if (kind == 1) {
                cheb_bwd_pass<R, NL, CH, 0, false>(c, S.data(), T, dst + og, rsb, valid, lane, src + oc, rsb, valid);
                cheb_bwd_pass<R, NL, CH, 1, true>(c, S.data(), T, dst2 + og, rsb, valid, lane, src + ocn, rsbn, validn);
            } else if (kind == 2) {
                cheb_bwd_pass<R, NL, CH, 2, true>(c, S.data(), T, dst + og, rsb, valid, lane, src + ocn, rsbn, validn);
            } else {
                cheb_bwd_pass<R, NL, CH, 0, true>(c, S.data(), T, dst + og, rsb, valid, lane, src + ocn, rsbn, validn);
            }
            oc = ocn;
            og = ogn;
            valid = validn;
        }
    });
}

template <int R, int NL, int CH>
static void cheb_fwd(const EmuTabs &e, const double *src, double *dst, long outer, long inner) {
    constexpr int N = 16 * R;
    const ChebTabs T = make_tabs(e, N, 16 * NL);
    const long npairs = inner / 2, tpo = (npairs + 3) / 4, ntiles = tpo * outer;
    std::vector<double2> S(ChebWaveLds<R, NL, CH>::size);
    run_wave([&](int lane) {
        const Lane L = make_lane(lane);
        for (long tile = 0; tile < ntiles; ++tile) {
            const long o = tile / tpo, tb = tile % tpo;
            const bool valid = 4 * tb + L.p < npairs;
            cheb_fwd_tile<R, NL, CH>(src + (o * N) * inner + 8 * tb, dst + (o * e.M) * inner + 8 * tb, (unsigned)(inner * 8), (unsigned)(inner * 8), valid, S.data(), T, lane);
        }
    });
}

// contiguous axis (wave_cheb_contig_kernel): lines [nlines][M] -> [nlines][N]; the staged grid lines are NP doubles apart
template <int R, int NL, int CH>
static void cheb_bwd_contig(const EmuTabs &e, int kind, const double *src, double *dst, double *dst2, long nlines) {
    constexpr int N = 16 * R, NP = N + 2;
    const int M = e.M;
    const ChebTabs T = make_tabs(e, N, 16 * NL);
    const long ntiles = (nlines + 7) / 8;
    std::vector<double2> S(ChebWaveLds<R, NL, CH>::size);
    std::vector<double> stage(8 * NP);
    run_wave([&](int lane) {
        const Lane L = make_lane(lane);
        double2 c[NL];
        cheb_bwd_load_contig<NL>(c, src, (unsigned)(M * 8), 2 * L.p < nlines, L);
        auto copy_out = [&](double *out, long l0) {
            WF_SYNC();
            for (int it = 0; it < (8 * (N / 2) + 63) / 64; ++it) {
                const int ch = it * 64 + lane;
                const int l = ch / (N / 2), j = ch - l * (N / 2);
                if (ch < 8 * (N / 2) && l0 + l < nlines) {
                    out[(l0 + l) * N + 2 * j] = stage[l * NP + 2 * j];
                    out[(l0 + l) * N + 2 * j + 1] = stage[l * NP + 2 * j + 1];
                }
            }
            WF_SYNC();
        };
        for (long tile = 0; tile < ntiles; ++tile) {
            const long l0 = 8 * tile;
            const bool valid = l0 + 2 * L.p < nlines;
            const bool more = tile + 1 < ntiles;
            const long l0n = more ? l0 + 8 : l0;
            const bool validn = l0n + 2 * L.p < nlines;
            const unsigned lsc = (unsigned)(M * 8), lscn = more ? lsc : 0u;
            if (kind == 1) {
                cheb_bwd_pass<R, NL, CH, 0, false, true>(c, S.data(), T, stage.data(), (unsigned)(NP * 8), valid, lane, src + l0 * M, lsc, valid);
                copy_out(dst, l0);
                cheb_bwd_pass<R, NL, CH, 1, true, true>(c, S.data(), T, stage.data(), (unsigned)(NP * 8), valid, lane, src + l0n * M, lscn, validn);
                copy_out(dst2, l0);
            } else {
                if (kind == 2)
                    cheb_bwd_pass<R, NL, CH, 2, true, true>(c, S.data(), T, stage.data(), (unsigned)(NP * 8), valid, lane, src + l0n * M, lscn, validn);
                else
                    cheb_bwd_pass<R, NL, CH, 0, true, true>(c, S.data(), T, stage.data(), (unsigned)(NP * 8), valid, lane, src + l0n * M, lscn, validn);
                copy_out(dst, l0);
            }
        }
    });
}

template <int R, int NL, int CH>
static void cheb_fwd_contig(const EmuTabs &e, const double *src, double *dst, long nlines) {
    constexpr int N = 16 * R;
    const int M = e.M;
    const ChebTabs T = make_tabs(e, N, 16 * NL);
    const long ntiles = (nlines + 7) / 8;
    std::vector<double2> S(ChebWaveLds<R, NL, CH>::size);
    run_wave([&](int lane) {
        const Lane L = make_lane(lane);
        for (long tile = 0; tile < ntiles; ++tile) {
            const long l0 = 8 * tile;
            const bool valid = l0 + 2 * L.p < nlines;
            cheb_fwd_tile<R, NL, CH, true>(src + l0 * N, dst + l0 * M, (unsigned)(N * 8), (unsigned)(M * 8), valid, S.data(), T, lane);
        }
    });
}

// plain complex FFT of 4 interleaved lines through wfft: x [N][4][2] -> X [N][4][2]
template <int R, int SIGN, int CH>
static void fft4(const double *tw, const double *x, double *X) {
    constexpr int N = 16 * R, RQ = R / 4;
    std::vector<double2> xb(WfftBuf<R, CH>::size);
    const double2 *xin = reinterpret_cast<const double2 *>(x);
    double2 *xout = reinterpret_cast<double2 *>(X);
    run_wave([&](int lane) {
        const Lane L = make_lane(lane);
        double2 v[R];
        for (int t = 0; t < R; ++t) v[t] = xin[(L.q + 16 * t) * 4 + L.p];
        wfft<R, SIGN, CH>(v, xb.data(), reinterpret_cast<const double2 *>(tw), L);
        for (int a0 = 0; a0 < 4; ++a0)
            for (int i = 0; i < RQ; ++i) xout[(R * (L.q0 + 4 * a0) + RQ * L.q1 + i) * 4 + L.p] = v[a0 * RQ + i];
        (void)N;
    });
}

// the sizes of csrc/ddh_fftwave.hip (DDH_CHEB_WAVE_SIZES)
#define EMU_CHEB_SIZES(X) X(24, 16) X(12, 8) X(16, 16) X(12, 12) X(8, 8) X(4, 4) X(16, 8) X(8, 4)

extern "C" {
int emu_fft4(int N, int sign, const double *tw, const double *x, double *X) {
    if (N == 384 && sign < 0) { fft4<24, -1, 2>(tw, x, X); return 0; }
    if (N == 384 && sign > 0) { fft4<24, +1, 2>(tw, x, X); return 0; }
    if (N == 256 && sign < 0) { fft4<16, -1, 2>(tw, x, X); return 0; }
    if (N == 256 && sign > 0) { fft4<16, +1, 2>(tw, x, X); return 0; }
    if (N == 192 && sign < 0) { fft4<12, -1, 1>(tw, x, X); return 0; }
    if (N == 192 && sign > 0) { fft4<12, +1, 1>(tw, x, X); return 0; }
    if (N == 128 && sign < 0) { fft4<8, -1, 2>(tw, x, X); return 0; }
    if (N == 128 && sign > 0) { fft4<8, +1, 2>(tw, x, X); return 0; }
    if (N == 64 && sign < 0) { fft4<4, -1, 1>(tw, x, X); return 0; }
    if (N == 64 && sign > 0) { fft4<4, +1, 1>(tw, x, X); return 0; }
    return 1;
}
int emu_cheb_bwd(int N, int M, int kind, const double *tw, const double *half, const double *bsub, const double *dvec,
                 int gcd_off, const double *src, double *dst, double *dst2, long outer, long inner) {
    EmuTabs e;
    memset(&e, 0, sizeof(e));
    e.tw = tw; e.half = half; e.bsub = bsub; e.dvec = dvec; e.M = M; e.gcd_off = gcd_off;
#define EMU_X(RV, NLV) \
    if (N == 16 * RV && M == 16 * NLV) { cheb_bwd<RV, NLV, WaveCH<RV>::ch>(e, kind, src, dst, dst2, outer, inner); return 0; }
    EMU_CHEB_SIZES(EMU_X)
#undef EMU_X
    return 1;
}
int emu_cheb_bwd_contig(int N, int M, int kind, const double *tw, const double *half, const double *bsub, int gcd_off,
                        const double *src, double *dst, long nlines, const double *dvec, double *dst2) {
    EmuTabs e;
    memset(&e, 0, sizeof(e));
    std::vector<double> zero(M, 0.0);
    e.tw = tw; e.half = half; e.bsub = bsub; e.dvec = dvec ? dvec : zero.data(); e.M = M; e.gcd_off = gcd_off;
#define EMU_X(RV, NLV) \
    if (N == 16 * RV && M == 16 * NLV) { cheb_bwd_contig<RV, NLV, WaveCH<RV>::ch>(e, kind, src, dst, dst2, nlines); return 0; }
    EMU_CHEB_SIZES(EMU_X)
#undef EMU_X
    return 1;
}
int emu_cheb_fwd_contig(int N, int M, const double *tw, const double *half, int nbands, const int *boff, const double *bands,
                        const double *src, double *dst, long nlines) {
    EmuTabs e;
    memset(&e, 0, sizeof(e));
    e.tw = tw; e.half = half; e.bands = bands; e.M = M; e.nbands = nbands; e.gcd_off = 1;
    for (int d = 0; d < nbands && d < 4; ++d) e.boff[d] = boff[d];
#define EMU_X(RV, NLV) \
    if (N == 16 * RV && M == 16 * NLV) { cheb_fwd_contig<RV, NLV, WaveCH<RV>::ch>(e, src, dst, nlines); return 0; }
    EMU_CHEB_SIZES(EMU_X)
#undef EMU_X
    return 1;
}
int emu_cheb_fwd(int N, int M, const double *tw, const double *half, int nbands, const int *boff, const double *bands,
                 const double *src, double *dst, long outer, long inner) {
    EmuTabs e;
    memset(&e, 0, sizeof(e));
    e.tw = tw; e.half = half; e.bands = bands; e.M = M; e.nbands = nbands; e.gcd_off = 1;
    for (int d = 0; d < nbands && d < 4; ++d) e.boff[d] = boff[d];
#define EMU_X(RV, NLV) \
    if (N == 16 * RV && M == 16 * NLV) { cheb_fwd<RV, NLV, WaveCH<RV>::ch>(e, src, dst, outer, inner); return 0; }
    EMU_CHEB_SIZES(EMU_X)
#undef EMU_X
    return 1;
}
}

template <int R>
static void rfft_bwd(const double *tw, bool dual, double dsc, double dsc2, const double *src, double *dst, double *dst2,
                     long outer, long inner) {
    constexpr int H = 16 * R, N = 3 * H, M = 2 * H;
    const long npairs = inner / 2, tpo = (npairs + 3) / 4, ntiles = tpo * outer;
    std::vector<double2> S(RfftWaveLds<R>::size);
    run_wave([&](int lane) {
        for (long tile = 0; tile < ntiles; ++tile) {
            const long o = tile / tpo, tb = tile % tpo;
            const bool valid = 4 * tb + (lane & 3) < npairs;
            const double *st = src + (o * M) * inner + 8 * tb;
            double *d1 = dst + (o * N) * inner + 8 * tb, *d2 = dual ? dst2 + (o * N) * inner + 8 * tb : nullptr;
            const double2 *twp = reinterpret_cast<const double2 *>(tw);
            if (dual) rfft_bwd_tile<R, 2>(st, d1, d2, (unsigned)(inner * 8), (unsigned)(inner * 8 * 64), valid, dsc2, S.data(), twp, lane);
            else if (dsc != 0.0) rfft_bwd_tile<R, 1>(st, d1, d2, (unsigned)(inner * 8), (unsigned)(inner * 8 * 64), valid, dsc, S.data(), twp, lane);
            else rfft_bwd_tile<R, 0>(st, d1, d2, (unsigned)(inner * 8), (unsigned)(inner * 8 * 64), valid, 0.0, S.data(), twp, lane);
        }
    });
}

template <int R>
static void rfft_fwd(const double *tw, const double *src, double *dst, long outer, long inner) {
    constexpr int H = 16 * R, N = 3 * H, M = 2 * H;
    const long npairs = inner / 2, tpo = (npairs + 3) / 4, ntiles = tpo * outer;
    std::vector<double2> S(RfftWaveLds<R>::size);
    run_wave([&](int lane) {
        for (long tile = 0; tile < ntiles; ++tile) {
            const long o = tile / tpo, tb = tile % tpo;
            const bool valid = 4 * tb + (lane & 3) < npairs;
            rfft_fwd_tile<R>(src + (o * N) * inner + 8 * tb, dst + (o * M) * inner + 8 * tb, (unsigned)(inner * 8), (unsigned)(inner * 8 * 64), valid, S.data(),
                             reinterpret_cast<const double2 *>(tw), lane);
        }
    });
}

// blocked coefficient layout [kx / B][z][kx % B][ky] (one component, gz planes): backward into the natural grid array
// [z][N][inner], forward back into the blocked layout
template <int R>
static void rfft_blocked(const double *tw, int B, long gz, const double *src, double *grid, double *back, long inner) {
    constexpr int H = 16 * R, N = 3 * H, M = 2 * H;
    const long npairs = inner / 2, tpo = (npairs + 3) / 4, ntiles = tpo * gz;
    const int bsh = (B == 64) ? 1 : (B == 128 ? 2 : 3);
    std::vector<double2> S(RfftWaveLds<R>::size);
    const double2 *twp = reinterpret_cast<const double2 *>(tw);
    run_wave([&](int lane) {
        for (long tile = 0; tile < ntiles; ++tile) {
            const long z = tile / tpo, tb = tile % tpo;
            const bool valid = 4 * tb + (lane & 3) < npairs;
            const long oc = ((long)B * z) * inner + 8 * tb, og = (z * N) * inner + 8 * tb;
            const unsigned rsb = (unsigned)(inner * 8), rsbB = (unsigned)((long)B * gz * inner * 8);
            rfft_bwd_tile<R, 0>(src + oc, grid + og, nullptr, rsb, rsbB, valid, 0.0, S.data(), twp, lane, bsh);
            rfft_fwd_tile<R>(grid + og, back + oc, rsb, rsbB, valid, S.data(), twp, lane, bsh);
        }
        (void)M;
    });
}

extern "C" {
int emu_rfft_blocked(int N, int B, long gz, const double *tw, const double *src, double *grid, double *back, long inner) {
    if (N == 768) { rfft_blocked<16>(tw, B, gz, src, grid, back, inner); return 0; }
    if (N == 384) { rfft_blocked<8>(tw, B, gz, src, grid, back, inner); return 0; }
    return 1;
}
int emu_rfft_bwd(int N, int dual, double dsc, double dsc2, const double *tw, const double *src, double *dst, double *dst2,
                 long outer, long inner) {
    if (N == 768) { rfft_bwd<16>(tw, dual != 0, dsc, dsc2, src, dst, dst2, outer, inner); return 0; }
    if (N == 576) { rfft_bwd<12>(tw, dual != 0, dsc, dsc2, src, dst, dst2, outer, inner); return 0; }
    if (N == 384) { rfft_bwd<8>(tw, dual != 0, dsc, dsc2, src, dst, dst2, outer, inner); return 0; }
    if (N == 192) { rfft_bwd<4>(tw, dual != 0, dsc, dsc2, src, dst, dst2, outer, inner); return 0; }
    return 1;
}
int emu_rfft_fwd(int N, const double *tw, const double *src, double *dst, long outer, long inner) {
    if (N == 768) { rfft_fwd<16>(tw, src, dst, outer, inner); return 0; }
    if (N == 576) { rfft_fwd<12>(tw, src, dst, outer, inner); return 0; }
    if (N == 384) { rfft_fwd<8>(tw, src, dst, outer, inner); return 0; }
    if (N == 192) { rfft_fwd<4>(tw, src, dst, outer, inner); return 0; }
    return 1;
}
}
