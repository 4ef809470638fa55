// Strided-axis Chebyshev transforms, one wavefront per four line pairs (ddh_wavefft.h): kernels and launch.
//
// A workgroup is 8 independent wavefronts that share the read-only tables in LDS (twiddles, half-angle factors,
// conversion bands / back-substitution table / derivative vector) and nothing else: no workgroup barrier after the
// table fill.  At step i the 8 waves of workgroup g work on 8 neighbouring tiles (512 contiguous bytes per row) and a
// wave requests the coefficient rows of its next tile while it transforms the current one.
// Replaces core/transforms.py:715-902 for the strided z axis (the reference: scipy DCT + scale / pad / conversion passes).
#include "ddh_fft_dev.h"
#include "ddh_wavefft.h"

#include <cstdlib>

namespace ddh {

constexpr int WV_WAVES = 8;
#ifndef DDH_WR_WAVES
#define DDH_WR_WAVES 8
#endif
constexpr int WR_WAVES = DDH_WR_WAVES;     // waves per workgroup of the real-FFT kernels

struct WaveArgs {
    const double *src;
    double *dst, *dst2;
    long inner, npairs;
    FastDiv fd_tpo;          // tiles per outer index
    unsigned ntiles, tpw;    // tiles in all, tiles per wave
    int kind;                // backward: 0 plain, 1 dual (plain + derivative pass), 2 conversion solve
    int wsync;               // 1: the waves of a workgroup start every tile together (one s_barrier per tile)
};

template <int KIND, int R, int NL, int CH>      // KIND: 0 backward plain, 1 backward dual, 2 backward conversion, 3 forward
__global__ void __launch_bounds__(64 * WV_WAVES, 2)
wave_cheb_kernel(FftDev p, WaveArgs a) {
    constexpr bool FWD = (KIND == 3);
    extern __shared__ double2 lds[];
    constexpr int N = 16 * R;
    const int M = p.M, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: tile bases live in scalar registers
    double2 *s_tw = lds, *s_half = lds + N;
    double *s_d = reinterpret_cast<double *>(lds + 2 * N);
    // doubles: forward [nbands][M] bands; backward [2][M] back-substitution table, [M] derivative vector
    const int nd = FWD ? p.nbands * M : 3 * M;
    double2 *S = reinterpret_cast<double2 *>(s_d + ((nd + 1) & ~1)) + wave * wf::ChebWaveLds<R, NL, CH>::size;
    for (int i = tid; i < N; i += 64 * WV_WAVES) {
        s_tw[i] = p.tw[i];
        s_half[i] = p.half[i];
    }
    if (FWD) {
        for (int i = tid; i < p.nbands * M; i += 64 * WV_WAVES) s_d[i] = p.bands[i];
    } else {
        for (int i = tid; i < 2 * M; i += 64 * WV_WAVES) s_d[i] = p.bsub ? p.bsub[i] : 0.0;
        for (int i = tid; i < M; i += 64 * WV_WAVES) s_d[2 * M + i] = p.dvec ? p.dvec[i] : 0.0;
    }
    __syncthreads();                    // the only workgroup barrier
    wf::ChebTabs T;
    T.tw = s_tw;
    T.half = s_half;
    T.bands = s_d;
    T.bsub = s_d;
    T.dvec = s_d + 2 * M;
    T.M = M;
    T.Mk = 16 * NL;
    T.nbands = p.nbands;
    T.gcd_off = p.gcd_off;
    { T.boff1 = p.boff[1]; T.boff2 = p.boff[2]; T.boff3 = p.boff[3]; }
    const double kSqPi = 1.7724538509055160272981674833411, kSqPi2 = 1.2533141373155002512078826424055;
    T.fs0 = kSqPi / (2.0 * (double)N);
    T.fs1 = kSqPi2 / (double)N;
    T.bs0 = 1.0 / kSqPi;
    T.bs1 = 0.5 / kSqPi2;
    const wf::Lane L = wf::make_lane(lane);
    const unsigned g = xcd_swizzle(blockIdx.x, gridDim.x);
    const long inner = a.inner;
    // bytes between coefficient rows (kx-band-major state vector: 8 ny doubles)
    const unsigned rsb = (p.ctile_nseg && p.cband) ? (unsigned)(p.ctile_nseg * 512u) : (unsigned)(inner * 8);
    // grid side (the stage array between the z and the x transforms): natural [z][kx][ky], or x-blocked
    // [kx / 64][z][kx % 64][ky] (p.xb = ny): rows 64 ny doubles apart (256 KiB at ny = 512, not one 2 MiB page per row)
    const unsigned rsg = p.xb ? (unsigned)(64u * p.xb * 8u) : rsb;
    // i-th tile of this wave
    auto tile_of = [&](unsigned i) -> unsigned { return (g * a.tpw + i) * WV_WAVES + wave; };
    auto locate = [&](unsigned tile, long &off_c, long &off_g, bool &valid) {
        unsigned o, tb;
        a.fd_tpo.divmod(tile, o, tb);
        long seg = tb;                       // 64-byte segment of the coefficient row [nx][ny] ...
        if (p.ctile_nseg) {           // ... tile-major: [kx / 8][ky / 8][kx % 8] (ddh_cheb_forward_tiled, ddh_fft_set_coeff_tiled)
            const unsigned kxrow = tb / p.ctile_nseg, sg = tb - kxrow * p.ctile_nseg;
            seg = ((long)(kxrow >> 3) * p.ctile_nseg + sg) * 8 + (kxrow & 7);
        }
        const long pair0 = 4L * tb;
        off_c = ((long)o * M) * inner + 8 * seg;
        if (p.ctile_nseg && p.cband) {       // ... kx-band-major: [kx / 8][row][ky / 8][kx % 8]
            const unsigned kxrow = tb / p.ctile_nseg, sg = tb - kxrow * p.ctile_nseg;
            off_c = ((long)o * M) * (64L * p.ctile_nseg) + (long)(kxrow >> 3) * (long)p.cband + 64L * sg + 8L * (kxrow & 7);
        }
        off_g = ((long)o * N) * inner + 2 * pair0;
        if (p.xb) {
            const unsigned nsg = p.xb >> 3, kxrow = tb / nsg, sg = tb - kxrow * nsg;
            off_g = ((long)o * N) * inner + ((long)(kxrow >> 6) * N * 64 + (kxrow & 63)) * p.xb + 8L * sg;
        }
        valid = pair0 + L.p < a.npairs;
    };
    // The 8 waves of the workgroup take 8 neighbouring tiles per step.  With wsync they enter every step together, so
    // that their row requests (8 x 64 contiguous bytes per row) reach the memory controllers at the same time.
    const unsigned step0 = g * a.tpw * WV_WAVES;
    if (step0 >= a.ntiles) return;
    if (FWD) {
        for (unsigned i = 0; i < a.tpw; ++i) {
            if (step0 + i * WV_WAVES >= a.ntiles) break;                 // workgroup-uniform
            if (a.wsync && i > 0) __syncthreads();
            const unsigned tile = tile_of(i);
            if (tile >= a.ntiles) continue;
            long oc, og;
            bool valid;
            locate(tile, oc, og, valid);
            wf::cheb_fwd_tile<R, NL, CH>(a.src + og, a.dst + oc, rsg, rsb, valid, S, T, lane);
        }
    } else {
        double2 c[NL];
        long oc = 0, og = 0;
        bool valid = false;
        const bool have0 = tile_of(0) < a.ntiles;
        if (have0) {
            locate(tile_of(0), oc, og, valid);
            wf::cheb_bwd_load<NL>(c, a.src + oc, rsb, valid, L);
        }
        for (unsigned i = 0; i < a.tpw; ++i) {
            if (step0 + i * WV_WAVES >= a.ntiles) break;                 // workgroup-uniform
            if (a.wsync && i > 0) __syncthreads();
            if (tile_of(i) >= a.ntiles) continue;                        // this wave has run out of tiles (it still syncs)
            const unsigned tn = tile_of(i + 1);
            const bool more = (i + 1 < a.tpw) && (tn < a.ntiles);
            long ocn = oc, ogn = og;
            bool validn = valid;
            if (more) locate(tn, ocn, ogn, validn);
            const unsigned rsbn = more ? rsb : 0u;
            if (KIND == 1) {
                // the plain pass keeps c for the derivative pass
                wf::cheb_bwd_pass<R, NL, CH, 0, false>(c, S, T, a.dst + og, rsg, valid, lane, a.src + oc, rsb, valid);
                wf::cheb_bwd_pass<R, NL, CH, 1, true>(c, S, T, a.dst2 + og, rsg, valid, lane, a.src + ocn, rsbn, validn);
            } else {
                wf::cheb_bwd_pass<R, NL, CH, (KIND == 2 ? 2 : 0), true>(c, S, T, a.dst + og, rsg, valid, lane, a.src + ocn, rsbn,
                                                                      validn);
            }
            oc = ocn;
            og = ogn;
            valid = validn;
        }
    }
}

template <int KIND, int R, int NL, int CH>
static int launch_wave_cheb(const FftDev &d, const WaveArgs &a, unsigned nwg, hipStream_t st) {
    const int N = 16 * R;
    constexpr bool FWD = (KIND == 3);
    const int nd = FWD ? d.nbands * d.M : 3 * d.M;
    const size_t lds = (size_t)2 * N * sizeof(double2) + (size_t)((nd + 1) & ~1) * sizeof(double) +
                       (size_t)WV_WAVES * wf::ChebWaveLds<R, NL, CH>::size * sizeof(double2);
    if (lds > 160 * 1024) return 1;
    auto kern = wave_cheb_kernel<KIND, R, NL, CH>;
    if (lds > 64 * 1024)
        DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WV_WAVES), lds, st, d, a);
    DDH_HIP(hipGetLastError());
    return 0;
}

template <int R, int NL>
static int launch_wave_cheb_kind(int kind, const FftDev &d, const WaveArgs &a, unsigned nwg, hipStream_t st) {
    constexpr int CH = wf::WaveCH<R>::ch;
    switch (kind) {
        case 3: return launch_wave_cheb<3, R, NL, CH>(d, a, nwg, st);
        case 2: return launch_wave_cheb<2, R, NL, CH>(d, a, nwg, st);
        case 1: return launch_wave_cheb<1, R, NL, CH>(d, a, nwg, st);
        default: return launch_wave_cheb<0, R, NL, CH>(d, a, nwg, st);
    }
}

// ---- Chebyshev along the CONTIGUOUS axis (the shell's radial transforms: [lines][192] <-> [lines][128]) ------------------
// The same lane code on lines that are contiguous in memory: a wave takes 8 consecutive lines (4 pairs of lines 2 p, 2 p + 1).
// Its loads follow the lane map directly (sixteen lanes of a row group read 128 contiguous bytes of one line); the grid rows
// of a backward transform leave the FFT scattered over the line (row 2 n / 2 (N - 1 - n) + 1 of FFT slot n), so they are
// written into the wave's LDS region -- laid out like the global lines, NP doubles apart -- and copied out with linear
// 16-byte stores.  The staging area is the region the transform itself has just used for its exchanges: the LDS
// operations of a wave execute in program order.  Replaces core/transforms.py:715-902 for a contiguous axis (the
// workgroup-per-tile kernel of ddh_fft.hip ran these short lines at 0.19-0.27 of the HBM rate).
constexpr int WC_WAVES = 4;

struct ContigArgs {
    const double *src;
    double *dst, *dst2;
    long nlines;             // even
    unsigned ntiles, tpw;    // tiles of 8 lines, tiles per wave
    int kind;                // backward: 0 plain, 1 dual (plain + derivative pass), 2 conversion solve
};

template <int R, int NL, int CH>
struct ChebContigLds {
    static constexpr int NP = 16 * R + 2;                       // doubles between staged lines (16-byte aligned line starts)
    static constexpr int a = wf::ChebWaveLds<R, NL, CH>::size, b = 4 * NP;
    static constexpr int size = a > b ? a : b;                  // double2 per wave
};

template <int KIND, int R, int NL, int CH>      // KIND: 0 backward plain, 1 backward dual, 2 backward conversion, 3 forward
__global__ void __launch_bounds__(64 * WC_WAVES, 2)
wave_cheb_contig_kernel(FftDev p, ContigArgs a) {
    constexpr bool FWD = (KIND == 3);
    extern __shared__ double2 lds[];
    constexpr int N = 16 * R, NP = ChebContigLds<R, NL, CH>::NP;
    const int M = p.M, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2 *s_tw = lds, *s_half = lds + N;
    double *s_d = reinterpret_cast<double *>(lds + 2 * N);
    const int nd = FWD ? p.nbands * M : 3 * M;
    double2 *S = reinterpret_cast<double2 *>(s_d + ((nd + 1) & ~1)) + wave * ChebContigLds<R, NL, CH>::size;
    for (int i = tid; i < N; i += 64 * WC_WAVES) {
        s_tw[i] = p.tw[i];
        s_half[i] = p.half[i];
    }
    if (FWD) {
        for (int i = tid; i < p.nbands * M; i += 64 * WC_WAVES) s_d[i] = p.bands[i];
    } else {
        for (int i = tid; i < 2 * M; i += 64 * WC_WAVES) s_d[i] = p.bsub ? p.bsub[i] : 0.0;
        for (int i = tid; i < M; i += 64 * WC_WAVES) s_d[2 * M + i] = (KIND == 1 && p.dvec) ? p.dvec[i] : 0.0;
    }
    __syncthreads();                    // the only workgroup barrier
    wf::ChebTabs T;
    T.tw = s_tw;
    T.half = s_half;
    T.bands = s_d;
    T.bsub = s_d;
    T.dvec = s_d + 2 * M;
    T.M = M;
    T.Mk = 16 * NL;
    T.nbands = p.nbands;
    T.gcd_off = p.gcd_off;
    { T.boff1 = p.boff[1]; T.boff2 = p.boff[2]; T.boff3 = p.boff[3]; }
    const double kSqPi = 1.7724538509055160272981674833411, kSqPi2 = 1.2533141373155002512078826424055;
    T.fs0 = kSqPi / (2.0 * (double)N);
    T.fs1 = kSqPi2 / (double)N;
    T.bs0 = 1.0 / kSqPi;
    T.bs1 = 0.5 / kSqPi2;
    const wf::Lane L = wf::make_lane(lane);
    const unsigned g = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned lsc = (unsigned)(M * 8), lsg = (unsigned)(N * 8);      // bytes between coefficient / grid lines
    auto tile_of = [&](unsigned i) -> unsigned { return (g * a.tpw + i) * WC_WAVES + wave; };
    if (FWD) {
        for (unsigned i = 0; i < a.tpw; ++i) {
            const unsigned tile = tile_of(i);
            if (tile >= a.ntiles) break;
            const long l0 = 8L * tile;
            const bool valid = l0 + 2 * L.p < a.nlines;
            wf::cheb_fwd_tile<R, NL, CH, true>(a.src + l0 * N, a.dst + l0 * M, lsg, lsc, valid, S, T, lane);
        }
    } else {
        double2 c[NL];
        if (tile_of(0) >= a.ntiles) return;
        {
            const long l0 = 8L * tile_of(0);
            wf::cheb_bwd_load_contig<NL>(c, a.src + l0 * M, lsc, l0 + 2 * L.p < a.nlines, L);
        }
        double *stage = reinterpret_cast<double *>(S);
        // staged lines -> global, 16 bytes per lane and step
        auto copy_out = [&](double *out, long l0) {
            WF_SYNC();
#pragma unroll
            for (int it = 0; it < (8 * (N / 2) + 63) / 64; ++it) {
                const int ch = it * 64 + lane;
                const int l = ch / (N / 2), j = ch - l * (N / 2);
                if (ch < 8 * (N / 2) && l0 + l < a.nlines)
                    *reinterpret_cast<double2 *>(out + (long)l * N + 2 * j) = *reinterpret_cast<const double2 *>(stage + l * NP + 2 * j);
            }
            WF_SYNC();
        };
        for (unsigned i = 0; i < a.tpw; ++i) {
            const unsigned tile = tile_of(i);
            if (tile >= a.ntiles) break;
            const long l0 = 8L * tile;
            const bool valid = l0 + 2 * L.p < a.nlines;
            const unsigned tn = tile_of(i + 1);
            const bool more = (i + 1 < a.tpw) && (tn < a.ntiles);
            const long l0n = more ? 8L * tn : l0;
            const bool validn = l0n + 2 * L.p < a.nlines;
            const unsigned lscn = more ? lsc : 0u;
            if (KIND == 1) {
                // the plain pass keeps c for the derivative pass (field and d/dz from one read of the coefficients)
                wf::cheb_bwd_pass<R, NL, CH, 0, false, true>(c, S, T, stage, (unsigned)(NP * 8), valid, lane, a.src + l0 * M, lsc, valid);
                copy_out(a.dst + l0 * N, l0);
                wf::cheb_bwd_pass<R, NL, CH, 1, true, true>(c, S, T, stage, (unsigned)(NP * 8), valid, lane, a.src + l0n * M, lscn, validn);
                copy_out(a.dst2 + l0 * N, l0);
            } else {
                if (KIND == 2)
                    wf::cheb_bwd_pass<R, NL, CH, 2, true, true>(c, S, T, stage, (unsigned)(NP * 8), valid, lane, a.src + l0n * M, lscn, validn);
                else
                    wf::cheb_bwd_pass<R, NL, CH, 0, true, true>(c, S, T, stage, (unsigned)(NP * 8), valid, lane, a.src + l0n * M, lscn, validn);
                copy_out(a.dst + l0 * N, l0);
            }
        }
    }
}

template <int KIND, int R, int NL, int CH>
static int launch_wave_cheb_contig(const FftDev &d, const ContigArgs &a, unsigned nwg, hipStream_t st) {
    const int N = 16 * R;
    constexpr bool FWD = (KIND == 3);
    const int nd = FWD ? d.nbands * d.M : 3 * d.M;
    const size_t lds = (size_t)2 * N * sizeof(double2) + (size_t)((nd + 1) & ~1) * sizeof(double) +
                       (size_t)WC_WAVES * ChebContigLds<R, NL, CH>::size * sizeof(double2);
    if (lds > 160 * 1024) return 1;
    auto kern = wave_cheb_contig_kernel<KIND, R, NL, CH>;
    if (lds > 64 * 1024)
        DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WC_WAVES), lds, st, d, a);
    DDH_HIP(hipGetLastError());
    return 0;
}

template <int R, int NL>
static int launch_wave_cheb_contig_kind(int kind, const FftDev &d, const ContigArgs &a, unsigned nwg, hipStream_t st) {
    constexpr int CH = wf::WaveCH<R>::ch;
    switch (kind) {
        case 3: return launch_wave_cheb_contig<3, R, NL, CH>(d, a, nwg, st);
        case 2: return launch_wave_cheb_contig<2, R, NL, CH>(d, a, nwg, st);
        case 1: return launch_wave_cheb_contig<1, R, NL, CH>(d, a, nwg, st);
        default: return launch_wave_cheb_contig<0, R, NL, CH>(d, a, nwg, st);
    }
}

// The (grid, coefficient) sizes the wave Chebyshev kernels are instantiated for: N = 16 R grid points, M = 16 NL modes.
// 3/2 dealiasing of 128 / 256 modes (the configurations' radial / vertical bases), no dealiasing (N = M = 64 .. 256) and
// factor-two padding; everything else takes the workgroup-per-tile kernel of ddh_fft.hip.
#define DDH_CHEB_WAVE_SIZES(X) X(24, 16) X(12, 8) X(16, 16) X(12, 12) X(8, 8) X(4, 4) X(16, 8) X(8, 4)

// 0 = launched, 1 = shape not covered
int wave_contig_try(int mode, const FftDev &d, const double *src, double *dst, long outer, double *dst2, hipStream_t st) {
    static const int on = getenv("DDH_CHEB_CONTIG_WAVE") ? atoi(getenv("DDH_CHEB_CONTIG_WAVE")) : 1;
    if (!on || d.dbg || d.prof || d.xb || d.ctile_nseg) return 1;
    if (mode != CHEB_FWD && mode != CHEB_BWD) return 1;
    if ((outer & 1) || outer < 2) return 1;
    if (src == dst || src == dst2) return 1;
    ContigArgs a;
    a.src = src;
    a.dst = dst;
    a.dst2 = dst2;
    a.nlines = outer;
    const unsigned long ntiles = ((unsigned long)outer + 7) / 8;
    if (ntiles > 0x7fffffffUL) return 1;
    a.ntiles = (unsigned)ntiles;
    a.kind = 0;
    if (mode == CHEB_BWD) {
        if (dst2) {
            if (!(d.bsub && d.bsub_order == 1 && (d.gcd_off == 1 || d.gcd_off == 2) && d.dvec)) return 1;
            a.kind = 1;
        } else if (d.nbands > 0) {
            if (!(d.bsub && d.bsub_order == 1 && (d.gcd_off == 1 || d.gcd_off == 2))) return 1;
            a.kind = 2;
        }
    } else if (dst2) {
        return 1;
    }
    static const int env_tpw = getenv("DDH_CHEB_CONTIG_TPW") ? atoi(getenv("DDH_CHEB_CONTIG_TPW")) : 0;
    unsigned tpw = (unsigned)(env_tpw > 0 ? env_tpw : 4);
    while (tpw > 1 && ntiles / ((unsigned long)tpw * WC_WAVES) < 2048) tpw /= 2;
    a.tpw = tpw;
    const unsigned nwg = (unsigned)((ntiles + (unsigned long)tpw * WC_WAVES - 1) / ((unsigned long)tpw * WC_WAVES));
    const int kind = (mode == CHEB_FWD) ? 3 : a.kind;
#define DDH_X(RV, NLV) \
    if (d.N == 16 * RV && d.M == 16 * NLV) return launch_wave_cheb_contig_kind<RV, NLV>(kind, d, a, nwg, st);
    DDH_CHEB_WAVE_SIZES(DDH_X)
#undef DDH_X
    return 1;
}

// ---- real Fourier, 3/2 dealiasing (N = 48 R, M = 32 R): RKIND 0 backward, 1 backward differentiated, 2 backward dual
// (plain + differentiated), 3 forward
template <int RKIND, int R>
__global__ void __launch_bounds__(64 * WR_WAVES, 2)
wave_rfft_kernel(FftDev p, WaveArgs a) {
    extern __shared__ double2 lds[];
    constexpr int N = 48 * R, M = 32 * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2 *s_tw = lds;
    double2 *S = lds + N + wave * wf::RfftWaveLds<R>::size;
    for (int i = tid; i < N; i += 64 * WR_WAVES) s_tw[i] = p.tw[i];
    __syncthreads();                    // the only workgroup barrier
    const unsigned g = xcd_swizzle(blockIdx.x, gridDim.x);
    const long inner = a.inner;
    const unsigned rsb = (unsigned)(inner * 8);
    const int p4 = lane & 3;
    for (unsigned i = 0; i < a.tpw; ++i) {
        if ((g * a.tpw + i) * WR_WAVES >= a.ntiles) break;               // workgroup-uniform
        if (a.wsync && i > 0) __syncthreads();
        const unsigned tile = (g * a.tpw + i) * WR_WAVES + wave;
        if (tile >= a.ntiles) continue;
        unsigned o, tb;
        a.fd_tpo.divmod(tile, o, tb);
        const long pair0 = 4L * tb;
        long oc = ((long)o * M) * inner + 2 * pair0;
        const long og = ((long)o * N) * inner + 2 * pair0;
        unsigned rsb64 = 64u * rsb;
        int bsh = 1;
        if (p.xb) {
            // coefficient side = the stage array [comp][kx / B][z][kx % B][ky], o = comp * gz + z (p.xb = gz; B = 64 on one
            // rank, nx / P in a sharded run: the layout [p][z][nx / P][ky] the exchange delivers / takes)
            const unsigned B = p.xbB ? p.xbB : 64u;
            unsigned comp, z;
            if (p.xbwn) {                                // a window of every component's planes (ddh_fft_set_stage_window)
                comp = o / p.xbwn;
                z = p.xbw0 + (o - comp * p.xbwn);
            } else {
                comp = o / p.xb;
                z = o - comp * p.xb;
            }
            oc = ((long)comp * M * p.xb + (long)B * z) * inner + 2 * pair0;
            rsb64 = (unsigned)(B * p.xb) * rsb;
            bsh = (B == 64u) ? 1 : ((B == 128u) ? 2 : 3);
        }
        const bool valid = pair0 + p4 < a.npairs;
        if (RKIND == 3) wf::rfft_fwd_tile<R>(a.src + og, a.dst + oc, rsb, rsb64, valid, S, s_tw, lane, bsh);
        else wf::rfft_bwd_tile<R, (RKIND == 3 ? 0 : RKIND)>(a.src + oc, a.dst + og, (RKIND == 2) ? a.dst2 + og : nullptr, rsb, rsb64, valid,
                                                            (RKIND == 2) ? p.dscale2 : p.dscale, S, s_tw, lane, bsh);
    }
}

template <int RKIND, int R>
static int launch_wave_rfft(const FftDev &d, const WaveArgs &a, unsigned nwg, hipStream_t st) {
    const size_t lds = ((size_t)48 * R + (size_t)WR_WAVES * wf::RfftWaveLds<R>::size) * sizeof(double2);
    if (lds > 160 * 1024) return 1;
    auto kern = wave_rfft_kernel<RKIND, R>;
    if (lds > 64 * 1024)
        DDH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * WR_WAVES), lds, st, d, a);
    DDH_HIP(hipGetLastError());
    return 0;
}

template <int R>
static int launch_wave_rfft_kind(int rk, const FftDev &d, const WaveArgs &a, unsigned nwg, hipStream_t st) {
    switch (rk) {
        case 3: return launch_wave_rfft<3, R>(d, a, nwg, st);
        case 2: return launch_wave_rfft<2, R>(d, a, nwg, st);
        case 1: return launch_wave_rfft<1, R>(d, a, nwg, st);
        default: return launch_wave_rfft<0, R>(d, a, nwg, st);
    }
}

// Returns 0 when the transform was launched here, 1 when the shape is not covered (the caller then uses the
// workgroup-per-tile kernel of ddh_fft.hip), < 0 on error.
int wave_axis_try(int mode, const FftDev &d, const double *src, double *dst, long outer, long inner, double *dst2,
                  const double *dvec, double dscale, double dscale2, hipStream_t st) {
    static const int mask = getenv("DDH_FFT_WAVE") ? atoi(getenv("DDH_FFT_WAVE")) : 3;     // bit 0: Chebyshev, bit 1: real FFT
    // tiles per wave / per-tile workgroup sync: measured per kernel at 3 x 256 x 512^2 and 1152 x 512 x 512
    // (tools/bench_strided.py, profiles/r3_strided_sweep.txt); the environment overrides all of them
    static const int env_tpw = getenv("DDH_FFT_TPW") ? atoi(getenv("DDH_FFT_TPW")) : 0;
    static const int env_wsync = getenv("DDH_FFT_WSYNC") ? atoi(getenv("DDH_FFT_WSYNC")) : -1;
    if (d.dbg || d.prof) return 1;
    if (d.xb) {
        // x-blocked stage layout: shapes it is defined for (the caller falls back to an error, never to another layout)
        if (mode == CHEB_FWD || mode == CHEB_BWD) {
            if ((d.xb & 7) || inner % d.xb || (inner / d.xb) % 64) return 1;
        } else if (mode == RFFT_FWD || mode == RFFT_BWD) {
            const unsigned long B = d.xbB ? d.xbB : 64UL;
            if (!(B == 64 || B == 128 || B == 256)) return 1;
            // (outer: whole components of gz planes each, or of the xbwn planes of a window)
            if (d.xbwn && (d.xbw0 + d.xbwn > d.xb)) return 1;
            if (outer % (d.xbwn ? d.xbwn : d.xb) || d.M % B || (unsigned long)d.xb * B * (unsigned long)inner * 8UL * (unsigned long)(d.M / B) >= 0xffffffffUL) return 1;
        } else {
            return 1;
        }
    }
    if (d.ctile_nseg && ((mode != CHEB_FWD && mode != CHEB_BWD) || inner % (8L * d.ctile_nseg) || (inner / (8L * d.ctile_nseg)) % 8)) return 1;
    const bool cheb = (mode == CHEB_FWD || mode == CHEB_BWD), rfft = (mode == RFFT_FWD || mode == RFFT_BWD);
    if (!cheb && !rfft) return 1;
    if ((cheb && !(mask & 1)) || (rfft && !(mask & 2))) return 1;
    if (inner < 2 || (inner & 1)) return 1;
    if (cheb) {
        bool have = false;
#define DDH_X(RV, NLV) have = have || (d.N == 16 * RV && d.M == 16 * NLV);
        DDH_CHEB_WAVE_SIZES(DDH_X)
#undef DDH_X
        if (!have) return 1;
    }
    // real FFT with 3/2 padding: N = 48 R, M = 32 R for the instantiated R = 4, 8, 12, 16 (128 .. 512 modes)
    if (rfft && !(d.N % 48 == 0 && 3 * d.M == 2 * d.N && (d.N == 192 || d.N == 384 || d.N == 576 || d.N == 768))) return 1;
    if (rfft && d.K != d.M / 2 - 1) return 1;
    const long npairs = inner / 2;
    const long tpo = (npairs + 3) / 4;
    const unsigned long ntiles = (unsigned long)tpo * (unsigned long)outer;
    if (ntiles > 0x7fffffffUL) return 1;
    if ((unsigned long)d.N * (unsigned long)inner * 8UL >= 0xffffffffUL) return 1;     // 32-bit row offsets inside a tile
    WaveArgs a;
    a.src = src;
    a.dst = dst;
    a.dst2 = dst2;
    a.inner = inner;
    a.npairs = npairs;
    a.fd_tpo.set((unsigned)tpo);
    a.ntiles = (unsigned)ntiles;
    a.kind = 0;
    a.wsync = 0;
    FftDev dd = d;
    dd.dvec = dvec;
    dd.dscale = dscale;
    dd.dscale2 = dscale2;
    if (mode == CHEB_BWD) {
        if (dst2) {
            if (!(d.bsub && d.bsub_order == 1 && (d.gcd_off == 1 || d.gcd_off == 2) && dvec)) return 1;
            a.kind = 1;
        } else if (d.nbands > 0) {
            if (!(d.bsub && d.bsub_order == 1 && (d.gcd_off == 1 || d.gcd_off == 2))) return 1;
            a.kind = 2;
        }
    }
    int def_tpw = 1, def_wsync = 0;
    if (mode == CHEB_FWD) {
        def_tpw = 4;
        def_wsync = 1;
    } else if (mode == CHEB_BWD && a.kind != 1) {
        def_tpw = 2;
    }
    a.wsync = env_wsync >= 0 ? env_wsync : def_wsync;
    unsigned tpw = (unsigned)(env_tpw > 0 ? env_tpw : def_tpw);
    const unsigned long wpg = rfft ? WR_WAVES : WV_WAVES;
    while (tpw > 1 && ntiles / ((unsigned long)tpw * wpg) < 1024) tpw /= 2;   // several rounds of workgroups
    a.tpw = tpw;
    const unsigned nwg = (unsigned)((ntiles + (unsigned long)tpw * wpg - 1) / ((unsigned long)tpw * wpg));
    if (rfft) {
        if (dst2 && dscale != 0.0) return 1;                    // the dual entry point transforms plainly into dst
        const int rk = (mode == RFFT_FWD) ? 3 : (dst2 ? 2 : (dscale != 0.0 ? 1 : 0));
        switch (d.N / 48) {
            case 16: return launch_wave_rfft_kind<16>(rk, dd, a, nwg, st);
            case 12: return launch_wave_rfft_kind<12>(rk, dd, a, nwg, st);
            case 8: return launch_wave_rfft_kind<8>(rk, dd, a, nwg, st);
            case 4: return launch_wave_rfft_kind<4>(rk, dd, a, nwg, st);
        }
        return 1;
    }
    const int ck = (mode == CHEB_FWD) ? 3 : a.kind;
#define DDH_X(RV, NLV) \
    if (d.N == 16 * RV && d.M == 16 * NLV) return launch_wave_cheb_kind<RV, NLV>(ck, dd, a, nwg, st);
    DDH_CHEB_WAVE_SIZES(DDH_X)
#undef DDH_X
    return 1;
}

}  // namespace ddh
