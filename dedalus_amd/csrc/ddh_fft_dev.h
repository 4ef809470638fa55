// Device-side building blocks shared by the transform kernels (ddh_fft.hip) and the wave-per-line
// grid stage (ddh_gridwave.hip): plan descriptor, complex helpers, in-register DFT butterflies.
#pragma once
#include "ddh_common.h"
#include "ddh_butterfly.h"

namespace ddh {


enum FftKind { K_RFFT = 0, K_CHEB = 1, K_CFFT = 2 };
enum FftMode { RFFT_FWD = 0, RFFT_BWD, CHEB_FWD, CHEB_BWD, CFFT_FWD, CFFT_BWD };

constexpr int MAX_RADIX_PASSES = 16;
constexpr int MAX_BANDS = 4;

// division by a runtime constant without the ~40-instruction software divide
struct FastDiv {
    unsigned d, m, s;
    __host__ void set(unsigned dd) {
        d = dd ? dd : 1;
        if (d == 1) { m = 0; s = 0; return; }
        s = 0;
        while ((1ull << s) < d) ++s;
        m = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
    }
    __device__ __forceinline__ unsigned div(unsigned n) const {
        return d == 1 ? n : (unsigned)(((unsigned long long)__umulhi(n, m) + n) >> s);
    }
    __device__ __forceinline__ void divmod(unsigned n, unsigned &q, unsigned &r) const {
        q = div(n);
        r = n - q * d;
    }
};

struct FftDev {
    FastDiv fdN, fdM, fdMh, fdK1, fdB, fd_nb[MAX_RADIX_PASSES], fd_ns[MAX_RADIX_PASSES];
    int N;       // grid size = FFT length
    int M;       // coefficient size
    int K;       // Fourier: largest retained wavenumber
    int nradix;
    int radix[MAX_RADIX_PASSES];
    const double2 *tw;     // exp(-2 pi i q / N)
    const double2 *half;   // exp(-i pi k / (2N))
    const double *fscale;  // Chebyshev forward scale per k (includes (-1)^k)
    const double *bscale;  // Chebyshev backward scale per k
    int nbands;
    int gcd_off;           // gcd of the non-zero band offsets (independent back-substitution chains)
    int boff[MAX_BANDS];
    const double *bands;   // [nbands][M]
    const double *bsub;    // [3][M] back-substitution table (1/diag, bands over diag by chain distance) or null
    int bsub_order;        // 1: only the nearest chain neighbour enters (first-order recurrence: wavefront scan), 2: two
    int B;                 // line pairs per workgroup
    double dscale;         // RFFT_BWD: != 0 differentiates along the axis while loading (2 pi / L)
    double *dst2;          // RFFT_BWD dual output: second destination (null = single output) ...
    double dscale2;        // ... transformed with this derivative scale
    const double *dvec;    // CHEB_BWD dual output: [M] superdiagonal of the derivative operator (second pass input = dvec[k] c[k+1])
    int spread_s, spread_c; // strided kernels: workgroup spreading over the address range (see kernel)
    int dbg;               // timing ablations (debug): 1 skip butterfly math, 2 skip FFT passes, 4 skip unpack, 8 skip products
    int rot;               // fused kernel: rotate the butterfly->wave assignment per workgroup
    int twdirect;          // 1: full twiddle table in LDS (N entries) instead of the two-level table
    int ld;                // LDS leading dimension of the FFT buffer (>= N)
    unsigned long long *prof;   // optional phase timing (debug): [load, fft, store, count]
    unsigned xb;                // != 0: x-blocked stage layout on the INTERMEDIATE side of a strided wave transform
    unsigned xbw0, xbwn;        // xbwn != 0: the launch covers planes xbw0 .. xbw0 + xbwn of EVERY component (a window of this
                                // rank's z planes: outer = components x xbwn), the other side holding xbwn planes per component
    unsigned xbB;               // rows per block of that layout on the coefficient side of a real-Fourier transform (0 = 64)
                                // (ddh_fft_set_stage_layout): Chebyshev plans: row length ny; real-FFT plans: z planes gz
    unsigned long cband;        // != 0 (with ctile_nseg): the tile-major coefficient rows are kx-band-major, [kx / 8][row][ky / 8][8][8]:
                                // cband = doubles between the bands of 8 storage rows (all rows of the state vector x 8 ny);
                                // the rows of a line are then 8 ny doubles apart (ddh_fft_set_coeff_tiled)
    unsigned ctile_nseg;        // != 0: the coefficient rows [nx][ny] are written tile-major, ctile_nseg = ny / 8 64-byte
                                // segments per storage row (ddh_cheb_forward_tiled; wave kernel only)
};

struct FftPlan : HandleBase {
    FftDev dev;
    int tkind;
    void *d_tw = nullptr, *d_half = nullptr, *d_fscale = nullptr, *d_bscale = nullptr, *d_bands = nullptr;
    // ddh_rfft_bilinear_fused: the plan of the wider grid the fused stage works on when this plan's grid size has no
    // wave kernel (see there); 0 = none yet, owned by this plan
    ddh_handle fused_alt = 0;
    ~FftPlan() override {
        if (fused_alt) (void)ddh_destroy(fused_alt);
        (void)hipFree(d_tw);
        (void)hipFree(d_half);
        (void)hipFree(d_fscale);
        (void)hipFree(d_bscale);
        (void)hipFree(d_bands);
    }
};

constexpr int FUSED_NA = 3, FUSED_NC = 4, FUSED_NB = 12, FUSED_TERMS = 32;
constexpr int FUSED_LOADS = FUSED_NA + FUSED_TERMS;
constexpr int FUSED_T = 256;

struct FusedArgs {
    const double *src[FUSED_LOADS];   // line arrays in load order: the `a` operands first
    double dscale[FUSED_LOADS];       // != 0: differentiate along the axis while unpacking (2 pi / L)
    double *out[FUSED_NC];
    double coef[FUSED_TERMS];
    short tbeg[FUSED_LOADS + 1];      // terms [tbeg[l], tbeg[l+1]) multiply load l
    short bbeg[FUSED_LOADS + 1];      // batch i transforms loads [bbeg[i], bbeg[i+1]) together
    signed char flush[FUSED_LOADS];   // per batch: >= 0: result `flush` is complete after it
    signed char ia[FUSED_TERMS];
    int na, nbatch;
};

}  // namespace ddh
