// Second-generation wave-per-line fused grid stage along the contiguous real-Fourier axis (gfx950):
//
//     out[ic] = forward_rfft( sum_t coef_t * backward_rfft(a[ia_t]) * backward_rfft(b[ib_t]) )
//
// Same contract, argument block and launch shape as gw::gridwave_bilinear_kernel (ddh_gridwave.hip); the transforms are
// the H = C x 8 x 8 decomposition of ddh_gridwave2.h: spectrum values built from direct + lane-reversed global loads (no
// LDS staging), the radix-C pass on the coefficient side, 34 % fewer LDS stores per line.  The grid-point <-> (lane,
// register) map is the same for every operand, which is all the point-wise products need.
//
// Replaces the reference's backward transforms + DotProduct / MultiplyFields + forward transform along the last axis
// (core/transforms.py:469-565, core/arithmetic.py:666-674, 855-866).
#include "ddh_fft_dev.h"
#include "ddh_gridwave2.h"

#include <algorithm>
#include <cstdlib>

namespace ddh {

namespace gw2 {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) d2v *gptr;

__device__ __forceinline__ void gstore16(double *p, double2 v) {
    d2v r;
    r.x = v.x;
    r.y = v.y;
    *(gptr)(p) = r;
}

// Buffer descriptor of one coefficient line, valid for the retained wavenumbers k <= K only: loads of anything beyond
// return zero from the hardware range check, every load address is a lane offset plus an instruction immediate.
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

// the loads of one operand line: 2 x NT wave instructions of 1 KiB, the second set lane-reversed (the mirror pairs)
template <int NT>
__device__ __forceinline__ void issue_loads(Loads<NT> &ld, const double *line, int lane, int K) {
    const __amdgpu_buffer_rsrc_t r = line_rsrc(line, K);
#pragma unroll
    for (int t = 0; t < NT; ++t) ld.d[t] = bload16(r, 16 * lane + 1024 * t);
#pragma unroll
    for (int t = 0; t < NT; ++t) ld.m[t] = bload16(r, 16 * (64 - lane) + 1024 * t);
}

// the same loads as LDS-DMA (buffer_load_dwordx4 ... lds): lane l's 16 bytes land at st + 64 * t + l, no destination
// registers; the data are valid for ds_read after the wave's own s_waitcnt vmcnt(0)
typedef __attribute__((address_space(3))) void *ldsptr;
// Hand-issued (inline asm): the compiler orders every LDS read behind an LDS-DMA load it knows about (s_waitcnt vmcnt(0)
// before the first ds_read that follows, whatever it addresses), which would expose the whole HBM latency of the prefetch;
// these loads are invisible to its counters and are waited for explicitly (s_waitcnt vmcnt(0) before build_z_staged).
// Its own vmcnt waits stay safe: memory operations retire in order, an uncounted load can only make a wait longer.
template <int NT>
__device__ __forceinline__ void issue_loads_dma(double2 *st, const double *line, int lane, int K) {
    const unsigned long long a = (unsigned long long)line;
    u4v rs;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    rs.z = (unsigned)((K + 1) * 16);
    rs.w = 0x00020000u;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(ldsptr)st);   // wave-uniform LDS address
    const int vd = 16 * lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        unsigned keep;
        const unsigned dst = base + 1024u * t, soff = 1024u * t;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(vd), "s"(rs), "s"(dst), "s"(soff)
                     : "memory");
    }
}
// wait until at most the NT loads of the NEWEST operand are outstanding / until nothing is
template <int NT>
__device__ __forceinline__ void wait_older_loads() {
    static_assert(NT == 2 || NT == 4, "vmcnt immediates of the instantiated sizes");
    if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}

// WAVES lines (wavefronts) per workgroup.  DMA: operand pairs staged in LDS by LDS-DMA instead of in registers (frees the
// 8 NT prefetch registers for the register twiddles; needs every stored pair in range, K + 1 == 64 NT)
template <int C, int NT, int WAVES, bool TWREG, bool DMA = false>
__global__ void __launch_bounds__(64 * WAVES, 2)
gridwave2_bilinear_kernel(FftDev p, FusedArgs f, long nlines) {
    constexpr int GW_T = 64 * WAVES;
    using G = G2<C>;
    constexpr int STGB = Staging<NT>::SIZE;             // one staging buffer (double2); two per wave: the pairs of the next
    constexpr int STG = DMA ? 2 * STGB : 1;             // TWO operands are in flight while one is transformed
    extern __shared__ double2 lds[];
    __shared__ double2 s_stage[WAVES][STG];
    double2 *tw = lds;                                   // twiddle tables (G2<C>::T_*)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2 *wb = lds + G::TW + wave * G::LDW;          // this wave's exchange buffer
    double2 *st = s_stage[__builtin_amdgcn_readfirstlane(wave)];     // ... and its staging area
    if (DMA) {
        // the pair the loads never bring (st[64 NT] = pair 64 NT, lane 0's mirror of the last block: beyond the line) is zero
        for (int i = lane; i < STG; i += 64) st[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i < G::TW; i += GW_T) tw[i] = p.tw[G::table_q(i)];
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
    double2 t64r[7];
    if (TWREG) {
#pragma unroll
        for (int i = 0; i < 7; ++i) t64r[i] = tw[G::T_64 + (lane & 7) * (i + 1)];
    }
    // LDS-DMA pipeline (DMA): the operand sequence of this wave -- every operand of every line it takes -- is fetched two
    // operands ahead into two alternating staging buffers, across line boundaries
    const long line0 = (long)blockIdx.x * WAVES + wave, lstride = (long)gridDim.x * WAVES;
    long pf_line = line0;
    int pf_l = 0, cur_buf = 0;
    bool pf_newer = false;
    auto prefetch = [&](int buf) -> bool {
        const bool live = pf_line < nlines;
        if (live) issue_loads_dma<NT>(st + buf * STGB, s_src[pf_l] + pf_line * (long)M, lane, K);
        if (++pf_l == nloads) {
            pf_l = 0;
            pf_line += lstride;
        }
        return live;
    };
    if constexpr (DMA) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the zero fill of the staging buffers)
        prefetch(0);
        pf_newer = prefetch(1);
    }
#pragma unroll 1
  for (long line = line0; line < nlines; line += lstride) {
    const long off = line * (long)M;

    Loads<DMA ? 1 : NT> ld;
    if constexpr (!DMA) issue_loads<NT>(ld, s_src[0] + off, lane, K);
    // One backward transform: operand l (already staged) -> grid values g; prefetches the next operand(s).
    auto backward = [&](int l, double2 *g) {
        int ln = lane;
        WF_OPAQUE_LANE(ln);
        double2 z[C];
        const double ds = s_dscale[l];
        if constexpr (DMA) {
            // The loads of this operand were issued two transforms ago, those of the next one a transform ago: wait for
            // the older set only -- "at most NT memory operations outstanding".  That is enough whatever the stores of a
            // forward transform in between do (they share the counter and retire out of order with loads): loads retire
            // in order among themselves, so while any load of the older set is pending all NT of the newer set are too,
            // i.e. more than NT operations.  When no newer set was issued (the wave's last operand): wait for everything.
            if (pf_newer) wait_older_loads<NT>();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WF_SYNC();
            const double2 *cur = st + cur_buf * STGB;
            if (ds != 0.0) build_z_staged<C, NT, true>(cur, ds, tw, ln, z);
            else build_z_staged<C, NT, false>(cur, ds, tw, ln, z);
            WF_SYNC();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and have been read: the buffer is free again
            pf_newer = prefetch(cur_buf);                            // operand + 2 into the buffer just read; at the next
            cur_buf ^= 1;                                            // transform it is the newer of the two sets in flight
        } else {
            if (ds != 0.0) build_z<C, NT, true>(ld, ds, tw, ln, z);  // wave-uniform branch
            else build_z<C, NT, false>(ld, ds, tw, ln, z);
            if (l + 1 < nloads) issue_loads<NT>(ld, s_src[l + 1] + off, ln, K);
        }
        backward_line<C, TWREG>(z, wb, tw, ln, g, t64r);
    };
    // the `a` operands stay in registers (twice their grid values, like every transformed operand)
    double2 areg[FUSED_NA][8];
#pragma unroll
    for (int ia = 0; ia < FUSED_NA; ++ia) {
        if (ia < na) {
            backward(ia, areg[ia]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) areg[ia][i] = make_double2(0.0, 0.0);
        }
    }
    double2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int l = na; l < nloads; ++l) {
        double2 g[8];
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
                    for (int i = 0; i < 8; ++i) {
                        // packed even/odd samples: real parts multiply real parts, imaginary parts imaginary parts
                        acc[i].x += (cf * g[i].x) * areg[ia][i].x;
                        acc[i].y += (cf * g[i].y) * areg[ia][i].y;
                    }
                }
            }
        }
        const int oc = s_flush[l];
        if (oc >= 0) {
            int lf = lane;
            WF_OPAQUE_LANE(lf);
            double *dst = s_out[oc] + off;
            forward_line<C, NT, TWREG>(acc, wb, tw, lf, M, K, t64r, [&](int k, double2 v) { gstore16(dst + 2 * k, v); });
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = make_double2(0.0, 0.0);
        }
    }
  }
}

template <int C>
int launch_c(const FftDev &d, const FusedArgs &f_in, long nlines, hipStream_t st) {
    using G = G2<C>;
    constexpr int WAVES = 4;
    constexpr int NT32 = (2 * C + 2) / 3;                  // 64-pair blocks that hold M/2 = N/3 pairs (3/2 dealiasing)
    static const int lpw_env = getenv("DDH_GW_LPW") ? std::max(1, atoi(getenv("DDH_GW_LPW"))) : 0;
    const long lpw = lpw_env ? lpw_env : std::min<long>(8, std::max<long>(1, nlines / ((long)WAVES * 2048)));
    const long nwg = (nlines + (long)WAVES * lpw - 1) / ((long)WAVES * lpw);
    if ((unsigned long)nwg > 0x7fffffffUL) return fail("rfft_bilinear_fused: grid too large");
    const size_t lds = ((size_t)G::TW + (size_t)WAVES * G::LDW) * sizeof(double2);
    const dim3 grid((unsigned)nwg), block(64 * WAVES);
    static const bool twreg = getenv("DDH_GW_TWREG") ? atoi(getenv("DDH_GW_TWREG")) != 0 : false;   // (spills: see below)
    static const int dma = getenv("DDH_GW_DMA") ? atoi(getenv("DDH_GW_DMA")) : 1;     // 1: LDS-DMA staging + register twiddles, 2: without, 0: register loads
    FusedArgs f = f_in;
    for (int t = 0; t < FUSED_TERMS; ++t) f.coef[t] *= 0.25;     // both factors of a term arrive doubled
    if (dma && d.K + 1 == 64 * NT32) {
        if (dma == 1)
            hipLaunchKernelGGL((gridwave2_bilinear_kernel<C, NT32, WAVES, true, true>), grid, block, lds, st, d, f, nlines);
        else
            hipLaunchKernelGGL((gridwave2_bilinear_kernel<C, NT32, WAVES, false, true>), grid, block, lds, st, d, f, nlines);
    } else if (twreg)
        hipLaunchKernelGGL((gridwave2_bilinear_kernel<C, NT32, WAVES, true>), grid, block, lds, st, d, f, nlines);
    else
        hipLaunchKernelGGL((gridwave2_bilinear_kernel<C, NT32, WAVES, false>), grid, block, lds, st, d, f, nlines);
    DDH_HIP(hipGetLastError());
    return 0;
}

}  // namespace gw2

// Default for 3/2-padded lines whose stored pairs fill their 64-pair blocks exactly (K + 1 == 64 NT: 768 / 512, 384 / 256):
// the LDS-DMA variant with register twiddles.  Measured on MI355X at 768 x 384 lines of 768 points (tools/bench_fused.py,
// same box, round 5): first generation 6.71 ms; this kernel with register loads 6.83 ms (no room for register twiddles:
// with them 256 VGPRs + 44 B of scratch, 8.19 ms); LDS-DMA staging, one operand ahead 6.71 ms; two operands ahead 6.42 ms
// (235 VGPRs), without register twiddles 6.59 ms.  Counters (profiles/r5_fused_sq_counters.txt): 6405 instead of 6897 VALU
// and 1021 instead of 1135 LDS instructions per line, LDS bank-conflict cycles -64 %, s_waitcnt time -15 %.
// DDH_GW_V2=0 keeps the first generation; other sizes take it anyway.
bool gridwave2_supported(const FftDev &d) {
    static const bool off = getenv("DDH_GW_V2") != nullptr && atoi(getenv("DDH_GW_V2")) == 0;
    if (off) return false;
    if (d.N % 128 != 0 || (d.M & 1) || d.M < 2 || d.M > d.N) return false;
    const int C = d.N / 128;
    if (!(C == 3 || C == 6)) return false;
    const int NT32 = (2 * C + 2) / 3;
    static const bool any_k = getenv("DDH_GW_V2") != nullptr && atoi(getenv("DDH_GW_V2")) == 2;   // 2: also truncated spectra (register loads)
    if (any_k) return d.K + 1 <= 64 * NT32 && NT32 < C;
    return d.K + 1 == 64 * NT32 && NT32 < C;
}

int launch_gridwave2(const FftDev &d, const FusedArgs &f, long nlines, hipStream_t st) {
    switch (d.N / 128) {
        case 3: return gw2::launch_c<3>(d, f, nlines, st);
        case 6: return gw2::launch_c<6>(d, f, nlines, st);
    }
    return fail("gridwave2: unsupported size");
}

}  // namespace ddh
