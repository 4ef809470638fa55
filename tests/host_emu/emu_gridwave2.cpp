// Host emulation of the second-generation wave-per-line grid stage (dedalus_amd/csrc/ddh_gridwave2.h): the SAME lane
// code the GPU kernel runs, compiled with g++; the 64 lanes of a wavefront are 64 threads, WF_SYNC is a barrier and the
// LDS is a shared array.  Test infrastructure only (tests/test_host_emu_gridwave2.py).
#define DDH_HOST_EMU
#include <pthread.h>

#include <cstring>
#include <thread>
#include <vector>

#include "../../dedalus_amd/csrc/ddh_gridwave2.h"

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
using namespace ddh::gw2;

template <typename F>
static void run_wave(F body) {
    pthread_barrier_init(&g_bar, nullptr, 64);
    std::vector<std::thread> th;
    for (int lane = 0; lane < 64; ++lane) th.emplace_back([=]() { body(lane); });
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&g_bar);
}

template <int C>
static std::vector<double2> make_tables(const double *plan_tw) {
    using G = G2<C>;
    std::vector<double2> tw(G::TW);
    const double2 *p = reinterpret_cast<const double2 *>(plan_tw);
    for (int i = 0; i < G::TW; ++i) tw[i] = p[G::table_q(i)];
    return tw;
}

// pair k of the line (zero beyond K: the buffer range check of the device loads)
static double2 pair_at(const double *line, int k, int K) {
    if (k < 0 || k > K) return make_double2(0.0, 0.0);
    return make_double2(line[2 * k], line[2 * k + 1]);
}

template <int C, bool TWREG>
static void backward(int K, double dscale, const double *line, const double *plan_tw, double *grid, bool staged) {
    using G = G2<C>;
    constexpr int NT = (2 * C + 2) / 3;
    const std::vector<double2> tw = make_tables<C>(plan_tw);
    std::vector<double2> wb(G::LDW), st(Staging<NT>::SIZE, make_double2(0.0, 0.0));
    run_wave([&](int lane) {
        Loads<NT> ld;
        for (int t = 0; t < NT; ++t) {
            ld.d[t] = pair_at(line, lane + 64 * t, K);
            ld.m[t] = pair_at(line, 64 - lane + 64 * t, K);
            // (the LDS-DMA loads of the device: lane l's 16 bytes land at st + 64 t + l)
            st[t * 64 + lane] = ld.d[t];
        }
        WF_SYNC();
        double2 t64r[7];
        for (int i = 0; i < 7; ++i) t64r[i] = tw[G::T_64 + (lane & 7) * (i + 1)];
        double2 z[C], g[8];
        for (int i = 0; i < 8; ++i) g[i] = make_double2(0.0, 0.0);
        if (staged) {
            if (dscale != 0.0) build_z_staged<C, NT, true>(st.data(), dscale, tw.data(), lane, z);
            else build_z_staged<C, NT, false>(st.data(), dscale, tw.data(), lane, z);
        } else if (dscale != 0.0) build_z<C, NT, true>(ld, dscale, tw.data(), lane, z);
        else build_z<C, NT, false>(ld, dscale, tw.data(), lane, z);
        backward_line<C, TWREG>(z, wb.data(), tw.data(), lane, g, t64r);
        if (lane < G::NB) {
            const int mc = lane >> 3, p = lane & 7;
            for (int q = 0; q < 8; ++q) {
                const int m = mc + C * (p + 8 * q);
                grid[2 * m] = g[q].x;
                grid[2 * m + 1] = g[q].y;
            }
        }
    });
}

template <int C, bool TWREG>
static void forward(int K, int M, const double *grid, const double *plan_tw, double *line) {
    using G = G2<C>;
    constexpr int NT = (2 * C + 2) / 3;
    const std::vector<double2> tw = make_tables<C>(plan_tw);
    std::vector<double2> wb(G::LDW);
    run_wave([&](int lane) {
        double2 t64r[7];
        for (int i = 0; i < 7; ++i) t64r[i] = tw[G::T_64 + (lane & 7) * (i + 1)];
        double2 y[8];
        for (int q = 0; q < 8; ++q) y[q] = make_double2(0.0, 0.0);
        if (lane < G::NB) {
            const int mc = lane >> 3, p = lane & 7;
            for (int q = 0; q < 8; ++q) {
                const int m = mc + C * (p + 8 * q);
                y[q] = make_double2(grid[2 * m], grid[2 * m + 1]);
            }
        }
        forward_line<C, NT, TWREG>(y, wb.data(), tw.data(), lane, M, K, t64r, [&](int k, double2 v) {
            line[2 * k] = v.x;
            line[2 * k + 1] = v.y;
        });
    });
}

extern "C" {

// grid (N = 128 C doubles) = TWICE the c2r transform of the (cos, msin) pairs of `line` (derivative applied when
// dscale != 0); returns -1 for sizes that are not instantiated
int emu_gw2_backward(int C, int twreg, int K, double dscale, const double *line, const double *plan_tw, double *grid) {
    const bool staged = (twreg & 2) != 0;      // bit 1: pairs through the LDS staging area (the LDS-DMA variant)
    twreg &= 1;
    if (C == 6) { twreg ? backward<6, true>(K, dscale, line, plan_tw, grid, staged) : backward<6, false>(K, dscale, line, plan_tw, grid, staged); return 0; }
    if (C == 3) { twreg ? backward<3, true>(K, dscale, line, plan_tw, grid, staged) : backward<3, false>(K, dscale, line, plan_tw, grid, staged); return 0; }
    return -1;
}
int emu_gw2_forward(int C, int twreg, int K, int M, const double *grid, const double *plan_tw, double *line) {
    if (C == 6) { twreg ? forward<6, true>(K, M, grid, plan_tw, line) : forward<6, false>(K, M, grid, plan_tw, line); return 0; }
    if (C == 3) { twreg ? forward<3, true>(K, M, grid, plan_tw, line) : forward<3, false>(K, M, grid, plan_tw, line); return 0; }
    return -1;
}
}
