// Complex helpers and in-register DFT butterflies (radix 2, 3, 4, 5, 6, 7, 8, 16) shared by every transform kernel.
// Plain C++: with DDH_HOST_EMU defined the file compiles with g++ (no HIP), which is how tests/host_emu runs the
// wave-level transform code of ddh_wavefft.h lane by lane on the CPU.
#pragma once
#ifdef DDH_HOST_EMU
#include <cmath>
struct double2 {
    double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
#define DDH_DEV inline
#else
#include <hip/hip_runtime.h>
#define DDH_DEV __device__ __forceinline__
#endif

namespace ddh {

// ------------------------------------------------------------------------------------------------
// complex helpers
// ------------------------------------------------------------------------------------------------
DDH_DEV double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
DDH_DEV double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
DDH_DEV double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by (sign * i)
DDH_DEV double2 muli(double2 a, int sign) {
    return sign > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

template <int R>
DDH_DEV void butterfly(double2 *v, int sign);

template <>
DDH_DEV void butterfly<2>(double2 *v, int) {
    double2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
DDH_DEV void butterfly<3>(double2 *v, int sign) {
    const double s = 0.86602540378443864676372317075294;  // sqrt(3)/2
    double2 t = cadd(v[1], v[2]);
    double2 d = csub(v[1], v[2]);
    double2 m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y);
    double2 r = muli(make_double2(s * d.x, s * d.y), sign);
    v[0] = cadd(v[0], t);
    v[1] = cadd(m, r);
    v[2] = csub(m, r);
}
template <>
DDH_DEV void butterfly<4>(double2 *v, int sign) {
    double2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    double2 c = cadd(v[1], v[3]), d = muli(csub(v[1], v[3]), sign);
    v[0] = cadd(a, c);
    v[1] = cadd(b, d);
    v[2] = csub(a, c);
    v[3] = csub(b, d);
}
template <>
DDH_DEV void butterfly<5>(double2 *v, int sign) {
    const double c1 = 0.30901699437494742410229341718282;   // cos(2pi/5)
    const double c2 = -0.80901699437494742410229341718282;  // cos(4pi/5)
    const double s1 = 0.95105651629515357211643933337938;   // sin(2pi/5)
    const double s2 = 0.58778525229247312916870595463907;   // sin(4pi/5)
    double2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    double2 d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
    double2 m1 = make_double2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    double2 m2 = make_double2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    double2 r1 = muli(make_double2(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y), sign);
    double2 r2 = muli(make_double2(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y), sign);
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, r1);
    v[4] = csub(m1, r1);
    v[2] = cadd(m2, r2);
    v[3] = csub(m2, r2);
}
template <>
DDH_DEV void butterfly<6>(double2 *v, int sign) {
    // t = 2 t1 + t2, u = u1 + 3 u2: DFT3 over t1, twiddle W6^(t2 u1), DFT2 over t2
    const double h = 0.86602540378443864676372317075294;  // sqrt(3)/2
    double2 e[3] = {v[0], v[2], v[4]};
    double2 o[3] = {v[1], v[3], v[5]};
    butterfly<3>(e, sign);
    butterfly<3>(o, sign);
    const double si = sign > 0 ? h : -h;
    o[1] = make_double2(0.5 * o[1].x - si * o[1].y, 0.5 * o[1].y + si * o[1].x);
    o[2] = make_double2(-0.5 * o[2].x - si * o[2].y, -0.5 * o[2].y + si * o[2].x);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        v[u] = cadd(e[u], o[u]);
        v[u + 3] = csub(e[u], o[u]);
    }
}
template <>
DDH_DEV void butterfly<7>(double2 *v, int sign) {
    // direct 7-point DFT (rare size): X_u = sum_t v_t exp(sign 2 pi i t u / 7)
    const double c[7] = {1.0, 0.62348980185873353052500488400424, -0.22252093395631440428890256449679,
                         -0.90096886790241912623610231950745, -0.90096886790241912623610231950745,
                         -0.22252093395631440428890256449679, 0.62348980185873353052500488400424};
    const double s[7] = {0.0, 0.78183148246802980870844452667406, 0.97492791218182360701813168299393,
                         0.43388373911755812047576833284836, -0.43388373911755812047576833284836,
                         -0.97492791218182360701813168299393, -0.78183148246802980870844452667406};
    double2 x[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) x[t] = v[t];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        double2 acc = x[0];
#pragma unroll
        for (int t = 1; t < 7; ++t) {
            const int q = (t * u) % 7;
            double2 w = make_double2(c[q], sign > 0 ? s[q] : -s[q]);
            acc = cadd(acc, cmul(x[t], w));
        }
        v[u] = acc;
    }
}

DDH_DEV void dft4_inplace(double2 &v0, double2 &v1, double2 &v2, double2 &v3, int sign) {
    const double2 a = cadd(v0, v2), b = csub(v0, v2);
    const double2 c = cadd(v1, v3), d = muli(csub(v1, v3), sign);
    v0 = cadd(a, c);
    v1 = cadd(b, d);
    v2 = csub(a, c);
    v3 = csub(b, d);
}
// multiply by exp(sign * 2 pi i * q / 16), q = 0..9 (constants)
DDH_DEV double2 mul_w16(double2 a, int q, int sign) {
    const double c1 = 0.92387953251128675612818318939679;   // cos(pi/8)
    const double s1 = 0.38268343236508977172845998403040;   // sin(pi/8)
    const double r2 = 0.70710678118654752440084436210485;   // sqrt(1/2)
    double wr, wi;
    switch (q) {
        case 0: return a;
        case 1: wr = c1; wi = s1; break;
        case 2: wr = r2; wi = r2; break;
        case 3: wr = s1; wi = c1; break;
        case 4: return muli(a, sign);
        case 6: wr = -r2; wi = r2; break;
        default: wr = -c1; wi = -s1; break;   // q == 9
    }
    if (sign < 0) wi = -wi;
    return make_double2(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
}
template <>
DDH_DEV void butterfly<8>(double2 *v, int sign) {
    // n = 2 n1 + n2, k = k1 + 4 k2
    double2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    double2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4_inplace(e0, e1, e2, e3, sign);
    dft4_inplace(o0, o1, o2, o3, sign);
    o1 = mul_w16(o1, 2, sign);
    o2 = mul_w16(o2, 4, sign);
    o3 = mul_w16(o3, 6, sign);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
template <>
DDH_DEV void butterfly<16>(double2 *v, int sign) {
    // n = 4 n1 + n2, k = k1 + 4 k2: DFT4 over n1 for each n2, twiddle W16^(n2 k1), DFT4 over n2 for each k1
    double2 y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        y[n2][0] = v[n2]; y[n2][1] = v[4 + n2]; y[n2][2] = v[8 + n2]; y[n2][3] = v[12 + n2];
        dft4_inplace(y[n2][0], y[n2][1], y[n2][2], y[n2][3], sign);
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        double2 c0 = y[0][k1];
        double2 c1 = mul_w16(y[1][k1], k1, sign);
        double2 c2 = mul_w16(y[2][k1], 2 * k1, sign);
        double2 c3 = mul_w16(y[3][k1], 3 * k1, sign);
        dft4_inplace(c0, c1, c2, c3, sign);
        v[k1] = c0; v[k1 + 4] = c1; v[k1 + 8] = c2; v[k1 + 12] = c3;
    }
}

}  // namespace ddh
