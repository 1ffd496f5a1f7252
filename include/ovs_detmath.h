// ovs_detmath.h -- deterministic elementary functions for the float DECISIONS of the matcher / BA path.
//
// Why this exists ("weak" #2): upstream calls libm for four functions whose result decides a match pair
//   * landmark::predict_scale_level            ceil(std::log(float ratio) / log_scale_factor_)      (src/openvslam/data/landmark.cc)
//   * camera::equirectangular::reproject_to_image   -asin(bearing.y), atan2(bearing.x, bearing.z)   (src/openvslam/camera/equirectangular.cc)
//   * match::robust::check_epipolar_constraint      pi/2 - |acos(cos_residual)|                     (src/openvslam/match/robust.cc)
//   * optimize::g2o::se3::equirectangular_reproj_edge   the same asin / atan2                       (src/openvslam/optimize/g2o/se3/)
// glibc's and the HIP device library's versions of these differ by an ulp now and then, which flips a ceil() or a window edge.
// The functions below are ONE op sequence of individually rounded IEEE-754 double operations (+ - * / sqrt, no FMA contraction:
// every translation unit that includes this header is built with -ffp-contract=off), so g++ on the host and hipcc on gfx950
// produce identical bits. Both the kernels (openvslam_amd/csrc) and the CPU oracle (oracle/) include this header; the
// functions themselves are pinned independently against glibc through numpy (tests/test_detmath.py).
//
// ovs_det_logf IS glibc's logf (>= 2.27, x86-64 and aarch64 builds): same table, same polynomial, exhaustively bit-equal on all
// 2^31 positive floats to the libm in this container. asin / acos / atan2 (double) use the classic argument reductions with minimax
// polynomials (Sun fdlibm lineage: s_atan.c, e_atan2.c, e_asin.c, e_acos.c), restated here: < 1 ulp (atan2: < 2 ulp) from glibc's
// table-driven double routines, which cannot be restated from memory. Their results are only ever consumed after a cast to float
// (window centres) or against a tolerance (BA residuals), so a 1-ulp double difference matters with probability ~2^-29 per value.
#ifndef OVS_DETMATH_H_
#define OVS_DETMATH_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define OVS_DM_FN __host__ __device__ inline
#else
#define OVS_DM_FN inline
#endif

namespace ovs_dm {

OVS_DM_FN uint64_t bits_of(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
OVS_DM_FN double from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
OVS_DM_FN double abs_d(double x) { return from_bits(bits_of(x) & 0x7fffffffffffffffull); }
OVS_DM_FN bool is_nan(double x) { return (bits_of(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }
// correctly rounded square root on both sides (x86-64 sqrtsd; gfx950: the IEEE-correct f64 expansion)
OVS_DM_FN double sqrt_d(double x) { return __builtin_sqrt(x); }

// logf, glibc >= 2.27's algorithm (sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c; the ARM optimized-routines logf): 16-entry
// (1/c, log c) table on [sqrt(2)/2 .. sqrt(2)) subintervals, r = z/c - 1, degree-3 polynomial, everything in double, one rounding to
// float. EXHAUSTIVELY equal to this container's glibc 2.35 logf for every positive finite float, with or without FMA contraction
// (tests/test_detmath.py::test_logf_equals_glibc_exhaustively) -- so this rule is pinned to the third-party arithmetic upstream calls.
OVS_DM_FN float logf_glibc(float x) {
    const double T[16][2] = {{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
                             {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
                             {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
                             {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
                             {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
                             {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
                             {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
                             {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix;
    __builtin_memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {   // zero, subnormal, negative, inf, NaN
        if (ix * 2u == 0u) return -1.0f / 0.0f;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / 0.0f;
        const float xs = x * 8388608.0f;   // subnormal: scale by 2^23
        __builtin_memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf;
    __builtin_memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * T[i][0] - 1.0;
    const double y0 = T[i][1] + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

OVS_DM_FN double atan_d(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                              1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                              6.12323399573676603587e-17};
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    if (is_nan(x)) return x + x;
    const bool neg = (bits_of(x) >> 63) != 0;
    const double ax = abs_d(x);
    int id;
    double r;
    if (ax >= 73786976294838206464.0) {   // 2^66
        const double z = atanhi[3] + atanlo[3];
        return neg ? -z : z;
    }
    if (ax < 0.4375) {
        if (ax < 1.862645149230957e-09) return x;   // 2^-29
        id = -1;
        r = x;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) {
            id = 0;
            r = (2.0 * ax - 1.0) / (2.0 + ax);
        } else {
            id = 1;
            r = (ax - 1.0) / (ax + 1.0);
        }
    } else if (ax < 2.4375) {
        id = 2;
        r = (ax - 1.5) / (1.0 + 1.5 * ax);
    } else {
        id = 3;
        r = -1.0 / ax;
    }
    const double z = r * r;
    const double w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return r - r * (s1 + s2);
    const double res = atanhi[id] - ((r * (s1 + s2) - atanlo[id]) - r);
    return neg ? -res : res;
}

// rational kernel shared by asin / acos: R(t) = p(t) / q(t) ~ (asin(x) - x) / x^3 with t = x^2
OVS_DM_FN double asin_pq(double t) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    return p / q;
}

}   // namespace ovs_dm

// std::log(float) as glibc computes it (see logf_glibc above)
OVS_DM_FN float ovs_det_logf(float x) { return ovs_dm::logf_glibc(x); }

// sinf / cosf as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h + sincosf_data.c: the ARM
// optimized-routines algorithm): |x| < pi/4 -> the polynomial directly; otherwise n = round(x * 2/pi) through a 2^24-scaled multiply,
// r = x - n * (pi/2) in double, a degree-7 sine or degree-8 cosine polynomial of r by quadrant, everything in double, ONE rounding to float.
// EXHAUSTIVELY equal to this container's glibc 2.35 sinf and cosf on every float in [0, 6.3] (1 086 953 883 values, the range the
// descriptor's angles live in; tests/test_detmath.py::test_sinf_cosf_equal_glibc_exhaustively runs the check in C through the oracle
// library). Used by the `trig` variant of the rBRIEF steering (oracle/ORACLE_SPEC.md rule 11): what upstream computes IF it calls
// std::cos / std::sin on the float angle instead of util::cos / util::sin. Valid for |x| < 120 (the callers pass [0, 2 pi]).
namespace ovs_dm {
struct sincosf_tab {
    double sign[4];
    double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
OVS_DM_FN float sincosf_poly(double x, double x2, const sincosf_tab& p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = p.s2 + x2 * p.s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = p.c3 + x2 * p.c4;
    const double c1 = p.c0 + x2 * p.c1;
    const double x6 = x4 * x2;
    const double c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}
OVS_DM_FN uint32_t abstop12(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ffu;
}
// want_cos = 0: sinf(y), 1: cosf(y)
OVS_DM_FN float sincosf_glibc(float y, int want_cos) {
    const sincosf_tab T0 = {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
                            -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
    const sincosf_tab T1 = {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
                            0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13};
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {   // |y| < pi/4
        if (abstop12(y) < abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
        return sincosf_poly(x, x * x, T0, want_cos);
    }
    const double r = x * T0.hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * T0.hpi;
    const double s = T0.sign[n & 3];
    return sincosf_poly(x * s, x * x, (n & 2) ? T1 : T0, n ^ want_cos);
}
}   // namespace ovs_dm
// keypt.angle * M_PI / 180.0 evaluated in double and rounded to float once (ORACLE_SPEC rule 11), as ONE double multiplication by RN(pi / 180):
// bit-identical to the two-operation form for EVERY float angle in [0, 360] (all 1 135 869 953 of them: tests/test_detmath.py::
// test_deg2rad_single_multiply_is_exact runs the comparison in C), and a thirtieth of the instructions of an IEEE f64 division on the device
OVS_DM_FN float ovs_det_deg2rad(float angle_deg) { return (float)((double)angle_deg * 0x1.1df46a2529d39p-6); }
OVS_DM_FN float ovs_det_sinf(float x) { return ovs_dm::sincosf_glibc(x, 0); }
OVS_DM_FN float ovs_det_cosf(float x) { return ovs_dm::sincosf_glibc(x, 1); }


OVS_DM_FN double ovs_det_atan(double x) { return ovs_dm::atan_d(x); }

OVS_DM_FN double ovs_det_atan2(double y, double x) {
    using namespace ovs_dm;
    const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16, pi_o_2 = 1.5707963267948965580e+00,
                 pi_o_4 = 7.8539816339744827900e-01;
    if (is_nan(x) || is_nan(y)) return x + y;
    if (x == 1.0) return atan_d(y);
    const uint64_t ux = bits_of(x), uy = bits_of(y);
    const int m = (int)(uy >> 63) | ((int)(ux >> 63) << 1);   // bit 0: y negative, bit 1: x negative
    const double ax = abs_d(x), ay = abs_d(y);
    const double inf = from_bits(0x7ff0000000000000ull);
    if (ay == 0.0) {
        switch (m) {
            case 0:
            case 1: return y;      // atan(+-0, +anything) = +-0
            case 2: return pi;     // atan(+0, -anything) = pi
            default: return -pi;   // atan(-0, -anything) = -pi
        }
    }
    if (ax == 0.0) return (m & 1) ? -pi_o_2 : pi_o_2;
    if (ax == inf) {
        if (ay == inf) {
            switch (m) {
                case 0: return pi_o_4;
                case 1: return -pi_o_4;
                case 2: return 3.0 * pi_o_4;
                default: return -3.0 * pi_o_4;
            }
        }
        switch (m) {
            case 0: return 0.0;
            case 1: return -0.0;
            case 2: return pi;
            default: return -pi;
        }
    }
    if (ay == inf) return (m & 1) ? -pi_o_2 : pi_o_2;
    const int k = (int)((uy >> 52) & 0x7ff) - (int)((ux >> 52) & 0x7ff);   // exponent difference
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;            // |y / x| > 2^60
    else if ((m & 2) && k < -60) z = 0.0;            // |y| / x < -2^60
    else z = atan_d(abs_d(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

OVS_DM_FN double ovs_det_asin(double x) {
    using namespace ovs_dm;
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
    if (is_nan(x)) return x + x;
    const double ax = abs_d(x);
    const bool neg = (bits_of(x) >> 63) != 0;
    if (ax >= 1.0) {
        if (ax == 1.0) return x * pio2_hi + x * pio2_lo;
        return (x - x) / (x - x);   // NaN
    }
    if (ax < 0.5) {
        if (ax < 7.450580596923828e-09) return x;   // 2^-27
        const double t = x * x;
        return x + x * asin_pq(t);
    }
    const double w1 = 1.0 - ax;
    const double t = w1 * 0.5;
    const double r = asin_pq(t);
    const double s = sqrt_d(t);
    double res;
    if (ax >= 0.975) {
        res = pio2_hi - (2.0 * (s + s * r) - pio2_lo);
    } else {
        const double w = from_bits(bits_of(s) & 0xffffffff00000000ull);   // s with the low word cleared
        const double c = (t - w * w) / (s + w);
        const double p = 2.0 * s * r - (pio2_lo - 2.0 * c);
        const double q = pio4_hi - 2.0 * w;
        res = pio4_hi - (p - q);
    }
    return neg ? -res : res;
}

OVS_DM_FN double ovs_det_acos(double x) {
    using namespace ovs_dm;
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    if (is_nan(x)) return x + x;
    const double ax = abs_d(x);
    const bool neg = (bits_of(x) >> 63) != 0;
    if (ax >= 1.0) {
        if (ax == 1.0) return neg ? pi + 2.0 * pio2_lo : 0.0;
        return (x - x) / (x - x);   // NaN
    }
    if (ax < 0.5) {
        if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;   // 2^-57
        const double z = x * x;
        const double r = asin_pq(z);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (neg) {
        const double z = (1.0 + x) * 0.5;
        const double r = asin_pq(z);
        const double s = sqrt_d(z);
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt_d(z);
    const double df = from_bits(bits_of(s) & 0xffffffff00000000ull);
    const double c = (z - df * df) / (s + df);
    const double r = asin_pq(z);
    const double w = r * s + c;
    return 2.0 * (df + w);
}

#endif   // OVS_DETMATH_H_
