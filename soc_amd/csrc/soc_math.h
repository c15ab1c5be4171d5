// soc_math.h -- deterministic fp32 math for the photon-packet path.
//
// The reference (kernel_ASOC.c / kernel_ASOC_aux.c) calls the OpenCL device
// built-ins exp, log, sin, cos, sincos, acos, sqrt, fmod, ldexp, floor.  Their
// last-bit rounding differs between OpenCL devices, and a one-ulp difference
// flips a cell-boundary decision, after which a packet visits different cells
// (SURVEY.md 7.3-1).  To make fixed-seed results reproducible between the host
// CPU and gfx950, every transcendental used on the path is defined HERE, from
// IEEE-754 fp32 add/mul/fma, correctly rounded sqrt/div and integer bit
// operations only.  The same header is compiled
//   * by hipcc for the device kernels (soc_kernels.hip), and
//   * by gcc for the CPU oracle's "soc" math mode (oracle/soc_oracle.c),
// with -ffp-contract=off on both sides, so the two produce bit-identical values.
// Accuracy is ~1 ulp (checked against libm in tests/test_math.py).
//
// No function here reads or writes memory other than its arguments.
#ifndef SOC_MATH_H
#define SOC_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#  define SOC_HD __host__ __device__ static inline
#else
#  define SOC_HD static inline
#endif

#define SOC_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

SOC_HD uint32_t soc_f2u(float f)
{
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}
SOC_HD float soc_u2f(uint32_t u)
{
    union { float f; uint32_t u; } v;
    v.u = u;
    return v.f;
}

// correctly rounded on both targets (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt is the default)
SOC_HD float soc_sqrtf(float x) { return __builtin_sqrtf(x); }
SOC_HD float soc_floorf(float x) { return __builtin_floorf(x); }
SOC_HD float soc_fabsf(float x) { return __builtin_fabsf(x); }
SOC_HD float soc_fminf(float a, float b) { return (b < a) ? b : a; }
SOC_HD float soc_fmaxf(float a, float b) { return (b > a) ? b : a; }
SOC_HD float soc_clampf(float x, float lo, float hi) { return soc_fminf(soc_fmaxf(x, lo), hi); }

// fmod(x, 1.0f): exact, result has the sign of x (C99 fmodf semantics)
SOC_HD float soc_fmod1f(float x)
{
    float r = x - __builtin_truncf(x);
    return __builtin_copysignf(r, x);
}
SOC_HD double soc_fmod1d(double x)
{
    double r = x - __builtin_trunc(x);
    return __builtin_copysign(r, x);
}

// ldexp(x, -level) / ldexp(x, +level) for 0 <= level <= 30: exact power-of-two scaling
SOC_HD float soc_scale_down(float x, int level) { return x * soc_u2f((uint32_t)(127 - level) << 23); }
SOC_HD float soc_scale_up(float x, int level) { return x * soc_u2f((uint32_t)(127 + level) << 23); }

// exp(x).  Results below FLT_MIN are flushed to zero (x < -87), overflow gives +inf.
SOC_HD float soc_expf(float x)
{
    if (!(x >= -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.72f) return soc_u2f(0x7f800000u);
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = SOC_FMA(-n, 0.693359375f, x);          // ln2 high part (exact product for |n| < 2^12)
    r = SOC_FMA(-n, -2.12194440e-4f, r);             // ln2 low part
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = SOC_FMA(p, r, 1.3981999507e-3f);
    p = SOC_FMA(p, r, 8.3334519073e-3f);
    p = SOC_FMA(p, r, 4.1665795894e-2f);
    p = SOC_FMA(p, r, 1.6666665459e-1f);
    p = SOC_FMA(p, r, 5.0000001201e-1f);
    float y = SOC_FMA(p, z, r) + 1.0f;
    int   k  = (int)n;                                // -126 .. 128
    int   k1 = k >> 1, k2 = k - k1;                   // two normal factors, product never overflows early
    y = y * soc_u2f((uint32_t)(127 + k1) << 23);
    y = y * soc_u2f((uint32_t)(127 + k2) << 23);
    return y;
}

// soc_expf for -0.34 < x <= 0: the argument reduction of soc_expf gives n = 0, r = x and the final
// scaling multiplies by 1.0 twice, so leaving those steps out produces the same bits
// (tests/test_math.py checks the equality over the whole interval).
SOC_HD float soc_expf_small(float x)
{
    const float z = x * x;
    float p = 1.9875691500e-4f;
    p = SOC_FMA(p, x, 1.3981999507e-3f);
    p = SOC_FMA(p, x, 8.3334519073e-3f);
    p = SOC_FMA(p, x, 4.1665795894e-2f);
    p = SOC_FMA(p, x, 1.6666665459e-1f);
    p = SOC_FMA(p, x, 5.0000001201e-1f);
    return SOC_FMA(p, z, x) + 1.0f;
}

// n / u from the correctly rounded reciprocal r = 1/u: q = RN(n*r), e = n - q*u (exact, fma),
// result RN(q + e*r).  By Markstein's theorem this is the correctly rounded quotient, i.e. the
// same bits as n / u (no overflow/underflow in the ranges of the path: 5e-5 <= |u| <= 1,
// |n| <= 1.0001); 3.4e9 random and adversarial pairs were compared with the division
// instruction (tests/test_math.py repeats a sample).  Three instructions instead of ~12.
SOC_HD float soc_div_by_rcp(float n, float u, float r)
{
    const float q = n * r;
    const float e = SOC_FMA(-q, u, n);
    return SOC_FMA(e, r, q);
}

// natural logarithm.  log(0) = -inf, log(x<0) = NaN.
SOC_HD float soc_logf(float x)
{
    if (!(x > 0.0f)) {
        if (x == 0.0f) return soc_u2f(0xff800000u);
        return soc_u2f(0x7fc00000u);
    }
    uint32_t ix = soc_f2u(x);
    if (ix >= 0x7f800000u) return x;                  // +inf
    int e = 0;
    if (ix < 0x00800000u) {                           // subnormal: scale by 2^23
        x  = x * 8388608.0f;
        ix = soc_f2u(x);
        e  = -23;
    }
    e += (int)(ix >> 23) - 126;                       // x = m * 2^e, m in [0.5, 1)
    float m = soc_u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = SOC_FMA(p, m, -1.1514610310e-1f);
    p = SOC_FMA(p, m, 1.1676998740e-1f);
    p = SOC_FMA(p, m, -1.2420140846e-1f);
    p = SOC_FMA(p, m, 1.4249322787e-1f);
    p = SOC_FMA(p, m, -1.6668057665e-1f);
    p = SOC_FMA(p, m, 2.0000714765e-1f);
    p = SOC_FMA(p, m, -2.4999993993e-1f);
    p = SOC_FMA(p, m, 3.3333331174e-1f);
    float y  = (p * m) * z;
    float fe = (float)e;
    y = SOC_FMA(fe, -2.12194440e-4f, y);
    y = SOC_FMA(-0.5f, z, y);
    float r = m + y;
    r = SOC_FMA(fe, 0.693359375f, r);
    return r;
}

// double-precision natural logarithm for x in (0, +inf): the scattered-light kernels write
// log(1.0 - W*Rand) with a double literal, so the reference evaluates that one logarithm in
// fp64 (kernel_ASOC_sca.c:906).  Argument reduction x = 2^k * (1+f), s = f/(2+f),
// log(1+f) = f - s*(f - R(s^2)); degree-14 even polynomial, relative error < 1e-16.
SOC_HD double soc_logd(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    if (!(x > 0.0)) return (x == 0.0) ? -__builtin_huge_val() : (x - x) / (x - x);
    int k = 0;
    if (v.u < 0x0010000000000000ULL) { x = x * 18014398509481984.0; v.d = x; k = -54; }   // subnormal
    if (v.u >= 0x7ff0000000000000ULL) return x;
    k += (int)(v.u >> 52) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;                           // m in [1, 2)
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// expm1(x) for x <= 0 (the point-source scattered-light kernel writes W = -expm1(-tau),
// kernel_ASOC_sca.c:1737): Taylor series in fma form below 0.35, exp(x)-1 beyond, where the
// subtraction loses less than one bit.
SOC_HD float soc_expm1f(float x)
{
    if (!(x > -0.35f)) return soc_expf(x) - 1.0f;
    float p = 2.7557319224e-6f;                       // 1/9!
    p = SOC_FMA(p, x, 2.4801587302e-5f);
    p = SOC_FMA(p, x, 1.9841269841e-4f);
    p = SOC_FMA(p, x, 1.3888888889e-3f);
    p = SOC_FMA(p, x, 8.3333333333e-3f);
    p = SOC_FMA(p, x, 4.1666666667e-2f);
    p = SOC_FMA(p, x, 1.6666666667e-1f);
    p = SOC_FMA(p, x, 0.5f);
    return SOC_FMA(p * x, x, x);
}

// x^1.5 for x > 0 (Henyey-Greenstein denominator of the cell-emission peel-off,
// kernel_ASOC_sca.c:1390): two correctly rounded operations
SOC_HD float soc_pow15f(float x) { return x * soc_sqrtf(x); }

// log10(x) and integer power x^n (n >= 0): used by the equilibrium-temperature lookup
// (kernel_A2E.c:136-137 calls the OpenCL built-ins log10 and pown)
SOC_HD float soc_log10f(float x) { return soc_logf(x) * 0.43429448190325182765f; }
SOC_HD float soc_pownf(float x, int n)
{
    float r = 1.0f, b = x;
    unsigned int e = (unsigned int)(n < 0 ? -n : n);
    while (e) {                                       // binary exponentiation, fixed operation order
        if (e & 1u) r = r * b;
        b = b * b;
        e >>= 1;
    }
    return (n < 0) ? (1.0f / r) : r;
}

// sin and cos together.  Three-term Cody-Waite reduction: accurate for |x| < ~1e4,
// which covers every call on the path (arguments lie in [-2 pi, 2 pi]).
SOC_HD void soc_sincosf(float x, float *s, float *c)
{
    float q = __builtin_rintf(x * 0.636619772367581343f);   // x * 2/pi
    float r = SOC_FMA(-q, 1.5703125f, x);
    r = SOC_FMA(-q, 4.837512969970703125e-4f, r);
    r = SOC_FMA(-q, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = SOC_FMA(ps, z, 8.3321608736e-3f);
    ps = SOC_FMA(ps, z, -1.6666654611e-1f);
    float sr = SOC_FMA(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = SOC_FMA(pc, z, -1.388731625493765e-3f);
    pc = SOC_FMA(pc, z, 4.166664568298827e-2f);
    float cr = SOC_FMA(pc * z, z, SOC_FMA(-0.5f, z, 1.0f));
    int   iq = (int)q;
    float ss = (iq & 1) ? cr : sr;
    float cc = (iq & 1) ? sr : cr;
    if (iq & 2) ss = -ss;
    if ((iq + 1) & 2) cc = -cc;
    *s = ss;
    *c = cc;
}
SOC_HD float soc_sinf(float x) { float s, c; soc_sincosf(x, &s, &c); return s; }
SOC_HD float soc_cosf(float x) { float s, c; soc_sincosf(x, &s, &c); return c; }

// asin on |a| <= 0.5 (polynomial), building block of acos
SOC_HD float soc_asin_core(float a)
{
    float z = a * a;
    float p = 4.2163199048e-2f;
    p = SOC_FMA(p, z, 2.4181311049e-2f);
    p = SOC_FMA(p, z, 4.5470025998e-2f);
    p = SOC_FMA(p, z, 7.4953002686e-2f);
    p = SOC_FMA(p, z, 1.6666752422e-1f);
    return SOC_FMA(p * z, a, a);
}

// acos(x); arguments outside [-1, 1] are clamped (the path never produces them).
SOC_HD float soc_acosf(float x)
{
    if (x != x) return x;
    if (x >= 1.0f) return 0.0f;
    if (x <= -1.0f) return 3.14159265358979323846f;
    if (x > 0.5f) {
        float a = soc_sqrtf(0.5f * (1.0f - x));
        return 2.0f * soc_asin_core(a);
    }
    if (x < -0.5f) {
        float a = soc_sqrtf(0.5f * (1.0f + x));
        return 3.14159265358979323846f - 2.0f * soc_asin_core(a);
    }
    return 1.57079632679489661923f - soc_asin_core(x);
}

// atan2(y, x) (direction -> Healpix longitude in the scattered-light kernels,
// kernel_ASOC_sca.c:357).  atan on [0,1] by a degree-19 odd polynomial (least-squares fit reweighted
// towards minimax), then the usual reflections; error < 2 ulp (tests/test_math.py).  atan2(0,0) = 0.
SOC_HD float soc_atan2f(float y, float x)
{
    const float ax = soc_fabsf(x), ay = soc_fabsf(y);
    const float mx = soc_fmaxf(ax, ay), mn = soc_fminf(ax, ay);
    float a = (mx > 0.0f) ? (mn / mx) : 0.0f;                      // [0, 1]
    const float s = a * a;
    float p = -1.7934360529e-03f;
    p = SOC_FMA(p, s, 1.0913770722e-02f);
    p = SOC_FMA(p, s, -3.1176284444e-02f);
    p = SOC_FMA(p, s, 5.7956025310e-02f);
    p = SOC_FMA(p, s, -8.4033591855e-02f);
    p = SOC_FMA(p, s, 1.0952155088e-01f);
    p = SOC_FMA(p, s, -1.4264236043e-01f);
    p = SOC_FMA(p, s, 1.9998547802e-01f);
    p = SOC_FMA(p, s, -3.3333299077e-01f);
    float r = SOC_FMA(p * s, a, a);                                  // atan(a)
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return (y < 0.0f) ? -r : r;
}

#endif  // SOC_MATH_H
