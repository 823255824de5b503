/*
 * gaps_oracle.c -- CPU restatement (plain C) of the CoGAPS asynchronous Gibbs sampler:
 * ProposalQueue / ConcurrentAtomicDomain / DenseNormalModel / AsynchronousGibbsSampler /
 * runOnePhase / GapsStatistics, following the reference v3.27.4 under /root/reference/src.
 *
 * TEST INFRASTRUCTURE ONLY (see gaps_oracle.h).  The product never links this file.
 *
 * Parity pin: SURVEY.md section 8c fingerprints, reproduced by tests/test_oracle_pin.py with
 * math_mode = GO_MATH_LIBM and sequential reductions (redW = 1), i.e. the arithmetic of the
 * reference's default scalar -O2 build.  The Boost.Math distribution calls the reference makes
 * while filling its three lookup tables (Math.cpp:43-86) are restated here with libm erfc /
 * Newton solves in double precision ("LUT parity with a real Boost build is unpinned",
 * SURVEY.md section 8c).
 *
 * Build: gcc -O2 -std=gnu11 -ffp-contract=off -fno-fast-math [-fopenmp] -shared -fPIC
 */
#include "gaps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GO_EPSILON 1.0e-5f                     /* Math.h:11 */
#define GO_SQRT2F 1.4142135623730950488016887242097f /* Math.h:14 */
#define GO_PI_D 3.1415926535897932384626433832795    /* Math.h:13 */
#define ERF_N 3001
#define ERFINV_N 5001
#define QGAMMA_N 5001
#define GO_NONE 0xFFFFFFFFu

/* ======================================================================================
 * portable log / exp (spec shared with the HIP kernels; every op is an IEEE double op)
 * ====================================================================================== */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* natural log of a non-negative finite float, evaluated in double, rounded once to float.
 * x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1). */
float go_portable_logf(float x)
{
    uint32_t ux = f2u(x);
    if (ux == 0u) return -INFINITY;
    if (ux == 0x3f800000u) return 0.0f;
    int e = (int)((ux >> 23) & 0xffu);
    uint32_t man = ux & 0x7fffffu;
    double m;
    if (e == 0) { /* subnormal: normalise */
        double d = (double)x * 0x1p64;
        uint64_t ud; memcpy(&ud, &d, 8);
        e = (int)((ud >> 52) & 0x7ff) - 1023 - 64;
        m = u2d((ud & 0xfffffffffffffull) | 0x3ff0000000000000ull);
    } else {
        e -= 127;
        m = u2d(((uint64_t)man << 29) | 0x3ff0000000000000ull);
    }
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 2.0 / 19.0;
    p = p * z + 2.0 / 17.0;
    p = p * z + 2.0 / 15.0;
    p = p * z + 2.0 / 13.0;
    p = p * z + 2.0 / 11.0;
    p = p * z + 2.0 / 9.0;
    p = p * z + 2.0 / 7.0;
    p = p * z + 2.0 / 5.0;
    p = p * z + 2.0 / 3.0;
    p = p * z + 2.0;
    double r = (double)e * 0.6931471805599453094 + s * p;
    return (float)r;
}

/* e^x for float x, evaluated in double: x = k ln2 + r, |r| <= ln2/2, Taylor to degree 13 */
float go_portable_expf(float x)
{
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 89.0) return INFINITY;
    if (xd < -104.0) return 0.0f;
    double kf = floor(xd * 1.4426950408889634074 + 0.5);
    double r = xd - kf * 0.6931471805599453094;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int k = (int)kf;
    double scale = u2d((uint64_t)(k + 1023) << 52); /* k in [-151, 129] -> normal double */
    return (float)(p * scale);
}

/* ======================================================================================
 * glibc's logf / expf, restated.  The reference calls libm's float functions in its accept tests
 * (AsynchronousGibbsSampler.h:165,189), in exponential() (Random.cpp:172-175) and in truncGammaUpper()
 * (Random.cpp:194-200), so its chain depends on the C library it is linked with -- a third-party dependency
 * that is not under /root/reference.  Pinned version: GNU libc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11, this
 * image), sysdeps/ieee754/flt-32/e_logf.c, e_expf.c, math_config.h (the ARM optimized-routines algorithms:
 * a 16-entry (1/c, log c) table + degree-3 polynomial for logf, a 32-entry 2^(i/32) table + degree-3
 * polynomial for expf, every intermediate an IEEE double).  x86-64 glibc ships each function twice and
 * picks one at load time (ifunc): the generic build (separate multiply and add) and a build compiled with
 * -mfma -mavx2 in which the compiler fused multiply-add pairs -- read off the instruction sequences of
 * __logf_fma / __expf_fma of the pinned library.  `fused` selects that variant; the two differ in the last
 * bit on a fraction of the inputs.  tests/test_oracle_pin.py checks the restatement against the host's
 * logf / expf over all floats the sampler can feed them.
 * ====================================================================================== */
static const double glf_tab[16][2] = {     /* __logf_data.tab: { invc, logc } */
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2}, {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};
static const uint64_t gef_tab[32] = {      /* __exp2f_data.tab: bits of 2^(i/32) with the exponent adjusted */
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float go_glibc_logf(float x, int fused)
{
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -INFINITY;                 /* log(+-0) */
        if (ix == 0x7f800000u) return x;                     /* log(inf) */
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN;
        ix = f2u(x * 0x1p23f); ix -= 23u << 23;              /* subnormal: normalise */
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = glf_tab[i][0], logc = glf_tab[i][1];
    const double z = (double)u2f(iz);
    if (fused) {
        const double r = fma(z, invc, -1.0);
        const double y0 = fma((double)k, Ln2, logc);
        double y = fma(A1, r, A2);
        const double r2 = r * r;
        y = fma(A0, r2, y);
        return (float)fma(y, r2, y0 + r);
    }
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

float go_glibc_expf(float x, int fused)
{
    const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const uint32_t ux = f2u(x), abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                  /* |x| >= 88 or nan */
        if (ux == 0xff800000u) return 0.f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;              /* overflow */
        if (x < -0x1.9fe368p6f) return 0.f;                  /* underflow */
        if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;   /* __math_may_uflowf */
    }
    const double xd = (double)x;
    double kd, r;
    uint64_t ki;
    if (fused) {
        kd = fma(InvLn2N, xd, Shift); memcpy(&ki, &kd, 8); kd -= Shift;
        r = fma(InvLn2N, xd, -kd);
    } else {
        const double z = InvLn2N * xd;
        kd = z + Shift; memcpy(&ki, &kd, 8); kd -= Shift;
        r = z - kd;
    }
    const uint64_t t = gef_tab[ki % 32u] + (ki << 47);
    const double s = u2d(t);
    double zz, y; const double r2 = r * r;
    if (fused) { zz = fma(C0, r, C1); y = fma(C2, r, 1.0); y = fma(zz, r2, y); }
    else { zz = C0 * r + C1; y = C2 * r + 1.0; y = zz * r2 + y; }
    return (float)(y * s);
}

/* ======================================================================================
 * RNG: Xoroshiro128+ seeder, PCG-XSH-RR per-object generator  (math/Random.cpp)
 * ====================================================================================== */

typedef struct { uint64_t s[2], prev[2]; } go_seeder;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* Random.cpp:232-243 */
static uint64_t seeder_next(go_seeder *g)
{
    g->prev[0] = g->s[0]; g->prev[1] = g->s[1];
    const uint64_t s0 = g->s[0];
    uint64_t s1 = g->s[1];
    uint64_t result = s0 + s1;
    s1 ^= s0;
    g->s[0] = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
    g->s[1] = rotl64(s1, 37);
    return result;
}
/* Random.cpp:245-249 */
static void seeder_rollback(go_seeder *g) { g->s[0] = g->prev[0]; g->s[1] = g->prev[1]; }
/* Random.cpp:222-230 */
static void seeder_init(go_seeder *g, uint64_t seed)
{
    g->s[0] = seed | 1; g->s[1] = seed | 1;
    for (unsigned i = 0; i < 5000; ++i) seeder_next(g);
}

typedef struct go_randstate {
    go_seeder seeder;
    float erf[ERF_N], erfinv[ERFINV_N], qgamma[QGAMMA_N];
    int math_mode;
} go_randstate;

typedef struct { uint64_t state; const go_randstate *rs; } go_rng;

/* Random.cpp:46-49 */
static inline void rng_advance(go_rng *r) { r->state = r->state * 6364136223846793005ull + (54u | 1); }
/* Random.cpp:51-56 */
static inline uint32_t rng_get(const go_rng *r)
{
    uint32_t xorshifted = (uint32_t)(((r->state >> 18u) ^ r->state) >> 27u);
    uint32_t rot = (uint32_t)(r->state >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
/* Random.cpp:40-44 */
static inline uint32_t rng_u32(go_rng *r) { rng_advance(r); return rng_get(r); }
/* Random.cpp:32-38 */
static void rng_init(go_rng *r, go_randstate *rs) { r->rs = rs; r->state = seeder_next(&rs->seeder); rng_advance(r); }

/* test hook: how many floats with bit patterns lo, lo+step, ... <= hi does the restatement get differently from the host libm */
uint64_t go_glibc_mismatches(int fn, int fused, uint32_t lo_bits, uint32_t hi_bits, uint32_t step)
{
    uint64_t bad = 0;
    if (!step) step = 1;
    for (uint64_t b = lo_bits; b <= hi_bits; b += step) {
        const float x = u2f((uint32_t)b);
        const float a = fn ? go_glibc_expf(x, fused) : go_glibc_logf(x, fused);
        const float c = fn ? expf(x) : logf(x);
        if (f2u(a) != f2u(c) && !(a != a && c != c)) ++bad;
    }
    return bad;
}
float go_strtof(const char *s) { return strtof(s, NULL); } /* MatrixElement.cpp:15-23: text -> float, one rounding */

uint32_t go_pcg_next(uint64_t *state)
{
    go_rng r; r.state = *state; r.rs = NULL;
    uint32_t v = rng_u32(&r);
    *state = r.state;
    return v;
}

/* Random.cpp:11, 63-66: float(u32) / float(UINT32_MAX) */
static inline float rng_uniform(go_rng *r)
{
    const float maxU32AsFloat = (float)4294967295u;
    return (float)rng_u32(r) / maxU32AsFloat;
}
/* Random.cpp:12, 58-61 */
static inline double rng_uniformd(go_rng *r) { return (double)rng_u32(r) / (double)4294967295u; }
/* Random.cpp:68-71 */
static inline float rng_uniform_ab(go_rng *r, float a, float b) { return rng_uniform(r) * (b - a) + a; }

/* Random.cpp:79-96 */
static uint32_t rng_uniform32(go_rng *r, uint32_t a, uint32_t b)
{
    if (b == a) return a;
    uint32_t range = b + 1 - a;
    uint32_t x = rng_u32(r);
    uint32_t iPart = 0xFFFFFFFFu / range;
    while (x >= range * iPart) x = rng_u32(r);
    return x / iPart + a;
}
/* Random.cpp:98-103 */
static inline uint64_t rng_u64(go_rng *r)
{
    uint64_t high = ((uint64_t)rng_u32(r) << 32) & 0xFFFFFFFF00000000ull;
    uint64_t low = rng_u32(r);
    return high | low;
}
/* Random.cpp:105-123 */
static uint64_t rng_uniform64(go_rng *r, uint64_t a, uint64_t b)
{
    if (b == a) return a;
    uint64_t range = b + 1 - a;
    uint64_t x = rng_u64(r);
    uint64_t iPart = 0xFFFFFFFFFFFFFFFFull / range;
    while (x >= range * iPart) x = rng_u64(r);
    return x / iPart + a;
}

static inline float go_logf(const go_randstate *rs, float x)
{
    return rs->math_mode == GO_MATH_PORTABLE ? go_portable_logf(x) : (rs->math_mode >= GO_MATH_GLIBC_FMA ? go_glibc_logf(x, rs->math_mode == GO_MATH_GLIBC_FMA) : logf(x));
}
static inline float go_expf(const go_randstate *rs, float x)
{
    return rs->math_mode == GO_MATH_PORTABLE ? go_portable_expf(x) : (rs->math_mode >= GO_MATH_GLIBC_FMA ? go_glibc_expf(x, rs->math_mode == GO_MATH_GLIBC_FMA) : expf(x));
}

/* Random.cpp:132-143 */
static int rng_poisson_small(go_rng *r, double lambda)
{
    int x = 0;
    double p = rng_uniformd(r);
    double cutoff = exp(-lambda);
    while (p >= cutoff) { p *= rng_uniformd(r); ++x; }
    return x;
}
/* Random.cpp:146-170; gaps::lgamma = boost::math::lgamma (Math.cpp:83-86) restated with libm */
static int rng_poisson_large(go_rng *r, double lambda)
{
    double c = 0.767 - 3.36 / lambda;
    double beta = GO_PI_D / sqrt(3.0 * lambda);
    double alpha = beta * lambda;
    double k = log(c) - lambda - log(beta);
    for (;;) {
        double u = rng_uniformd(r);
        double x = (alpha - log((1.0 - u) / u)) / beta;
        double n = floor(x + 0.5);
        if (n < 0.0) continue;
        double v = rng_uniformd(r);
        double y = alpha - beta * x;
        double w = 1.0 + exp(y);
        double lhs = y + log(v / (w * w));
        double rhs = k + n * log(lambda) - lgamma(n + 1);
        if (lhs <= rhs) return (int)n;
    }
}
/* Random.cpp:125-128 */
static int rng_poisson(go_rng *r, double lambda)
{
    return lambda <= 5.0 ? rng_poisson_small(r, lambda) : rng_poisson_large(r, lambda);
}
/* Random.cpp:172-175 */
static float rng_exponential(go_rng *r, float lambda)
{
    return -1.f * go_logf(r->rs, rng_uniform(r)) / lambda;
}

static inline float fmin_ref(float a, float b) { return a < b ? a : b; } /* Math.cpp:13-16 */
static inline float fmax_ref(float a, float b) { return a < b ? b : a; } /* Math.cpp:28-31 */

/* Random.cpp:307-326 */
static float p_norm_fast(const go_randstate *rs, float p, float mean, float sd)
{
    float term = (p - mean) / (sd * GO_SQRT2F);
    float erf_ = 0.f;
    if (term < 0.f) {
        term = fmax_ref(term, -3.f);
        const unsigned ndx = (unsigned)(-term * 1000.f);
        erf_ = -rs->erf[ndx];
    } else {
        term = fmin_ref(term, 3.f);
        const unsigned ndx = (unsigned)(term * 1000.f);
        erf_ = rs->erf[ndx];
    }
    return 0.5f * (1.f + erf_);
}
/* Random.cpp:328-345 */
static float q_norm_fast(const go_randstate *rs, float q, float mean, float sd)
{
    float term = 2.f * q - 1.f;
    float erfinv_ = 0.f;
    if (term < 0.f) {
        const unsigned ndx = (unsigned)(-term * (float)(ERFINV_N - 1));
        erfinv_ = -rs->erfinv[ndx];
    } else {
        const unsigned ndx = (unsigned)(term * (float)(ERFINV_N - 1));
        erfinv_ = rs->erfinv[ndx];
    }
    return mean + sd * GO_SQRT2F * erfinv_;
}

typedef struct { float v; int has; } optf;
static inline optf optf_none(void) { optf o; o.v = 0.f; o.has = 0; return o; }
static inline optf optf_some(float v) { optf o; o.v = v; o.has = 1; return o; }

/* Random.cpp:178-191 */
static optf rng_trunc_normal(go_rng *r, float a, float b, float mean, float sd)
{
    float pLower = p_norm_fast(r->rs, a, mean, sd);
    float pUpper = p_norm_fast(r->rs, b, mean, sd);
    if (!(pLower > 0.95f || pUpper < 0.05f)) {
        float z = q_norm_fast(r->rs, rng_uniform_ab(r, pLower, pUpper), mean, sd);
        z = fmax_ref(a, fmin_ref(z, b));
        return optf_some(z);
    }
    return optf_none();
}
/* Random.cpp:194-200 */
static float rng_trunc_gamma_upper(go_rng *r, float b, float scale)
{
    float upper = 1.f - go_expf(r->rs, -b / scale) * (1.f + b / scale);
    const unsigned ndx = (unsigned)rng_uniform_ab(r, 0.f, upper * 5000.f);
    return r->rs->qgamma[ndx] * scale;
}

/* ---- lookup tables: Random.cpp:269-295 over Math.cpp:43-81 (Boost.Math in double) ---- */

/* normal cdf: boost normal cdf = erfc(-(x-mean)/(sd*sqrt2))/2 */
static double norm_cdf_d(double x) { return 0.5 * erfc(-x / 1.41421356237309504880); }

/* erfc_inv by Newton/Halley on libm erfc, to double precision */
static double erfc_inv_d(double y)
{
    if (y == 1.0) return 0.0;
    /* initial guess from the normal quantile (Acklam-style rational is overkill: bisection seed) */
    double lo = -6.0, hi = 6.0;
    for (int i = 0; i < 60; ++i) { double mid = 0.5 * (lo + hi); if (erfc(mid) > y) lo = mid; else hi = mid; }
    double x = 0.5 * (lo + hi);
    for (int i = 0; i < 4; ++i) {
        double f = erfc(x) - y;
        double fp = -2.0 / sqrt(GO_PI_D) * exp(-x * x);
        double fpp = -2.0 * x * fp;
        double dx = f / fp;
        x -= dx / (1.0 - 0.5 * dx * fpp / fp);
    }
    return x;
}
/* boost normal quantile: mean - sd*sqrt2*erfc_inv(2p) */
static double norm_quantile_d(double p) { double r = erfc_inv_d(2.0 * p); r = -r; r *= 1.41421356237309504880; return r + 0.0; }

/* regularised lower incomplete gamma P(2,x) = 1 - e^-x (1+x) */
static double gamma2_cdf_d(double x)
{
    if (x < 0.5) { /* series, no cancellation */
        double term = x * x / 2.0, sum = 0.0; /* k=2 term: x^2 (k-1)/k! */
        double xk = x * x, fact = 2.0; int k = 2;
        (void)term;
        for (; k < 40; ++k) {
            double t = xk * (double)(k - 1) / fact;
            sum += (k & 1) ? -t : t;
            xk *= x; fact *= (double)(k + 1);
        }
        return sum;
    }
    return 1.0 - exp(-x) * (1.0 + x);
}
static double gamma2_quantile_d(double p)
{
    double lo = 0.0, hi = 60.0;
    for (int i = 0; i < 80; ++i) { double mid = 0.5 * (lo + hi); if (gamma2_cdf_d(mid) < p) lo = mid; else hi = mid; }
    double x = 0.5 * (lo + hi);
    for (int i = 0; i < 3; ++i) {
        double f = gamma2_cdf_d(x) - p;
        double fp = x * exp(-x);
        if (fp > 0) x -= f / fp;
    }
    return x;
}

/* Math.cpp:43-47 / 55-63 / 71-81 : float in, double distribution call, float out */
static float ref_p_norm(float p) { return (float)norm_cdf_d((double)p); }
static float ref_q_norm(float q) { return (float)norm_quantile_d((double)q); }
static float ref_q_gamma2(float q) { if (q < 0.000001f) return 0.f; return (float)gamma2_quantile_d((double)q); }

void go_build_luts(float *erf_, float *erfinv_, float *qgamma_)
{
    for (unsigned i = 0; i < ERF_N; ++i) {
        float x = (float)i / 1000.f;
        erf_[i] = 2.f * ref_p_norm(x * GO_SQRT2F) - 1.f;
    }
    for (unsigned i = 0; i < ERFINV_N - 1; ++i) {
        float x = (float)i / (float)(ERFINV_N - 1);
        erfinv_[i] = ref_q_norm((1.f + x) / 2.f) / GO_SQRT2F;
    }
    erfinv_[ERFINV_N - 1] = ref_q_norm(1.9998f / 2.f) / GO_SQRT2F;
    qgamma_[0] = 0.f;
    for (unsigned i = 1; i < QGAMMA_N - 1; ++i) {
        float x = (float)i / (float)(QGAMMA_N - 1);
        qgamma_[i] = ref_q_gamma2(x);
    }
    qgamma_[QGAMMA_N - 1] = ref_q_gamma2(0.9998f);
}

static void randstate_init(go_randstate *rs, uint32_t seed, int math_mode)
{
    seeder_init(&rs->seeder, seed);
    go_build_luts(rs->erf, rs->erfinv, rs->qgamma);
    rs->math_mode = math_mode;
}

uint64_t go_seeder_stream(uint32_t seed, uint64_t *out, uint64_t n)
{
    go_seeder g; seeder_init(&g, seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = seeder_next(&g);
    return n;
}

/* ======================================================================================
 * Atomic domain (atomic/ConcurrentAtomicDomain.cpp, ConcurrentAtom.h, MutableMap.h)
 * atoms live in a pool addressed by handle; `vec` is the unsorted mAtoms vector; the sorted
 * std::map is restated as a two-level sorted array (blocks of <= BLK_CAP entries).
 * ====================================================================================== */

typedef struct { uint64_t pos; uint32_t left, right, index; float mass; } go_atom;
typedef struct { uint64_t pos; uint32_t h; } go_ent;
#define BLK_CAP 256
typedef struct { go_ent e[BLK_CAP]; uint32_t n; } go_blk;

typedef struct {
    go_atom *pool; uint32_t pool_cap, pool_hi; uint32_t *free_h; uint32_t n_free, free_cap;
    uint32_t *vec; uint32_t n, vec_cap;
    go_blk **blk; uint32_t nblk, blk_cap;
    uint32_t *erase; uint32_t n_erase, erase_cap;
    uint64_t domain_len;
} go_domain;

static void dom_init(go_domain *d, uint64_t nBins)
{
    memset(d, 0, sizeof(*d));
    uint64_t binLength = 0xFFFFFFFFFFFFFFFFull / nBins;   /* ConcurrentAtomicDomain.cpp:16-18 */
    d->domain_len = binLength * nBins;
    d->blk_cap = 16; d->blk = (go_blk **)calloc(d->blk_cap, sizeof(go_blk *));
    d->nblk = 1; d->blk[0] = (go_blk *)calloc(1, sizeof(go_blk));
}
static void dom_free(go_domain *d)
{
    for (uint32_t i = 0; i < d->nblk; ++i) free(d->blk[i]);
    free(d->blk); free(d->pool); free(d->free_h); free(d->vec); free(d->erase);
}
static uint32_t dom_alloc(go_domain *d)
{
    if (d->n_free) return d->free_h[--d->n_free];
    if (d->pool_hi == d->pool_cap) {
        d->pool_cap = d->pool_cap ? d->pool_cap * 2 : 1024;
        d->pool = (go_atom *)realloc(d->pool, (size_t)d->pool_cap * sizeof(go_atom));
    }
    return d->pool_hi++;
}
static void dom_release(go_domain *d, uint32_t h)
{
    if (d->n_free == d->free_cap) {
        d->free_cap = d->free_cap ? d->free_cap * 2 : 1024;
        d->free_h = (uint32_t *)realloc(d->free_h, (size_t)d->free_cap * sizeof(uint32_t));
    }
    d->free_h[d->n_free++] = h;
}
/* block whose range holds key: last block with first pos <= key (block 0 otherwise) */
static uint32_t sorted_find_blk(const go_domain *d, uint64_t key)
{
    uint32_t lo = 0, hi = d->nblk;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) / 2;
        if (d->blk[mid]->n && d->blk[mid]->e[0].pos <= key) lo = mid; else hi = mid;
    }
    return lo;
}
/* first slot in block with pos >= key */
static uint32_t blk_lower(const go_blk *b, uint64_t key)
{
    uint32_t lo = 0, hi = b->n;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (b->e[mid].pos < key) lo = mid + 1; else hi = mid; }
    return lo;
}
static int sorted_count(const go_domain *d, uint64_t key)
{
    uint32_t bi = sorted_find_blk(d, key);
    const go_blk *b = d->blk[bi];
    uint32_t s = blk_lower(b, key);
    return s < b->n && b->e[s].pos == key;
}
static void sorted_insert(go_domain *d, uint64_t key, uint32_t h)
{
    uint32_t bi = sorted_find_blk(d, key);
    go_blk *b = d->blk[bi];
    if (b->n == BLK_CAP) { /* split */
        if (d->nblk == d->blk_cap) { d->blk_cap *= 2; d->blk = (go_blk **)realloc(d->blk, d->blk_cap * sizeof(go_blk *)); }
        go_blk *nb = (go_blk *)calloc(1, sizeof(go_blk));
        uint32_t half = BLK_CAP / 2;
        memcpy(nb->e, b->e + half, (BLK_CAP - half) * sizeof(go_ent));
        nb->n = BLK_CAP - half; b->n = half;
        memmove(d->blk + bi + 2, d->blk + bi + 1, (d->nblk - bi - 1) * sizeof(go_blk *));
        d->blk[bi + 1] = nb; d->nblk++;
        if (key >= nb->e[0].pos) { b = nb; }
    }
    uint32_t s = blk_lower(b, key);
    memmove(b->e + s + 1, b->e + s, (b->n - s) * sizeof(go_ent));
    b->e[s].pos = key; b->e[s].h = h; b->n++;
}
static void sorted_erase(go_domain *d, uint64_t key)
{
    uint32_t bi = sorted_find_blk(d, key);
    go_blk *b = d->blk[bi];
    uint32_t s = blk_lower(b, key);
    memmove(b->e + s, b->e + s + 1, (b->n - s - 1) * sizeof(go_ent));
    b->n--;
    if (b->n == 0 && d->nblk > 1) {
        free(b);
        memmove(d->blk + bi, d->blk + bi + 1, (d->nblk - bi - 1) * sizeof(go_blk *));
        d->nblk--;
    }
}
/* MutableMap.h:79-82 updateKey: in-place key rewrite, order preserved by the caller */
static void sorted_update_key(go_domain *d, uint64_t oldKey, uint64_t newKey)
{
    uint32_t bi = sorted_find_blk(d, oldKey);
    go_blk *b = d->blk[bi];
    uint32_t s = blk_lower(b, oldKey);
    b->e[s].pos = newKey;
}
/* neighbours of key after it has been inserted */
static void sorted_neighbours(const go_domain *d, uint64_t key, uint32_t *left, uint32_t *right)
{
    uint32_t bi = sorted_find_blk(d, key);
    const go_blk *b = d->blk[bi];
    uint32_t s = blk_lower(b, key);
    *left = GO_NONE; *right = GO_NONE;
    if (s > 0) *left = b->e[s - 1].h;
    else { uint32_t j = bi; while (j > 0) { --j; if (d->blk[j]->n) { *left = d->blk[j]->e[d->blk[j]->n - 1].h; break; } } }
    if (s + 1 < b->n) *right = b->e[s + 1].h;
    else { for (uint32_t j = bi + 1; j < d->nblk; ++j) if (d->blk[j]->n) { *right = d->blk[j]->e[0].h; break; } }
}
/* ConcurrentAtomicDomain.cpp:20-24 front(): smallest position */
static uint32_t dom_front(const go_domain *d)
{
    for (uint32_t j = 0; j < d->nblk; ++j) if (d->blk[j]->n) return d->blk[j]->e[0].h;
    return GO_NONE;
}
/* ConcurrentAtomicDomain.cpp:82-106 insert */
static uint32_t dom_insert(go_domain *d, uint64_t pos, float mass)
{
    uint32_t h = dom_alloc(d);
    go_atom *a = &d->pool[h];
    a->pos = pos; a->mass = mass; a->left = GO_NONE; a->right = GO_NONE;
    sorted_insert(d, pos, h);
    if (d->n == d->vec_cap) { d->vec_cap = d->vec_cap ? d->vec_cap * 2 : 1024; d->vec = (uint32_t *)realloc(d->vec, (size_t)d->vec_cap * sizeof(uint32_t)); }
    a = &d->pool[h];
    a->index = d->n; d->vec[d->n++] = h;
    uint32_t l, r; sorted_neighbours(d, pos, &l, &r);
    if (r != GO_NONE) { a->right = r; d->pool[r].left = h; }
    if (l != GO_NONE) { a->left = l; d->pool[l].right = h; }
    return h;
}
/* ConcurrentAtomicDomain.cpp:109-124 erase: swap-with-last in the unsorted vector */
static void dom_erase(go_domain *d, uint32_t h)
{
    go_atom *a = &d->pool[h];
    sorted_erase(d, a->pos);
    d->vec[a->index] = d->vec[d->n - 1];
    d->pool[d->vec[a->index]].index = a->index;
    d->n--;
    if (a->left != GO_NONE) d->pool[a->left].right = a->right;
    if (a->right != GO_NONE) d->pool[a->right].left = a->left;
    dom_release(d, h);
}
/* ConcurrentAtomicDomain.cpp:62-69 cacheErase (omp critical) */
static void dom_cache_erase(go_domain *d, uint32_t h)
{
#pragma omp critical(AtomicInsertOrErase)
    {
        if (d->n_erase == d->erase_cap) { d->erase_cap = d->erase_cap ? d->erase_cap * 2 : 256; d->erase = (uint32_t *)realloc(d->erase, (size_t)d->erase_cap * sizeof(uint32_t)); }
        d->erase[d->n_erase++] = h;
    }
}
static __thread go_domain *g_sort_dom;      /* (per host thread: several sessions may be stepped from several threads at once -- bench.py --chains) */
static int cmp_erase(const void *a, const void *b)
{
    uint64_t pa = g_sort_dom->pool[*(const uint32_t *)a].pos, pb = g_sort_dom->pool[*(const uint32_t *)b].pos;
    return pa < pb ? -1 : (pa > pb ? 1 : 0);
}
/* ConcurrentAtomicDomain.cpp:71-79 flushEraseCache: sort by position, erase in that order */
static void dom_flush_erase(go_domain *d)
{
    if (d->n_erase > 1) { g_sort_dom = d; qsort(d->erase, d->n_erase, sizeof(uint32_t), cmp_erase); }
    for (uint32_t i = 0; i < d->n_erase; ++i) dom_erase(d, d->erase[i]);
    d->n_erase = 0;
}
/* ConcurrentAtomicDomain.cpp:126-132 move */
static void dom_move(go_domain *d, uint32_t h, uint64_t newPos)
{
    uint64_t old = d->pool[h].pos;
    d->pool[h].pos = newPos;
    sorted_update_key(d, old, newPos);
}
/* ConcurrentAtomicDomain.cpp:46-54 randomFreePosition */
static uint64_t dom_random_free_position(const go_domain *d, go_rng *rng)
{
    uint64_t pos = rng_uniform64(rng, 1, d->domain_len);
    while (sorted_count(d, pos)) pos = rng_uniform64(rng, 1, d->domain_len);
    return pos;
}

/* ======================================================================================
 * ProposalQueue (atomic/ProposalQueue.cpp) + hash sets (data_structures/HashSets.cpp)
 * ====================================================================================== */

typedef struct {
    go_rng rng; uint64_t pos; uint32_t atom1, atom2; /* handles */
    uint32_t r1, c1, r2, c2; char type;
} go_prop;

typedef struct { uint64_t a, b; } go_pair;

typedef struct {
    go_prop *q; uint32_t nq, qcap;
    uint32_t *usedRows; uint64_t rowKey; uint32_t nRows;          /* FixedHashSetU32 */
    uint64_t *usedAtoms; uint32_t nUsedAtoms, usedAtomsCap;       /* SmallHashSetU64 */
    go_pair *moves; uint32_t nMoves, movesCap;                    /* SmallPairedHashSetU64 */
    go_randstate *rs; go_rng rng;
    uint64_t minAtoms, maxAtoms, binLength, numCols;
    double alpha, domainLength, numBins;
    float lambda, u1, u2;
    unsigned numProcessed; int useCachedRng;
} go_queue;

/* ProposalQueue.cpp:19-36 */
static void queue_init(go_queue *q, uint64_t nElements, uint64_t nPatterns, go_randstate *rs)
{
    memset(q, 0, sizeof(*q));
    q->nRows = (uint32_t)(nElements / nPatterns);
    q->usedRows = (uint32_t *)calloc(q->nRows ? q->nRows : 1, sizeof(uint32_t));
    q->rowKey = 1;
    q->rs = rs;
    rng_init(&q->rng, rs);
    q->binLength = 0xFFFFFFFFFFFFFFFFull / nElements;
    q->numCols = nPatterns;
    q->domainLength = (double)(q->binLength * nElements);
    q->numBins = (double)nElements;
}
static void queue_free(go_queue *q) { free(q->q); free(q->usedRows); free(q->usedAtoms); free(q->moves); }

static inline int rows_contains(const go_queue *q, uint32_t r) { return (uint64_t)q->usedRows[r] == q->rowKey; } /* HashSets.cpp:19-22 (u32 store compared with the u64 key) */
static inline void rows_insert(go_queue *q, uint32_t r) { q->usedRows[r] = (uint32_t)q->rowKey; }
static int atoms_contains(const go_queue *q, uint64_t pos)
{
    for (uint32_t i = 0; i < q->nUsedAtoms; ++i) if (q->usedAtoms[i] == pos) return 1;
    return 0;
}
static void atoms_insert(go_queue *q, uint64_t pos)
{
    if (q->nUsedAtoms == q->usedAtomsCap) { q->usedAtomsCap = q->usedAtomsCap ? q->usedAtomsCap * 2 : 256; q->usedAtoms = (uint64_t *)realloc(q->usedAtoms, q->usedAtomsCap * sizeof(uint64_t)); }
    q->usedAtoms[q->nUsedAtoms++] = pos;
}
/* HashSets.cpp:85-96 overlap */
static int moves_overlap(const go_queue *q, uint64_t pos)
{
    for (uint32_t i = 0; i < q->nMoves; ++i) if (q->moves[i].a < pos && pos < q->moves[i].b) return 1;
    return 0;
}
/* HashSets.cpp:75-78 */
static void moves_insert(go_queue *q, uint64_t a, uint64_t b)
{
    if (q->nMoves == q->movesCap) { q->movesCap = q->movesCap ? q->movesCap * 2 : 256; q->moves = (go_pair *)realloc(q->moves, q->movesCap * sizeof(go_pair)); }
    if (a < b) { q->moves[q->nMoves].a = a; q->moves[q->nMoves].b = b; } else { q->moves[q->nMoves].a = b; q->moves[q->nMoves].b = a; }
    q->nMoves++;
}
static void queue_push(go_queue *q, const go_prop *p)
{
    if (q->nq == q->qcap) { q->qcap = q->qcap ? q->qcap * 2 : 256; q->q = (go_prop *)realloc(q->q, q->qcap * sizeof(go_prop)); }
    q->q[q->nq++] = *p;
}
/* ProposalQueue.cpp:78-85 clear */
static void queue_clear(go_queue *q) { q->nq = 0; ++q->rowKey; q->nUsedAtoms = 0; q->nMoves = 0; }

/* ProposalQueue.cpp:123-127 */
static float queue_death_prob(const go_queue *q, double nAtoms)
{
    double numer = nAtoms * q->domainLength;
    return (float)(numer / (numer + q->alpha * q->numBins * (q->domainLength - nAtoms)));
}

/* AtomicProposal ctor (ProposalQueue.cpp:12-15) */
static void prop_init(go_prop *p, char t, go_randstate *rs)
{
    rng_init(&p->rng, rs);
    p->pos = 0; p->atom1 = GO_NONE; p->atom2 = GO_NONE; p->r1 = p->c1 = p->r2 = p->c2 = 0; p->type = t;
}

/* static_cast<uint64_t>(double) as x86-64 gcc evaluates it when the double is 2^64 (-> 0) */
static inline uint64_t u64_from_double_x86(double d)
{
    if (d >= 18446744073709551616.0) return 0ull;
    return (uint64_t)d;
}

/* ProposalQueue.cpp:162-187 */
static int queue_birth(go_queue *q, go_domain *dom)
{
    go_prop prop; prop_init(&prop, 'B', q->rs);
    uint64_t pos = dom_random_free_position(dom, &prop.rng);
    if (moves_overlap(q, pos)) { seeder_rollback(&q->rs->seeder); return 0; }
    prop.r1 = (uint32_t)((pos / q->binLength) / q->numCols);
    prop.c1 = (uint32_t)((pos / q->binLength) % q->numCols);
    if (rows_contains(q, prop.r1)) { seeder_rollback(&q->rs->seeder); return 0; }
    prop.atom1 = dom_insert(dom, pos, 0.f);
    rows_insert(q, prop.r1);
    atoms_insert(q, dom->pool[prop.atom1].pos);
    queue_push(q, &prop);
    ++q->maxAtoms;
    return 1;
}
/* ProposalQueue.cpp:189-207 */
static int queue_death(go_queue *q, go_domain *dom)
{
    go_prop prop; prop_init(&prop, 'D', q->rs);
    prop.atom1 = dom->vec[rng_uniform32(&prop.rng, 0, dom->n - 1)]; /* ConcurrentAtomicDomain.cpp:32-37 */
    uint64_t p1 = dom->pool[prop.atom1].pos;
    prop.r1 = (uint32_t)((p1 / q->binLength) / q->numCols);
    prop.c1 = (uint32_t)((p1 / q->binLength) % q->numCols);
    if (rows_contains(q, prop.r1)) { seeder_rollback(&q->rs->seeder); return 0; }
    rows_insert(q, prop.r1);
    atoms_insert(q, p1);
    queue_push(q, &prop);
    --q->minAtoms;
    return 1;
}
/* ProposalQueue.cpp:209-248 */
static int queue_move(go_queue *q, go_domain *dom)
{
    go_prop prop; prop_init(&prop, 'M', q->rs);
    uint32_t idx = rng_uniform32(&prop.rng, 0, dom->n - 1); /* ConcurrentAtomicDomain.cpp:39-44 */
    prop.atom1 = dom->vec[idx];
    const go_atom *c = &dom->pool[prop.atom1];
    uint64_t lbound = c->left != GO_NONE ? dom->pool[c->left].pos : 0;
    uint64_t rbound = c->right != GO_NONE ? dom->pool[c->right].pos : u64_from_double_x86(q->domainLength);
    if (atoms_contains(q, lbound) || atoms_contains(q, rbound)) { seeder_rollback(&q->rs->seeder); return 0; }
    prop.pos = rng_uniform64(&prop.rng, lbound + 1, rbound - 1);
    prop.r1 = (uint32_t)((c->pos / q->binLength) / q->numCols);
    prop.c1 = (uint32_t)((c->pos / q->binLength) % q->numCols);
    prop.r2 = (uint32_t)((prop.pos / q->binLength) / q->numCols);
    prop.c2 = (uint32_t)((prop.pos / q->binLength) % q->numCols);
    if (rows_contains(q, prop.r1) || rows_contains(q, prop.r2)) { seeder_rollback(&q->rs->seeder); return 0; }
    if (prop.r1 == prop.r2 && prop.c1 == prop.c2) { dom_move(dom, prop.atom1, prop.pos); return 1; }
    queue_push(q, &prop);
    rows_insert(q, prop.r1);
    rows_insert(q, prop.r2);
    atoms_insert(q, c->pos);
    moves_insert(q, c->pos, prop.pos);
    return 1;
}
/* ProposalQueue.cpp:250-283 */
static int queue_exchange(go_queue *q, go_domain *dom)
{
    go_prop prop; prop_init(&prop, 'E', q->rs);
    uint32_t idx = rng_uniform32(&prop.rng, 0, dom->n - 1);
    prop.atom1 = dom->vec[idx];
    go_atom *a1 = &dom->pool[prop.atom1];
    prop.atom2 = a1->right != GO_NONE ? a1->right : dom_front(dom);
    go_atom *a2 = &dom->pool[prop.atom2];
    prop.r1 = (uint32_t)((a1->pos / q->binLength) / q->numCols);
    prop.c1 = (uint32_t)((a1->pos / q->binLength) % q->numCols);
    prop.r2 = (uint32_t)((a2->pos / q->binLength) / q->numCols);
    prop.c2 = (uint32_t)((a2->pos / q->binLength) % q->numCols);
    if (rows_contains(q, prop.r1) || rows_contains(q, prop.r2)) { seeder_rollback(&q->rs->seeder); return 0; }
    if (prop.r1 == prop.r2 && prop.c1 == prop.c2) {
        float newMass = rng_trunc_gamma_upper(&prop.rng, a1->mass + a2->mass, 1.f / q->lambda);
        float delta = (a1->mass > a2->mass) ? newMass - a1->mass : a2->mass - newMass;
        if (a1->mass + delta > GO_EPSILON && a2->mass - delta > GO_EPSILON) {
            float m1 = a1->mass + delta, m2 = a2->mass - delta;
            a1->mass = m1; a2->mass = m2;
        }
        return 1;
    }
    queue_push(q, &prop);
    rows_insert(q, prop.r1);
    rows_insert(q, prop.r2);
    return 1;
}
/* ProposalQueue.cpp:129-160 */
static int queue_make_proposal(go_queue *q, go_domain *dom)
{
    q->u1 = q->useCachedRng ? q->u1 : rng_uniform(&q->rng);
    q->u2 = q->useCachedRng ? q->u2 : rng_uniform(&q->rng);
    q->useCachedRng = 0;
    if (q->minAtoms < 2 && q->maxAtoms >= 2) return 0;
    if (q->maxAtoms < 2) return queue_birth(q, dom);
    float lowerBound = queue_death_prob(q, (double)q->minAtoms);
    float upperBound = queue_death_prob(q, (double)q->maxAtoms);
    if (q->u1 < 0.5f) {
        if (q->u2 < lowerBound) return queue_death(q, dom);
        if (q->u2 >= upperBound) return queue_birth(q, dom);
        return 0;
    }
    return (q->u1 < 0.75f) ? queue_move(q, dom) : queue_exchange(q, dom);
}
/* ProposalQueue.cpp:53-76 */
static void queue_populate(go_queue *q, go_domain *dom, unsigned limit)
{
    int success = 1;
    q->numProcessed = 0;
    while (q->numProcessed < limit && success) {
        if (!queue_make_proposal(q, dom)) { success = 0; q->useCachedRng = 1; }
        else ++q->numProcessed;
    }
}

/* ======================================================================================
 * DenseNormalModel + AsynchronousGibbsSampler
 * ====================================================================================== */

typedef struct go_sampler {
    uint32_t M, N, K;       /* factor rows, data-vector length, patterns */
    float *D, *S, *AP;      /* [M][N]: vector r = column r of mDMatrix (DenseNormalModel.h:56-60) */
    float *mat;             /* column-major M x K: mat[k*M + r] */
    const float *other;     /* other sampler's mat: column-major N x K */
    float maxGibbsMass, annealTemp, lambda, alpha;
    uint32_t redW, redG;
    go_domain dom; go_queue queue;
    float avgQueue, nQueueSamples;
    /* ---- SparseNormalModel (SparseNormalModel.h:56-64): D as bit flags + packed non-zeros per vector, mMatrix as a
     * HybridMatrix = row copy `rows` [M][K] + column copy `mat` [K][M] with bit flags `mflags` [K][Mw] ---- */
    int sparse;
    uint64_t *dflags; uint32_t Wn;      /* [M][Wn], Wn = N/64 + 1 (SparseVector.cpp:14-18) */
    uint32_t *dptr; float *dvals;       /* packed values of vector r: dvals[dptr[r] .. dptr[r+1]) */
    float *rows; uint64_t *mflags; uint32_t Mw;
    const struct go_sampler *oth;       /* mOtherMatrix */
    float *Z1, *Z2;                     /* lookup tables, Z2 column-major K x K */
    float beta;
} go_sampler;

/* lane-strided reduction (generalises SIMD.h PackedFloat: SIMD_INC lanes, here W lanes with G
 * consecutive elements per lane slot) followed by an ascending xor butterfly (the AVX hadd tree
 * of SIMD.h:102-107 widened to W lanes).  W <= 1 is the scalar build's sequential order. */
#define GO_MAXW 16384
typedef struct { float s[GO_MAXW], m[GO_MAXW]; } lane_acc;

static void lanes_finish(lane_acc *a, uint32_t W, float *s, float *smu)
{
    for (uint32_t off = 1; off < W; off <<= 1) {
        for (uint32_t L = 0; L < W; ++L) {
            if ((L & off) == 0) {
                float ts = a->s[L] + a->s[L ^ off], tm = a->m[L] + a->m[L ^ off];
                a->s[L] = ts; a->s[L ^ off] = ts; a->m[L] = tm; a->m[L ^ off] = tm;
            }
        }
    }
    *s = a->s[0]; *smu = a->m[0];
}

/* DenseNormalModel.cpp:162-183 (ch == NULL) and :217-240 (with change) */
static void sp_alpha_one(const go_sampler *sm, uint32_t row, uint32_t col, const float *ch, float *s_out, float *smu_out);
static void sp_alpha_two(const go_sampler *sm, uint32_t r1, uint32_t c1, uint32_t r2, uint32_t c2, float *s_out, float *smu_out);
static void alpha_one(const go_sampler *sm, uint32_t row, uint32_t col, const float *ch, float *s_out, float *smu_out)
{
    if (sm->sparse) { sp_alpha_one(sm, row, col, ch, s_out, smu_out); return; }
    const uint32_t N = sm->N;
    const float *D = sm->D + (size_t)row * N, *S = sm->S + (size_t)row * N, *AP = sm->AP + (size_t)row * N;
    const float *mat = sm->other + (size_t)col * N;
    if (sm->redW <= 1) {
        float s = 0.f, smu = 0.f;
        if (ch) { const float c = *ch; for (uint32_t i = 0; i < N; ++i) { float ratio = mat[i] / (S[i] * S[i]); s += mat[i] * ratio; smu += ratio * (D[i] - (AP[i] + c * mat[i])); } }
        else { for (uint32_t i = 0; i < N; ++i) { float ratio = mat[i] / (S[i] * S[i]); s += mat[i] * ratio; smu += ratio * (D[i] - AP[i]); } }
        *s_out = s; *smu_out = smu; return;
    }
    lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
    for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
    for (uint32_t i = 0; i < N; ++i) {
        uint32_t L = (i / G) % W;
        float ratio = mat[i] / (S[i] * S[i]);
        a.s[L] += mat[i] * ratio;
        a.m[L] += ch ? ratio * (D[i] - (AP[i] + *ch * mat[i])) : ratio * (D[i] - AP[i]);
    }
    lanes_finish(&a, W, s_out, smu_out);
}
/* DenseNormalModel.cpp:186-214 */
static void alpha_two(const go_sampler *sm, uint32_t r1, uint32_t c1, uint32_t r2, uint32_t c2, float *s_out, float *smu_out)
{
    if (sm->sparse) { sp_alpha_two(sm, r1, c1, r2, c2, s_out, smu_out); return; }
    if (r1 == r2) {
        const uint32_t N = sm->N;
        const float *D = sm->D + (size_t)r1 * N, *S = sm->S + (size_t)r1 * N, *AP = sm->AP + (size_t)r1 * N;
        const float *m1 = sm->other + (size_t)c1 * N, *m2 = sm->other + (size_t)c2 * N;
        if (sm->redW <= 1) {
            float s = 0.f, smu = 0.f;
            for (uint32_t i = 0; i < N; ++i) { float d = m1[i] - m2[i]; float ratio = d / (S[i] * S[i]); s += d * ratio; smu += ratio * (D[i] - AP[i]); }
            *s_out = s; *smu_out = smu; return;
        }
        lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
        for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
        for (uint32_t i = 0; i < N; ++i) {
            uint32_t L = (i / G) % W;
            float d = m1[i] - m2[i]; float ratio = d / (S[i] * S[i]);
            a.s[L] += d * ratio; a.m[L] += ratio * (D[i] - AP[i]);
        }
        lanes_finish(&a, W, s_out, smu_out);
        return;
    }
    float sa, ma, sb, mb;
    alpha_one(sm, r1, c1, NULL, &sa, &ma);
    alpha_one(sm, r2, c2, NULL, &sb, &mb);
    *s_out = sa + sb; *smu_out = ma - mb; /* AlphaParameters.cpp:11-14 "minus sign not a typo" */
}
/* ======================================================================================
 * SparseNormalModel
 * ====================================================================================== */

/* gaps::dot in the scalar build (VectorMath.h:41-134, SIMD_INC = 1): for size <= 25 the fall-through switch adds
 * element size-1 first and element 0 last; longer vectors are added front to back */
static float sp_dot(const float *a, const float *b, uint32_t n)
{
    float d = 0.f;
    if (n <= 25) { for (uint32_t i = n; i-- > 0;) d = d + a[i] * b[i]; }
    else { for (uint32_t i = 0; i < n; ++i) d = d + a[i] * b[i]; }
    return d;
}
/* VectorMath.h:137-155 */
static float sp_dot_diff(const float *a, const float *b, const float *c, uint32_t n)
{
    float d = 0.f;
    for (uint32_t i = 0; i < n; ++i) d += a[i] * (b[i] - c[i]);
    return d;
}
/* virtual lanes of the sparse reductions in lane mode: one per 64-bit flag word up to 256 */
static uint32_t sp_width(uint32_t N)
{
    uint32_t need = N / 64 + 1, w = 64;
    while (w < need && w < 256) w <<= 1;
    return w;
}
/* HybridVector::add / set (HybridVector.cpp:55-86) on column `col` of the column copy */
static void hv_add(go_sampler *sm, uint32_t row, uint32_t col, float v)
{
    float *e = &sm->mat[(size_t)col * sm->M + row];
    uint64_t *f = &sm->mflags[(size_t)col * sm->Mw + row / 64];
    if (*e + v < GO_EPSILON) {
#pragma omp atomic
        *f &= ~(1ull << (row % 64));
        *e = 0.f;
    } else {
#pragma omp atomic
        *f |= (1ull << (row % 64));
        *e += v;
    }
}
static void hv_set(go_sampler *sm, uint32_t row, uint32_t col, float v)
{
    float *e = &sm->mat[(size_t)col * sm->M + row];
    uint64_t *f = &sm->mflags[(size_t)col * sm->Mw + row / 64];
    if (v < GO_EPSILON) {
#pragma omp atomic
        *f &= ~(1ull << (row % 64));
        *e = 0.f;
    } else {
#pragma omp atomic
        *f |= (1ull << (row % 64));
        *e = v;
    }
}
/* SparseNormalModel.cpp:153-193 (ch == NULL) and :196-239.  Lane mode: virtual lane L of sp_width(N) takes the flag
 * words w = L, L+W, ... in increasing w (bits in increasing order) and accumulates its terms from +0; the lanes are
 * folded by the ascending xor butterfly and the total is added to the table terms once. */
static void sp_alpha_one(const go_sampler *sm, uint32_t row, uint32_t col, const float *ch, float *s_out, float *smu_out)
{
    const go_sampler *ot = sm->oth;
    const uint32_t K = sm->K;
    const uint64_t *fD = sm->dflags + (size_t)row * sm->Wn, *fV = ot->mflags + (size_t)col * ot->Mw;
    const float *data = sm->dvals + sm->dptr[row];
    const float *V = ot->mat + (size_t)col * ot->M;
    const float *arow = sm->rows + (size_t)row * K;
    float s = sm->Z1[col];
    float s_mu = -1.f * sp_dot(arow, sm->Z2 + (size_t)col * K, K);
    if (ch) s_mu -= *ch * sm->Z2[(size_t)col * K + col];
    const int lanes = sm->redW > 1;
    const uint32_t W = sp_width(sm->N);
    lane_acc a;
    if (lanes) for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
    unsigned sparseIndex = 0;
    for (uint32_t i = 0; i < sm->Wn; ++i) {
        uint64_t d_flags = fD[i];
        uint64_t common = d_flags & fV[i];
        float *ps = lanes ? &a.s[i % W] : &s, *pm = lanes ? &a.m[i % W] : &s_mu;
        while (common != 0u) {
            unsigned index = (unsigned)__builtin_ffsll((long long)common) - 1;
            sparseIndex += (unsigned)__builtin_popcountll(d_flags & ((1ull << index) - 1ull));
            d_flags = (index == 63) ? 0 : d_flags & ~((1ull << (index + 1ull)) - 1ull);
            common &= d_flags;
            unsigned v_ndx = 64 * i + index;
            float v_val = V[v_ndx];
            float d_val = data[sparseIndex++];
            float term1 = v_val / d_val;
            float term2 = v_val - term1 / d_val;
            *ps += term1 * term1 - v_val * v_val;
            *pm += term1 + term2 * sp_dot(arow, ot->rows + (size_t)v_ndx * K, K);
            if (ch) *pm += term2 * ot->rows[(size_t)v_ndx * K + col] * *ch;
        }
        sparseIndex += (unsigned)__builtin_popcountll(d_flags);
    }
    if (lanes) { float ts, tm; lanes_finish(&a, W, &ts, &tm); s += ts; s_mu += tm; }
    *s_out = s * sm->beta; *smu_out = s_mu * sm->beta;
}
/* SparseNormalModel.cpp:242-292 */
static void sp_alpha_two(const go_sampler *sm, uint32_t r1, uint32_t c1, uint32_t r2, uint32_t c2, float *s_out, float *smu_out)
{
    if (r1 == r2) {
        const go_sampler *ot = sm->oth;
        const uint32_t K = sm->K;
        const uint64_t *fD = sm->dflags + (size_t)r1 * sm->Wn, *fV1 = ot->mflags + (size_t)c1 * ot->Mw, *fV2 = ot->mflags + (size_t)c2 * ot->Mw;
        const float *data = sm->dvals + sm->dptr[r1];
        const float *V1 = ot->mat + (size_t)c1 * ot->M, *V2 = ot->mat + (size_t)c2 * ot->M;
        const float *arow = sm->rows + (size_t)r1 * K;
        float s = sm->Z1[c1] - 2.f * sm->Z2[(size_t)c2 * K + c1] + sm->Z1[c2];
        float s_mu = -1.f * sp_dot_diff(arow, sm->Z2 + (size_t)c1 * K, sm->Z2 + (size_t)c2 * K, K);
        const int lanes = sm->redW > 1;
        const uint32_t W = sp_width(sm->N);
        lane_acc a;
        if (lanes) for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
        unsigned sparseIndex = 0;
        for (uint32_t i = 0; i < sm->Wn; ++i) {
            uint64_t d_flags = fD[i];
            uint64_t common = d_flags & (fV1[i] | fV2[i]);
            float *ps = lanes ? &a.s[i % W] : &s, *pm = lanes ? &a.m[i % W] : &s_mu;
            while (common != 0u) {
                unsigned index = (unsigned)__builtin_ffsll((long long)common) - 1;
                sparseIndex += (unsigned)__builtin_popcountll(d_flags & ((1ull << index) - 1ull));
                d_flags = (index == 63) ? 0 : d_flags & ~((1ull << (index + 1ull)) - 1ull);
                common &= d_flags;
                unsigned v_ndx = 64 * i + index;
                float v1_val = V1[v_ndx], v2_val = V2[v_ndx];
                float d_val = data[sparseIndex++];
                float d_recip = 1.f / d_val;
                float term1 = 1.f - d_recip * d_recip;
                float v_diff = v1_val - v2_val;
                float ap = sp_dot(arow, ot->rows + (size_t)v_ndx * K, K);
                *ps -= v_diff * v_diff * term1;
                *pm += v_diff * (ap * term1 + d_recip);
            }
            sparseIndex += (unsigned)__builtin_popcountll(d_flags);
        }
        if (lanes) { float ts, tm; lanes_finish(&a, W, &ts, &tm); s += ts; s_mu += tm; }
        *s_out = s * sm->beta; *smu_out = s_mu * sm->beta;
        return;
    }
    float sa, ma, sb, mb;
    sp_alpha_one(sm, r1, c1, NULL, &sa, &ma);
    sp_alpha_one(sm, r2, c2, NULL, &sb, &mb);
    *s_out = sa + sb; *smu_out = ma - mb;
}
/* SparseNormalModel::generateLookupTables (SparseNormalModel.cpp:294-311).  Z1 reads the other matrix through
 * operator() = its row copy, Z2 through the column copies.  Lane mode: the dense lane order over the N elements. */
static void sp_tables(go_sampler *sm)
{
    const go_sampler *ot = sm->oth;
    const uint32_t K = sm->K, N = ot->M;
    for (uint32_t i = 0; i < K; ++i) {
        if (sm->redW <= 1) {
            float z = 0.f;
            for (uint32_t k = 0; k < N; ++k) { float v = ot->rows[(size_t)k * K + i]; z += v * v; }
            sm->Z1[i] = z;
        } else {
            lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
            for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
            for (uint32_t k = 0; k < N; ++k) { float v = ot->rows[(size_t)k * K + i]; a.s[(k / G) % W] += v * v; }
            float ts, tm; lanes_finish(&a, W, &ts, &tm); sm->Z1[i] = ts;
        }
        for (uint32_t j = i; j < K; ++j) {
            const float *ci = ot->mat + (size_t)i * N, *cj = ot->mat + (size_t)j * N;
            float d;
            if (sm->redW <= 1) d = sp_dot(ci, cj, N);
            else {
                lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
                for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
                for (uint32_t k = 0; k < N; ++k) a.s[(k / G) % W] += ci[k] * cj[k];
                float tm; lanes_finish(&a, W, &d, &tm);
            }
            sm->Z2[(size_t)j * K + i] = d; sm->Z2[(size_t)i * K + j] = d;
        }
    }
}
/* SparseNormalModel::chiSq (SparseNormalModel.cpp:40-62).  Lane mode: per vector j, lane (i/G)%W takes element i
 * (the dense term, then the non-zero correction of the same element), butterfly, vectors added in order. */
static float sp_chisq(const go_sampler *sm)
{
    const go_sampler *ot = sm->oth;
    const uint32_t K = sm->K;
    float chisq = 0.f;
    for (uint32_t j = 0; j < sm->M; ++j) {
        const float *arow = sm->rows + (size_t)j * K;
        const uint64_t *fD = sm->dflags + (size_t)j * sm->Wn;
        const float *data = sm->dvals + sm->dptr[j];
        if (sm->redW <= 1) {
            for (uint32_t i = 0; i < sm->N; ++i) { float dot = sp_dot(arow, ot->rows + (size_t)i * K, K); chisq += dot * dot; }
            unsigned si = 0;
            for (uint32_t w = 0; w < sm->Wn; ++w) {
                uint64_t fl = fD[w];
                while (fl) {
                    unsigned b = (unsigned)__builtin_ffsll((long long)fl) - 1; fl &= fl - 1;
                    float d = data[si++];
                    float dot = sp_dot(arow, ot->rows + (size_t)(64 * w + b) * K, K);
                    float dsq = d * d;
                    chisq += 1 + dot * (dot - 2 * d - dsq * dot) / dsq;
                }
            }
        } else {
            lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
            for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
            unsigned si = 0;
            for (uint32_t i = 0; i < sm->N; ++i) {
                float dot = sp_dot(arow, ot->rows + (size_t)i * K, K);
                float *acc = &a.s[(i / G) % W];
                *acc += dot * dot;
                if ((fD[i / 64] >> (i % 64)) & 1ull) {
                    float d = data[si++];
                    float dsq = d * d;
                    *acc += 1 + dot * (dot - 2 * d - dsq * dot) / dsq;
                }
            }
            float ps, pm; lanes_finish(&a, W, &ps, &pm);
            chisq += ps;
        }
    }
    return chisq * sm->beta;
}

/* DenseNormalModel.cpp:243-258 */
static void update_ap(go_sampler *sm, uint32_t row, uint32_t col, float delta)
{
    const float *other = sm->other + (size_t)col * sm->N;
    float *ap = sm->AP + (size_t)row * sm->N;
    for (uint32_t i = 0; i < sm->N; ++i) ap[i] += delta * other[i];
}
/* DenseNormalModel.cpp:110-115 */
static void change_matrix(go_sampler *sm, uint32_t row, uint32_t col, float delta)
{
    if (sm->sparse) { sm->rows[(size_t)row * sm->K + col] += delta; hv_add(sm, row, col, delta); return; }   /* HybridMatrix::add, HybridMatrix.cpp:25-31 */
    sm->mat[(size_t)col * sm->M + row] += delta;
    update_ap(sm, row, col, delta);
}
/* DenseNormalModel.cpp:117-123 */
static void safely_change_matrix(go_sampler *sm, uint32_t row, uint32_t col, float delta)
{
    if (sm->sparse) {   /* SparseNormalModel.cpp:117-122: mMatrix(row,col) is the row copy; HybridMatrix::set */
        float newVal = fmax_ref(sm->rows[(size_t)row * sm->K + col] + delta, 0.f);
        sm->rows[(size_t)row * sm->K + col] = newVal; hv_set(sm, row, col, newVal); return;
    }
    float *e = &sm->mat[(size_t)col * sm->M + row];
    float newVal = fmax_ref(*e + delta, 0.f);
    update_ap(sm, row, col, newVal - *e);
    *e = newVal;
}
/* DenseNormalModel.cpp:100-108, VectorMath.cpp:113-123 */
static int can_use_gibbs(const go_sampler *sm, uint32_t col)
{
    if (sm->sparse) {   /* isVectorZero(HybridVector) = empty(): no flag set (VectorMath.cpp:125-128) */
        const uint64_t *f = sm->oth->mflags + (size_t)col * sm->oth->Mw;
        for (uint32_t i = 0; i < sm->oth->Mw; ++i) if (f[i]) return 1;
        return 0;
    }
    const float *v = sm->other + (size_t)col * sm->N;
    for (uint32_t i = 0; i < sm->N; ++i) if (v[i] > 0.f) return 1;
    return 0;
}
/* AlphaParameters.cpp:27-36 / 38-48 */
static optf gibbs_mass(float s, float s_mu, float a, float b, go_rng *rng, int useLambda, float lambda)
{
    if (s > GO_EPSILON) {
        float mean = useLambda ? (s_mu - lambda) / s : s_mu / s;
        float sd = 1.f / sqrtf(s);
        return rng_trunc_normal(rng, a, b, mean, sd);
    }
    return optf_none();
}

/* AsynchronousGibbsSampler.h:127-144 */
static void eval_birth(go_sampler *sm, go_prop *p)
{
    optf mass;
    if (can_use_gibbs(sm, p->c1)) {
        float s, smu; alpha_one(sm, p->r1, p->c1, NULL, &s, &smu);   /* sampleBirth, DenseNormalModel.cpp:132-136 */
        s *= sm->annealTemp; smu *= sm->annealTemp;
        mass = gibbs_mass(s, smu, 0.f, sm->maxGibbsMass, &p->rng, 1, sm->lambda);
    } else {
        mass = optf_some(rng_exponential(&p->rng, sm->lambda));
    }
    if (mass.has && mass.v >= GO_EPSILON) {
#pragma omp atomic
        ++sm->queue.minAtoms;  /* acceptBirth */
        sm->dom.pool[p->atom1].mass = mass.v;
        change_matrix(sm, p->r1, p->c1, mass.v);
        return;
    }
#pragma omp atomic
    --sm->queue.maxAtoms;      /* rejectBirth */
    dom_cache_erase(&sm->dom, p->atom1);
}
/* AsynchronousGibbsSampler.h:148-180 */
static void eval_death(go_sampler *sm, go_prop *p)
{
    go_atom *a = &sm->dom.pool[p->atom1];
    float rebirthMass = a->mass;
    float ch = -1.f * a->mass;
    float s, smu; alpha_one(sm, p->r1, p->c1, &ch, &s, &smu);
    s *= sm->annealTemp; smu *= sm->annealTemp;
    if (can_use_gibbs(sm, p->c1)) {
        optf g = gibbs_mass(s, smu, 0.f, sm->maxGibbsMass, &p->rng, 1, sm->lambda);
        if (g.has) rebirthMass = g.v;
    }
    float deltaLL = rebirthMass * (smu - s * rebirthMass / 2.f);
    if (go_logf(p->rng.rs, rng_uniform(&p->rng)) < deltaLL) {
#pragma omp atomic
        ++sm->queue.minAtoms;  /* rejectDeath */
        if (rebirthMass != a->mass) {
            safely_change_matrix(sm, p->r1, p->c1, rebirthMass - a->mass);
            a->mass = rebirthMass;
        }
    } else {
#pragma omp atomic
        --sm->queue.maxAtoms;  /* acceptDeath */
        safely_change_matrix(sm, p->r1, p->c1, -1.f * a->mass);
        dom_cache_erase(&sm->dom, p->atom1);
    }
}
/* AsynchronousGibbsSampler.h:184-196 + DenseNormalModel.cpp:125-130 */
static void eval_move(go_sampler *sm, go_prop *p)
{
    go_atom *a = &sm->dom.pool[p->atom1];
    float s, smu; alpha_two(sm, p->r1, p->c1, p->r2, p->c2, &s, &smu);
    s *= sm->annealTemp; smu *= sm->annealTemp;
    float mass = a->mass;
    float deltaLL = -1.f * mass * (smu + s * mass / 2.f);
    if (go_logf(p->rng.rs, rng_uniform(&p->rng)) < deltaLL) {
        dom_move(&sm->dom, p->atom1, p->pos);
        safely_change_matrix(sm, p->r1, p->c1, -a->mass);
        change_matrix(sm, p->r2, p->c2, a->mass);
    }
}
/* AsynchronousGibbsSampler.h:201-219 + DenseNormalModel.cpp:154-159 */
static void eval_exchange(go_sampler *sm, go_prop *p)
{
    go_atom *a1 = &sm->dom.pool[p->atom1], *a2 = &sm->dom.pool[p->atom2];
    if (can_use_gibbs(sm, p->c1) || can_use_gibbs(sm, p->c2)) {
        float s, smu; alpha_two(sm, p->r1, p->c1, p->r2, p->c2, &s, &smu);
        s *= sm->annealTemp; smu *= sm->annealTemp;
        optf mass = gibbs_mass(s, smu, -a1->mass, a2->mass, &p->rng, 0, 0.f);
        float newMass1 = a1->mass + mass.v;
        float newMass2 = a2->mass - mass.v;
        if (mass.has && newMass1 > GO_EPSILON && newMass2 > GO_EPSILON) {
            safely_change_matrix(sm, p->r1, p->c1, newMass1 - a1->mass);
            safely_change_matrix(sm, p->r2, p->c2, newMass2 - a2->mass);
            a1->mass = newMass1;
            a2->mass = newMass2;
        }
    }
}

static void trace_batch(go_trace *t, const go_sampler *sm, uint32_t batch)
{
    if (!t) return;
    if (t->n_batches < t->batch_cap) { t->batch_nproc[t->n_batches] = sm->queue.numProcessed; t->batch_qlen[t->n_batches] = sm->queue.nq; }
    t->n_batches++;
    for (uint32_t i = 0; i < sm->queue.nq; ++i) {
        if (t->n < t->cap) {
            const go_prop *p = &sm->queue.q[i]; go_trace_rec *r = &t->rec[t->n];
            r->pos = p->pos; r->rng_state = p->rng.state;
            r->atom1 = p->atom1 != GO_NONE ? sm->dom.pool[p->atom1].index : GO_NONE;
            r->atom2 = p->atom2 != GO_NONE ? sm->dom.pool[p->atom2].index : GO_NONE;
            r->r1 = p->r1; r->c1 = p->c1; r->r2 = p->r2; r->c2 = p->c2; r->type = (uint32_t)p->type; r->batch = batch;
        }
        t->n++;
    }
}

/* AsynchronousGibbsSampler.h:88-122 */
static void sampler_update(go_sampler *sm, unsigned nSteps, unsigned nThreads, go_trace *trace)
{
    unsigned n = 0; uint32_t batch = 0;
    while (n < nSteps) {
        queue_populate(&sm->queue, &sm->dom, nSteps - n);
        n += sm->queue.numProcessed;
        if (n < nSteps) {
            sm->nQueueSamples += 1.f;
            sm->avgQueue *= (sm->nQueueSamples - 1.f) / sm->nQueueSamples;
            sm->avgQueue += (float)sm->queue.nq / sm->nQueueSamples;
        }
        trace_batch(trace, sm, batch++);
        const int nq = (int)sm->queue.nq;
        (void)nThreads;
#pragma omp parallel for num_threads(nThreads) schedule(static)
        for (int i = 0; i < nq; ++i) {
            go_prop *p = &sm->queue.q[i];
            switch (p->type) {
                case 'B': eval_birth(sm, p); break;
                case 'D': eval_death(sm, p); break;
                case 'M': eval_move(sm, p); break;
                case 'E': eval_exchange(sm, p); break;
            }
        }
        queue_clear(&sm->queue);
        dom_flush_erase(&sm->dom);
    }
}

/* ---- sampler construction: Matrix(mat, genesInCols, subsetGenes, indices) (Matrix.cpp:30-69) into
 * D[vector j][element i], then DenseNormalModel ctor (DenseNormalModel.h:66-88) ---- */
static void sampler_init(go_sampler *sm, const float *data, uint32_t nrow, uint32_t ncol, const float *unc,
                         int genesInCols, int subsetGenes, const uint32_t *indices, uint32_t nIdx,
                         uint32_t K, float alpha, float maxGibbsMass, go_randstate *rs, uint32_t redW, uint32_t redG, int sparse)
{
    memset(sm, 0, sizeof(*sm));
    sm->sparse = sparse; sm->beta = 100.f;
    int subsetData = nIdx != 0;
    uint32_t nGenes = (subsetData && subsetGenes) ? nIdx : (genesInCols ? ncol : nrow);
    uint32_t nSamples = (subsetData && !subsetGenes) ? nIdx : (genesInCols ? nrow : ncol);
    /* resulting Matrix is nGenes x nSamples (rows x cols); vector j (a column) has nGenes elements */
    sm->N = nGenes; sm->M = nSamples; sm->K = K;
    size_t tot = (size_t)sm->M * sm->N;
    sm->D = (float *)malloc(tot * sizeof(float));
    sm->S = (float *)malloc(tot * sizeof(float));
    sm->AP = (float *)calloc(tot, sizeof(float));
    sm->mat = (float *)calloc((size_t)sm->M * K, sizeof(float));
    for (uint32_t j = 0; j < nSamples; ++j) {
        for (uint32_t i = 0; i < nGenes; ++i) {
            uint32_t dataRow = (subsetData && (subsetGenes != genesInCols)) ? indices[genesInCols ? j : i] - 1 : (genesInCols ? j : i);
            uint32_t dataCol = (subsetData && (subsetGenes == genesInCols)) ? indices[genesInCols ? i : j] - 1 : (genesInCols ? i : j);
            sm->D[(size_t)j * sm->N + i] = data[(size_t)dataRow * ncol + dataCol];
            if (unc && !sparse) sm->S[(size_t)j * sm->N + i] = unc[(size_t)dataRow * ncol + dataCol];
        }
    }
    if (sparse) {
        /* SparseMatrix(mat, ...) (SparseMatrix.cpp:10-47): SparseVector keeps v > 0 only (SparseVector.cpp:20-33);
         * the uncertainty is always the default one (SparseNormalModel.h:90-96) */
        sm->Wn = sm->N / 64 + 1; sm->Mw = sm->M / 64 + 1;
        sm->dflags = (uint64_t *)calloc((size_t)sm->M * sm->Wn, 8);
        sm->dptr = (uint32_t *)calloc((size_t)sm->M + 1, 4);
        size_t nz = 0;
        for (size_t t = 0; t < tot; ++t) { if (!(sm->D[t] > 0.f)) sm->D[t] = 0.f; else ++nz; }
        sm->dvals = (float *)malloc((nz ? nz : 1) * 4);
        nz = 0;
        for (uint32_t j = 0; j < sm->M; ++j) {
            sm->dptr[j] = (uint32_t)nz;
            for (uint32_t i = 0; i < sm->N; ++i) {
                float v = sm->D[(size_t)j * sm->N + i];
                if (v > 0.f) { sm->dvals[nz++] = v; sm->dflags[(size_t)j * sm->Wn + i / 64] |= 1ull << (i % 64); }
            }
        }
        sm->dptr[sm->M] = (uint32_t)nz;
        sm->rows = (float *)calloc((size_t)sm->M * K, 4);
        sm->mflags = (uint64_t *)calloc((size_t)K * sm->Mw, 8);
        sm->Z1 = (float *)calloc(K, 4); sm->Z2 = (float *)calloc((size_t)K * K, 4);
    }
    if (!unc || sparse) { /* gaps::pmax(mDMatrix, 0.1f), MatrixMath.cpp:74-84 */
        for (size_t t = 0; t < tot; ++t) sm->S[t] = fmax_ref(sm->D[t] * 0.1f, 0.1f);
    }
    /* gaps::nonZeroMean, MatrixMath.cpp:39-55: column-major sequential sum */
    float sum = 0.f; unsigned nnz = 0;
    for (size_t t = 0; t < tot; ++t) { sum += sm->D[t]; if (sm->D[t] > 0.f) ++nnz; }
    float meanD = sum / (float)nnz;
    sm->alpha = alpha;
    sm->lambda = alpha * sqrtf((float)(uint64_t)K / meanD);  /* DenseNormalModel.h:79-80 */
    sm->maxGibbsMass = maxGibbsMass / sm->lambda;             /* :81 */
    sm->annealTemp = 1.f;
    sm->redW = redW; sm->redG = redG ? redG : 1;
    uint64_t nElements = (uint64_t)sm->M * K;
    dom_init(&sm->dom, nElements);                            /* AsynchronousGibbsSampler.h:66-67 */
    queue_init(&sm->queue, nElements, K, rs);
    sm->queue.alpha = (double)alpha;                          /* setAlpha */
    sm->queue.lambda = sm->lambda;                            /* setLambda */
}
static void sampler_free(go_sampler *sm)
{
    free(sm->D); free(sm->S); free(sm->AP); free(sm->mat);
    free(sm->dflags); free(sm->dptr); free(sm->dvals); free(sm->rows); free(sm->mflags); free(sm->Z1); free(sm->Z2);
    dom_free(&sm->dom); queue_free(&sm->queue);
}
/* DenseNormalModel.cpp:20-36 */
static void sampler_sync(go_sampler *dst, const go_sampler *src)
{
    if (dst->sparse) { dst->oth = src; dst->other = src->mat; sp_tables(dst); return; }   /* SparseNormalModel.cpp:27-31 */
    const uint32_t nc = src->M, nr = src->N; /* src AP: nr x nc; vector j of src has nr elements */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < (int64_t)nc; ++j)
        for (uint32_t i = 0; i < nr; ++i)
            dst->AP[(size_t)i * dst->N + (size_t)j] = src->AP[(size_t)j * src->N + i];
    dst->other = src->mat;
}
/* DenseNormalModel.cpp:38-54 */
static void sampler_extra_init(go_sampler *sm)
{
    if (sm->sparse) return;    /* SparseNormalModel.cpp:34-37 */
    for (uint32_t j = 0; j < sm->M; ++j)
        for (uint32_t i = 0; i < sm->N; ++i) {
            float acc = 0.f;
            for (uint32_t k = 0; k < sm->K; ++k) acc += sm->other[(size_t)k * sm->N + i] * sm->mat[(size_t)k * sm->M + j];
            sm->AP[(size_t)j * sm->N + i] = acc;
        }
}
/* DenseNormalModel.cpp:56-68: i over rows (element index) outer, j over columns (vectors) inner */
static float sampler_chisq(const go_sampler *sm)
{
    if (sm->sparse) return sp_chisq(sm);
    float chisq = 0.f;
    if (sm->redW <= 1) {
        for (uint32_t i = 0; i < sm->N; ++i)
            for (uint32_t j = 0; j < sm->M; ++j) {
                size_t t = (size_t)j * sm->N + i;
                float q = (sm->D[t] - sm->AP[t]) / sm->S[t];
                chisq += q * q;
            }
        return chisq;
    }
    /* lane mode: per-vector lane-strided partial, then sequential over vectors */
    for (uint32_t j = 0; j < sm->M; ++j) {
        lane_acc a; const uint32_t W = sm->redW, G = sm->redG;
        for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
        for (uint32_t i = 0; i < sm->N; ++i) {
            size_t t = (size_t)j * sm->N + i;
            float q = (sm->D[t] - sm->AP[t]) / sm->S[t];
            a.s[(i / G) % W] += q * q;
        }
        float ps, pm; lanes_finish(&a, W, &ps, &pm);
        chisq += ps;
    }
    return chisq;
}

/* ======================================================================================
 * Session = runCoGAPSAlgorithm state (GapsRunner.cpp:382-499) + GapsStatistics
 * ====================================================================================== */

struct go_session {
    go_params p;
    go_randstate *rs;
    go_sampler A, P;
    go_rng rng;
    uint32_t nGenes, nSamples, K;
    float *Amean, *Astd, *Pmean, *Pstd; /* column-major running sums */
    unsigned statUpdates;
    float *chisqHist; uint32_t *atomHistA, *atomHistP; uint32_t nHist, histCap;
    uint64_t totalUpdates;
    double samplerSeconds;
    float *pump; unsigned pumpUpdates;           /* mPumpMatrix [nGenes][K] row-major, mPumpUpdates */
    float *snapA[2], *snapP[2]; uint32_t nSnap[2], capSnap[2];   /* [0] equilibration, [1] sampling */
};

void go_default_params(go_params *p)
{
    memset(p, 0, sizeof(*p));
    p->nPatterns = 3; p->nIterations = 1000; p->seed = 0; p->outputFrequency = 500; p->maxThreads = 1;
    p->alphaA = 0.01f; p->alphaP = 0.01f; p->maxGibbsMassA = 100.f; p->maxGibbsMassP = 100.f;
    p->whichMatrixFixed = 'N'; p->math_mode = GO_MATH_LIBM; p->redW_A = 1; p->redW_P = 1; p->redG = 1;
}

go_session *go_create(const float *data, uint32_t nrow, uint32_t ncol, const go_params *p, const float *unc)
{
    go_session *s = (go_session *)calloc(1, sizeof(go_session));
    s->p = *p;
    s->rs = (go_randstate *)malloc(sizeof(go_randstate));
    randstate_init(s->rs, p->seed, p->math_mode);               /* Cogaps.cpp:158 */
    const uint32_t *idx = p->subsetData ? p->subsetIndices : NULL;
    uint32_t nIdx = p->subsetData ? p->nSubset : 0;
    /* GapsRunner.cpp:402-406: A sampler on the transposed data, flags flipped */
    sampler_init(&s->A, data, nrow, ncol, unc, !p->transposeData, !p->subsetGenes, idx, nIdx, p->nPatterns,
                 p->alphaA, p->maxGibbsMassA, s->rs, p->redW_A, p->redG, p->useSparseOptimization);
    sampler_init(&s->P, data, nrow, ncol, unc, p->transposeData, p->subsetGenes, idx, nIdx, p->nPatterns,
                 p->alphaP, p->maxGibbsMassP, s->rs, p->redW_P, p->redG, p->useSparseOptimization);
    s->nGenes = s->A.M; s->nSamples = s->P.M; s->K = p->nPatterns;
    /* processFixedMatrix, GapsRunner.cpp:329-350 */
    if ((p->whichMatrixFixed == 'A' || p->whichMatrixFixed == 'P') && p->fixedPatterns) {
        go_sampler *fx = p->whichMatrixFixed == 'A' ? &s->A : &s->P;
        for (uint32_t r = 0; r < fx->M; ++r) for (uint32_t k = 0; k < s->K; ++k) {
            const float v = p->fixedPatterns[(size_t)r * s->K + k];
            if (fx->sparse) { fx->rows[(size_t)r * s->K + k] = v; hv_add(fx, r, k, -1.f * fx->mat[(size_t)k * fx->M + r]); hv_add(fx, r, k, v); }   /* HybridMatrix::operator=(Matrix), HybridMatrix.cpp:70-84 */
            else fx->mat[(size_t)k * fx->M + r] = v;
        }
    }
    size_t na = (size_t)s->nGenes * s->K, np = (size_t)s->nSamples * s->K;
    s->Amean = (float *)calloc(na, 4); s->Astd = (float *)calloc(na, 4);
    s->Pmean = (float *)calloc(np, 4); s->Pstd = (float *)calloc(np, 4);
    s->pump = (float *)calloc(na, 4);
    rng_init(&s->rng, s->rs);                                   /* GapsRunner.cpp:437 */
    sampler_sync(&s->A, &s->P);                                 /* :444-447 */
    sampler_sync(&s->P, &s->A);
    sampler_extra_init(&s->A);
    sampler_extra_init(&s->P);
    return s;
}
void go_destroy(go_session *s)
{
    if (!s) return;
    sampler_free(&s->A); sampler_free(&s->P);
    free(s->Amean); free(s->Astd); free(s->Pmean); free(s->Pstd);
    free(s->pump); for (int w = 0; w < 2; ++w) { free(s->snapA[w]); free(s->snapP[w]); }
    free(s->chisqHist); free(s->atomHistA); free(s->atomHistP); free(s->rs); free(s);
}
static go_sampler *pick(go_session *s, char which) { return which == 'A' ? &s->A : &s->P; }
static const go_sampler *pickc(const go_session *s, char which) { return which == 'A' ? &s->A : &s->P; }

void go_set_annealing(go_session *s, float temp) { s->A.annealTemp = temp; s->P.annealTemp = temp; }
uint32_t go_natoms(const go_session *s, char which) { return pickc(s, which)->dom.n; }
void go_draw_steps(go_session *s, uint32_t *nA, uint32_t *nP)
{
    unsigned a = s->A.dom.n < 10 ? 10 : s->A.dom.n, p = s->P.dom.n < 10 ? 10 : s->P.dom.n;
    *nA = (uint32_t)rng_poisson(&s->rng, (double)a);
    *nP = (uint32_t)rng_poisson(&s->rng, (double)p);
}
void go_update(go_session *s, char which, uint32_t nSteps, go_trace *trace) { sampler_update(pick(s, which), nSteps, s->p.maxThreads ? s->p.maxThreads : 1, trace); }
void go_sync(go_session *s, char which) { if (which == 'A') sampler_sync(&s->A, &s->P); else sampler_sync(&s->P, &s->A); }

/* updateSampler, GapsRunner.cpp:201-222 */
uint64_t go_iterate(go_session *s, uint32_t nA, uint32_t nP)
{
    const char f = s->p.whichMatrixFixed;
    if (f != 'A') { go_update(s, 'A', nA, NULL); if (f != 'P') go_sync(s, 'P'); }
    if (f != 'P') { go_update(s, 'P', nP, NULL); if (f != 'A') go_sync(s, 'A'); }
    return (uint64_t)nA + nP;
}
/* GapsStatistics.h:130-185 */
void go_stats_update(go_session *s)
{
    const char f = s->p.whichMatrixFixed;
    ++s->statUpdates;
    for (uint32_t j = 0; j < s->K; ++j) {
        const float *pc = s->P.mat + (size_t)j * s->P.M, *ac = s->A.mat + (size_t)j * s->A.M;
        float norm = 0.f;
        for (uint32_t i = 0; i < s->P.M; ++i) norm = (pc[i] > norm) ? pc[i] : norm; /* gaps::max(Vector) */
        if (f == 'N') norm = (norm == 0.f) ? 1.f : norm; else norm = 1.f;
        if (f != 'P') for (uint32_t i = 0; i < s->P.M; ++i) { float q = pc[i] / norm; s->Pmean[(size_t)j * s->P.M + i] += q; s->Pstd[(size_t)j * s->P.M + i] += q * q; }
        if (f != 'A') for (uint32_t i = 0; i < s->A.M; ++i) { float q = ac[i] * norm; s->Amean[(size_t)j * s->A.M + i] += q; s->Astd[(size_t)j * s->A.M + i] += q * q; }
    }
}
float go_chisq(const go_session *s, char which) { return sampler_chisq(pickc(s, which)); }
void go_get_matrix(const go_session *s, char which, float *out)
{
    const go_sampler *sm = pickc(s, which);
    for (uint32_t r = 0; r < sm->M; ++r) for (uint32_t k = 0; k < sm->K; ++k) out[(size_t)r * sm->K + k] = sm->mat[(size_t)k * sm->M + r];
}
/* test hooks: load both factor matrices (row-major [rows][K]) as setMatrix would and rebuild what depends on them;
 * evaluate the alpha parameters of one (mode 0), one with change (mode 1) or two (mode 2) matrix entries */
void go_debug_set_matrices(go_session *s, const float *A, const float *P)
{
    for (int w = 0; w < 2; ++w) {
        go_sampler *sm = w ? &s->P : &s->A; const float *src = w ? P : A;
        for (uint32_t r = 0; r < sm->M; ++r) for (uint32_t k = 0; k < sm->K; ++k) {
            const float v = src[(size_t)r * sm->K + k];
            if (sm->sparse) { sm->rows[(size_t)r * sm->K + k] = v; hv_add(sm, r, k, -1.f * sm->mat[(size_t)k * sm->M + r]); hv_add(sm, r, k, v); }
            else sm->mat[(size_t)k * sm->M + r] = v;
        }
    }
    sampler_sync(&s->A, &s->P); sampler_sync(&s->P, &s->A);
    sampler_extra_init(&s->A); sampler_extra_init(&s->P);
}
/* bench hook: put a freshly created session into a given chain state -- the atoms of both domains (position, mass; inserted in the
 * given order, which becomes the order of the unsorted vector) and the factor matrices (row-major [rows][K]); the A*P caches / lookup
 * tables are rebuilt from the matrices.  Lets bench.py time the CPU port on the SAME iterations of the chain the GPU is timed on
 * (the generators' states are the session's own: the same distribution of work, not the same draws). */
int go_import_state(go_session *s, const uint64_t *posA, const float *massA, uint32_t nA, const float *A,
                    const uint64_t *posP, const float *massP, uint32_t nP, const float *P)
{
    if (s->A.dom.n != 0 || s->P.dom.n != 0) return 1;
    for (uint32_t i = 0; i < nA; ++i) dom_insert(&s->A.dom, posA[i], massA[i]);
    for (uint32_t i = 0; i < nP; ++i) dom_insert(&s->P.dom, posP[i], massP[i]);
    go_debug_set_matrices(s, A, P);
    return 0;
}
void go_debug_alpha(const go_session *s, char which, int mode, uint32_t r1, uint32_t c1, uint32_t r2, uint32_t c2, float ch, float *out2)
{
    const go_sampler *sm = pickc(s, which);
    if (mode == 0) alpha_one(sm, r1, c1, NULL, &out2[0], &out2[1]);
    else if (mode == 1) alpha_one(sm, r1, c1, &ch, &out2[0], &out2[1]);
    else alpha_two(sm, r1, c1, r2, c2, &out2[0], &out2[1]);
}
void go_get_rows(const go_session *s, char which, float *out) { const go_sampler *sm = pickc(s, which); if (sm->sparse) memcpy(out, sm->rows, (size_t)sm->M * sm->K * 4); else go_get_matrix(s, which, out); }
void go_get_ap(const go_session *s, char which, float *out) { const go_sampler *sm = pickc(s, which); memcpy(out, sm->AP, (size_t)sm->M * sm->N * 4); }
void go_get_atoms(const go_session *s, char which, uint64_t *pos, float *mass, uint32_t *left, uint32_t *right)
{
    const go_domain *d = &pickc(s, which)->dom;
    for (uint32_t i = 0; i < d->n; ++i) {
        const go_atom *a = &d->pool[d->vec[i]];
        if (pos) pos[i] = a->pos;
        if (mass) mass[i] = a->mass;
        if (left) left[i] = a->left != GO_NONE ? d->pool[a->left].index : GO_NONE;
        if (right) right[i] = a->right != GO_NONE ? d->pool[a->right].index : GO_NONE;
    }
}
void go_get_dims(const go_session *s, char which, uint32_t *M, uint32_t *N, uint32_t *K) { const go_sampler *sm = pickc(s, which); *M = sm->M; *N = sm->N; *K = sm->K; }
float go_lambda(const go_session *s, char which) { return pickc(s, which)->lambda; }
float go_max_gibbs_mass(const go_session *s, char which) { return pickc(s, which)->maxGibbsMass; }
float go_avg_queue(const go_session *s, char which) { return pickc(s, which)->avgQueue; }
void go_get_luts(const go_session *s, float *e, float *ei, float *qg)
{
    memcpy(e, s->rs->erf, sizeof(s->rs->erf)); memcpy(ei, s->rs->erfinv, sizeof(s->rs->erfinv)); memcpy(qg, s->rs->qgamma, sizeof(s->rs->qgamma));
}

static void hist_push(go_session *s, float cs, uint32_t nA, uint32_t nP)
{
    if (s->nHist == s->histCap) {
        s->histCap = s->histCap ? s->histCap * 2 : 64;
        s->chisqHist = (float *)realloc(s->chisqHist, s->histCap * 4);
        s->atomHistA = (uint32_t *)realloc(s->atomHistA, s->histCap * 4);
        s->atomHistP = (uint32_t *)realloc(s->atomHistP, s->histCap * 4);
    }
    s->chisqHist[s->nHist] = cs; s->atomHistA[s->nHist] = nA; s->atomHistP[s->nHist] = nP; s->nHist++;
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* runOnePhase, GapsRunner.cpp:272-327 (no checkpoints, snapshots or PUMP) */
/* pumpMatrixUniqueThreshold == pumpMatrixCutThreshold (GapsStatistics.h:65-111): per row the first column holding the
 * row maximum (strictly greater than everything before, starting from 0) gets +1.  src == NULL: the A sampler's
 * mMatrix through operator(); else a row-major [nGenes][K] matrix (meanPattern on Amean). */
static void pump_update(const go_session *s, const float *src, float *stat)
{
    const go_sampler *A = &s->A; const uint32_t K = s->K;
    for (uint32_t i = 0; i < s->nGenes; ++i) {
        float maxV = 0.f; uint32_t maxI = 0;
        for (uint32_t j = 0; j < K; ++j) {
            const float v = src ? src[(size_t)i * K + j] : (A->sparse ? A->rows[(size_t)i * K + j] : A->mat[(size_t)j * A->M + i]);
            if (maxV < v) { maxV = v; maxI = j; }
        }
        stat[(size_t)i * K + maxI] += 1.f;
    }
}
/* iterations [first, first + n) of a phase; stepsA / stepsP (may be NULL): the Poisson step counts drawn per iteration */
void go_run_iterations(go_session *s, int phase, uint32_t first, uint32_t n, uint32_t *stepsA, uint32_t *stepsP)
{
    const go_params *p = &s->p;
    for (unsigned iter = first; iter < first + n; ++iter) {
        if (phase == 1) {
            float temp = (float)(2 * iter) / (float)p->nIterations;
            go_set_annealing(s, fmin_ref(1.f, temp));
        }
        uint32_t nA, nP; go_draw_steps(s, &nA, &nP);
        if (stepsA) stepsA[iter - first] = nA;
        if (stepsP) stepsP[iter - first] = nP;
        s->totalUpdates += go_iterate(s, nA, nP);
        if (phase == 2) {
            go_stats_update(s);
            if (p->whichMatrixFixed == 'N' && p->takePumpSamples) pump_update(s, NULL, s->pump), ++s->pumpUpdates;   /* GapsRunner.cpp:308-313 */
        }
        if ((p->snapshotPhase == 0 || p->snapshotPhase == phase) && p->snapshotFrequency > 0 && ((iter + 1) % p->snapshotFrequency) == 0) {
            /* takeSnapshot, GapsStatistics.h:188-202: AModel.mMatrix.getMatrix() = operator() values */
            const int w = phase - 1;
            if (s->nSnap[w] == s->capSnap[w]) {
                s->capSnap[w] = s->capSnap[w] ? 2 * s->capSnap[w] : 4;
                s->snapA[w] = (float *)realloc(s->snapA[w], (size_t)s->capSnap[w] * s->nGenes * s->K * 4);
                s->snapP[w] = (float *)realloc(s->snapP[w], (size_t)s->capSnap[w] * s->nSamples * s->K * 4);
            }
            go_get_rows(s, 'A', s->snapA[w] + (size_t)s->nSnap[w] * s->nGenes * s->K);
            go_get_rows(s, 'P', s->snapP[w] + (size_t)s->nSnap[w] * s->nSamples * s->K);
            ++s->nSnap[w];
        }
        if (p->outputFrequency > 0 && ((iter + 1) % p->outputFrequency) == 0) { /* displayStatus :162-199 */
            float cs = (p->whichMatrixFixed == 'P') ? sampler_chisq(&s->A) : sampler_chisq(&s->P);
            hist_push(s, cs, s->A.dom.n, s->P.dom.n);
        }
    }
}
static void run_phase(go_session *s, int phase) { go_run_iterations(s, phase, 0, s->p.nIterations, NULL, NULL); }
/* GapsStatistics.cpp:63-86 (model = P sampler: mDMatrix is genes x samples) */
static float mean_chisq(const go_session *s)
{
    const go_sampler *P = &s->P;
    float chisq = 0.f;
    const float n2 = (float)s->statUpdates * (float)s->statUpdates;
    if (P->redW > 1) {
        /* lane mode (as sampler_chisq): per sample vector j a lane-strided partial over the genes, then
         * the vectors are added sequentially */
        for (uint32_t j = 0; j < P->M; ++j) {
            lane_acc a; const uint32_t W = P->redW, G = P->redG;
            for (uint32_t L = 0; L < W; ++L) { a.s[L] = 0.f; a.m[L] = 0.f; }
            for (uint32_t i = 0; i < P->N; ++i) {
                float m = 0.f;
                for (uint32_t k = 0; k < s->K; ++k) m += s->Amean[(size_t)k * s->nGenes + i] * s->Pmean[(size_t)k * s->nSamples + j];
                m /= n2;
                float d = P->D[(size_t)j * P->N + i], sd = P->S[(size_t)j * P->N + i];
                a.s[(i / G) % W] += ((d - m) * (d - m)) / (sd * sd);
            }
            float ps, pm; lanes_finish(&a, W, &ps, &pm);
            chisq += ps;
        }
        return chisq;
    }
    for (uint32_t i = 0; i < P->N; ++i)          /* genes */
        for (uint32_t j = 0; j < P->M; ++j) {    /* samples */
            float m = 0.f;
            for (uint32_t k = 0; k < s->K; ++k) m += s->Amean[(size_t)k * s->nGenes + i] * s->Pmean[(size_t)k * s->nSamples + j];
            m /= n2;
            float d = P->D[(size_t)j * P->N + i], sd = P->S[(size_t)j * P->N + i];
            chisq += ((d - m) * (d - m)) / (sd * sd);
        }
    return chisq;
}
void go_finish(go_session *s, go_result *out)
{
    memset(out, 0, sizeof(*out));
    out->nGenes = s->nGenes; out->nSamples = s->nSamples; out->nPatterns = s->K;
    size_t na = (size_t)s->nGenes * s->K, np = (size_t)s->nSamples * s->K;
    out->Amean = (float *)malloc(na * 4); out->Asd = (float *)malloc(na * 4);
    out->Pmean = (float *)malloc(np * 4); out->Psd = (float *)malloc(np * 4);
    const float n = (float)s->statUpdates;
    /* GapsStatistics.cpp:13-59 */
    for (uint32_t i = 0; i < s->nGenes; ++i) for (uint32_t k = 0; k < s->K; ++k) {
        float sum = s->Amean[(size_t)k * s->nGenes + i], sq = s->Astd[(size_t)k * s->nGenes + i];
        out->Amean[(size_t)i * s->K + k] = sum / n;
        float meanTerm = (sum * sum) / n; float numer = fmax_ref(0.f, sq - meanTerm);
        out->Asd[(size_t)i * s->K + k] = sqrtf(numer / (n - 1.f));
    }
    for (uint32_t i = 0; i < s->nSamples; ++i) for (uint32_t k = 0; k < s->K; ++k) {
        float sum = s->Pmean[(size_t)k * s->nSamples + i], sq = s->Pstd[(size_t)k * s->nSamples + i];
        out->Pmean[(size_t)i * s->K + k] = sum / n;
        float meanTerm = (sum * sum) / n; float numer = fmax_ref(0.f, sq - meanTerm);
        out->Psd[(size_t)i * s->K + k] = sqrtf(numer / (n - 1.f));
    }
    out->nHistory = s->nHist;
    out->chisqHistory = (float *)malloc((s->nHist + 1) * 4); out->atomHistoryA = (uint32_t *)malloc((s->nHist + 1) * 4); out->atomHistoryP = (uint32_t *)malloc((s->nHist + 1) * 4);
    memcpy(out->chisqHistory, s->chisqHist, s->nHist * 4); memcpy(out->atomHistoryA, s->atomHistA, s->nHist * 4); memcpy(out->atomHistoryP, s->atomHistP, s->nHist * 4);
    out->totalUpdates = s->totalUpdates;
    out->averageQueueLengthA = s->A.avgQueue; out->averageQueueLengthP = s->P.avgQueue;
    out->meanChiSq = (s->p.whichMatrixFixed != 'N') ? 0.f : mean_chisq(s); /* GapsRunner.cpp:478-484 */
    out->samplerSeconds = s->samplerSeconds;
    if (s->p.takePumpSamples) {   /* GapsRunner.cpp:487-492, GapsStatistics.cpp:113-131 */
        const float denom = s->pumpUpdates != 0 ? (float)s->pumpUpdates : 1.f;
        out->pumpMatrix = (float *)malloc(na * 4); out->meanPatternAssignment = (float *)calloc(na, 4);
        for (size_t t = 0; t < na; ++t) out->pumpMatrix[t] = s->pump[t] / denom;
        pump_update(s, out->Amean, out->meanPatternAssignment);
    }
    out->nEquilibrationSnapshots = s->nSnap[0]; out->nSamplingSnapshots = s->nSnap[1];
    float **dst[4] = {&out->equilibrationSnapshotsA, &out->equilibrationSnapshotsP, &out->samplingSnapshotsA, &out->samplingSnapshotsP};
    for (int w = 0; w < 2; ++w) {
        size_t sa = (size_t)s->nSnap[w] * na, sp = (size_t)s->nSnap[w] * np;
        *dst[2 * w] = (float *)malloc(sa * 4 + 4); *dst[2 * w + 1] = (float *)malloc(sp * 4 + 4);
        if (sa) memcpy(*dst[2 * w], s->snapA[w], sa * 4);
        if (sp) memcpy(*dst[2 * w + 1], s->snapP[w], sp * 4);
    }
}
int go_run(const float *data, uint32_t nrow, uint32_t ncol, const go_params *p, const float *unc, go_result *out)
{
    go_session *s = go_create(data, nrow, ncol, p, unc);
    double t0 = now_s();
    run_phase(s, 1);
    run_phase(s, 2);
    s->samplerSeconds = now_s() - t0;
    go_finish(s, out);
    go_destroy(s);
    return 0;
}
void go_result_free(go_result *r)
{
    free(r->Amean); free(r->Asd); free(r->Pmean); free(r->Psd); free(r->chisqHistory); free(r->atomHistoryA); free(r->atomHistoryP);
    free(r->pumpMatrix); free(r->meanPatternAssignment);
    free(r->equilibrationSnapshotsA); free(r->equilibrationSnapshotsP); free(r->samplingSnapshotsA); free(r->samplingSnapshotsP);
    memset(r, 0, sizeof(*r));
}
