// gaps_math.h -- per-proposal random numbers and the scalar math of the Gibbs step, as the kernels
// use them.  Behavioural contract: reference src/math/Random.cpp (cited per function) -- PCG-XSH-RR
// state carried per proposal, rejection-sampled integer ranges, LUT-based truncated normal / gamma.
//
// log/exp: the reference calls libm logf/expf, whose last-bit results differ between libms (and
// between glibc ifunc variants).  Both the kernels and the oracle's "portable" mode evaluate the
// algorithm specified in DESIGN.md section "portable log/exp" -- IEEE double ops only, no
// contraction -- so CPU and GPU agree bit for bit.  Compile with -ffp-contract=off.
#pragma once
#include "platform.h"

#define GAPS_EPSILON 1.0e-5f
#define GAPS_SQRT2F 1.4142135623730950488016887242097f
#define GAPS_ERF_N 3001
#define GAPS_ERFINV_N 5001
#define GAPS_QGAMMA_N 5001

struct GapsLuts {            // device pointers (Random.cpp:269-295 tables)
    const float *erf, *erfinv, *qgamma;
};

CG_HD uint32_t gm_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
CG_HD double gm_u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
CG_HD uint64_t gm_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
CG_HD float gm_neg_inf() { union { float f; uint32_t u; } c; c.u = 0xff800000u; return c.f; }

// ---- portable log / exp (spec: DESIGN.md) ------------------------------------------------------
CG_HD float gm_logf(float x)
{
    uint32_t ux = gm_f2u(x);
    if (ux == 0u) return gm_neg_inf();
    if (ux == 0x3f800000u) return 0.0f;
    int e = (int)((ux >> 23) & 0xffu);
    uint32_t man = ux & 0x7fffffu;
    double m;
    if (e == 0) {
        double d = (double)x * 18446744073709551616.0;
        uint64_t ud = gm_d2u(d);
        e = (int)((ud >> 52) & 0x7ff) - 1023 - 64;
        m = gm_u2d((ud & 0xfffffffffffffull) | 0x3ff0000000000000ull);
    } else {
        e -= 127;
        m = gm_u2d(((uint64_t)man << 29) | 0x3ff0000000000000ull);
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

CG_HD double gm_floor_pos_neg(double v)   // floor for |v| < 2^31 without libm
{
    long long i = (long long)v;
    double fi = (double)i;
    return (fi > v) ? fi - 1.0 : fi;
}

CG_HD float gm_expf(float x)
{
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 89.0) { union { float f; uint32_t u; } c; c.u = 0x7f800000u; return c.f; }
    if (xd < -104.0) return 0.0f;
    double kf = gm_floor_pos_neg(xd * 1.4426950408889634074 + 0.5);
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
    double scale = gm_u2d((uint64_t)(k + 1023) << 52);
    return (float)(p * scale);
}

// ---- glibc's logf / expf restated: the "reference libm" math mode -----------------------------
// The reference calls libm logf / expf (accept tests AsynchronousGibbsSampler.h:165,189; Random.cpp:172-175, :194-200), so
// its chain is a function of the C library it runs on.  This is GNU libc 2.35's algorithm (sysdeps/ieee754/flt-32/e_logf.c,
// e_expf.c: table + degree-3 polynomial, every intermediate an IEEE double) with the multiply-add pairs fused exactly where
// the library's -mfma build (__logf_fma / __expf_fma, what x86-64 glibc selects on FMA-capable hosts) fuses them; fused =
// false gives the generic build.  Used by the verification mode (cogaps_params.mathMode) so that the GPU chain can be
// compared with numbers the reference binary itself produced.  (The test suite keeps its own restatement of the algorithm
// and checks it against a glibc 2.35 host's libm over every float.)
#define GM_MATH_PORTABLE 0u
#define GM_MATH_GLIBC_FMA 1u
#define GM_MATH_GLIBC_SSE2 2u
CG_HD float gm_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
CG_HD void gm_logf_tab(uint32_t i, double &invc, double &logc)
{
    // __logf_data.tab; a switch so that host, emulator and device builds share one definition without a global table
    switch (i & 15u) {
    case 0: invc = 0x1.661ec79f8f3bep+0; logc = -0x1.57bf7808caadep-2; break;
    case 1: invc = 0x1.571ed4aaf883dp+0; logc = -0x1.2bef0a7c06ddbp-2; break;
    case 2: invc = 0x1.49539f0f010bp+0; logc = -0x1.01eae7f513a67p-2; break;
    case 3: invc = 0x1.3c995b0b80385p+0; logc = -0x1.b31d8a68224e9p-3; break;
    case 4: invc = 0x1.30d190c8864a5p+0; logc = -0x1.6574f0ac07758p-3; break;
    case 5: invc = 0x1.25e227b0b8eap+0; logc = -0x1.1aa2bc79c81p-3; break;
    case 6: invc = 0x1.1bb4a4a1a343fp+0; logc = -0x1.a4e76ce8c0e5ep-4; break;
    case 7: invc = 0x1.12358f08ae5bap+0; logc = -0x1.1973c5a611cccp-4; break;
    case 8: invc = 0x1.0953f419900a7p+0; logc = -0x1.252f438e10c1ep-5; break;
    case 9: invc = 0x1p+0; logc = 0x0p+0; break;
    case 10: invc = 0x1.e608cfd9a47acp-1; logc = 0x1.aa5aa5df25984p-5; break;
    case 11: invc = 0x1.ca4b31f026aap-1; logc = 0x1.c5e53aa362eb4p-4; break;
    case 12: invc = 0x1.b2036576afce6p-1; logc = 0x1.526e57720db08p-3; break;
    case 13: invc = 0x1.9c2d163a1aa2dp-1; logc = 0x1.bc2860d22477p-3; break;
    case 14: invc = 0x1.886e6037841edp-1; logc = 0x1.1058bc8a07ee1p-2; break;
    default: invc = 0x1.767dcf5534862p-1; logc = 0x1.4043057b6ee09p-2; break;
    }
}
CG_HD uint64_t gm_expf_tab(uint32_t i)      // __exp2f_data.tab
{
    switch (i & 31u) {
    case 0: return 0x3ff0000000000000ull; case 1: return 0x3fefd9b0d3158574ull; case 2: return 0x3fefb5586cf9890full; case 3: return 0x3fef9301d0125b51ull;
    case 4: return 0x3fef72b83c7d517bull; case 5: return 0x3fef54873168b9aaull; case 6: return 0x3fef387a6e756238ull; case 7: return 0x3fef1e9df51fdee1ull;
    case 8: return 0x3fef06fe0a31b715ull; case 9: return 0x3feef1a7373aa9cbull; case 10: return 0x3feedea64c123422ull; case 11: return 0x3feece086061892dull;
    case 12: return 0x3feebfdad5362a27ull; case 13: return 0x3feeb42b569d4f82ull; case 14: return 0x3feeab07dd485429ull; case 15: return 0x3feea47eb03a5585ull;
    case 16: return 0x3feea09e667f3bcdull; case 17: return 0x3fee9f75e8ec5f74ull; case 18: return 0x3feea11473eb0187ull; case 19: return 0x3feea589994cce13ull;
    case 20: return 0x3feeace5422aa0dbull; case 21: return 0x3feeb737b0cdc5e5ull; case 22: return 0x3feec49182a3f090ull; case 23: return 0x3feed503b23e255dull;
    case 24: return 0x3feee89f995ad3adull; case 25: return 0x3feeff76f2fb5e47ull; case 26: return 0x3fef199bdd85529cull; case 27: return 0x3fef3720dcef9069ull;
    case 28: return 0x3fef5818dcfba487ull; case 29: return 0x3fef7c97337b9b5full; case 30: return 0x3fefa4afa2a490daull; default: return 0x3fefd0765b6e4540ull;
    }
}
CG_HD float gm_logf_glibc(float x, bool fused)
{
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = gm_f2u(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return gm_neg_inf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return gm_u2f(0x7fc00000u);
        ix = gm_f2u(x * 8388608.0f); ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    double invc, logc; gm_logf_tab(tmp >> 19, invc, logc);
    const double z = (double)gm_u2f(iz);
    if (fused) {
        const double r = __builtin_fma(z, invc, -1.0);
        const double y0 = __builtin_fma((double)k, Ln2, logc);
        double y = __builtin_fma(A1, r, A2);
        const double r2 = r * r;
        y = __builtin_fma(A0, r2, y);
        return (float)__builtin_fma(y, r2, y0 + r);
    }
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
CG_HD float gm_expf_glibc(float x, bool fused)
{
    const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const uint32_t ux = gm_f2u(x), abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {
        if (ux == 0xff800000u) return 0.f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return gm_u2f(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.f;
        if (x < -0x1.9d1d9ep6f) return gm_u2f(1u);        // __math_may_uflowf: 0x1.4p-75f squared, the smallest subnormal
    }
    const double xd = (double)x;
    double kd, r;
    if (fused) kd = __builtin_fma(InvLn2N, xd, Shift); else kd = InvLn2N * xd + Shift;
    const uint64_t ki = gm_d2u(kd);
    kd = kd - Shift;
    if (fused) r = __builtin_fma(InvLn2N, xd, -kd); else r = InvLn2N * xd - kd;
    const double s = gm_u2d(gm_expf_tab((uint32_t)ki) + (ki << 47));
    const double r2 = r * r;
    double zz, y;
    if (fused) { zz = __builtin_fma(C0, r, C1); y = __builtin_fma(C2, r, 1.0); y = __builtin_fma(zz, r2, y); }
    else { zz = C0 * r + C1; y = C2 * r + 1.0; y = zz * r2 + y; }
    return (float)(y * s);
}
// the math mode of a session (cogaps_params.mathMode): product kernels pass the constant GM_MATH_PORTABLE
CG_HD float gm_logf_m(float x, uint32_t mode) { return mode == GM_MATH_PORTABLE ? gm_logf(x) : gm_logf_glibc(x, mode == GM_MATH_GLIBC_FMA); }
CG_HD float gm_expf_m(float x, uint32_t mode) { return mode == GM_MATH_PORTABLE ? gm_expf(x) : gm_expf_glibc(x, mode == GM_MATH_GLIBC_FMA); }

// ---- PCG-XSH-RR (Random.cpp:40-56) -------------------------------------------------------------
#define GAPS_PCG_MULT 6364136223846793005ull
#define GAPS_PCG_INC 55ull

CG_HD void pcg_advance(uint64_t &s) { s = s * GAPS_PCG_MULT + GAPS_PCG_INC; }
CG_HD uint32_t pcg_output(uint64_t s)
{
    uint32_t xorshifted = (uint32_t)(((s >> 18u) ^ s) >> 27u);
    uint32_t rot = (uint32_t)(s >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
CG_HD uint32_t pcg_u32(uint64_t &s) { pcg_advance(s); return pcg_output(s); }
// GapsRng ctor (Random.cpp:32-38): state = seed, then one advance
CG_HD uint64_t pcg_from_seed(uint64_t seed) { uint64_t s = seed; pcg_advance(s); return s; }
// Random.cpp:63-66: float(u32) / float(UINT32_MAX)  (the divisor rounds to 2^32)
CG_HD float pcg_uniform(uint64_t &s) { return (float)pcg_u32(s) / 4294967296.0f; }
// Random.cpp:68-71
CG_HD float pcg_uniform_ab(uint64_t &s, float a, float b) { return pcg_uniform(s) * (b - a) + a; }
// Random.cpp:79-96
CG_HD uint32_t pcg_uniform32(uint64_t &s, uint32_t a, uint32_t b)
{
    if (b == a) return a;
    uint32_t range = b + 1u - a;
    uint32_t x = pcg_u32(s);
    uint32_t iPart = 0xFFFFFFFFu / range;
    while (x >= range * iPart) x = pcg_u32(s);
    return x / iPart + a;
}
// Random.cpp:98-103 (high word first)
CG_HD uint64_t pcg_u64(uint64_t &s)
{
    uint64_t high = ((uint64_t)pcg_u32(s) << 32) & 0xFFFFFFFF00000000ull;
    uint64_t low = pcg_u32(s);
    return high | low;
}
// a / b for b >= 1, exact.  The compiler's 64-bit division is a ~130-instruction routine on gfx950; this is two
// double-precision quotient estimates (each leaves an error far below the next one's range) and a +-1 fix-up.
CG_HD uint64_t gm_udiv64(uint64_t a, uint64_t b)
{
    if (b >> 62) { uint64_t q = 0; while (a >= b) { a -= b; ++q; } return q; }      // quotient <= 3
    const double inv = 1.0 / (double)b;
    double qd = (double)a * inv;
    qd = qd < 18446744073709549568.0 ? qd : 18446744073709549568.0;                  // largest double below 2^64
    uint64_t q = (uint64_t)qd;
    int64_t r = (int64_t)(a - q * b);                                                // |r| < 2^13 + b
    const int64_t q2 = (int64_t)((double)r * inv);
    q += (uint64_t)q2; r -= q2 * (int64_t)b;
    for (int k = 0; k < 2; ++k) { const uint64_t neg = r < 0; q -= neg; r += neg ? (int64_t)b : 0; }
    for (int k = 0; k < 2; ++k) { const uint64_t big = r >= (int64_t)b; q += big; r -= big ? (int64_t)b : 0; }
    return q;
}
// Random.cpp:105-123
CG_HD uint64_t pcg_uniform64(uint64_t &s, uint64_t a, uint64_t b)
{
    if (b == a) return a;
    uint64_t range = b + 1ull - a;
    uint64_t x = pcg_u64(s);
    uint64_t iPart = gm_udiv64(0xFFFFFFFFFFFFFFFFull, range);
    while (x >= range * iPart) x = pcg_u64(s);
    return gm_udiv64(x, iPart) + a;
}
// LCG jump: state after k advances = mulK * s + incK  (k-step affine map)
CG_HD void pcg_jump_coeffs(uint64_t k, uint64_t &mulK, uint64_t &incK)
{
    uint64_t accM = 1, accI = 0, curM = GAPS_PCG_MULT, curI = GAPS_PCG_INC;
    while (k) {
        if (k & 1) { accM *= curM; accI = accI * curM + curI; }
        curI = (curM + 1) * curI; curM *= curM; k >>= 1;
    }
    mulK = accM; incK = accI;
}

CG_HD float gm_min(float a, float b) { return a < b ? a : b; }   // Math.cpp:13-16
CG_HD float gm_max(float a, float b) { return a < b ? b : a; }   // Math.cpp:28-31

// Random.cpp:172-175
CG_HD float pcg_exponential(uint64_t &s, float lambda, uint32_t mathMode = GM_MATH_PORTABLE) { return -1.f * gm_logf_m(pcg_uniform(s), mathMode) / lambda; }

// Random.cpp:307-326
// (selects instead of branches: the two table reads of a truncated normal's bounds then travel together)
CG_HD float gm_p_norm_fast(const GapsLuts &L, float p, float mean, float sd)
{
    const float term = (p - mean) / (sd * GAPS_SQRT2F);
    const bool neg = term < 0.f;
    const float mag = neg ? -gm_max(term, -3.f) : gm_min(term, 3.f);
    const unsigned ndx = (unsigned)(mag * 1000.f);
#if defined(GM_FAKE_LUT)
    // dev probe (timing only, wrong bits): closed form instead of the table read
    const float xx = (float)ndx * 0.001f, x2 = xx * xx;
    const float e = cg_sqrtf(1.f - gm_expf(-x2 * (1.2732395f + 0.147f * x2) / (1.f + 0.147f * x2)));
#else
    const float e = L.erf[ndx];
#endif
    return 0.5f * (1.f + (neg ? -e : e));
}
// Random.cpp:328-345
CG_HD float gm_q_norm_fast(const GapsLuts &L, float q, float mean, float sd)
{
    const float term = 2.f * q - 1.f;
    const bool neg = term < 0.f;
    const unsigned ndx = (unsigned)((neg ? -term : term) * (float)(GAPS_ERFINV_N - 1));
#if defined(GM_FAKE_LUT)
    const float xq = gm_min((float)ndx * 0.0002f, 0.9998f);
    const float lg = gm_logf(1.f - xq * xq), aa = 4.3307467f + 0.5f * lg;
    const float e = cg_sqrtf(cg_sqrtf(aa * aa - lg / 0.147f) - aa);
#else
    const float e = L.erfinv[ndx];
#endif
    return mean + sd * GAPS_SQRT2F * (neg ? -e : e);
}

struct OptF { float v; bool has; };

// Random.cpp:178-191
CG_HD OptF pcg_trunc_normal(uint64_t &s, const GapsLuts &L, float a, float b, float mean, float sd)
{
    OptF o; o.v = 0.f; o.has = false;
    float pLower = gm_p_norm_fast(L, a, mean, sd);
    float pUpper = gm_p_norm_fast(L, b, mean, sd);
    if (!(pLower > 0.95f || pUpper < 0.05f)) {
        float z = gm_q_norm_fast(L, pcg_uniform_ab(s, pLower, pUpper), mean, sd);
        z = gm_max(a, gm_min(z, b));
        o.v = z; o.has = true;
    }
    return o;
}
// Random.cpp:194-200 (shape 2)
CG_HD float pcg_trunc_gamma_upper(uint64_t &s, const GapsLuts &L, float b, float scale, uint32_t mathMode = GM_MATH_PORTABLE)
{
    float upper = 1.f - gm_expf_m(-b / scale, mathMode) * (1.f + b / scale);
    const unsigned ndx = (unsigned)pcg_uniform_ab(s, 0.f, upper * 5000.f);
    return L.qgamma[ndx] * scale;
}
// AlphaParameters.cpp:27-36 (useLambda=false) and :38-48 (true)
CG_HD OptF gm_gibbs_mass(float s_, float s_mu, float a, float b, uint64_t &rng, const GapsLuts &L, bool useLambda, float lambda)
{
    OptF o; o.v = 0.f; o.has = false;
    if (s_ > GAPS_EPSILON) {
        float mean = useLambda ? (s_mu - lambda) / s_ : s_mu / s_;
        float sd = 1.f / cg_sqrtf(s_);
        return pcg_trunc_normal(rng, L, a, b, mean, sd);
    }
    return o;
}
// ProposalQueue.cpp:123-127
CG_HD float gm_death_prob(double nAtoms, double domainLength, double alpha, double numBins)
{
    double numer = nAtoms * domainLength;
    return (float)(numer / (numer + alpha * numBins * (domainLength - nAtoms)));
}
// static_cast<uint64_t>(double) as the reference's x86-64 build evaluates it (2^64 -> 0)
CG_HD uint64_t gm_u64_from_double_x86(double d)
{
    if (d >= 18446744073709551616.0) return 0ull;
    return (uint64_t)d;
}
