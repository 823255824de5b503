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
CG_HD float pcg_exponential(uint64_t &s, float lambda) { return -1.f * gm_logf(pcg_uniform(s)) / lambda; }

// Random.cpp:307-326
// (selects instead of branches: the two table reads of a truncated normal's bounds then travel together)
CG_HD float gm_p_norm_fast(const GapsLuts &L, float p, float mean, float sd)
{
    const float term = (p - mean) / (sd * GAPS_SQRT2F);
    const bool neg = term < 0.f;
    const float mag = neg ? -gm_max(term, -3.f) : gm_min(term, 3.f);
    const unsigned ndx = (unsigned)(mag * 1000.f);
    const float e = L.erf[ndx];
    return 0.5f * (1.f + (neg ? -e : e));
}
// Random.cpp:328-345
CG_HD float gm_q_norm_fast(const GapsLuts &L, float q, float mean, float sd)
{
    const float term = 2.f * q - 1.f;
    const bool neg = term < 0.f;
    const unsigned ndx = (unsigned)((neg ? -term : term) * (float)(GAPS_ERFINV_N - 1));
    const float e = L.erfinv[ndx];
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
CG_HD float pcg_trunc_gamma_upper(uint64_t &s, const GapsLuts &L, float b, float scale)
{
    float upper = 1.f - gm_expf(-b / scale) * (1.f + b / scale);
    const unsigned ndx = (unsigned)pcg_uniform_ab(s, 0.f, upper * 5000.f);
    return L.qgamma[ndx] * scale;
}
// AlphaParameters.cpp:27-36 (useLambda=false) and :38-48 (true)
CG_HD OptF gm_gibbs_mass(float s_, float s_mu, float a, float b, uint64_t &rng, const GapsLuts &L, bool useLambda, float lambda)
{
    OptF o; o.v = 0.f; o.has = false;
    if (s_ > GAPS_EPSILON) {
        float mean = useLambda ? (s_mu - lambda) / s_ : s_mu / s_;
#if defined(COGAPS_EMUL)
        float sd = 1.f / __builtin_sqrtf(s_);
#else
        float sd = 1.f / sqrtf(s_);
#endif
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
