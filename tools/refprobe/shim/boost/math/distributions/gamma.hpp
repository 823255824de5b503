// stand-in, see normal.hpp
#pragma once
#include "normal.hpp"
namespace boost { namespace math {
template <class T = double> struct gamma_distribution { T k, th; gamma_distribution(T shape, T scale = 1) : k(shape), th(scale) {} };
template <class T, class U> inline T cdf(const gamma_distribution<T> &d, U x) { return (T)refprobe_detail::gamma_cdf(d.k, (long double)x / d.th); }
template <class T, class U> inline T pdf(const gamma_distribution<T> &d, U x) { return (T)(refprobe_detail::gamma_pdf(d.k, (long double)x / d.th) / d.th); }
template <class T, class U> inline T quantile(const gamma_distribution<T> &d, U p)
{
    const long double q = (long double)p;
    if (q <= 0.0L) return (T)0;
    long double lo = 0.0L, hi = 1.0L;
    while (refprobe_detail::gamma_cdf(d.k, hi) < q && hi < 1e6L) hi *= 2.0L;
    for (int i = 0; i < 200; ++i) { const long double mid = 0.5L * (lo + hi); if (refprobe_detail::gamma_cdf(d.k, mid) < q) lo = mid; else hi = mid; }
    long double x = 0.5L * (lo + hi);
    for (int i = 0; i < 3; ++i) { const long double f = refprobe_detail::gamma_pdf(d.k, x); if (f > 0.0L) x -= (refprobe_detail::gamma_cdf(d.k, x) - q) / f; }
    return (T)(x * d.th);
}
} }
