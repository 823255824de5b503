// stand-in, see normal.hpp (the reference includes the header and never instantiates the distribution)
#pragma once
#include "normal.hpp"
namespace boost { namespace math {
template <class T = double> struct exponential_distribution { T l; exponential_distribution(T lambda = 1) : l(lambda) {} };
template <class T, class U> inline T cdf(const exponential_distribution<T> &d, U x) { return (T)(-expm1l(-(long double)d.l * (long double)x)); }
template <class T, class U> inline T pdf(const exponential_distribution<T> &d, U x) { return (T)((long double)d.l * expl(-(long double)d.l * (long double)x)); }
template <class T, class U> inline T quantile(const exponential_distribution<T> &d, U p) { return (T)(-log1pl(-(long double)p) / (long double)d.l); }
} }
