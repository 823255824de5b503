// stand-in for the three things GapsRunner.cpp takes from Boost.DateTime (ptime, time_duration, microsec_clock::local_time):
// clock_gettime.  Written for this repository, no Boost text.  The last difference of two ptimes is kept in
// refprobe_last_interval_us so that the driver can report the sampler's own wall time in microseconds (the reference keeps whole
// seconds: result.totalRunningTime, GapsRunner.cpp:473).
#pragma once
#include <time.h>
#include <stdint.h>
inline int64_t &refprobe_last_interval_us() { static int64_t v = 0; return v; }
namespace boost { namespace posix_time {
struct time_duration {
    int64_t us;
    explicit time_duration(int64_t u = 0) : us(u) {}
    long total_seconds() const { return (long)(us / 1000000); }
    long total_milliseconds() const { return (long)(us / 1000); }
    long total_microseconds() const { return (long)us; }
};
struct ptime { int64_t us; ptime() : us(0) {} explicit ptime(int64_t u) : us(u) {} };
inline time_duration operator-(const ptime &a, const ptime &b) { refprobe_last_interval_us() = a.us - b.us; return time_duration(a.us - b.us); }
struct microsec_clock {
    static ptime local_time() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ptime((int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000); }
};
} }
