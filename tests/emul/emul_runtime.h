// emul_runtime.h -- TEST-ONLY workgroup emulator (x86-64, single OS thread, one cooperative fiber
// per lane).  Lets the HIP kernel sources under cogaps_amd/csrc run on the CPU for the "not gpu"
// unit tests.  Never part of the product library.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <functional>
#include <string.h>

#define CG_HD inline
#define CG_DEVICE inline
#define CG_KERNEL
#define CG_SHARED static
#define CG_LAUNCH_BOUNDS(n)
#define CG_LAUNCH_BOUNDS2(n, w)

namespace cgemu {
struct LaneCtx { unsigned tid, bid, bdim, gdim; };
extern LaneCtx g_lane;
void block_barrier();                       // __syncthreads()
void wave_barrier();                        // all lanes of one wave (they run in lockstep on the device)
float wave_exchange_f32(float v, int src_lane_xor); // shfl_xor across the 64-lane wave
float wave_read_f32(float v, int src_lane);         // shfl: read lane src_lane of the wave
unsigned long long wave_ballot(bool p);
void launch(unsigned grid, unsigned block, const std::function<void()> &body);
}

inline unsigned cg_tid() { return cgemu::g_lane.tid; }
inline unsigned cg_bid() { return cgemu::g_lane.bid; }
inline unsigned cg_bdim() { return cgemu::g_lane.bdim; }
inline unsigned cg_gdim() { return cgemu::g_lane.gdim; }
inline void cg_sync() { cgemu::block_barrier(); }
inline void cg_sync_lds() { cg_sync(); }
inline void cg_wave_sync() { cgemu::wave_barrier(); }
template <int BYTES> inline void cg_kernarg_warm() {}
struct cg_const_lines { };
template <int BYTES, class T> inline void cg_const_warm_begin(const T *, cg_const_lines &) {}
template <class T> inline const T *cg_const_warm_end(const T *p, cg_const_lines &) { return p; }
#define CG_CONSTANT
template <int BYTES, class T> inline void cg_const_warm(const T *) {}
inline void cg_keep_f32(float) {}
inline void cg_keep_u32(uint32_t) {}
inline void cg_sched_fence() {}
inline uint32_t cg_uniform_u32(uint32_t x) { return x; }
inline uint32_t cg_fresh_u32(uint32_t x) { return x; }
struct cg_f4 { float x, y, z, w; };
inline cg_f4 cg_ld4_stream(const float *base, uint32_t j) { return reinterpret_cast<const cg_f4 *>(base)[j]; }
inline float cg_sqrtf(float x) { return __builtin_sqrtf(x); }
#define CG_PLATFORM_NAME "TEST-ONLY emulator"

// fibers only switch at barriers / wave exchanges, so plain read-modify-write is atomic here
inline uint32_t cg_atomic_add_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline uint32_t cg_atomic_sub_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o - v; return o; }
inline uint32_t cg_atomic_min_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
inline uint32_t cg_atomic_max_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t cg_atomic_or_u32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
inline uint32_t cg_atomic_cas_u32(uint32_t *p, uint32_t cmp, uint32_t v) { uint32_t o = *p; if (o == cmp) *p = v; return o; }
inline unsigned long long cg_atomic_add_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline unsigned long long cg_atomic_max_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline unsigned long long cg_atomic_or_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o | v; return o; }
inline unsigned long long cg_atomic_and_u64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o & v; return o; }

inline unsigned long long cg_ballot(bool p) { return cgemu::wave_ballot(p); }
inline int cg_popc64(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned long long cg_load_l2_u64(const unsigned long long *p) { return *p; }
inline void cg_store_agent_u64(unsigned long long *p, unsigned long long v) { *p = v; }
inline void cg_poll_pause() {}
inline unsigned long long cg_realtime() { return 0ull; }
inline bool cg_poll_expired(uint32_t spins) { return spins > (1u << 16); }      // (no clock here: the emulator's producers have always finished)
inline float cg_shfl_xor_f32(float v, int mask) { return cgemu::wave_exchange_f32(v, mask); }
inline float cg_wave_allsum_f32(float x) { for (int off = 1; off < 64; off <<= 1) x = x + cgemu::wave_exchange_f32(x, off); return x; }
inline float cg_shfl_f32(float v, int lane) { return cgemu::wave_read_f32(v, lane); }
inline float cg_lane_read_f32(float v, int lane) { return cgemu::wave_read_f32(v, lane); }
inline uint32_t cg_wave_sum_u32(uint32_t x)
{
    for (int off = 1; off < 64; off <<= 1) { float f; memcpy(&f, &x, 4); const float g = cgemu::wave_exchange_f32(f, off); uint32_t y; memcpy(&y, &g, 4); x += y; }
    return x;
}
inline uint32_t cg_wave_bcast_u32_from(uint32_t x, int srcLane) { float f; memcpy(&f, &x, 4); const float g = cgemu::wave_read_f32(f, srcLane < 0 ? 0 : srcLane); uint32_t y; memcpy(&y, &g, 4); return y; }
inline uint32_t cg_wave_bcast_u32(uint32_t x, int lane) { float f; memcpy(&f, &x, 4); const float g = cgemu::wave_read_f32(f, lane); uint32_t y; memcpy(&y, &g, 4); return y; }
inline uint32_t cg_wave_excl_scan_u32(uint32_t x, uint32_t &total)
{
    // inclusive Hillis-Steele by lane reads (a lane below the wave's first reads nothing)
    const unsigned lane = cgemu::g_lane.tid & 63u;
    uint32_t v = x;
    for (unsigned off = 1; off < 64u; off <<= 1) { const uint32_t o = cg_wave_bcast_u32_from(v, (int)lane - (int)off); v += lane >= off ? o : 0u; }
    total = cg_wave_bcast_u32(v, 63);
    return v - x;
}
inline unsigned long long cg_clock() { return 0; }
inline int cg_clz64(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline int cg_ctz64(unsigned long long x) { return x ? __builtin_ctzll(x) : -1; }
