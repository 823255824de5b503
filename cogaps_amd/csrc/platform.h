// platform.h -- the thin layer between the sampler kernels and the machine they run on.
//
// Product build (hipcc, gfx950): everything maps straight onto HIP / CDNA4 intrinsics.
// Test build (-DCOGAPS_EMUL, g++, tests/emul only): the SAME kernel source runs on a cooperative
// fiber-per-lane workgroup emulator so that the device-side populate / flush logic can be unit
// tested against the oracle on a machine without a GPU.  The emulator is test infrastructure; the
// product library never contains it and the Python package never loads it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(COGAPS_EMUL)
#include "emul_runtime.h"   // tests/emul/
#else
#include <hip/hip_runtime.h>

#define CG_HD __host__ __device__ __forceinline__
#define CG_DEVICE __device__ __forceinline__
#define CG_KERNEL __global__
#define CG_SHARED __shared__
#define CG_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define CG_LAUNCH_BOUNDS2(n, wavesPerSimd) __launch_bounds__(n, wavesPerSimd)

CG_DEVICE unsigned cg_tid() { return threadIdx.x; }
CG_DEVICE unsigned cg_bid() { return blockIdx.x; }
CG_DEVICE unsigned cg_bdim() { return blockDim.x; }
CG_DEVICE unsigned cg_gdim() { return gridDim.x; }
CG_DEVICE void cg_sync() { __syncthreads(); }
// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight across it
// (__syncthreads waits for vmcnt(0) as well).  Only where the lanes exchange nothing through global memory.
CG_DEVICE void cg_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ordering point inside ONE wave (all its lanes call it): LDS traffic issued before it has completed before anything after it is
// issued -- what a lane needs to read another lane's LDS write when both belong to the same wave (no workgroup barrier involved)
CG_DEVICE void cg_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// Pull the whole kernel-argument segment into the scalar cache with one memory trip.  The scalar cache is
// cold at kernel entry and the compiler loads the fields of a by-value argument struct where they are first
// used: without this a kernel that walks through a 500-byte struct pays one full-latency miss per 64-byte line,
// one after the other.
template <int BYTES>
CG_DEVICE void cg_kernarg_warm()
{
    static_assert(BYTES <= 640, "extend the line list");
    const auto p = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8, d9;
    constexpr int L = BYTES - 4;      // last dword of the explicit arguments
#define CG_KA_OFF(i) ((i) * 64 < L ? (i) * 64 : L)
    asm volatile("s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
                 "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
                 "s_load_dword %8, %10, %19\n\ts_load_dword %9, %10, %20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9)
                 : "s"(p), "n"(CG_KA_OFF(0)), "n"(CG_KA_OFF(1)), "n"(CG_KA_OFF(2)), "n"(CG_KA_OFF(3)), "n"(CG_KA_OFF(4)), "n"(CG_KA_OFF(5)),
                   "n"(CG_KA_OFF(6)), "n"(CG_KA_OFF(7)), "n"(CG_KA_OFF(8)), "n"(CG_KA_OFF(9))
                 : "memory");
#undef CG_KA_OFF
}

// Constant address space: loads through such a pointer are invariant for the kernel's lifetime and, at a wave-uniform address,
// scalar (s_load through the scalar cache) -- what a by-value kernel argument gets.  For records the host writes before the launch.
#define CG_CONSTANT __attribute__((address_space(4)))
// one memory trip for all the 64-byte lines of such a record (cg_kernarg_warm for a record in memory)
template <int BYTES, class T>
CG_DEVICE void cg_const_warm(const T CG_CONSTANT *p)
{
    static_assert(BYTES <= 640, "extend the line list");
    const uint32_t CG_CONSTANT *w = (const uint32_t CG_CONSTANT *)p;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) acc |= w[(i * 64 < BYTES - 4 ? i * 64 : BYTES - 4) / 4];
    asm volatile("" :: "s"(acc));
}

// The same in two halves, for a kernel that has other work to do while the record's lines are on their way: _begin issues one load per
// 64-byte line, _end waits for them and returns the pointer THROUGH the wait: every load the compiler derives from the returned
// pointer is ordered behind it and finds the lines in the scalar cache.  (A by-value kernel argument cannot be fenced like this: the
// compiler is free to load its fields at the kernel's entry, before any statement of the source -- with the 540-byte SamplerDev it
// did, group after group, each waiting for its own cold miss because the scalar registers had to be spilled in between: seven
// serial misses in front of the generator's first instruction.)  The destination registers hold nothing anyone reads; they are
// outputs of the first statement and inputs of the second only so that the compiler keeps them allocated in between.
struct cg_const_lines { uint32_t d[10]; };
template <int BYTES, class T>
CG_DEVICE void cg_const_warm_begin(const T CG_CONSTANT *p, cg_const_lines &k)
{
    static_assert(BYTES <= 640, "extend the line list");
    constexpr int L = BYTES - 4;
#define CG_KA_OFF(i) ((i) * 64 < L ? (i) * 64 : L)
    asm volatile("s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
                 "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
                 "s_load_dword %8, %10, %19\n\ts_load_dword %9, %10, %20"
                 : "=&s"(k.d[0]), "=&s"(k.d[1]), "=&s"(k.d[2]), "=&s"(k.d[3]), "=&s"(k.d[4]), "=&s"(k.d[5]), "=&s"(k.d[6]), "=&s"(k.d[7]), "=&s"(k.d[8]), "=&s"(k.d[9])
                 : "s"(p), "n"(CG_KA_OFF(0)), "n"(CG_KA_OFF(1)), "n"(CG_KA_OFF(2)), "n"(CG_KA_OFF(3)), "n"(CG_KA_OFF(4)), "n"(CG_KA_OFF(5)),
                   "n"(CG_KA_OFF(6)), "n"(CG_KA_OFF(7)), "n"(CG_KA_OFF(8)), "n"(CG_KA_OFF(9))
                 : "memory");
#undef CG_KA_OFF
}
template <class T>
CG_DEVICE const T CG_CONSTANT *cg_const_warm_end(const T CG_CONSTANT *p, cg_const_lines &k)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(p) : "s"(k.d[0]), "s"(k.d[1]), "s"(k.d[2]), "s"(k.d[3]), "s"(k.d[4]), "s"(k.d[5]), "s"(k.d[6]), "s"(k.d[7]), "s"(k.d[8]), "s"(k.d[9]) : "memory");
    return p;
}

// four packed floats; a read of four that bypasses the caches' retention (non-temporal: rows used once per batch)
typedef float4 cg_f4;
typedef float cg_v4f __attribute__((ext_vector_type(4)));
CG_DEVICE cg_f4 cg_ld4_stream(const float *base, uint32_t j)
{
    const cg_v4f v = __builtin_nontemporal_load(reinterpret_cast<const cg_v4f *>(base) + j);
    cg_f4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
}
CG_HD float cg_sqrtf(float x) { return sqrtf(x); }        // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
#define CG_PLATFORM_NAME "HIP gfx950 (MI355X)"
// a value every lane of the wave holds alike, moved to a scalar register
CG_DEVICE uint32_t cg_uniform_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
// A wave-uniform value the compiler must treat as new from here on (it stays in a scalar register).  Tests on a kernel-wide invariant --
// "chunk c of the row exists", "wave w exists" -- are otherwise evaluated once at kernel entry, sixteen at a time, as 64-bit lane masks
// that stay live for the whole kernel: the sparse evaluation spilled 280 scalar registers into vector lanes before looking at its record.
CG_DEVICE uint32_t cg_fresh_u32(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
// nothing is scheduled across this point: what was issued before it stays before
CG_DEVICE void cg_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// keeps a loaded value (and so the load) alive without using it
CG_DEVICE void cg_keep_f32(float x) { asm volatile("" :: "v"(x)); }
CG_DEVICE void cg_keep_u32(uint32_t x) { asm volatile("" :: "v"(x)); }

// global-memory atomics (device scope)
CG_DEVICE uint32_t cg_atomic_add_u32(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
CG_DEVICE uint32_t cg_atomic_sub_u32(uint32_t *p, uint32_t v) { return atomicSub(p, v); }
CG_DEVICE uint32_t cg_atomic_min_u32(uint32_t *p, uint32_t v) { return atomicMin(p, v); }
CG_DEVICE uint32_t cg_atomic_max_u32(uint32_t *p, uint32_t v) { return atomicMax(p, v); }
CG_DEVICE uint32_t cg_atomic_or_u32(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
CG_DEVICE uint32_t cg_atomic_cas_u32(uint32_t *p, uint32_t cmp, uint32_t v) { return atomicCAS(p, cmp, v); }
CG_DEVICE unsigned long long cg_atomic_add_u64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
CG_DEVICE unsigned long long cg_atomic_max_u64(unsigned long long *p, unsigned long long v) { return atomicMax(p, v); }
CG_DEVICE unsigned long long cg_atomic_or_u64(unsigned long long *p, unsigned long long v) { return atomicOr(p, v); }
CG_DEVICE unsigned long long cg_atomic_and_u64(unsigned long long *p, unsigned long long v) { return atomicAnd(p, v); }

// wave64 cross-lane
CG_DEVICE unsigned long long cg_ballot(bool p) { return __ballot(p); }
CG_DEVICE int cg_popc64(unsigned long long x) { return __popcll(x); }
// loads of words that other waves of the workgroup update with L2 atomics: bypass the CU's L1
CG_DEVICE unsigned long long cg_load_l2_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A hand-off between workgroups INSIDE a launch (eval_kernel.h, split evaluation): one naturally aligned 8-byte {data, tag} granule,
// written through to the device-coherent level by one store (agent scope: `sc1`), read with cg_load_l2_u64 past the reader's
// non-coherent caches.  The tag says the data is this batch's, so no flag, counter, fence or ordering between two stores is needed
// (MI355X_MICROARCH.md, persistent-kernel price list, handoff-1to1: ~0.8 us on an idle chip).
CG_DEVICE void cg_store_agent_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CG_DEVICE void cg_poll_pause() { __builtin_amdgcn_s_sleep(1); }
CG_DEVICE unsigned long long cg_realtime() { return __builtin_amdgcn_s_memrealtime(); }      // chip-wide constant 100 MHz clock
// A poll inside a launch is bounded, generously: every turn is a load past the caches (>= 0.5 us while the word is not there) plus a
// sleep, so 2^22 turns are at least two seconds -- a waiting workgroup may share the GPU with a foreign kernel, a debugger or a preempted
// queue -- and a bound that is hit is an error of the whole update (GAPS_ERR_SPIN), never a hang.  (Reading the clock inside the loop
// cost the deciding evaluation kernel ten spilled vector registers and 2.8 us per launch: the turn count is the clock.)
CG_DEVICE bool cg_poll_expired(uint32_t spins) { return spins > (1u << 22); }
CG_DEVICE float cg_shfl_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }
CG_DEVICE float cg_shfl_f32(float v, int lane) { return __shfl(v, lane, 64); }
// the value lane `lane` holds (lane: the same in every lane of the wave) -- a lane read instead of an LDS-crossbar permute
CG_DEVICE float cg_lane_read_f32(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// Sum over the 64 lanes in the order of the ascending xor butterfly (x += x[lane ^ 1], ^ 2, ... ^ 32), the same
// bits in every lane.  After the level-1 and level-2 steps the four lanes of a quad hold one value, so the partner
// of the ^4 (^8) step may be any lane of the other quad (other half row): DPP row_half_mirror / row_mirror instead
// of two more LDS-crossbar permutes (~100 cycles each against a VALU op); the rows' sums are then combined from
// four lane reads as (r0 + r1) + (r2 + r3), which is what the ^16 and ^32 steps compute.  a + b == b + a bitwise.
CG_DEVICE float cg_wave_allsum_f32(float x)
{
#define CG_DPP_ADD(ctrl) x = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), (ctrl), 0xF, 0xF, false))
    CG_DPP_ADD(0xB1);      // quad_perm [1,0,3,2]
    CG_DPP_ADD(0x4E);      // quad_perm [2,3,0,1]
    CG_DPP_ADD(0x141);     // row_half_mirror
    CG_DPP_ADD(0x140);     // row_mirror
#undef CG_DPP_ADD
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}
// the sum of x over the wave's 64 lanes (inactive lanes of a divergent caller must not call: all lanes call it), in every lane: DPP
// row operations + four lane reads.  (An LDS atomicAdd at one address from every lane is turned by the compiler into a serial loop
// over the active lanes -- s_ff1 / v_readlane / s_add, 64 iterations, 2-3 k cycles per wave.)
CG_DEVICE uint32_t cg_wave_sum_u32(uint32_t x)
{
#define CG_DPP_ADDU(ctrl) x = x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, (ctrl), 0xF, 0xF, false)
    CG_DPP_ADDU(0xB1); CG_DPP_ADDU(0x4E); CG_DPP_ADDU(0x141); CG_DPP_ADDU(0x140);
#undef CG_DPP_ADDU
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 0) + (uint32_t)__builtin_amdgcn_readlane((int)x, 16) + (uint32_t)__builtin_amdgcn_readlane((int)x, 32) + (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
}
// exclusive prefix sum of x over the wave's lanes, and the wave's total (all 64 lanes call it with exec full; a lane that has nothing
// passes 0): Hillis-Steele inside the rows of 16 by DPP row shifts, then the rows' totals by the two row broadcasts.  What an LDS
// atomicAdd-with-return at one address from every lane computes, without the compiler's serial scan over the active lanes.
CG_DEVICE uint32_t cg_wave_excl_scan_u32(uint32_t x, uint32_t &total)
{
    uint32_t v = x;
#define CG_DPP_SCAN(ctrl, rows) v = v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, (ctrl), (rows), 0xF, false)
    CG_DPP_SCAN(0x111, 0xF); CG_DPP_SCAN(0x112, 0xF); CG_DPP_SCAN(0x114, 0xF); CG_DPP_SCAN(0x118, 0xF);      // row_shr:1,2,4,8
    CG_DPP_SCAN(0x142, 0xA);      // row_bcast:15 -> rows 1 and 3
    CG_DPP_SCAN(0x143, 0xC);      // row_bcast:31 -> rows 2 and 3
#undef CG_DPP_SCAN
    total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return v - x;
}
CG_DEVICE uint32_t cg_wave_bcast_u32(uint32_t x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, lane); }
CG_DEVICE unsigned long long cg_clock() { return __builtin_readcyclecounter(); }
CG_DEVICE int cg_clz64(unsigned long long x) { return __clzll((long long)x); }
CG_DEVICE int cg_ctz64(unsigned long long x) { return __ffsll((long long)x) - 1; }

#endif // COGAPS_EMUL

#define CG_NONE 0xFFFFFFFFu
