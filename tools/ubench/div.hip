// dev: cycles of 64-bit division variants for one wave per SIMD (dependent chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t udiv64_d(uint64_t a, uint64_t b)
{
    if (b >> 63) return a >= b ? 1ull : 0ull;
    const double inv = __builtin_amdgcn_rcp((double)b);
    double qd = (double)a * inv;
    qd = qd < 18446744073709549568.0 ? qd : 18446744073709549568.0;
    uint64_t q = (uint64_t)qd;
    int64_t r = (int64_t)(a - q * b);
    const int64_t q2 = (int64_t)((double)r * inv);
    q += (uint64_t)q2; r -= q2 * (int64_t)b;
    while (r < 0) { --q; r += (int64_t)b; }
    while (r >= (int64_t)b) { ++q; r -= (int64_t)b; }
    return q;
}
#define N_IT 64
__global__ void k(uint64_t *out, uint64_t seed, int mode)
{
    uint64_t a = seed * (threadIdx.x + 12345u) * 0x9E3779B97F4A7C15ull, b = (seed ^ (threadIdx.x * 77u)) | 1ull, acc = 0;
    uint64_t c0, c1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "+v"(a) :: "memory");
    for (int i = 0; i < N_IT; ++i) {
        uint64_t q;
        if (mode == 0) q = a / b;
        else if (mode == 1) q = udiv64_d(a, b);
        else if (mode == 2) q = (uint32_t)a / (uint32_t)(b | 1u);
        else if (mode == 3) q = (uint64_t)((double)a / (double)b);
        else q = a * b + 1;
        acc += q; a = a * 6364136223846793005ull + q; b = (b >> 1) | 1ull;
        if ((i & 7) == 7) b = (seed ^ a) | 1ull;
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "+v"(acc) :: "memory");
    if (threadIdx.x == 0) out[0] = c1 - c0;
    out[1 + threadIdx.x] = acc;
}
int main()
{
    uint64_t *out; (void)hipMalloc(&out, 8 * 300);
    const char *names[] = {"native u64 /", "double-reciprocal exact u64 /", "native u32 /", "f64 / + cvt", "u64 mul (loop overhead)"};
    uint64_t h[257], ref[257];
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, 0x123456789abcdefull, mode);
        (void)hipMemcpy(h, out, 8 * 257, hipMemcpyDeviceToHost);
        if (mode == 0) for (int i = 0; i < 257; ++i) ref[i] = h[i];
        int same = 1; for (int i = 1; i < 257; ++i) same &= (h[i] == ref[i]);
        printf("%-34s %8.1f cycles/iter%s\n", names[mode], (double)h[0] / N_IT, mode == 1 ? (same ? "  (results == native)" : "  (MISMATCH)") : "");
    }
    return 0;
}
