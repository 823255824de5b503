// dev: LDS access patterns of one 256-lane workgroup on gfx950 (cycles per step, 4 waves concurrently).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 64
struct alignas(16) V4 { uint32_t x, y, z, w; };
__global__ void k(uint64_t *out, int mode)
{
    __shared__ V4 tab[4096];      // 64 KB
    __shared__ uint32_t w32[4096];
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < 4096; i += 256) { tab[i] = V4{i * 2654435761u, i ^ 0x5555u, i + 7u, i * 3u}; w32[i] = i * 2654435761u; }
    __syncthreads();
    uint32_t p = t * 2654435761u + 12345u, acc = 0;
    uint64_t c0, c1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "+v"(p) :: "memory");
    for (int i = 0; i < N_IT; ++i) {
        if (mode == 0) { V4 a = tab[(p >> 7) & 4095u]; p = p * 1664525u + a.x + a.w; }                                   // 1 random b128, dependent
        else if (mode == 1) { uint32_t s = 0; for (int q = 0; q < 6; ++q) { V4 a = tab[((p >> 7) + q * 977u) & 4095u]; s += a.x ^ a.y ^ a.z ^ a.w; } p = p * 1664525u + s; }   // 6 independent random b128
        else if (mode == 2) { uint32_t s = 0; for (int q = 0; q < 6; ++q) s += w32[((p >> 7) + q * 977u) & 4095u]; p = p * 1664525u + s; }      // 6 independent random b32
        else if (mode == 3) { V4 a = tab[(t + i * 256u) & 4095u]; p = p * 1664525u + a.x + a.w; }                          // 1 linear b128
        else if (mode == 4) { uint32_t s = 0; for (int q = 0; q < 3; ++q) s += atomicCAS(&w32[((p >> 7) + q * 977u) & 4095u], 0xFFFFFFFFu, p); p = p * 1664525u + s; }   // 3 independent CAS
        else if (mode == 5) { for (int q = 0; q < 3; ++q) atomicMin(&w32[((p >> 7) + q * 977u) & 4095u], p); p = p * 1664525u + 1u; }          // 3 fire-and-forget min
        else if (mode == 6) { for (uint32_t j = t; j < 4096; j += 256) tab[j] = V4{p, p, p, p}; p = p * 1664525u + 1u; __syncthreads(); }     // clear 64 KB + barrier
        else if (mode == 7) { p = p * 1664525u + 1013904223u; }                                                             // loop overhead
        else if (mode == 8) { uint32_t s = 0; for (int q = 0; q < 6; ++q) { uint32_t key = p + q * 977u; V4 a = tab[(key >> 7) & 4095u];
                                  uint32_t m = (a.x == key) ? 0u : ((a.y == key) ? 1u : ((a.z == key) ? 2u : ((a.w == key) ? 3u : 4u))); s += m; } p = p * 1664525u + s; }   // 6 bucket reads + match logic
        else if (mode == 9) { uint32_t s = 0; for (int q = 0; q < 6; ++q) { uint32_t key = p + q * 977u; V4 a = tab[(key >> 7) & 4095u];
                                  const uint32_t e0 = a.x == key, e1 = a.y == key, e2 = a.z == key, e3 = a.w == key;
                                  const uint32_t m = (e1 + 2u * e2 + 3u * e3) | ((e0 | e1 | e2 | e3) ^ 1u) << 2; s += m; } p = p * 1664525u + s; }   // same, arithmetic
        acc += p;
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "+v"(acc) :: "memory");
    if (t == 0) out[0] = c1 - c0;
    if (acc == 0xdeadbeef) out[1] = acc;
}
int main()
{
    uint64_t *out; (void)hipMalloc(&out, 64);
    const char *names[] = {"1 random b128 (dependent)", "6 independent random b128", "6 independent random b32", "1 linear b128", "3 independent CAS (return)", "3 atomicMin (no return)",
                           "clear 64 KB + barrier", "loop overhead", "6 bucket reads + 4-way match", "6 bucket reads + arithmetic match"};
    for (int mode = 0; mode < 10; ++mode) {
        uint64_t best = ~0ull;
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, mode); uint64_t c; (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); if (c < best) best = c; }
        printf("%-36s %8.1f cycles/step\n", names[mode], (double)best / N_IT);
    }
    return 0;
}
