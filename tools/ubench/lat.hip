// dev: primitive latencies seen by ONE workgroup of 256 lanes on gfx950 (cycles of s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N_IT 256
__global__ void k(uint32_t *chain, unsigned long long *stamps, uint64_t *out, int mode)
{
    __shared__ uint32_t lds[4096];
    __shared__ uint32_t ctr;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < 4096; i += 256) lds[i] = (i * 97u + 13u) & 4095u;
    if (t == 0) ctr = 0;
    __syncthreads();
    uint32_t p = t;
    uint64_t c0 = clock64();
    if (mode == 0) { for (int i = 0; i < N_IT; ++i) p = lds[p]; }                               // dependent LDS loads
    else if (mode == 1) { for (int i = 0; i < N_IT; ++i) p = atomicMin(&lds[(p * 31u + i) & 4095u], p) + 1u; }   // dependent LDS atomics w/ return
    else if (mode == 2) { for (int i = 0; i < N_IT; ++i) p = chain[p]; }                         // dependent global loads (L2-resident)
    else if (mode == 3) { for (int i = 0; i < N_IT; ++i) p = (uint32_t)atomicMax(&stamps[(p * 31u + i) & 65535u], (unsigned long long)p) + 1u; }   // dependent global atomics w/ return
    else if (mode == 4) { for (int i = 0; i < N_IT; ++i) { __syncthreads(); p += i; } }          // barrier only
    else if (mode == 5) { for (int i = 0; i < N_IT; ++i) { p = chain[(p + i) & 65535u]; __syncthreads(); } }  // global load + barrier
    else if (mode == 6) { for (int i = 0; i < N_IT; ++i) { p += (uint32_t)clock64(); } }       // s_memtime
    else if (mode == 7) { for (int i = 0; i < N_IT; ++i) { atomicMax(&stamps[(p * 31u + i) & 65535u], (unsigned long long)p); p = p * 5u + 1u; } __syncthreads(); }  // fire-and-forget global atomics
    else if (mode == 8) { for (int i = 0; i < N_IT; ++i) { p = __builtin_nontemporal_load(&chain[(p + 7u) & 65535u]); } }
    else if (mode == 9) { for (int i = 0; i < N_IT; ++i) { p = __hip_atomic_load(&chain[(p + 7u) & 65535u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
    else if (mode == 10) { for (int i = 0; i < N_IT; ++i) { if (t == 0) p = atomicAdd(&ctr, 1u); __syncthreads(); p += ctr; __syncthreads(); } }  // lane0 LDS write + 2 barriers
    else if (mode == 11) { for (int i = 0; i < N_IT; ++i) { uint64_t x = __ballot(p & 1); p += (uint32_t)__popcll(x) + (uint32_t)__shfl_xor((int)p, 1 << (i & 5)); } }
    uint64_t c1 = clock64();
    if (t == 0) out[0] = c1 - c0;
    if (p == 0xdeadbeef) out[1] = p;
}
int main()
{
    uint32_t *chain; unsigned long long *st; uint64_t *out;
    std::vector<uint32_t> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)((i * 1000003ull + 12345ull) & 65535ull);
    hipMalloc(&chain, h.size() * 4); hipMemcpy(chain, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&st, 65536 * 8); hipMemset(st, 0, 65536 * 8);
    hipMalloc(&out, 16);
    const char *names[] = {"LDS load (dependent)", "LDS atomicMin w/ return (dependent)", "global load, L2-resident 256KB set (dependent)", "global atomicMax u64 w/ return (dependent)",
                           "__syncthreads (4 waves)", "global load + __syncthreads", "s_memtime", "global atomic no return (x256 then barrier)", "nontemporal load", "agent-scope relaxed atomic load",
                           "lane0 LDS atomic + 2 barriers", "ballot+popc+shfl"};
    for (int mode = 0; mode < 12; ++mode) {
        uint64_t best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, chain, st, out, mode);
            uint64_t c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
            if (c < best) best = c;
        }
        printf("%-50s %8.1f cycles/op\n", names[mode], (double)best / N_IT);
    }
    // clock rate: time a long kernel
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, chain, st, out, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); uint64_t c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("100 launches mode0: %.3f ms total, %.2f us each, kernel body %llu cycles\n", ms, ms * 10.0, (unsigned long long)c);
    return 0;
}
