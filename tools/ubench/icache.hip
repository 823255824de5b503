// dev: cost of cold instruction fetch for a single-workgroup kernel on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define S1 p = (p * 1664525u + 1013904223u) ^ (p >> 7);
#define S4 S1 S1 S1 S1
#define S16 S4 S4 S4 S4
#define S64 S16 S16 S16 S16
#define S256 S64 S64 S64 S64
#define S1024 S256 S256 S256 S256
__global__ void big(uint64_t *out, uint32_t seed, int iters)
{
    uint32_t p = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        uint64_t c0, c1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "+v"(p) :: "memory");
        S1024 S1024
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "+v"(p) :: "memory");
        if (threadIdx.x == 0) out[it] = c1 - c0;
        __syncthreads();
    }
    if (p == 0xdeadbeef) out[15] = p;
}
__global__ void other(float *x) { x[threadIdx.x + blockIdx.x * blockDim.x] += 1.f; }
int main()
{
    uint64_t *out; float *x; (void)hipMalloc(&out, 128); (void)hipMalloc(&x, 4 * 256 * 4096);
    uint64_t h[4];
    for (int rep = 0; rep < 8; ++rep) {
        if (rep >= 4) hipLaunchKernelGGL(other, dim3(4096), dim3(256), 0, 0, x);   // a different kernel in between (like gen/eval alternation)
        hipLaunchKernelGGL(big, dim3(1), dim3(256), 0, 0, out, (uint32_t)rep, 3);
        (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        printf("launch %d%s: pass0 %llu  pass1 %llu  pass2 %llu cycles (2048 steps, 4 waves)\n", rep, rep >= 4 ? " (after another kernel)" : "", (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2]);
    }
    return 0;
}
