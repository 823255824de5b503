// dev: the shader clock a lone workgroup runs at (s_memtime ticks per s_memrealtime tick x 100 MHz), with 1 or 240 workgroups resident,
// short and long kernels -- is the latency-bound chain running at the chip's full clock?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)x; }
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 4096 * 24); unsigned long long h[3];
    for (int rep = 0; rep < 3; ++rep)
    for (int grid : {1, 240, 2048}) for (int iters : {2000, 200000}) {
        for (int k = 0; k < 200; ++k) probe<<<grid, 512>>>(d, iters);      // back to back, like the chain's launches
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("grid %5d iters %7d: %8llu shader ticks in %7llu x 10 ns -> %.0f MHz\n", grid, iters, h[0], h[1], 100.0 * (double)h[0] / (double)h[1]);
    }
    return 0;
}
