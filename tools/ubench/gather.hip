// dev: what do FETCH_SIZE / TCC_EA0_RDREQ count for GATHERS?  
// The MI355X guide calibrates FETCH_SIZE for wide coalesced streaming reads only (16 B per lane: the counter shows half the bytes).  The sparse
// evaluation's traffic is mostly short gathers -- 4-, 8- and 16-byte words and K = 50 float rows (208 bytes with the padding) at data-dependent addresses -- so
// its counter figure cannot be read without knowing what the counter does with those.  One kernel per access shape (the kernel's NAME carries
// the shape: rocprofv3 reports counters per kernel), each reads a KNOWN number of segments at pseudo-random addresses spread over a buffer far
// larger than L2 + Infinity Cache (4 GiB against 32 MiB + 256 MiB), every segment once:
//   gather_seg<B, false>: segment of B bytes (4, 8, 16: one load per lane; 208: a lane reads a K = 50 row of the product's row copy -- Kpad = 52
//                         floats -- as 13 x 16 B, the way the sparse evaluation's K-length dots do: sp_row_load) at a 256-byte-granular random
//                         offset (+ an in-line offset for the short ones)
//   gather_seg<208, true>: rows packed back to back at 208 bytes (rows straddle 64- and 128-byte lines as the product's [M][52] copy does)
//   stream16:              the guide's reference shape, 16 B per lane, consecutive lanes consecutive addresses
// Run under `rocprofv3 --pmc FETCH_SIZE`, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (separate passes): tools/pmc_gather.sh.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/gather.hip -o gather_DEV.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// every lane reads ONE segment of B bytes; segments = gridDim.x * blockDim.x
template <int B, bool PACKED>
__global__ void __launch_bounds__(256) gather_seg(const unsigned char *buf, uint64_t bytes, uint32_t *sink)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t h = mix(i * 0x9E3779B97F4A7C15ull + 12345u);
    uint64_t off;
    if (PACKED) off = (h % (bytes / B - 1)) * (uint64_t)B;                       // row r of a packed [rows][B bytes] array
    else off = (h % (bytes / 256 - 2)) * 256ull + (B <= 16 ? ((h >> 40) % (256 / B)) * B : 0);   // a segment inside its own 256-byte granule
    uint32_t acc = 0;
    const unsigned char *p = buf + off;
    if (B == 4) acc = *(const uint32_t *)p;
    else if (B == 8) { const uint2 v = *(const uint2 *)p; acc = v.x ^ v.y; }
    else if (B == 16) { const uint4 v = *(const uint4 *)p; acc = v.x ^ v.y ^ v.z ^ v.w; }
    else {
        // 208 bytes = Kpad floats of a K = 50 row: thirteen float4 loads issued together (sparse_kernels.h, sp_row_load)
#pragma unroll
        for (int k = 0; k < B / 16; ++k) { const uint4 v = *(const uint4 *)(p + 16 * k); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345677u) sink[0] = acc;      // (never true for the zero-filled buffer's pattern; keeps the loads)
}

__global__ void __launch_bounds__(256) stream16(const uint4 *buf, uint64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345677u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const uint64_t bytes = 4ull << 30;
    const uint32_t segs = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 22);      // segments per gather kernel (4 Mi)
    unsigned char *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes); (void)hipMemset(sink, 0, 64);
    (void)hipDeviceSynchronize();
    const dim3 grid(segs / 256), block(256);
    // (a 1 GiB streaming pass between the gathers evicts what the previous kernel left in the Infinity Cache)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timed = [&](const char *name, double algBytes, auto launch) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)(buf + (3ull << 30)), (1ull << 30) / 16, sink);      // evict
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s segments %9u  algorithmic bytes %12.0f  %8.3f ms  %8.1f GB/s\n", name, segs, algBytes, ms, algBytes / 1e6 / ms);
    };
    timed("gather_seg<4>", 4.0 * segs, [&] { hipLaunchKernelGGL((gather_seg<4, false>), grid, block, 0, 0, buf, bytes, sink); });
    timed("gather_seg<8>", 8.0 * segs, [&] { hipLaunchKernelGGL((gather_seg<8, false>), grid, block, 0, 0, buf, bytes, sink); });
    timed("gather_seg<16>", 16.0 * segs, [&] { hipLaunchKernelGGL((gather_seg<16, false>), grid, block, 0, 0, buf, bytes, sink); });
    timed("gather_seg<208,aligned>", 208.0 * segs, [&] { hipLaunchKernelGGL((gather_seg<208, false>), grid, block, 0, 0, buf, bytes, sink); });
    timed("gather_seg<208,packed>", 208.0 * segs, [&] { hipLaunchKernelGGL((gather_seg<208, true>), grid, block, 0, 0, buf, bytes, sink); });
    // the reference shape: 1 GiB streamed at 16 B per lane (kept as the LAST stream16 dispatch: tools/pmc_gather.py reads it by position)
    timed("stream16 (1 GiB)", (double)(1ull << 30), [&] { hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, (1ull << 30) / 16, sink); });
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}
