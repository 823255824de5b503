// dev: how fast can Q workgroups each pull 4 rows of N floats (random rows of big matrices) and reduce them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
template <int C>   // float4 chunks per thread
__global__ void __launch_bounds__(1024) rows(const float *D, const float *S2, const float *AP, const float *O, const uint32_t *rowIdx, const uint32_t *colIdx, uint32_t Npad, float *out, int split)
{
    const uint32_t q = blockIdx.x / split, part = blockIdx.x % split;
    const uint32_t nq = Npad >> 2, BS = blockDim.x, t = threadIdx.x;
    const float4 *d = (const float4 *)(D + (size_t)rowIdx[q] * Npad), *s = (const float4 *)(S2 + (size_t)rowIdx[q] * Npad), *a = (const float4 *)(AP + (size_t)rowIdx[q] * Npad), *o = (const float4 *)(O + (size_t)colIdx[q] * Npad);
    float4 vd[C], vs[C], va[C], vo[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { const uint32_t j = (part * C + c) * BS + t; if (j < nq) { vd[c] = d[j]; vs[c] = s[j]; va[c] = a[j]; vo[c] = o[j]; } else { vd[c] = vs[c] = va[c] = vo[c] = make_float4(0, 1, 0, 0); } }
    float acc = 0.f, acc2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        { float r = vo[c].x / vs[c].x; acc += vo[c].x * r; acc2 += r * (vd[c].x - va[c].x); }
        { float r = vo[c].y / vs[c].y; acc += vo[c].y * r; acc2 += r * (vd[c].y - va[c].y); }
        { float r = vo[c].z / vs[c].z; acc += vo[c].z * r; acc2 += r * (vd[c].z - va[c].z); }
        { float r = vo[c].w / vs[c].w; acc += vo[c].w * r; acc2 += r * (vd[c].w - va[c].w); }
    }
    for (int off = 1; off < 64; off <<= 1) { acc += __shfl_xor(acc, off, 64); acc2 += __shfl_xor(acc2, off, 64); }
    __shared__ float l[32];
    if (BS > 64) { if ((t & 63) == 0) { l[t >> 6] = acc; l[16 + (t >> 6)] = acc2; } __syncthreads(); if (t == 0) { float x = 0, y = 0; for (uint32_t w = 0; w < (BS >> 6); ++w) { x += l[w]; y += l[16 + w]; } acc = x; acc2 = y; } }
    if (t == 0) out[blockIdx.x] = acc + acc2;
}
template <int C> static float run(int Q, int BS, int split, const float *D, const float *S2, const float *AP, const float *O, const uint32_t *ri, const uint32_t *ci, uint32_t Npad, float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(rows<C>, dim3(Q * split), dim3(BS), 0, 0, D, S2, AP, O, ri + (rep * 20 + k) * 256, ci + (rep * 20 + k) * 256, Npad, out, split);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms / 20.f);
    }
    return best * 1e3f;
}
int main()
{
    for (int cfg = 0; cfg < 2; ++cfg) {
        const uint32_t M = cfg == 0 ? 20000 : 2000, N = cfg == 0 ? 2000 : 20000, K = 50, Q = cfg == 0 ? 157 : 50;
        float *D, *S2, *AP, *O, *out; uint32_t *ri, *ci;
        (void)hipMalloc(&D, (size_t)M * N * 4); (void)hipMalloc(&S2, (size_t)M * N * 4); (void)hipMalloc(&AP, (size_t)M * N * 4); (void)hipMalloc(&O, (size_t)K * N * 4); (void)hipMalloc(&out, 65536 * 4);
        (void)hipMemset(D, 0, (size_t)M * N * 4); (void)hipMemset(AP, 0, (size_t)M * N * 4); (void)hipMemset(O, 0, (size_t)K * N * 4);
        std::vector<float> ones((size_t)M * N, 1.f); (void)hipMemcpy(S2, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
        std::vector<uint32_t> hr(256 * 200), hc(256 * 200); uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < hr.size(); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hr[i] = (uint32_t)(x % M); hc[i] = (uint32_t)((x >> 32) % K); }
        (void)hipMalloc(&ri, hr.size() * 4); (void)hipMalloc(&ci, hc.size() * 4); (void)hipMemcpy(ri, hr.data(), hr.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ci, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
        printf("N = %u floats per row, %u workgroups (4 rows each, %.1f MB per launch); time per launch incl. ~2-3 us launch gap\n", N, Q, Q * 4.0 * N * 4 / 1e6);
        if (cfg == 0) {
            printf("  64 thr x 8 chunks : %6.2f us\n", run<8>(Q, 64, 1, D, S2, AP, O, ri, ci, N, out));
            printf(" 128 thr x 4 chunks : %6.2f us\n", run<4>(Q, 128, 1, D, S2, AP, O, ri, ci, N, out));
            printf(" 256 thr x 2 chunks : %6.2f us\n", run<2>(Q, 256, 1, D, S2, AP, O, ri, ci, N, out));
            printf(" 512 thr x 1 chunk  : %6.2f us\n", run<1>(Q, 512, 1, D, S2, AP, O, ri, ci, N, out));
            printf("  64 thr x 1 chunk, 8 workgroups per proposal : %6.2f us\n", run<1>(Q, 64, 8, D, S2, AP, O, ri, ci, N, out));
            printf(" 128 thr x 1 chunk, 4 workgroups per proposal : %6.2f us\n", run<1>(Q, 128, 4, D, S2, AP, O, ri, ci, N, out));
        } else {
            printf(" 1024 thr x 5 chunks: %6.2f us\n", run<5>(Q, 1024, 1, D, S2, AP, O, ri, ci, N, out));
            printf("  512 thr x 10 chunks: %6.2f us\n", run<10>(Q, 512, 1, D, S2, AP, O, ri, ci, N, out));
            printf("  256 thr x 20 chunks: %6.2f us\n", run<20>(Q, 256, 1, D, S2, AP, O, ri, ci, N, out));
            printf(" 1024 thr x 1 chunk, 5 workgroups per proposal : %6.2f us\n", run<1>(Q, 1024, 5, D, S2, AP, O, ri, ci, N, out));
            printf("  256 thr x 1 chunk, 20 workgroups per proposal: %6.2f us\n", run<1>(Q, 256, 20, D, S2, AP, O, ri, ci, N, out));
            printf("  256 thr x 4 chunks, 5 workgroups per proposal: %6.2f us\n", run<4>(Q, 256, 5, D, S2, AP, O, ri, ci, N, out));
            printf("   64 thr x 4 chunks, 20 workgroups per proposal: %6.2f us\n", run<4>(Q, 64, 20, D, S2, AP, O, ri, ci, N, out));
        }
        (void)hipFree(D); (void)hipFree(S2); (void)hipFree(AP); (void)hipFree(O); (void)hipFree(out); (void)hipFree(ri); (void)hipFree(ci);
    }
    // empty-kernel reference
    return 0;
}
