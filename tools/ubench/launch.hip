// dev: cost of a dependent (gen, eval)-shaped kernel pair on one stream: plain launches vs a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
struct Big { uint64_t w[56]; uint32_t *p; };     // ~456-byte kernarg like SamplerDev
__global__ void one(Big b) { if (threadIdx.x == 0 && b.w[0] == 12345) b.p[0] = 1; }
__global__ void many(Big b) { if (threadIdx.x == 0 && b.w[1] == 12345) b.p[blockIdx.x] = 1; }
__global__ void one_s(uint32_t *p, uint64_t w) { if (threadIdx.x == 0 && w == 12345) p[0] = 1; }
__global__ void many_s(uint32_t *p, uint64_t w) { if (threadIdx.x == 0 && w == 12345) p[blockIdx.x] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    uint32_t *p; (void)hipMalloc(&p, 4096 * 4);
    Big b{}; b.p = p;
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipStreamSynchronize(s);
            double t0 = now();
            for (int i = 0; i < N; ++i) {
                if (variant == 0) { hipLaunchKernelGGL(one, dim3(1), dim3(256), 0, s, b); hipLaunchKernelGGL(many, dim3(157), dim3(64), 0, s, b); }
                else if (variant == 1) { hipLaunchKernelGGL(one_s, dim3(1), dim3(256), 0, s, p, 0ull); hipLaunchKernelGGL(many_s, dim3(157), dim3(64), 0, s, p, 0ull); }
                else { hipLaunchKernelGGL(one, dim3(1), dim3(256), 0, s, b); hipLaunchKernelGGL(many, dim3(50), dim3(1024), 0, s, b); }
            }
            double t1 = now();
            (void)hipStreamSynchronize(s);
            double t2 = now();
            if (rep) printf("variant %d (%s): %.2f us per pair on the GPU, host enqueue %.2f us per pair\n", variant, variant == 0 ? "456-B kernarg, 1x256 + 157x64" : variant == 1 ? "16-B kernarg" : "456-B kernarg, 1x256 + 50x1024", (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
    }
    // graph of 64 pairs, replayed
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 64; ++i) { hipLaunchKernelGGL(one, dim3(1), dim3(256), 0, s, b); hipLaunchKernelGGL(many, dim3(157), dim3(64), 0, s, b); }
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < N / 64; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        double t2 = now();
        if (rep) printf("graph of 64 pairs: %.2f us per pair\n", (t2 - t0) / (N / 64 * 64) * 1e6);
    }
    return 0;
}
