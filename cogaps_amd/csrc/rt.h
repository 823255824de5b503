// rt.h -- host-side runtime shim: device memory, copies, launches.  HIP in the product build; plain
// host memory + the fiber emulator in the test-only COGAPS_EMUL build (tests/emul).
#pragma once
#include "platform.h"
#if !defined(COGAPS_EMUL)
#include <hip/hip_ext.h>
#endif
#include <stdlib.h>
#include <string.h>
#include <string>
#include <stdexcept>

// device memory exhausted (hipErrorOutOfMemory): its own type, so that the C ABI can report it as a code (cogaps_last_error_code) and a
// caller can retry with fewer sessions in flight without reading message texts
struct rt_out_of_memory : std::runtime_error { explicit rt_out_of_memory(const std::string &m) : std::runtime_error(m) {} };

#if defined(COGAPS_EMUL)

typedef int rt_stream_t;
inline void *rt_malloc(size_t n) { void *p = calloc(n ? n : 1, 1); if (!p) throw rt_out_of_memory("out of memory"); return p; }
struct rt_alloc_scope { explicit rt_alloc_scope(rt_stream_t) {} };
inline void rt_free(void *p) { free(p); }
inline void *rt_malloc_host(size_t n) { return rt_malloc(n); }
inline void rt_free_host(void *p) { free(p); }
inline void rt_h2d(void *d, const void *h, size_t n, rt_stream_t) { memcpy(d, h, n); }
inline void rt_d2h(void *h, const void *d, size_t n, rt_stream_t) { memcpy(h, d, n); }
inline void rt_d2d(void *d, const void *s, size_t n, rt_stream_t) { memcpy(d, s, n); }
inline void rt_memset(void *d, int v, size_t n, rt_stream_t) { memset(d, v, n); }
inline void rt_sync(rt_stream_t) {}
inline void rt_set_device(int) {}
inline int rt_get_device() { return 0; }
inline void rt_mem_info(size_t *freeB, size_t *totalB) { *freeB = *totalB = (size_t)1 << 40; }
inline unsigned rt_compute_units() { const char *e = getenv("COGAPS_TEST_COMPUTE_UNITS"); return e ? (unsigned)atoi(e) : (1u << 20); }      // (test-only build: a small device on request)
inline rt_stream_t rt_stream_create() { return 0; }
inline void rt_stream_destroy(rt_stream_t) {}
#define RT_LAUNCH(kernel, grid, block, stream, ...) cgemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
struct rt_event_pair { };
#define RT_LAUNCH_TIMED(kernel, grid, block, stream, ev, ...) RT_LAUNCH(kernel, grid, block, stream, __VA_ARGS__)
inline void rt_event_create(rt_event_pair &) {}
inline void rt_event_destroy(rt_event_pair &) {}
inline void rt_event_start(rt_event_pair &, rt_stream_t) {}
inline void rt_event_stop(rt_event_pair &, rt_stream_t) {}
inline float rt_event_ms(rt_event_pair &) { return 0.f; }
struct rt_graph { };
inline bool rt_graphs_supported() { return false; }
inline void rt_capture_begin(rt_stream_t) {}
inline void rt_capture_end(rt_stream_t, rt_graph &) {}
inline void rt_graph_launch(rt_graph &, rt_stream_t) {}
inline void rt_graph_destroy(rt_graph &) {}
inline const char *rt_platform_name() { return "emulator (test only)"; }

#else

#define RT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); throw rt_out_of_memory(std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
typedef hipStream_t rt_stream_t;
// zero-filled device memory.  The fill runs on the stream the calling thread has announced (rt_alloc_scope: the session's
// own stream) or, outside such a scope, on a temporary non-blocking stream, and is waited for: the legacy null stream is
// never used (sessions on other host threads may be capturing a graph, which makes any legacy-stream operation fail),
// an upload enqueued right after the allocation cannot be overtaken by the fill, and no extra long-lived stream takes
// one of the process's four hardware queues away from the sessions.
inline hipStream_t &rt_alloc_stream_slot() { static thread_local hipStream_t s = nullptr; return s; }
struct rt_alloc_scope {
    hipStream_t prev;
    explicit rt_alloc_scope(hipStream_t s) : prev(rt_alloc_stream_slot()) { rt_alloc_stream_slot() = s; }
    ~rt_alloc_scope() { rt_alloc_stream_slot() = prev; }
};
inline void *rt_malloc(size_t n)
{
    void *p = nullptr; RT_CHECK(hipMalloc(&p, n ? n : 1));
    hipStream_t f = rt_alloc_stream_slot();
    const bool temp = f == nullptr;
    if (temp) RT_CHECK(hipStreamCreateWithFlags(&f, hipStreamNonBlocking));
    RT_CHECK(hipMemsetAsync(p, 0, n ? n : 1, f)); RT_CHECK(hipStreamSynchronize(f));
    if (temp) (void)hipStreamDestroy(f);
    return p;
}
inline void rt_free(void *p) { if (p) (void)hipFree(p); }
inline void *rt_malloc_host(size_t n) { void *p = nullptr; RT_CHECK(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault)); return p; }
inline void rt_free_host(void *p) { if (p) (void)hipHostFree(p); }
inline void rt_h2d(void *d, const void *h, size_t n, rt_stream_t s) { RT_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); }
inline void rt_d2h(void *h, const void *d, size_t n, rt_stream_t s) { RT_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); }
inline void rt_d2d(void *d, const void *sr, size_t n, rt_stream_t s) { RT_CHECK(hipMemcpyAsync(d, sr, n, hipMemcpyDeviceToDevice, s)); }
inline void rt_memset(void *d, int v, size_t n, rt_stream_t s) { RT_CHECK(hipMemsetAsync(d, v, n, s)); }
inline void rt_sync(rt_stream_t s) { RT_CHECK(hipStreamSynchronize(s)); }
inline void rt_set_device(int d) { if (d >= 0) RT_CHECK(hipSetDevice(d)); }
// (hipGetDevice as a process's FIRST runtime call reports "no ROCm-capable device" on ROCm 7.2: initialise explicitly)
inline int rt_get_device() { int d = 0; RT_CHECK(hipInit(0)); RT_CHECK(hipGetDevice(&d)); return d; }
inline void rt_mem_info(size_t *freeB, size_t *totalB) { RT_CHECK(hipMemGetInfo(freeB, totalB)); }
// compute units of the current device (a partitioned MI355X shows a fraction of the 256)
inline unsigned rt_compute_units() { int d = 0, n = 0; RT_CHECK(hipGetDevice(&d)); RT_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d)); return n > 0 ? (unsigned)n : 0u; }
inline rt_stream_t rt_stream_create() { hipStream_t s; RT_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); return s; }
inline void rt_stream_destroy(rt_stream_t s) { (void)hipStreamDestroy(s); }
#define RT_LAUNCH(kernel, grid, block, stream, ...) do { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (stream), __VA_ARGS__); RT_CHECK(hipGetLastError()); } while (0)
struct rt_event_pair { hipEvent_t a, b; };
// start / stop events carried by the kernel's own dispatch packet
#define RT_LAUNCH_TIMED(kernel, grid, block, stream, ev, ...) do { hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (stream), (ev).a, (ev).b, 0, __VA_ARGS__); RT_CHECK(hipGetLastError()); } while (0)
inline void rt_event_create(rt_event_pair &e) { RT_CHECK(hipEventCreate(&e.a)); RT_CHECK(hipEventCreate(&e.b)); }
inline void rt_event_destroy(rt_event_pair &e) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
inline void rt_event_start(rt_event_pair &e, rt_stream_t s) { RT_CHECK(hipEventRecord(e.a, s)); }
inline void rt_event_stop(rt_event_pair &e, rt_stream_t s) { RT_CHECK(hipEventRecord(e.b, s)); }
inline float rt_event_ms(rt_event_pair &e) { float ms = 0.f; RT_CHECK(hipEventSynchronize(e.b)); RT_CHECK(hipEventElapsedTime(&ms, e.a, e.b)); return ms; }
struct rt_graph { hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; };
inline bool rt_graphs_supported() { return true; }
inline void rt_capture_begin(rt_stream_t s) { RT_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); }
inline void rt_capture_end(rt_stream_t s, rt_graph &gr) { RT_CHECK(hipStreamEndCapture(s, &gr.g)); RT_CHECK(hipGraphInstantiate(&gr.e, gr.g, nullptr, nullptr, 0)); }
inline void rt_graph_launch(rt_graph &gr, rt_stream_t s) { RT_CHECK(hipGraphLaunch(gr.e, s)); }
inline void rt_graph_destroy(rt_graph &gr) { if (gr.e) (void)hipGraphExecDestroy(gr.e); if (gr.g) (void)hipGraphDestroy(gr.g); gr.e = nullptr; gr.g = nullptr; }
inline const char *rt_platform_name() { return "HIP gfx950"; }

#endif
