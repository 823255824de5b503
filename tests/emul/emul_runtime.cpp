// emul_runtime.cpp -- TEST-ONLY cooperative fiber scheduler behind emul_runtime.h.
#include "emul_runtime.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <vector>
#include <mutex>

extern "C" void cgemu_ctx_switch(void **from_sp, void **to_sp);
asm(R"(
.text
.globl cgemu_ctx_switch
.type cgemu_ctx_switch,@function
cgemu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size cgemu_ctx_switch,.-cgemu_ctx_switch
)");

namespace cgemu {

LaneCtx g_lane;

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber { void *sp; char *stack; State st; float xch; };

static const size_t STACK_BYTES = 64 * 1024;
static std::vector<Fiber> g_fibers;
static std::vector<char *> g_stack_pool;
static void *g_sched_sp;
static unsigned g_cur;
static const std::function<void()> *g_body;
static unsigned g_bdim, g_bid, g_gdim;

static void fiber_main()
{
    (*g_body)();
    g_fibers[g_cur].st = DONE;
    cgemu_ctx_switch(&g_fibers[g_cur].sp, &g_sched_sp);
    abort();
}

static void yield_to_sched(State st)
{
    unsigned me = g_cur;
    g_fibers[me].st = st;
    cgemu_ctx_switch(&g_fibers[me].sp, &g_sched_sp);
    g_lane.tid = me; g_lane.bid = g_bid; g_lane.bdim = g_bdim; g_lane.gdim = g_gdim;
}

void block_barrier() { yield_to_sched(WAIT_BLOCK); }
void wave_barrier() { yield_to_sched(WAIT_WAVE); }

float wave_exchange_f32(float v, int mask)
{
    unsigned me = g_cur;
    g_fibers[me].xch = v;
    yield_to_sched(WAIT_WAVE);            // everyone in the wave has deposited
    unsigned src = (me & ~63u) | ((me ^ (unsigned)mask) & 63u);
    float r = src < g_bdim ? g_fibers[src].xch : v;
    yield_to_sched(WAIT_WAVE);            // everyone has read before the next deposit
    return r;
}

float wave_read_f32(float v, int lane)
{
    unsigned me = g_cur;
    g_fibers[me].xch = v;
    yield_to_sched(WAIT_WAVE);
    unsigned src = (me & ~63u) | ((unsigned)lane & 63u);
    float r = src < g_bdim ? g_fibers[src].xch : v;
    yield_to_sched(WAIT_WAVE);
    return r;
}

unsigned long long wave_ballot(bool p)
{
    unsigned me = g_cur;
    g_fibers[me].xch = p ? 1.f : 0.f;
    yield_to_sched(WAIT_WAVE);
    unsigned lo = me & ~63u;
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && lo + i < g_bdim; ++i) if (g_fibers[lo + i].st != DONE && g_fibers[lo + i].xch != 0.f) m |= 1ull << i;
    yield_to_sched(WAIT_WAVE);
    return m;
}

static char *get_stack(unsigned i)
{
    while (g_stack_pool.size() <= i) {
        void *p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); abort(); }
        g_stack_pool.push_back((char *)p);
    }
    return g_stack_pool[i];
}

static void run_block(unsigned bid, unsigned bdim, unsigned gdim, const std::function<void()> &body)
{
    g_body = &body; g_bdim = bdim; g_bid = bid; g_gdim = gdim;
    g_fibers.assign(bdim, Fiber());
    for (unsigned i = 0; i < bdim; ++i) {
        Fiber &f = g_fibers[i];
        f.stack = get_stack(i); f.st = READY; f.xch = 0.f;
        // initial frame: 6 callee-saved zeros + return address = fiber_main; entry rsp % 16 == 8
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void **sp = (void **)(top - 8);     // slot that makes rsp%16==8 after ret
        *--sp = (void *)fiber_main;
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        f.sp = sp;
    }
    unsigned live = bdim;
    while (live) {
        bool progressed = false;
        for (unsigned i = 0; i < bdim; ++i) {
            if (g_fibers[i].st != READY) continue;
            g_cur = i;
            g_lane.tid = i; g_lane.bid = bid; g_lane.bdim = bdim; g_lane.gdim = gdim;
            cgemu_ctx_switch(&g_sched_sp, &g_fibers[i].sp);
            progressed = true;
            if (g_fibers[i].st == DONE) --live;
        }
        // release wave collectives: all non-done lanes of a wave waiting
        for (unsigned w = 0; w * 64 < bdim; ++w) {
            unsigned lo = w * 64, hi = lo + 64 < bdim ? lo + 64 : bdim;
            bool all = true, any = false;
            for (unsigned i = lo; i < hi; ++i) {
                if (g_fibers[i].st == WAIT_WAVE) any = true;
                else if (g_fibers[i].st != DONE) all = false;
            }
            if (any && all) { for (unsigned i = lo; i < hi; ++i) if (g_fibers[i].st == WAIT_WAVE) g_fibers[i].st = READY; progressed = true; }
        }
        // release block barrier: all non-done lanes waiting
        bool all = true, any = false;
        for (unsigned i = 0; i < bdim; ++i) {
            if (g_fibers[i].st == WAIT_BLOCK) any = true;
            else if (g_fibers[i].st != DONE) all = false;
        }
        if (any && all) { for (unsigned i = 0; i < bdim; ++i) if (g_fibers[i].st == WAIT_BLOCK) g_fibers[i].st = READY; progressed = true; }
        if (!progressed && live) { fprintf(stderr, "cgemu: deadlock (divergent barrier) in block %u\n", bid); abort(); }
    }
}

void launch(unsigned grid, unsigned block, const std::function<void()> &body)
{
    // one emulated device: sessions driven from several host threads (shards in flight) take turns
    static std::mutex device;
    std::lock_guard<std::mutex> hold(device);
    for (unsigned b = 0; b < grid; ++b) run_block(b, block, grid, body);
}

} // namespace cgemu
