// chain_kernel.h -- the chained launch: ONE launch per batch.
//
// AsynchronousGibbsSampler::update (AsynchronousGibbsSampler.h:88-122) alternates populate and the evaluation of the queue; as two
// launches per batch (gen_kernel, eval_kernel) every batch pays two kernel boundaries and, between the last decision and the first
// attempt of the next batch, everything the generator does that never needed the decisions.  Here workgroups 0 .. gridDim-2 evaluate
// batch n exactly as the stand-alone evaluation kernels do, and the LAST workgroup is the generator of batch n + 1 (gen_populate.h,
// gen_body<.., CHAIN>).  It has the evaluation's workgroup size and splits by wave:
//   * attempt waves: prologue, classification of the next window with both ends' birth / death thresholds, and -- round 5 -- the window's
//     DRAWS (picks, records, matrix entries, a birth's gap: a round's three dependent memory trips) against the domain as this
//     workgroup's own commit left it;
//   * applier waves (the helper wave and the waves beyond it): fetch what each decision will rewrite, receive the decision as a pair of
//     tagged 8-byte granules (one write-through store each by the evaluation workgroup, polled past the caches: MI355X guide,
//     handoff-1to1), carry it out on the atomic domain and the factor matrix, and note which atom records, matrix cells, vector slots
//     and bitmap words change;
//   * behind the join a lane that read none of these has drawn what it would draw now; the others draw again (gen_draw_valid,
//     gen_round<.., AHEAD>) -- the same code against the current domain: bit-identical to the serial procedure.
// The evaluation workgroups write nothing but the granules and their A*P rows; the A*P updates run beside the generator.
//
// What an evaluation workgroup reads when it starts -- its queue record, the queue length, the batch tag -- exists in two copies, one
// per launch PARITY (a kernel argument: consecutive launches alternate): a launch of parity p evaluates copy p and its generator writes
// copy 1 - p, so nothing in a launch reads what the same launch writes, however late a workgroup starts.  Only the generator workgroup
// ever waits, and only for evaluation workgroups, which never wait: in whatever order the dispatcher starts the workgroups, on however few
// compute units, every one of them ends -- the launch cannot deadlock; the bounded poll (two seconds at least) is there for a workgroup the
// GPU does not schedule at all, and what happens then is chain_recover_kernel's business (below).  Round 6: the fused form's generator is
// the launch's FIRST workgroup (it enters ~0.25 us earlier than as the last of 241, and its window drawn ahead is what the join waits for:
// +0.3 %, profiles/r06_ab_small_experiments.txt); the test-only emulator, which runs workgroups one after the other in index order, keeps it
// last (a first workgroup there would wait for workgroups that have not run), as does the split form, whose update items are dealt out by
// index.  The grid is kept at one workgroup per compute unit or less and the host takes the chained form only for an update that runs alone:
// for speed, not for correctness.
// Bit-identical to the two-launch form: the same reductions, decisions and stores, and the erase cache's order does not matter (the
// flush sorts it by position, ConcurrentAtomicDomain.cpp:71-79).
#pragma once
#include "eval_kernel.h"

#define CHAIN_MAX_THREADS 512        // 8 waves of up to 256 VGPRs: one workgroup per compute unit, whichever role it plays
static_assert(CHAIN_MAX_THREADS == GEN_CHAIN_THREADS, "gen_kernel.h knows the chained launch's workgroup size");
#if defined(COGAPS_EMUL)
#define CHAIN_EVAL_GRID 7u           // (test-only emulator: a workgroup is a set of fibers; few of them keep the tests quick and make every workgroup evaluate several proposals)
#else
#define CHAIN_EVAL_GRID 240u         // evaluation workgroups of a chained launch (+ the generator: 241 <= 256 compute units)
#endif

#if defined(GEN_TIMELINE)
// dev: one chained launch on the chip-wide 100 MHz clock -- per workgroup {entry, decision published, end}; the generator workgroup's marks in g_chain_gen
__device__ unsigned long long g_chain_rt[256 * 4];
#endif
// SPLIT: the sampler's data vectors are evaluated in slices (more than 4096 elements: eval_kernel.h, EVAL_CHAIN_SPLIT) -- `slices` workgroups per
// proposal, the grid's evaluation workgroups a multiple of it; every workgroup takes its items in increasing order, a deciding
// workgroup waits only for items before its own and an update only for deciding workgroups: nothing waits in a circle.
template <int WIN, bool SPLIT>
CG_KERNEL void CG_LAUNCH_BOUNDS(CHAIN_MAX_THREADS) chain_kernel(const uint64_t *lcgMul, const uint64_t *lcgInc, GenScalars *gs, PropRec *queue, unsigned long long *grans, ChainSlot *slots,
                                                                uint32_t queueCap, uint32_t parity, uint32_t slices, const SamplerDev CG_CONSTANT *sp)
{
    constexpr int PH = SPLIT ? EVAL_CHAIN_SPLIT : EVAL_CHAIN;
#if defined(COGAPS_EMUL)
    constexpr bool GEN_FIRST = false;
#else
    constexpr bool GEN_FIRST = !SPLIT;
#endif
    if (GEN_FIRST ? cg_bid() == 0u : cg_bid() + 1u == cg_gdim()) {
        GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = nullptr; hot.queueUnits = nullptr; hot.eraseCap = 0; hot.queueCap = queueCap;
        hot.queueRd = queue + (size_t)parity * queueCap; hot.queueWr = queue + (size_t)(1u - parity) * queueCap; hot.grans = grans; hot.slotWr = &slots[1u - parity];
        gen_body<WIN, true, true, 0>(sp, hot);
        return;
    }
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = gs; hot.queueCap = queueCap; hot.slot = &slots[parity]; hot.grans = grans;
#if defined(GEN_TIMELINE)
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t vb = GEN_FIRST ? cg_bid() - 1u : cg_bid();      // this workgroup's number among the evaluation workgroups
    const unsigned long long clk0 = (vb == 0u && cg_tid() == 0u) ? cg_realtime() : 0ull;
    const EvalFirst first = eval_first<PH>(hot, slices, vb);
    const SamplerDev &S = eval_record<PH>(sp);
    if (vb == 0u && cg_tid() == 0u && S.launchClock) S.launchClock[2u * (first.tag % GAPS_CLOCK_RING)] = clk0;      // (launch clock: gaps_state.h)
    eval_body<PH, true>(S, slices, vb, cg_gdim() - 1u, hot, first);
#if defined(GEN_TIMELINE)
    if (cg_tid() == 0u && first.qlen >= 140u && gs->nSteps - gs->nDone >= 512u && cg_bid() < 255u) { g_chain_rt[cg_bid() * 4u] = rt0; g_chain_rt[cg_bid() * 4u + 2u] = __builtin_amdgcn_s_memrealtime(); g_chain_rt[cg_bid() * 4u + 3u] = first.qlen; }
#endif
}

// ---- the chained launch for a BATCH of chains (round 6; cogaps_batch_*, cogaps_hip.cpp: run_update_multi) ----------------------------------
// The chains of a batch step in lock-step; until round 6 every step was a generator launch (one workgroup per chain) and an evaluation
// launch.  Here one launch serves all chains: workgroups [c * wgPerChain, (c + 1) * wgPerChain) belong to chain c, the last of them is its
// generator, the others evaluate its queue -- chain_kernel's two roles, each chain's scalars, queue copies, slots and granules taken from
// its own record (the decisions' granules live in SamplerDev::grans, which the fused evaluation has no other use for).  Every workgroup
// of the launch is resident at once (C * wgPerChain <= compute units: the host sees to it), so a chain's evaluation workgroups take
// several proposals each -- in pairs, eval_chain_pair -- where the batched evaluation launch packs three workgroups per compute unit:
// the launch pays off for few chains (profiles/r06_ab_chained_batch.txt).
template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(CHAIN_MAX_THREADS) chain_kernel_multi(const SamplerDev CG_CONSTANT *arr, uint32_t parity, uint32_t wgPerChain)
{
    const uint32_t chain = cg_bid() / wgPerChain, vbid = cg_bid() - chain * wgPerChain;
    const SamplerDev CG_CONSTANT *sp = arr + chain;
    cg_const_warm<sizeof(SamplerDev)>(sp);
    const SamplerDev &S = *(const SamplerDev *)sp;
    PropRec *const queue = S.queue; const uint32_t queueCap = S.queueCap;
    if (vbid + 1u == wgPerChain) {
        GenHot hot; hot.lcgMul = S.lcgMul; hot.lcgInc = S.lcgInc; hot.gs = S.gs; hot.eraseList = nullptr; hot.queueUnits = nullptr; hot.eraseCap = 0; hot.queueCap = queueCap;
        hot.queueRd = queue + (size_t)parity * queueCap; hot.queueWr = queue + (size_t)(1u - parity) * queueCap; hot.grans = S.grans; hot.slotWr = &S.chainSlots[1u - parity];
        gen_body<WIN, false, true, 0>(sp, hot);
        return;
    }
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = S.gs; hot.queueCap = queueCap; hot.slot = &S.chainSlots[parity]; hot.grans = S.grans;
    const EvalFirst first = eval_first<EVAL_CHAIN>(hot, 1u, vbid);
    eval_body<EVAL_CHAIN, true>(S, 1u, vbid, wgPerChain - 1u, hot, first);
}

// ---- a hand-over that never arrived: the batch is completed, the update goes on (round 6) --------------------------------------------------
// A generator lane that gave up waiting for a decision (GAPS_ERR_SPIN: the bounded poll of gen_body_sh) applied nothing of that proposal,
// marked it (CHAIN_DROPPED_MARK in SamplerDev::queueUnits) and the workgroup left without generating; the launches enqueued behind it found
// empty queues.  By the time the host reads the error word every evaluation workgroup of the failed launch has ended -- the fused evaluation
// never waits for anything -- so every proposal's granules ARE there, and its A*P rows are updated.  This launch (one workgroup, behind the
// stream's synchronisation) carries out the marked proposals' decisions exactly as the generator's lanes would have (chain_fetch /
// chain_apply, erase cache entries appended), hands the traffic units to the two-launch generator's bookkeeping, and puts the scalars back:
// the state is the one the two-launch form has behind the evaluation of this batch, and the host goes on from it with gen_kernel /
// eval_kernel pairs.  A granule that is still missing (the split form's deciding workgroup gave up as well) leaves the error standing.
CG_KERNEL void CG_LAUNCH_BOUNDS(256) chain_recover_kernel(GenScalars *gs, const PropRec *queueRd, const unsigned long long *grans, uint32_t nSteps, const SamplerDev CG_CONSTANT *sp)
{
    cg_const_warm<sizeof(SamplerDev)>(sp);
    const SamplerDev &S = *(const SamplerDev *)sp;
    CG_SHARED uint32_t missing;
    const bool sparse = S.sparse != 0u;
    const uint32_t qlen = gs->applyCount, tag = (uint32_t)gs->batchEpoch, mark = CHAIN_DROPPED_MARK(tag);
    if (cg_tid() == 0u) { missing = 0u; gs->eraseCount = gs->savedErase; }
    cg_sync();
    for (uint32_t q = cg_tid(); q < qlen; q += cg_bdim()) {
        const unsigned long long g0 = cg_load_l2_u64(&grans[(size_t)q * CHAIN_GRAN_STRIDE]), g1 = cg_load_l2_u64(&grans[(size_t)q * CHAIN_GRAN_STRIDE + 1u]);
        if ((uint32_t)(g0 >> 32) != tag || (uint32_t)(g1 >> 32) != tag) { missing = 1u; continue; }
        if (S.queueUnits[q] == mark) {
            ChainItem it; chain_fetch(S, queueRd, q, it, sparse);
            const uint32_t code = (uint32_t)g0 & 0xFFu;
            if (code == CHAIN_ERASE) {
                const uint32_t k = cg_atomic_add_u32(&gs->eraseCount, 1u);
                if (k < S.eraseCap) S.eraseList[k] = it.eraseEntry; else missing = 1u;
            }
            chain_apply(it, code, gm_u2f((uint32_t)g1));
#if defined(COGAPS_EMUL)
            cg_atomic_add_u64(&gs->prof[5], 1ull);      // test-only build: decisions carried out by the recovery
#endif
        }
        S.queueUnits[q] = ((uint32_t)g0 >> 8) << (sparse ? 5u : 0u);      // (the two-launch generator adds the slots up: units of 4N bytes; sparse model: bytes)
    }
    cg_sync();
    if (cg_tid() == 0u && missing == 0u) { gs->qlen = qlen; gs->nSteps = nSteps; gs->applyCount = 0u; gs->savedErase = 0u; gs->updateFlushed = 0u; gs->error = GAPS_OK; }
}

#if defined(COGAPS_EMUL)
// test-only emulator (workgroups run one after the other): the split form's A*P updates as a launch of their own behind the chained one
CG_KERNEL void chain_updates_kernel(PropRec *queue, unsigned long long *grans, ChainSlot *slots, uint32_t queueCap, uint32_t parity, const SamplerDev CG_CONSTANT *sp)
{
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = nullptr; hot.queueCap = queueCap; hot.slot = &slots[parity]; hot.grans = grans;
    const ChainSlot cs = *hot.slot;
    eval_chain_updates(*(const SamplerDev *)sp, hot, cs.tag, cs.qlen, cg_bid(), cg_gdim());
}
#endif
