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
// copy 1 - p, so nothing in a launch reads what the same launch writes, however late a workgroup starts.  The generator workgroup is
// the last one so that every evaluation workgroup has been dispatched when it begins to wait (the dispatcher starts workgroups in index
// order: observed, not promised -- a generator that waited for a workgroup not yet started would give up after two seconds with
// GAPS_ERR_SPIN, applying nothing; and the test-only emulator, which runs workgroups in index order, never spins); the grid is kept at
// one workgroup per compute unit or less, all resident at once, and the host takes the chained form only for an update that runs alone.
// Bit-identical to the two-launch form: the same reductions, decisions and stores, and the erase cache's order does not matter (the
// flush sorts it by position, ConcurrentAtomicDomain.cpp:71-79).
#pragma once
#include "eval_kernel.h"

#define CHAIN_MAX_THREADS 512        // 8 waves of up to 256 VGPRs: one workgroup per compute unit, whichever role it plays
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
    if (cg_bid() + 1u == cg_gdim()) {
        GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = nullptr; hot.queueUnits = nullptr; hot.eraseCap = 0; hot.queueCap = queueCap;
        hot.queueRd = queue + (size_t)parity * queueCap; hot.queueWr = queue + (size_t)(1u - parity) * queueCap; hot.grans = grans; hot.slotWr = &slots[1u - parity]; hot.slotRd = &slots[parity];
        gen_body<WIN, true, true>(sp, hot);
        return;
    }
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = gs; hot.queueCap = queueCap; hot.slot = &slots[parity]; hot.grans = grans; hot.pub = 0u; hot.pubBase = nullptr; hot.pubBytes = 0u; hot.pubTag = 0u;
#if defined(GEN_TIMELINE)
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned long long clk0 = (cg_bid() == 0u && cg_tid() == 0u) ? cg_realtime() : 0ull;
    const EvalFirst first = eval_first<PH>(hot, slices, cg_bid());
    const SamplerDev &S = eval_record<PH>(sp);
    if (cg_bid() == 0u && cg_tid() == 0u && S.launchClock) S.launchClock[2u * (first.tag % GAPS_CLOCK_RING)] = clk0;      // (launch clock: gaps_state.h)
    eval_body<PH, true>(S, slices, cg_bid(), cg_gdim() - 1u, hot, first);
#if defined(GEN_TIMELINE)
    if (cg_tid() == 0u && first.qlen >= 140u && gs->nSteps - gs->nDone >= 512u && cg_bid() < 255u) { g_chain_rt[cg_bid() * 4u] = rt0; g_chain_rt[cg_bid() * 4u + 2u] = __builtin_amdgcn_s_memrealtime(); g_chain_rt[cg_bid() * 4u + 3u] = first.qlen; }
#endif
}

// ---- The persistent generator beside evaluation launches of their own (round 5) ------------------------------------------------------------
// The chained launch still pays, on the decide -> generate -> decide cycle, a kernel boundary and the generator workgroup's start-up per
// batch, and its evaluation workgroups begin only when the launch does.  Here the generator is ONE workgroup in a launch of its own that
// stays for `nBatches` batches (chain_gen_kernel: gen_body_sh<.., PERSIST> in a loop -- its LDS, its compute unit and the domain's lines
// in that unit's caches stay with it), and every batch's evaluation is a launch of its own on a SECOND stream (chain_eval_kernel), enqueued
// ahead: its workgroups are resident and poll the batch's slot when the generator publishes it -- records written through, stores
// drained, one barrier, the slot (gen_helper) -- evaluate, hand the decisions back as the chained launch's do (tagged granules) and
// update their A*P rows; the launch's end (the next evaluation launch follows it on the same stream) makes the rows visible to the next batch.
// Nothing else crosses: the evaluation reads the data, the other factor, A*P and its record; the generator reads the domain, the matrix
// cells (which its own lanes write) and the granules.  Which batch a launch evaluates: ctl[0] (the batch tag it waits for), advanced by
// the launch's last workgroup to leave (ctl[1] counts them) -- a kernel argument cannot count inside a replayed graph.  An update that
// is over marks both slots CHAIN_FIN: the evaluation launches still enqueued leave at once.  Needs the two launches to RUN at the same
// time: the host takes this form only where a probe at start-up saw two streams' kernels overlap (cogaps_hip.cpp), and every wait is
// bounded (GAPS_ERR_SPIN).  The test-only emulator and the counter tools run the same two kernels one batch at a time on one stream.
template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(CHAIN_MAX_THREADS) chain_gen_kernel(const uint64_t *lcgMul, const uint64_t *lcgInc, GenScalars *gs, PropRec *queue, unsigned long long *grans, ChainSlot *slots,
                                                                    unsigned char *pub, uint32_t queueCap, uint32_t parity0, uint32_t nBatches, const SamplerDev CG_CONSTANT *sp)
{
    CG_SHARED GenShared<WIN> sh;
    if (cg_tid() == 0u) { sh.barSeq = 0u; sh.endGen = (uint32_t)gs->batchEpoch; }      // (the tag of the batch in the queue: not the one any batch of this launch ends with)
    cg_sync();
    const uint32_t pubBytes = queueCap * CHAIN_PUB_BYTES;
    for (uint32_t k = 0; k < nBatches; ++k) {
        const uint32_t parity = (parity0 + k) & 1u;
        GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = nullptr; hot.queueUnits = nullptr; hot.eraseCap = 0; hot.queueCap = queueCap;
        hot.queueRd = queue + (size_t)parity * queueCap; hot.queueWr = queue + (size_t)(1u - parity) * queueCap; hot.grans = grans; hot.slotWr = &slots[2u * (1u - parity)]; hot.slotRd = &slots[2u * parity];
        hot.pubWr = pub + (size_t)(1u - parity) * pubBytes; hot.pubBytes = pubBytes;
        if (gen_body_sh<WIN, true, true, true>(sp, hot, sh) != 0u) break;
    }
}
// (the slots of this form lie 16 bytes apart: an evaluation workgroup's first wave reads the slot and its record's granules with ONE 16-byte
// load per turn -- lanes 0-6 a granule each, lane 7 the slot -- until the slot carries the batch's tag and, if the queue is long enough
// to hold a proposal for this workgroup, every granule does)
CG_KERNEL void CG_LAUNCH_BOUNDS(CHAIN_MAX_THREADS) chain_eval_kernel(GenScalars *gs, const unsigned char *pub, unsigned long long *grans, const ChainSlot *slots, uint32_t *ctl, uint32_t queueCap, uint32_t parity,
                                                                     const SamplerDev CG_CONSTANT *sp)
{
    CG_SHARED uint32_t first_w[24]; CG_SHARED uint32_t slotSeen[2];
    const uint32_t expect = ctl[0], pubBytes = queueCap * CHAIN_PUB_BYTES;
    const unsigned char *pubBase = pub + (size_t)parity * pubBytes;
    const uint32_t t = cg_tid(), qMine = cg_bid() < queueCap ? cg_bid() : 0u;
    if (t < 64u) {
        const cg_pub pb = cg_pub_open(pubBase, pubBytes), ps = cg_pub_open(slots, 32u);
        cg_u4 v; v.x = 0u; v.y = 0u; v.z = 0u; v.w = 0u; uint32_t spins = 0, qlen = 0u;
        // (this launch starts while the generator is at the join of the batch it will evaluate, some 6 us or more before the slot is written:
        // 240 workgroups reading past the caches all that time slow every memory trip of the generator -- MI355X guide, polling-cost --
        // so the first look comes late and the turns are a fifth of a microsecond apart)
#if !defined(EXP_NO_NAP)
        cg_nap_long();
#endif
        for (;;) {
            // (lanes 8-63 repeat lane 7's load)
            v = t < 7u ? cg_pub_load(pb, qMine * CHAIN_PUB_BYTES + 16u * t) : cg_pub_load(ps, 16u * parity);
            const uint32_t sq = cg_wave_bcast_u32(v.x, 7), stag = cg_wave_bcast_u32(v.y, 7);
            const bool fin = sq == CHAIN_FIN, slotOk = stag == expect;
            const bool recOk = t >= 7u || v.w == expect;
            qlen = fin ? 0u : sq;
            if (fin || (slotOk && (cg_bid() >= qlen || cg_ballot(!recOk) == 0ull))) break;
            if (cg_poll_expired(++spins)) { if (t == 0u) gs->error = GAPS_ERR_SPIN_EVAL; qlen = 0u; break; }      // (bounded; nothing is evaluated)
            if ((spins & 4095u) == 0u && cg_load_agent_u32(&gs->error) != 0u) { qlen = 0u; break; }      // (the generator gave up: the launches still enqueued leave)
            cg_nap_short();
        }
        if (t < 7u) { first_w[3u * t] = v.x; first_w[3u * t + 1u] = v.y; first_w[3u * t + 2u] = v.z; }
        if (t == 7u) { slotSeen[0] = qlen; slotSeen[1] = expect; }
    }
    const SamplerDev &S = eval_record<EVAL_CHAIN>(sp);
    cg_sync_lds();
    const uint32_t qlen = slotSeen[0];
    if (cg_bid() < qlen) {
        EvalHot hot; hot.queue = nullptr; hot.gs = gs; hot.queueCap = queueCap; hot.slot = nullptr; hot.grans = grans; hot.pub = 1u; hot.pubBase = pubBase; hot.pubBytes = pubBytes; hot.pubTag = expect;
        uint32_t w[24];
#pragma unroll
        for (int k = 0; k < 20; ++k) w[k] = first_w[k];
        w[20] = 0u; w[21] = 0u; w[22] = 0u; w[23] = 0u;
        EvalFirst first; first.qlen = qlen; first.tag = expect; first.T = gs->annealTemp; __builtin_memcpy(&first.p, w, sizeof(PropRec));
        eval_body<EVAL_CHAIN, true>(S, 1u, cg_bid(), cg_gdim(), hot, first);
    }
    if (t == 0u) {      // the last workgroup to leave moves the launches on to the next batch
        const uint32_t old = cg_atomic_add_u32(&ctl[1], 1u);
        if (old + 1u == cg_gdim()) { ctl[1] = 0u; ctl[0] = expect + 1u; }
    }
}

#if defined(COGAPS_EMUL)
// test-only emulator (workgroups run one after the other): the split form's A*P updates as a launch of their own behind the chained one
CG_KERNEL void chain_updates_kernel(PropRec *queue, unsigned long long *grans, ChainSlot *slots, uint32_t queueCap, uint32_t parity, const SamplerDev CG_CONSTANT *sp)
{
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = nullptr; hot.queueCap = queueCap; hot.slot = &slots[parity]; hot.grans = grans; hot.pub = 0u; hot.pubBase = nullptr; hot.pubBytes = 0u; hot.pubTag = 0u;
    const ChainSlot cs = *hot.slot;
    eval_chain_updates(*(const SamplerDev *)sp, hot, cs.tag, cs.qlen, cg_bid(), cg_gdim());
}
#endif
