// gen_populate.h -- the generator kernel body (included from gen_kernel.h, which documents the method).
//
// Lane assignment: attempts are classified in attempt order (one lane = one attempt), then SORTED BY
// TYPE so that a wavefront executes (mostly) one of the birth/death, move or exchange code paths instead
// of all four under divergence -- the attempt ordinal `ct` travels with the lane and is what the
// conflict stamps compare.  Per-attempt prefix counts (queue slot, birth rank) come from LDS bit masks.
//
// Wave specialisation (round 3).  A launch has WIN attempt lanes (WIN / 64 waves) plus ONE HELPER WAVE (lanes WIN .. WIN + 63) that
// owns the launch's serial chores, so that they run BESIDE the attempt waves instead of before and after them:
//   * the flush of the erase cache (ConcurrentAtomicDomain.cpp:71-79) -- <= FLUSH_MAX atoms, one helper lane each, the steps of one
//     wave need no workgroup barrier between them -- runs while the attempt waves classify their attempts (A1) and draw what needs
//     only the window's scalars (first stage of A2); the two sides meet at one barrier before the first load that reads the domain;
//   * the conflict table's value words are preset while the launch's first memory trip is in flight (so registration needs no
//     opener / second barrier), and
//   * the round bookkeeping (counts, generator state, queue-length mean) and the write-back of the generator's scalars run while the
//     attempt lanes commit; a batch's last round ends without a closing barrier.
// A workgroup barrier counts every wave, so the helper executes exactly the barriers the attempt waves execute (the same number, at
// points chosen so that neither side waits long); what it does between them is its own.
#pragma once

// The generator by role, in the order the pieces build on each other (round 6: one file of 1 700 lines until then):
#include "gen_flush.h"        // flushEraseCache + the helper wave
#include "gen_draw.h"         // classification ahead, draws, notes, validation
#include "gen_round.h"        // one round: registration, look-ups, commit
#include "gen_chain_apply.h"  // the chained launch's applier lanes

// sp: the sampler's record in device memory (constant address space: scalar loads).  ASYNC: the launch's hot pointers arrived as
// preloaded kernel arguments, the record's lines are requested behind the first trip and waited for after the conflict table has been
// emptied (one-chain launch); otherwise the caller has read the record already (batched launch: the hot pointers come from it).
// CHAIN: the chained launch's generator workgroup (above; chain_kernel.h).
// launch clock (gaps_state.h): every wave of the chained launch's generator workgroup leaves the chip-wide clock at its own end in the
// launch's ring slot (a non-returning maximum: the waves end without a closing barrier, the last one's stamp stays)
struct GenClockEnd {
    unsigned long long *slot; unsigned t;
    CG_DEVICE GenClockEnd(unsigned t_) : slot(nullptr), t(t_) {}
    CG_DEVICE ~GenClockEnd() { if (slot && (t & 63u) == 0u) cg_atomic_max_u64(slot, cg_realtime()); }
};
// SP: the model the kernel serves -- 0 dense, 1 sparse (the HybridMatrix branches and a decision's sparse fields fold away: -0.38 us per chained
// launch of the dense model, profiles/r06_ab_model_as_a_template_parameter.txt), -1 either (read from the record)
template <int WIN, bool ASYNC, bool CHAIN = false, int SP = -1>
CG_DEVICE void gen_body_sh(const SamplerDev CG_CONSTANT *sp, const GenHot hot, GenShared<WIN> &sh)
{
    GenClockEnd clockEnd(cg_tid());
    constexpr unsigned TPB = (unsigned)WIN + 64u;       // attempt lanes + the helper wave
    const unsigned t = cg_tid();
    // wave-uniform roles: attempt lanes, the helper wave, and -- chained launch only, which has the evaluation's workgroup size -- the
    // waves beyond it, which with the helper wave apply the previous batch's decisions
    const bool attempt = t < (unsigned)WIN, helper = t >= (unsigned)WIN && t < TPB, spare = t >= TPB, applier = CHAIN && !attempt;
    const unsigned ht = t - (unsigned)WIN, ta = attempt ? t : 0u;
    GenScalars *gs = hot.gs;

    GEN_TS_INIT(); GEN_TS(0); GEN_TS(0);
    GEN_RT(0);
    // k-step PCG jumps for this lane's (u1,u2): k = 2t, or 2(t-1) when attempt 0 replays cached values
    const uint64_t jm0 = hot.lcgMul[2u * ta], ji0 = hot.lcgInc[2u * ta];
    const uint64_t jm1 = hot.lcgMul[ta ? 2u * (ta - 1u) : 0u], ji1 = hot.lcgInc[ta ? 2u * (ta - 1u) : 0u];
    // first memory trip of the launch, everything independent: the scalars every lane needs (one lane per word into LDS, where they
    // live for the whole launch), the erase cache (helper lanes) and the traffic-unit slots (attempt lanes) read speculatively
    // (the chained launch has neither: its lanes collect both from the decisions they apply)
    unsigned long long specE = (!CHAIN && helper && ht < (unsigned)FLUSH_MAX && ht < hot.eraseCap) ? hot.eraseList[ht] : 0ull;
    // chained launch: the lane's queue record of the batch being evaluated (slot t always exists), with the launch's first trip
    PropRec p0; p0.type = 0; p0.h1 = 0; p0.h2 = 0; p0.pos = 0; p0.curPos = 0; p0.r1 = 0; p0.c1 = 0; p0.r2 = 0; p0.c2 = 0; p0.m1 = 0.f; p0.m2 = 0.f; p0.old1 = 0.f; p0.old2 = 0.f;
    if (applier && ht < hot.queueCap) p0 = hot.queueRd[ht];

    uint32_t units = (!CHAIN && attempt && t < hot.queueCap) ? hot.queueUnits[t] : 0u;
    const uint64_t jmW = hot.lcgMul[2 * WIN], jiW = hot.lcgInc[2 * WIN];
    constexpr uint32_t GSW = (uint32_t)(sizeof(GenScalars) / 4u);
    static_assert(GSW <= 2u * TPB, "at most two words of GenScalars per lane");
    const uint32_t gword = (t < GSW) ? reinterpret_cast<const uint32_t *>(gs)[t] : 0u;
    const uint32_t gword2 = (t + TPB < GSW) ? reinterpret_cast<const uint32_t *>(gs)[t + TPB] : 0u;
    cg_sched_fence();
    GEN_TS(26);
    // the record's lines are requested, the conflict table is emptied while they and the first trip are on their way
    cg_const_lines lines;
    if (ASYNC) cg_const_warm_begin<sizeof(SamplerDev)>(sp, lines);
    {   // empty conflict table: keys, and the value words ("nobody" = all ones: whoever registers first under a key needs no
        // opener) -- by ALL lanes: one wave alone stores to LDS at a fraction of the workgroup's rate (the helper wave presetting
        // the 64 KB of value words by itself took 5 k cycles, longer than the trip)
        // keys and value words are one contiguous region of 5 * GEN_TAB_NB 16-byte units: a fixed number of stores per lane at constant
        // offsets from one address (a run-time loop bound made every store a loop iteration with its own branch)
        GenTabKeys none; none.k[0] = none.k[1] = none.k[2] = none.k[3] = GEN_TAB_EMPTY;
        static_assert(offsetof(GenShared<WIN>, bval) == offsetof(GenShared<WIN>, bkey) + 16u * (size_t)GEN_TAB_NB, "keys and value words are contiguous");
        constexpr uint32_t UNITS = 5u * (uint32_t)GEN_TAB_NB, ROUNDS = (UNITS + TPB - 1u) / TPB;
        GenTabKeys *tab = reinterpret_cast<GenTabKeys *>(&sh.bkey[0]) + t;
        // (round 6, chained launch: the table is needed behind the join only -- its 80 KB are emptied by the APPLIER waves while they wait for the
        // decisions, see below, and the workgroup's first barrier, which the window drawn ahead starts from, comes ~0.4 us earlier)
#pragma unroll
        for (uint32_t k = 0; k < ROUNDS; ++k) { if (!CHAIN && !spare && ((k + 1u) * TPB <= UNITS || t + k * TPB < UNITS)) tab[k * TPB] = none; }
        if (CHAIN) {      // ... and the notes of what the decisions change (atom records, matrix cells), by every lane of the launch's workgroup
            static_assert(offsetof(GenShared<WIN>, dCell) == offsetof(GenShared<WIN>, dAtom) + 4u * (size_t)GEN_DIRTY_ATOMS
                          && offsetof(GenShared<WIN>, dErase) == offsetof(GenShared<WIN>, dCell) + 4u * (size_t)GEN_DIRTY_CELLS, "the note tables are contiguous");
            GenTabKeys *dt = reinterpret_cast<GenTabKeys *>(&sh.dAtom[0]);
            GenTabKeys zero; zero.k[0] = zero.k[1] = zero.k[2] = zero.k[3] = 0u;
            for (uint32_t i = t; i < (uint32_t)(GEN_DIRTY_ATOMS + GEN_DIRTY_CELLS + GEN_DIRTY_ERASE) / 4u; i += cg_bdim()) dt[i] = zero;
        }
    }
    GEN_TS(27);
    if (ASYNC) sp = cg_const_warm_end(sp, lines);
    const SamplerDev &S = *(const SamplerDev *)sp;
    const bool isSparse = SP < 0 ? S.sparse != 0u : SP != 0;
    GEN_TS(28);
    if (t < GSW) reinterpret_cast<uint32_t *>(&sh.g)[t] = gword;
    if (t + TPB < GSW) reinterpret_cast<uint32_t *>(&sh.g)[t + TPB] = gword2;
    if (t == 0) { sh.newFront = CG_KEEP; sh.unitSum = 0; sh.jmul[WIN] = jmW; sh.jinc[WIN] = jiW; if (CHAIN) { sh.eraseN = 0; sh.specBad = 0; sh.spinFail = 0; sh.anyRedo = 0; } }
    if (attempt) { sh.jmul[t] = jm0; sh.jinc[t] = ji0; }        // even-step PCG jumps, for the round bookkeeping
    GEN_TS(29);
    cg_sync_lds();
    // the scalars every lane needs, from the LDS copy (wave-uniform: kept in scalar registers)
    uint32_t e_m = CHAIN ? 0u : cg_uniform_u32(sh.g.eraseCount);
    const uint32_t e_n = cg_uniform_u32(sh.g.nAtoms), e_fc = cg_uniform_u32(sh.g.freeCount), e_prevQ = cg_uniform_u32(sh.g.qlen),
                   e_nDone = cg_uniform_u32(sh.g.nDone), e_nSteps = cg_uniform_u32(sh.g.nSteps);
    GEN_TS(1);
    const bool updateDone = e_nDone >= e_nSteps;
    uint64_t seedC = 0ull; bool dpStaged = false, trySpec = false, specDone = false; uint32_t dpBase = 0;
    GenSpec specKeep; specKeep.bBefore = 0; specKeep.dBefore = 0; specKeep.guess = 0; specKeep.active = 0; specKeep.u1 = 0.f; specKeep.u2 = 0.f; specKeep.go = 0; specKeep.ct = 0; specKeep.info = 0;
    specKeep.rng = 0; specKeep.pos = 0; specKeep.bin = 0; specKeep.r1 = 0; specKeep.c1 = 0;
    uint64_t epoch0 = 0;
    GenDraw drawKeep; gen_draw_clear(drawKeep);      // chained launch: the lane's attempt of the next window, drawn ahead of the decisions
    GenCheck checkKeep; checkKeep.atomA.a = checkKeep.atomA.b = checkKeep.atomB.a = checkKeep.atomB.b = checkKeep.slot.a = checkKeep.slot.b = 0u;
    checkKeep.cellA.a = checkKeep.cellA.b = checkKeep.cellB.a = checkKeep.cellB.b = 0u; checkKeep.iPartS = 1u;
    if (!CHAIN) {
        GEN_TS_ZERO(7u, 13u);
#if defined(GEN_TIMELINE)
        if (t == 0u) { sh.rtOn = 0u; sh.rtLog = 0u; }
#endif
        if (attempt) {   // roofline bookkeeping: add up the traffic units the evaluation kernel left per queue slot (the helper wave adds
            // the sum to evalBytes at the end of the batch)
            if (t >= e_prevQ) units = 0;
            for (uint32_t q = t + WIN; q < e_prevQ; q += WIN) units += S.queueUnits[q];
            const uint32_t waveUnits = cg_wave_sum_u32(units);
            if ((t & 63u) == 0u && waveUnits) cg_atomic_add_u32(&sh.unitSum, waveUnits);
        }
    } else {
        // ---- Everything that does not depend on the previous batch's decisions, while its evaluation workgroups run.  The workgroup's
        // waves split (round 5): the ATTEMPT waves classify the next window (gen_spec_a1) and DRAW it -- picks, records, matrix entries,
        // a birth's gap: the three dependent memory trips of a round -- against the domain as this workgroup's own commit left it;
        // the APPLIER waves (the helper wave and the waves beyond it: the launch has the evaluation's workgroup size) fetch what the
        // decisions will rewrite, receive the decisions, carry them out and note every atom record, matrix cell, vector slot and
        // bitmap word they or the flush change (sh.dAtom / dCell / dSlot / dirty).  Behind the join an attempt lane whose reads
        // touched none of these has drawn exactly what it would draw now; the others draw again (gen_round<.., AHEAD>).
        unsigned long long *const eraseList = S.eraseList; const uint32_t eraseCap = S.eraseCap;      // (read here: nothing of the record is read behind the wait)
        GEN_TS(30);
        GEN_RT(1);
#if defined(GEN_TIMELINE)
        if (t == 0u) { sh.rtOn = (e_prevQ >= 140u && e_nSteps - e_nDone >= 512u) ? 1u : 0u; sh.rt[6] = 0ull; sh.rt[7] = 0ull; sh.rtLog = (WIN == 256 && e_prevQ >= 100u && e_nSteps - e_nDone >= 512u) ? 1u : 0u; }
#endif
#if defined(GEN_TEST_APPLIER_LANES)
        const uint32_t NA = (uint32_t)GEN_TEST_APPLIER_LANES, al = t - (uint32_t)WIN;      // test-only variant of the emulator build: few applier lanes, so that short queues take several passes
        const bool applying = applier && al < NA;
#else
        const uint32_t NA = cg_bdim() - (uint32_t)WIN, al = t - (uint32_t)WIN;      // applier lanes (al: this lane's number among them)
        const bool applying = applier;
#endif
        const bool have0 = applying && al < e_prevQ;
        // (sparse model, the full window: the attempt lanes take part in the hand-over -- below; the half window's launch has 384 applier lanes for
        // queues of 128 per round, its attempt lanes would only pay for the pass: P side +0.5 us, profiles/r06_ab_sparse_chained_launch.txt)
#if defined(GEN_TEST_APPLIER_LANES)
        constexpr bool ATTEMPTS_APPLY = SP == 1;
#else
        constexpr bool ATTEMPTS_APPLY = SP == 1 && 2 * WIN >= GEN_CHAIN_THREADS;
#endif
        const uint32_t passLanes = ATTEMPTS_APPLY ? NA + (uint32_t)WIN : NA;      // proposals per pass of the hand-over
        // second trip (the first brought the scalars and an applier's record): what the decision will rewrite; the seeds, the table's window
        ChainMid mid0; mid0.l2 = CG_NONE; mid0.head1 = CG_NONE; mid0.b1 = 0; mid0.b2 = 0; mid0.x1 = 0ull; mid0.x2 = 0ull;
        mid0.a.pos = 0; mid0.a.lpos = 0; mid0.a.rpos = 0; mid0.a.left = CG_NONE; mid0.a.right = CG_NONE; mid0.a.mass = 0.f; mid0.a.rmass = 0.f; mid0.a.idx = 0; mid0.a.pad0 = 0;
        // (the free-handle stack's top entries, which a committing birth pops: nothing the decisions change -- only the flush pushes)
        const uint32_t freeTopAhead = (helper && ht < 16u && ht < e_fc) ? S.freeHandles[e_fc - 1u - ht] : CG_NONE;
        if (applier) mid0 = chain_fetch_mid(S, p0, isSparse);      // (p0 of a lane without a proposal: a slot of the queue copy, whatever it holds -- handles and positions of an older batch: valid addresses)
        if (!updateDone && attempt) seedC = S.seeds[e_nDone + t < e_nSteps ? e_nDone + t : e_nSteps - 1u];
        const uint32_t span = e_prevQ + (uint32_t)(WIN - 1);
        dpBase = e_n > span ? e_n - span : 0u;
        const uint32_t dpCnt = e_n + (uint32_t)WIN - dpBase;                  // entries dpBase .. nAtoms + WIN - 1
        dpStaged = !updateDone && dpCnt <= 4u * (uint32_t)WIN;
        const uint32_t nLo = e_n > e_prevQ ? e_n - e_prevQ : 0u, nHi = e_n;
        trySpec = dpStaged && nLo >= 2u;                                      // (tiny domains: the type also depends on the count itself)
        float dpw[4] = {0.f, 0.f, 0.f, 0.f};
        if (dpStaged && !spare) {
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) { const uint32_t i = t + k * TPB; dpw[k] = S.deathProb[dpBase + (i < dpCnt ? i : dpCnt - 1u)]; }
        }
        cg_sched_fence();
        GEN_TS(31);
        // ... and under it: the classification (the two thresholds computed -- the table's entries would arrive with the trip)
        GenRoundCtx rcS; GenSpec spS;
        rcS.t = t; rcS.jm0 = jm0; rcS.ji0 = ji0; rcS.jm1 = jm1; rcS.ji1 = ji1; rcS.seed1 = 0ull; rcS.g_qrng = sh.g.qrng; rcS.g_skip = sh.g.useCached ? 1u : 0u;
        rcS.g_u1 = sh.g.u1; rcS.g_u2 = sh.g.u2; rcS.remaining = e_nSteps - e_nDone; rcS.K = S.K; rcS.sparse = isSparse ? 1u : 0u;
        spS.bBefore = 0; spS.dBefore = 0; spS.guess = 0; spS.active = 0; spS.u1 = 0.f; spS.u2 = 0.f; spS.go = 0; spS.ct = 0; spS.info = 0; spS.rng = 0; spS.pos = 0; spS.bin = 0; spS.r1 = 0; spS.c1 = 0;
        if (trySpec && attempt) {
            const float dpAtLo = gm_death_prob((double)(uint64_t)nLo, S.domainLenD, S.alphaD, S.numBins), dpAtHi = gm_death_prob((double)(uint64_t)nHi, S.domainLenD, S.alphaD, S.numBins);
            gen_spec_a1<WIN>(S, sh, rcS, nLo, nHi, dpAtLo, dpAtHi, spS);      // (the first of A1's two barriers inside)
        } else if (trySpec) cg_sync_lds();
        if (trySpec) { for (uint32_t i = t; i < 512u; i += cg_bdim()) sh.dirty[i] = 0u; }
        // the trip has landed: the attempt's seed and the table's window go to LDS; A1's second barrier
        if (dpStaged && !spare) {
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) { const uint32_t i = t + k * TPB; if (i < dpCnt) sh.dpWin[i] = dpw[k]; }
        }
        ChainItem it; chain_item_clear(it);
        if (trySpec) {
            if (attempt && spS.guess != (uint32_t)GEN_T_NONE) sh.seed[t] = seedC;      // consumed after the type sort
            cg_sync_lds();
            if (attempt) gen_spec_slot<WIN>(S, sh, rcS, spS);
        }
        const bool drawAhead = trySpec && cg_uniform_u32(sh.specBad) == 0u;      // (a window with an attempt between the two thresholds: classified and drawn the usual way, behind the decisions)
        if (have0) chain_fetch_build(S, p0, mid0, it, isSparse);
        if (helper && ht < 16u) sh.freeTop[ht] = freeTopAhead;
        GEN_TS(32);
        epoch0 = sh.g.batchEpoch;
        const uint32_t tag = (uint32_t)epoch0;      // the batch in the queue: the one this workgroup generated in the previous launch
        if (S.launchClock) clockEnd.slot = S.launchClock + 2u * (tag % GAPS_CLOCK_RING) + 1u;
#if defined(COGAPS_EMUL)
        if (t == 0 && e_prevQ) cg_atomic_add_u64(&gs->prof[13], 1ull);      // test-only build: batches whose decisions arrived inside a chained launch
#endif
        if (attempt) {
            // ---- the window drawn ahead: the domain's size taken as it is now (no atom erased: true of every second batch; otherwise the
            // picks are checked against the size the flush leaves, gen_draw_valid)
#if !defined(EXP_NO_AHEAD)
            if (drawAhead) {
                const uint32_t typeS = spS.info & 0xFFu;
                gen_draw_a<WIN, true, true>(S, rcS, &spS, spS.go != 0u, typeS, spS.info >> 8, spS.rng, e_n, drawKeep);
                gen_draw_b<WIN, true>(S, sh, rcS, typeS, drawKeep, [&]() {});
            }
#endif
            if (drawAhead) checkKeep = gen_draw_check(spS, drawKeep, e_n, rcS.K);
            GEN_PIN(drawKeep.flags); GEN_PIN(drawKeep.old1); GEN_PIN(drawKeep.old2);
            GEN_TS(36);
            GEN_RT(2);
            // Sparse model (round 6): the window drawn ahead is done ~8 us before the sparse evaluation's decisions arrive -- the attempt lanes take
            // the queue slots behind the applier lanes'.  A queue longer than the applier lanes (the batch behind a generator launch of two
            // rounds: 39 % of the launches at BASELINE configs[4]'s shard shape) used to take a second pass -- three dependent trips, the wait, the
            // stores -- behind the first one.
            if constexpr (ATTEMPTS_APPLY) {
#define HP_STRIDE passLanes
#define HP_LANE (NA + t)
#define HP_MINE true
#define HP_FETCH(b) true
#include "gen_handover_pass.h"
#undef HP_STRIDE
#undef HP_LANE
#undef HP_MINE
#undef HP_FETCH
            }
        } else {
            // ---- the appliers: one proposal per lane and pass -- wait for its two granules (read past this workgroup's caches until both
            // carry the batch's tag), note an erased atom in the erase cache, carry the decision out, note what changed
            {   // the conflict table of the round behind the join (keys and value words: one contiguous region), emptied here by the applier lanes:
                // their fetches are out, the decisions some microseconds away; the join's barrier orders these stores before the registrations
                GenTabKeys none; none.k[0] = none.k[1] = none.k[2] = none.k[3] = GEN_TAB_EMPTY;
                constexpr uint32_t UNITS = 5u * (uint32_t)GEN_TAB_NB;
                GenTabKeys *tab = reinterpret_cast<GenTabKeys *>(&sh.bkey[0]);
                const uint32_t nAll = cg_bdim() - (uint32_t)WIN;
#pragma unroll 4
                for (uint32_t i = t - (uint32_t)WIN; i < UNITS; i += nAll) tab[i] = none;
            }
#define HP_STRIDE passLanes
#define HP_LANE al
#define HP_MINE applying
#define HP_FETCH(b) (b)
#include "gen_handover_pass.h"
#undef HP_STRIDE
#undef HP_LANE
#undef HP_MINE
#undef HP_FETCH
        }
        GEN_TS(34);
        // The join: the decisions are issued to the domain, the erase cache, the notes and the unit sum are complete.  The window drawn
        // ahead: nobody reads the domain before the lanes have validated their draws, so the join waits for LDS traffic only and the
        // appliers' stores are acknowledged further down -- the spare waves stay for two more barriers (the validation's; then the join
        // of the lanes that draw again, or the registration barrier -- cg_sync: this wave's stores are acknowledged before either).
        // The usual way: the attempt lanes read the domain next, behind stores that are acknowledged here.
        if (drawAhead) cg_sync_lds(); else cg_sync();
        GEN_TS(35);
        GEN_RT(4);
        const uint32_t sfRaw = sh.spinFail, emRaw = sh.eraseN;      // (both words in one LDS trip)
        // A decision never arrived (GAPS_ERR_SPIN is set): every wave leaves, nothing is generated.  What the host's recovery needs is parked
        // (the batch's queue length, the erase cache's fill: its entries are in the list) and the launches already enqueued behind this one are
        // made harmless: they find an empty queue in both copies (nothing is evaluated, nothing applied a second time) and an update that
        // is over (nSteps = nDone: the generator only reports), until the host reads the error word.
        if (cg_uniform_u32(sfRaw) != 0u) {
            if (t == 0u) { gs->applyCount = e_prevQ; gs->savedErase = emRaw; gs->qlen = 0u; gs->nSteps = e_nDone; ChainSlot cs; cs.qlen = 0u; cs.tag = tag + 1u; *hot.slotWr = cs; }
            return;
        }
        if (spare) {                                        // (the waves beyond the helper wave only applied)
            if (drawAhead && !updateDone) { cg_sync_lds(); cg_sync(); }
            { const bool ts_ok = e_prevQ >= 140u && e_nSteps - e_nDone >= 512u; (void)ts_ok; GEN_TS_DUMP_WAVE(); }
            return;
        }
        e_m = cg_uniform_u32(emRaw);
        if (e_m > eraseCap) e_m = eraseCap;
        if (helper) specE = (ht < (unsigned)FLUSH_MAX && ht < e_m) ? sh.eraseTmp[ht] : 0ull;
        specDone = drawAhead;
#if defined(GEN_TIMELINE)
        if (t == 0u) sh.rtInfo = (unsigned long long)e_prevQ | ((unsigned long long)e_m << 16) | ((unsigned long long)(specDone ? 1u : 0u) << 32);
#endif
#if defined(COGAPS_EMUL)
        if (t == 0 && !updateDone) cg_atomic_add_u64(&gs->prof[specDone ? 12 : 11], 1ull);      // test-only build: windows classified and drawn ahead of the decisions / the usual way
#endif
        if (specDone) specKeep = spS;
    }
    if (updateDone) {
        // a launch past the end of the update: the last erase cache is flushed (by the helper wave alone) and the progress words reported
        if (!CHAIN) cg_sync_lds();              // (the unit sum is complete)
        if (!helper) return;                    // (attempt lanes; the chained launch's spare waves left behind the join)
        GenFlushRegs fr;
        gen_flush_fetch<WIN>(S, fr, ht, e_m, e_n, specE, e_fc);
        if (ht == 0) { sh.flushM = 0; sh.flushBase = e_fc; sh.nLow = 0; }
        cg_wave_sync();
        for (int part = 0; part < 4; ++part) { gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, part); cg_wave_sync(); }
        if (ht == 0) { gs->nAtoms = sh.g.nAtoms; gs->front = sh.g.front; gs->freeCount = sh.g.freeCount; gs->eraseCount = 0; gs->qlen = 0; gs->batchNproc = 0; gs->updateFlushed = 1;
                       gs->evalBytes = sh.g.evalBytes + (unsigned long long)sh.unitSum * S.unitBytes; gs->evalProps = sh.g.evalProps + e_prevQ;
                       if (CHAIN) { ChainSlot cs; cs.qlen = 0; cs.tag = (uint32_t)sh.g.batchEpoch; *hot.slotWr = cs; } }
        return;
    }
    if (helper) { gen_helper<WIN>(S, sh, gs, ht, specE, e_m, e_n, e_fc, e_prevQ, e_nDone, e_nSteps, CHAIN ? hot.slotWr : nullptr, CHAIN && specDone); return; }

    // ================================================================================ attempt lanes
    // second trip (addresses from the first): this round's seeds
    const uint64_t seed1 = CHAIN ? seedC : ((e_nDone + t < e_nSteps) ? S.seeds[e_nDone + t] : 0ull);
    const uint32_t n0 = e_n - e_m;                  // after the flush (which the helper wave runs meanwhile) the domain holds this many atoms
    // death probability (ProposalQueue::deathProb) for every atom count an attempt of this window can see: lane t takes the entries for
    // t births / t deaths ahead of it from the session's table (the same gm_death_prob, evaluated once per session).  They are needed
    // after the first count of the classification, and are parked in LDS just before it: the trip runs under the draws and the
    // first guess, which needs only the entry of the count itself (computed here: the load would be on the critical path).
    // (The chained launch staged the table's window in LDS while it waited for the decisions: no trip, no division.)
    const bool fromWin = CHAIN && dpStaged;
    // (a window classified ahead reads the staged table where it needs it: none of the three values below is used there -- and the
    // compiler would compute the division of the third whichever way the select goes)
    float tabHi = 0.f, tabLo = 0.f, dp0 = 0.f;
    if (!(CHAIN && specDone)) {
        tabHi = fromWin ? sh.dpWin[n0 + t - dpBase] : S.deathProb[n0 + t]; tabLo = (n0 >= t) ? (fromWin ? sh.dpWin[n0 - t - dpBase] : S.deathProb[n0 - t]) : 0.f;
        dp0 = fromWin ? sh.dpWin[n0 - dpBase] : gm_death_prob((double)(uint64_t)n0, S.domainLenD, S.alphaD, S.numBins);
    }
    const uint64_t batchEpoch = (CHAIN ? epoch0 : sh.g.batchEpoch) + 1;
    const uint32_t updBase = e_nDone;           // attempts consumed by earlier batches of this update
    const uint32_t remaining = e_nSteps - e_nDone;
    const uint32_t K = S.K;
    // round 1 takes its scalars from the LDS copy of GenScalars (complete since the first barrier); the helper wave writes the round
    // variables' LDS copies, which later phases and rounds read
    // (a window drawn ahead has consumed these already: nothing of them is read behind the decisions)
    uint64_t g_qrng = 0; uint32_t g_skip = 0; float g_u1 = 0.f, g_u2 = 0.f;
    if (!(CHAIN && specDone)) { g_qrng = sh.g.qrng; g_skip = sh.g.useCached ? 1u : 0u; g_u1 = sh.g.u1; g_u2 = sh.g.u2; }

    GenRoundCtx rc; rc.t = t; rc.jm0 = jm0; rc.ji0 = ji0; rc.jm1 = jm1; rc.ji1 = ji1; rc.seed1 = seed1; rc.batchEpoch = batchEpoch; rc.g_qrng = g_qrng; rc.n0 = n0; rc.updBase = updBase;
    rc.remaining = remaining; rc.K = K; rc.g_skip = g_skip; rc.e_prevQ = e_prevQ; rc.dp0 = dp0; rc.g_u1 = g_u1; rc.g_u2 = g_u2; rc.gs = gs; rc.tabHi = tabHi; rc.tabLo = tabLo;
    rc.queueOut = CHAIN ? hot.queueWr : S.queue; rc.dpBase = dpBase; rc.sparse = isSparse ? 1u : 0u;
    if (CHAIN && specDone) {
        // which lanes drew what they would draw now (gen_draw_valid).  If every lane of the window did, the round goes straight into its
        // conflict phases and the helper wave's flush runs beside them (gen_helper); otherwise the join with the flush first
        const uint32_t again = gen_draw_valid<WIN>(S, sh, specKeep, drawKeep, checkKeep, n0, e_m);
        const bool valid = again == 0u;
        {   // the window's level: 2 if some lane waits for the flush, else 1 if some lane draws again at all
            const uint32_t lv = cg_ballot(again == 2u) != 0ull ? 2u : (cg_ballot(again == 1u) != 0ull ? 1u : 0u);
            if (lv != 0u && (t & 63u) == 0u) cg_atomic_max_u32(&sh.anyRedo, lv);
        }
        GEN_PIN(drawKeep.flags);
        GEN_TS(37);
#if defined(GEN_TIMELINE)
        { const unsigned long long bad_ = cg_ballot(!valid); if ((t & 63u) == 0u && bad_) cg_atomic_add_u64(&sh.rt[6], (unsigned long long)cg_popc64(bad_)); }
#endif
        cg_sync_lds();
        // (level 1: the barrier only acknowledges the appliers' stores -- the lanes that draw again keep their picks and read records and
        // matrix cells the flush leaves alone, which runs beside them as it does when no lane draws again; level 2: the join with the flush)
        const uint32_t redoLevel = cg_uniform_u32(sh.anyRedo);
        if (redoLevel != 0u) cg_sync();
        GEN_TS(38);
#if defined(GEN_TIMELINE)
        if (t == 0u) { sh.rt[7] = __builtin_amdgcn_s_memrealtime(); sh.rtInfo |= ((sh.rt[6] & 0xFFull) << 40) | (((sh.rt[7] - sh.rt[4]) & 0xFFFFull) << 48) | ((unsigned long long)(redoLevel & 3u) << 36); }
#endif
        if (gen_round<WIN, true, true, true>(S, sh, rc, 1u, &specKeep, &drawKeep, valid, redoLevel == 1u)) return;
    }
    else if (gen_round<WIN, true>(S, sh, rc, 1u)) return;
    for (uint32_t roundNo = 2; ; ++roundNo) {
        // ------------------------------------------------------------------ set-up of the next round of this batch (the helper wave has published
        // sh.nR / sh.minAtoms and resets the masks and the stop key between the two barriers)
        cg_sync_lds();
        {
            const uint32_t nn = sh.nR, m0 = sh.minAtoms;
            sh.dpHi[t] = gm_death_prob((double)((uint64_t)nn + t), S.domainLenD, S.alphaD, S.numBins);
            sh.dpLo[t] = (m0 >= t) ? gm_death_prob((double)(uint64_t)(m0 - t), S.domainLenD, S.alphaD, S.numBins) : 0.f;
        }
        cg_sync();
        if (gen_round<WIN, false>(S, sh, rc, roundNo)) return;
    }
}

// (the generator's LDS as the kernel's own static block; the chained launch of the sparse model places it in a block it shares with the
// evaluation workgroups' -- a launch's workgroups all carry the kernel's static LDS, whichever role they play: chain_kernel.h)
template <int WIN, bool ASYNC, bool CHAIN = false, int SP = -1>
CG_DEVICE void gen_body(const SamplerDev CG_CONSTANT *sp, const GenHot hot)
{
    CG_SHARED GenShared<WIN> sh;
    gen_body_sh<WIN, ASYNC, CHAIN, SP>(sp, hot, sh);
}
// WIN attempt lanes + the helper wave.  The launch's first loads need only the leading scalar arguments (preloaded into SGPRs); the
// sampler's record is read from device memory through `sp`
template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(WIN + 64) gen_kernel(const uint64_t *lcgMul, const uint64_t *lcgInc, GenScalars *gs, const unsigned long long *eraseList, const uint32_t *queueUnits,
                                                    uint32_t eraseCap, uint32_t queueCap, const SamplerDev CG_CONSTANT *sp)
{
    GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = eraseList; hot.queueUnits = queueUnits; hot.eraseCap = eraseCap; hot.queueCap = queueCap;
    hot.queueRd = nullptr; hot.queueWr = nullptr; hot.grans = nullptr; hot.slotWr = nullptr;
    gen_body<WIN, true>(sp, hot);
}
// batched multi-chain launch (eval_kernel.h): one workgroup per chain
template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(WIN + 64) gen_kernel_multi(const SamplerDev CG_CONSTANT *arr)
{
    const SamplerDev CG_CONSTANT *sp = arr + cg_bid();
    cg_const_warm<sizeof(SamplerDev)>(sp);
    const SamplerDev &S = *(const SamplerDev *)sp;
    GenHot hot; hot.lcgMul = S.lcgMul; hot.lcgInc = S.lcgInc; hot.gs = S.gs; hot.eraseList = S.eraseList; hot.queueUnits = S.queueUnits; hot.eraseCap = S.eraseCap; hot.queueCap = S.queueCap;
    hot.queueRd = nullptr; hot.queueWr = nullptr; hot.grans = nullptr; hot.slotWr = nullptr;
    gen_body<WIN, false>(sp, hot);
}
