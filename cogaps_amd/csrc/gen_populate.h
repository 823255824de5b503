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

// ---- flushEraseCache (ConcurrentAtomicDomain.cpp:71-79 + erase :109-124), by the helper wave --------------------------
// The reference sorts the erase cache by position and erases one atom after the other.  Here: rank sort in
// LDS; the list surgery and the bin-head index are done by one lane per erased atom (after the sort, an
// erased neighbour of erased atom k can only be k-1 / k+1, so runs of adjacent erased atoms are walked in
// LDS); the swap-with-last sequence on the unsorted vector -- order dependent -- is replayed by one lane on
// indices held in LDS (no memory traffic), and only its net effect (<= m slots) is written back.
// ht: helper lane 0..63.  The steps are separate functions because the caller interleaves them with the barriers it owes the
// attempt waves; inside the one wave a step sees the previous step's LDS writes after cg_wave_sync().
struct GenFlushRegs { uint32_t myH, myBin, myHead; AtomRec rec; uint32_t vtail, freeTop; };
// step 1: request the erased atoms' records, their bins' heads and the tail of the unsorted vector (no wait)
template <int WIN>
CG_DEVICE void gen_flush_fetch(const SamplerDev &S, GenFlushRegs &f, const unsigned ht, const uint32_t m, const uint32_t n, const unsigned long long specE, const uint32_t fc, const bool haveFreeTop = false)
{
    // (the stack's top sixteen entries ride along: half of the launches commit a birth that pops below what the flush pushed, and its
    // wave -- the launch's last phase -- waited a memory trip for the handle)
    f.freeTop = (!haveFreeTop && ht < 16u && ht < fc) ? S.freeHandles[fc - 1u - ht] : CG_NONE;      // (haveFreeTop: the chained launch parked them in sh.freeTop ahead of the decisions)
    f.myH = 0; f.myBin = 0; f.myHead = CG_NONE; f.vtail = CG_NONE;
    f.rec.pos = 0; f.rec.lpos = 0; f.rec.rpos = 0; f.rec.left = CG_NONE; f.rec.right = CG_NONE; f.rec.mass = 0.f; f.rec.rmass = 0.f; f.rec.idx = 0; f.rec.pad0 = 0;
    // (the bin travels with the handle in the erase cache: the bin's head is asked for in the same trip as the record)
    if (m <= (uint32_t)FLUSH_MAX && ht < m) { f.myH = (uint32_t)specE; f.myBin = (uint32_t)(specE >> 32); f.rec = S.atoms[f.myH]; f.myHead = S.binHead[f.myBin]; f.vtail = S.vec[n - m + ht]; }
}
// steps 2-5.  part 0: sort (waits for the records); part 1: list surgery + bin heads; part 2: index replay; part 3: write-back
template <int WIN>
CG_DEVICE void gen_flush_part(const SamplerDev &S, GenShared<WIN> &sh, const GenFlushRegs &f, const unsigned ht, const uint32_t m, const uint32_t n, const uint32_t fc0, const int part)
{
    GenScalars &g = sh.g;
    if (m == 0) return;                      // uniform across the wave
    if (m > (uint32_t)FLUSH_MAX) {           // rare: serial fallback, exactly the reference's procedure (one lane, in the last part)
        if (part == 3 && ht == 0) {
            for (uint32_t i = 1; i < m; ++i) {
                const unsigned long long e = S.eraseList[i]; uint64_t p = S.atoms[(uint32_t)e].pos; uint32_t j = i;
                while (j > 0 && S.atoms[(uint32_t)S.eraseList[j - 1]].pos > p) { S.eraseList[j] = S.eraseList[j - 1]; --j; }
                S.eraseList[j] = e;
            }
            uint32_t nn = n, fc = g.freeCount, fr = g.front;
            for (uint32_t i = 0; i < m; ++i) gen_erase_one(S, (uint32_t)S.eraseList[i], nn, fc, fr);
            g.nAtoms = nn; g.freeCount = fc; g.front = fr; g.eraseCount = 0;
        }
        return;
    }
    if (part == 0) {
        if (ht < m) { sh.fpos[ht] = f.rec.pos; sh.vt[ht] = f.vtail; }
        cg_wave_sync();
        // rank sort by position (positions are unique)
        if (ht < m) {
            uint32_t r = 0;
            for (uint32_t j = 0; j < m; ++j) r += (sh.fpos[j] < f.rec.pos) ? 1u : 0u;
            sh.fh[r] = f.myH; sh.fl[r] = f.rec.left; sh.fr[r] = f.rec.right; sh.fidx[r] = f.rec.idx; sh.fbin[r] = f.myBin; sh.fhead[r] = f.myHead;
            sh.flpos[r] = f.rec.lpos; sh.frpos[r] = f.rec.rpos; sh.frmass[r] = f.rec.rmass;
        }
        cg_wave_sync();
        return;
    }
    if (part == 1) {
        // list surgery + bin heads (reads the pre-flush links only)
        if (ht < m) {
            const uint32_t k = ht, h = sh.fh[k];
            const bool leftErased = (k > 0) && (sh.fh[k - 1] == sh.fl[k]);
            if (!leftErased) {                    // head of a run of adjacent erased atoms
                uint32_t j = k;
                while (j + 1 < m && sh.fh[j + 1] == sh.fr[j]) ++j;
                const uint32_t L = sh.fl[k], R = sh.fr[j];
                // (the run's survivors take over each other's cached position / mass: the first erased atom knows L's, the last R's)
                if (L != CG_NONE) { S.atoms[L].right = R; S.atoms[L].rpos = sh.frpos[j]; S.atoms[L].rmass = sh.frmass[j]; } else sh.newFront = R;
                if (R != CG_NONE) { S.atoms[R].left = L; S.atoms[R].lpos = sh.flpos[k]; }
            }
            const uint32_t b = sh.fbin[k];
            if (sh.fhead[k] == h) {               // the lowest atom of its bin goes: the next surviving atom of the bin takes over
                uint32_t j = k;
                while (j + 1 < m && sh.fh[j + 1] == sh.fr[j]) ++j;
                const uint32_t cand = sh.fr[j];
                if (cand != CG_NONE && gen_bin_of(S, sh.frpos[j]) == b) S.binHead[b] = cand;      // (the survivor's position is cached in the run's last record: no trip)
                else { S.binHead[b] = CG_NONE; bm_clear(S, b); }
            }
            S.freeHandles[fc0 + k] = h;           // pushed in erase order
        }
        return;
    }
    if (part == 2) {
        // swap-with-last replay on indices (mAtoms[idx] = mAtoms.back(); pop_back), one lane, LDS only
        if (ht == 0) {
            uint32_t curN = n, nl = 0;
            const uint32_t base = n - m;
            for (uint32_t k = 0; k < m; ++k) {
                const uint32_t i = sh.fidx[k];
                const uint32_t hl = sh.vt[curN - 1u - base];           // occupant of the last slot
                if (i >= base) sh.vt[i - base] = hl;
                else {
                    uint32_t e = 0; while (e < nl && sh.lowSlot[e] != i) ++e;
                    sh.lowSlot[e] = i; sh.lowH[e] = hl; if (e == nl) ++nl;
                }
                for (uint32_t q = k + 1; q < m; ++q) if (sh.fh[q] == hl) sh.fidx[q] = i;   // a later victim was moved
                --curN;
            }
            sh.nLow = nl; sh.flushM = m;
            g.nAtoms = n - m; g.freeCount += m; g.eraseCount = 0;
        }
        cg_wave_sync();
        return;
    }
    if (ht < sh.nLow) { const uint32_t slot = sh.lowSlot[ht], h = sh.lowH[ht]; S.vec[slot] = h; S.atoms[h].idx = slot; }
    if (ht == 0 && sh.newFront != CG_KEEP) { g.front = sh.newFront; }
}

// hot: what the launch's first memory trip reads, passed as leading scalar kernel arguments so that the dispatcher preloads them into
// SGPRs (-amdgpu-kernarg-preload-count): the trip starts at once and the by-value SamplerDev's kernel-argument lines (WARM
// bytes; 0 = the caller warmed them) come in under it instead of before it.
struct GenHot { const uint64_t *lcgMul, *lcgInc; GenScalars *gs; const unsigned long long *eraseList; const uint32_t *queueUnits; uint32_t eraseCap, queueCap;
                // chained launch only: the queue copy the previous batch sits in, the copy and slot this launch writes, the decision granules
                const PropRec *queueRd; PropRec *queueWr; const unsigned long long *grans; ChainSlot *slotWr; };

// ---- the helper wave: flush, table presets, round bookkeeping, write-back.  Mirrors the attempt waves' barriers one for one. ----
template <int WIN>
CG_DEVICE void gen_helper(const SamplerDev &S, GenShared<WIN> &sh, GenScalars *gs, const unsigned ht, const unsigned long long specE,
                          const uint32_t e_m, const uint32_t e_n, const uint32_t e_fc, const uint32_t e_prevQ, const uint32_t e_nDone, const uint32_t e_nSteps, ChainSlot *slotWr, const bool specDone = false)
{
    const unsigned t = (unsigned)WIN + ht;
    GEN_TS_INIT(); GEN_TS_RESUME(13);      // (marks 0, 0, 26-29, 1 and the chained launch's 30-35 were left by gen_body)
    GenFlushRegs fr;
    gen_flush_fetch<WIN>(S, fr, ht, e_m, e_n, specE, e_fc, specDone);          // the flush's one memory trip: under the attempt waves' A1
    const uint32_t n0 = e_n - e_m;                              // the domain holds this many atoms after the flush
    const uint64_t batchEpoch = sh.g.batchEpoch + 1;
    const uint32_t remaining = e_nSteps - e_nDone;
    if (ht == 0) {
        // the round scalars of round 1 (the attempt lanes derive the same values in registers and read these copies only later)
        sh.batchEpoch = batchEpoch; sh.roundNo = 1; sh.stopKey = 0xFFFFFFFFu; sh.frontPending = 0;
        sh.qrngRound = sh.g.qrng; sh.nR = n0; sh.minAtoms = n0; sh.processed = 0; sh.qlen = 0; sh.skip = sh.g.useCached ? 1u : 0u;
        sh.remaining = remaining; sh.u1c = sh.g.u1; sh.u2c = sh.g.u2; sh.updBase = e_nDone;
        sh.flushM = 0; sh.flushBase = e_fc; sh.nLow = 0;
    }
    if (ht < (unsigned)(WIN / 64)) { sh.mq[ht] = 0ull; sh.mb[ht] = 0ull; sh.md[ht] = 0ull; }
    GEN_TS(2);
    for (uint32_t roundNo = 1; ; ++roundNo) {
        const bool first = roundNo == 1u;
        const bool ldsRound = roundNo <= (uint32_t)GEN_LDS_ROUNDS;      // (as the attempt lanes decide it, gen_round)
        // ---- A1's two barriers (the classification's one count exchange, then the sorted slots); round 1: the flush goes on between them
        // (the sort -- it waits for the records -- while the attempt waves draw and guess; the list surgery and the index replay during
        // the type sort; the write-back during the first stage of A2)
        // (specDone: the chained launch classified this window before the decisions arrived and executed A1's two barriers then -- the
        // flush runs straight through to the join)
        if (first && specDone) {
            // Chained launch, the window drawn ahead of the decisions (gen_body): the attempt lanes have validated their draws when they
            // arrive at the first barrier.  No lane draws again (every second launch): the domain is not read before the commit, and the
            // flush runs BESIDE the conflict phases -- it is complete, its stores acknowledged (cg_sync waits for this wave's), at the
            // look-up barrier, behind which the attempt lanes read its LDS results for the commit.
            // Some lane draws again: it reads the domain as the flush leaves it -- the whole flush, then the join, as in the other forms.
            cg_sync_lds();
            const uint32_t redoLevel = cg_uniform_u32(sh.anyRedo);
            const bool beside = redoLevel != 2u;      // the flush runs beside the attempt lanes' phases (nobody waits for it before the look-up barrier)
            if (redoLevel == 1u) cg_sync();           // (lanes draw again, keeping their picks: this wave's own applied decisions are acknowledged first)
            gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 0);
            gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 1); gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 2);
            if (beside) cg_sync_lds();        // (the registration barrier, which the attempt lanes reach about now)
            gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 3); GEN_TS(3);
            cg_sync();                        // (beside: the look-up barrier; otherwise the join)
            if (!beside) { cg_sync_lds(); cg_sync_lds(); }
        } else {
            if (first) { gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 0); if (ht < 16u) sh.freeTop[ht] = fr.freeTop; }
            cg_sync_lds();
            if (first) { gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 1); gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 2); }
            cg_sync_lds();
            if (first) { gen_flush_part<WIN>(S, sh, fr, ht, e_m, e_n, e_fc, 3); GEN_TS(3); cg_sync(); }      // the join: the flush's stores are acknowledged (vmcnt(0)) before any lane reads the domain
            // ---- B1 / B2 barriers
            if (ldsRound) cg_sync_lds(); else cg_sync();
            if (ldsRound) cg_sync_lds(); else cg_sync();
        }
        // ---- C: masks complete behind this barrier; the attempt lanes commit, this wave keeps the books
        const uint32_t nR = sh.nR, minR = sh.minAtoms, skip = sh.skip, processed = sh.processed;
        const uint32_t left_ = remaining - processed;
        const uint32_t winN = left_ < (uint32_t)WIN ? left_ : (uint32_t)WIN;
        cg_sync_lds();
        GEN_TS(20);
        const uint32_t stopKey = sh.stopKey;
        const uint32_t stopT = (stopKey == 0xFFFFFFFFu) ? winN : (stopKey >> 1);
        const bool stopFail = (stopKey != 0xFFFFFFFFu) && (stopKey & 1u);
        const bool endB = stopFail || (processed + stopT >= remaining);
        const bool frontPending = sh.frontPending != 0u;
        if (ht == 0) {
            uint32_t totQ = 0, totB = 0, totD = 0;
            for (uint32_t w = 0; w < (uint32_t)(WIN / 64); ++w) { totQ += (uint32_t)cg_popc64(sh.mq[w]); totB += (uint32_t)cg_popc64(sh.mb[w]); totD += (uint32_t)cg_popc64(sh.md[w]); }
            if (totB) { const uint32_t fc = sh.g.freeCount; if (totB <= fc) sh.g.freeCount = fc - totB; else { sh.g.freeCount = 0; sh.g.handleHi += totB - fc; } sh.g.nAtoms = nR + totB; }
            sh.nR = nR + totB; sh.minAtoms = minR - totD;
            const uint32_t qlen = sh.qlen + totQ;
            sh.qlen = qlen; sh.processed = processed + stopT;
            const uint32_t attempted = stopT + (stopFail ? 1u : 0u);
            const uint32_t draws = 2u * (attempted - ((skip && attempted) ? 1u : 0u));
            const uint64_t jm = sh.jmul[draws >> 1], ji = sh.jinc[draws >> 1];
            const uint64_t qr = jm * sh.qrngRound + ji;
            sh.qrngRound = qr;
            if (attempted) sh.skip = 0;
            if (endB) {
                // final values of the scalars the generator owns, in the LDS copy; the lanes of this wave write it back below
                GenScalars &g = sh.g;
                g.qrng = qr;
                if (stopFail) { g.useCached = 1; g.u1 = sh.u1[stopT]; g.u2 = sh.u2[stopT]; }
                else g.useCached = 0;
                const uint32_t nDone = e_nDone + processed + stopT;
                g.nDone = nDone;
                g.qlen = qlen; g.batchNproc = processed + stopT;
                g.batchEpoch = batchEpoch; g.eraseCount = 0;
                if (nDone < g.nSteps) {           // n < nSteps: AsynchronousGibbsSampler.h:97-102
                    const float ns = g.nQueueSamples + 1.f;
                    float avg = g.avgQueue;
                    avg *= (ns - 1.f) / ns;
                    avg += (float)qlen / ns;
                    g.nQueueSamples = ns; g.avgQueue = avg;
                }
                if (g.traceOn) {
                    const uint32_t bi = g.traceBatchCount;
                    if (bi < g.traceCap) { S.traceBatchNproc[bi] = processed + stopT; S.traceBatchQlen[bi] = qlen; }
                    g.traceBatchCount = bi + 1; g.traceCount += qlen;
                }
                g.nBatches += 1;
                g.evalBytes = g.evalBytes + (unsigned long long)sh.unitSum * S.unitBytes; g.evalProps = g.evalProps + e_prevQ;
            }
        }
        cg_wave_sync();
        GEN_TS(23);
        if (endB) {
            // write back the leading words of GenScalars (everything the generator owns) one lane per word; the sticky error word is
            // only ever written in place, and a new front atom's handle is written by the birth that made it (it may still be on its way
            // into the LDS copy)
            const uint32_t frontWord = (uint32_t)(offsetof(GenScalars, front) / 4u);
            for (uint32_t w = ht; w < GEN_GS_WORDS; w += 64u)
                if (w != GEN_GS_ERROR_WORD && !(frontPending && w == frontWord)) reinterpret_cast<uint32_t *>(gs)[w] = reinterpret_cast<const uint32_t *>(&sh.g)[w];
            // chained launch: what the next launch's evaluation workgroups start from (the queue copy they read was filled by this launch's commit)
            if (slotWr && ht == 0) { ChainSlot cs; cs.qlen = sh.g.qlen; cs.tag = (uint32_t)batchEpoch; *slotWr = cs; }
            GEN_TS(24);
            { const bool ts_ok = e_prevQ >= 140u && remaining >= 512u && GEN_TS_ROUND_OK(roundNo); (void)ts_ok; GEN_TS_DUMP_WAVE(); }
            return;
        }
        // ---- another round of this batch: its set-up once every lane is done with this round's masks
        cg_sync_lds();
        if (ht == 0) { sh.roundNo = roundNo + 1u; sh.stopKey = 0xFFFFFFFFu; sh.frontPending = 0; if (roundNo + 1u >= 4094u) gs->error = GAPS_ERR_SPIN; }
        if (ht < (unsigned)(WIN / 64)) { sh.mq[ht] = 0ull; sh.mb[ht] = 0ull; sh.md[ht] = 0ull; }
        cg_sync();
    }
}

// One round of a batch for the attempt lanes: the window's attempts are classified (A1), drawn (A2), checked against each other (B1, B2)
// and committed up to the first one that ends the batch or has to be redrawn (C).  Returns whether the batch ends with this round.
// FIRST: round 1, compiled as its own straight-line copy.  92 % of all launches are one round long; as the body of a loop the round
// had every loop-invariant of its rare paths hoisted in front of it by the compiler -- the reciprocal of a 64-bit division that only
// tiny domains perform, 1 / lambda and the glibc exponential's table for same-bin exchanges, four hundred instructions before the
// first attempt was looked at, and the wait for the seeds at the loop's head -- and a dozen scalar registers spilled to carry them.
struct GenRoundCtx {
    unsigned t; uint64_t jm0, ji0, jm1, ji1, seed1, batchEpoch, g_qrng; uint32_t n0, updBase, remaining, K, g_skip, e_prevQ; float dp0, g_u1, g_u2; GenScalars *gs;
    float tabHi, tabLo;      // round 1: this lane's entries of the window's death-probability rows, on their way from SamplerDev::deathProb
    PropRec *queueOut;       // where the batch's queue records go (S.queue; the chained launch: the copy of the other parity)
    uint32_t dpBase;         // chained launch: first entry of the death-probability table's window in sh.dpWin
    uint32_t sparse;         // the model (SamplerDev::sparse) -- a compile-time constant where the kernel serves one model only (gen_body_sh<.., SP>)
};
// The chained launch classifies and sorts its first window BEFORE the previous batch's decisions are in (gen_spec_a1, while the
// evaluation workgroups of the same launch run): what the lane keeps of that in registers.  First half: lane = attempt; second half:
// lane = sorted slot.
struct GenSpec {
    uint32_t bBefore, dBefore, guess, active; float u1, u2;
    uint32_t go, ct, info; uint64_t rng, pos; uint32_t bin, r1, c1;
};
// A1 of round 1 without the domain's size (chained launch).  The type of an attempt depends on the atom count n only through the
// birth / death threshold deathProb(n) (ProposalQueue.cpp:129-160), which is monotone in n and moves by ~1e-9 per atom; the count
// after the flush lies in [nLo, nHi] = [nAtoms - queue length, nAtoms] (a proposal erases at most one atom).  The lanes classify with
// both ends' thresholds: where every attempt gets the same type from both -- practically always -- that is its type for the true
// count too, and the count exchange, the sorted slots, the attempt's generator state and a birth's position follow without it.  A
// window with an attempt between the two thresholds (sh.specBad) is classified again the usual way once the count is known.
template <int WIN>
CG_DEVICE void gen_spec_a1(const SamplerDev &S, GenShared<WIN> &sh, const GenRoundCtx &c, const uint32_t nLo, const uint32_t nHi, const float dpAtLo, const float dpAtHi, GenSpec &sp)
{
    const unsigned t = c.t;
    const uint32_t winN = c.remaining < (uint32_t)WIN ? c.remaining : (uint32_t)WIN;
    const uint32_t active = t < winN;
    uint64_t s = (c.g_skip ? c.jm1 : c.jm0) * c.g_qrng + (c.g_skip ? c.ji1 : c.ji0);
    float u1 = pcg_uniform(s), u2 = pcg_uniform(s);
    const uint32_t cached = (c.g_skip != 0u) & (uint32_t)(t == 0u);       // attempt 0 replays the cached pair
    u1 = cached ? c.g_u1 : u1; u2 = cached ? c.g_u2 : u2;
    const uint32_t gLo = gen_decide(u1, u2, nLo, nLo, dpAtLo, dpAtLo), gHi = gen_decide(u1, u2, nHi, nHi, dpAtHi, dpAtHi);
    if (cg_ballot(active && gLo != gHi) != 0ull && (t & 63u) == 0u) sh.specBad = 1u;
#if defined(GEN_SPEC_BAD_EVERY)
    if (t == 0u && (sh.g.batchEpoch % (uint64_t)GEN_SPEC_BAD_EVERY) == 0ull) sh.specBad = 1u;      // test-only variant: the fall-back path, regularly
#endif
    const uint32_t guess = active ? gHi : (uint32_t)GEN_T_NONE;
    sh.u1[t] = u1; sh.u2[t] = u2;
    uint32_t eX[4], tX[4];
    gen_count4<WIN>(sh.wtot4, t, guess == 'B', guess == 'D', guess == 'M', guess == 'E', eX, tX);
    sp.bBefore = eX[0]; sp.dBefore = eX[1]; sp.u1 = u1; sp.u2 = u2; sp.guess = guess; sp.active = active;
    const uint32_t goA = (uint32_t)(guess != GEN_T_NONE);
    const uint32_t k0 = (uint32_t)(guess == 'B') | (uint32_t)(guess == 'D'), k1 = (uint32_t)(guess == 'M');
    const uint32_t T0 = tX[0] + tX[1], T1 = tX[2], T2 = tX[3];
    if (goA) {
        uint32_t slot = T0 + T1 + eX[3];
        slot = k1 ? T0 + eX[2] : slot;
        slot = k0 ? eX[0] + eX[1] : slot;
        sh.perm[slot] = (uint16_t)t;
        sh.info[t] = guess | (eX[0] << 8);
    }
    if (t == 0) { sh.nWork = T0 + T1 + T2; sh.nBD = T0; }
    // (the caller parks the attempt's seed in sh.seed[t] once the trip that brings it has landed, then closes with the second barrier)
}
// ... second half (lane = sorted slot): the attempt's generator state, a birth's position
template <int WIN>
CG_DEVICE void gen_spec_slot(const SamplerDev &S, GenShared<WIN> &sh, const GenRoundCtx &c, GenSpec &sp)
{
    const unsigned t = c.t;
    const bool go = t < sh.nWork;
    sp.go = go ? 1u : 0u;
    sp.ct = go ? (uint32_t)sh.perm[t] : 0u;
    sp.info = go ? sh.info[sp.ct] : 0u;
    sp.rng = go ? pcg_from_seed(sh.seed[sp.ct]) : 0ull;
    sp.pos = 0; sp.bin = 0; sp.r1 = 0; sp.c1 = 0;
    if (go && (sp.info & 0xFFu) == 'B') {
        uint64_t x = pcg_u64(sp.rng);
        while (x >= S.limitL) x = pcg_u64(sp.rng);
        sp.pos = (S.iPartL == 1ull ? x : gm_udiv64(x, S.iPartL)) + 1ull;      // (exact; the compiler's 64-bit division is a ~130-instruction routine the whole wave waits for)
        sp.bin = gen_bin_of(S, sp.pos); sp.r1 = gen_div_k(S, sp.bin); sp.c1 = sp.bin - sp.r1 * c.K;
    }
}

// one bit per level-0 bitmap word (mod 16384) that a decision being applied, or the flush of an atom it erases, changes: a birth drawn
// ahead of the decisions checks the words it read (gen_draw_valid)
CG_DEVICE void gen_mark_dirty(uint32_t *dirty, uint32_t bin)
{
    const uint32_t w = (bin >> 6) & 16383u;
    cg_atomic_or_u32(&dirty[w >> 5], 1u << (w & 31u));
}

// ---- notes of what the previous batch's decisions change (chained launch) ------------------------------------------------------------
// Two bit sets in LDS -- atom records (by handle; the vector slots the flush refills share it under complemented keys) and matrix cells
// (by bin) --, two hash positions per key: the lanes that apply the decisions set bits with non-returning LDS atomics (nothing to wait
// for; an exact hash set's compare-and-swap chains cost the applying waves 3 k cycles per launch), the lanes that drew the next window
// ahead read their keys' bits behind the join.  A key that was never noted reads as noted with probability ~2e-5 (two of ~600 set bits
// among 131072): the lane then draws again, which is always correct.
// (a key's two bit numbers are computed where the key is known -- ahead of the decisions, on both sides --, so that behind the wait only
// the LDS operations themselves remain)
struct GenNotePos { uint32_t a, b; };
template <int WORDS>
CG_DEVICE GenNotePos gen_note_pos(uint32_t key)
{
    constexpr uint32_t LOG2 = WORDS == 4096 ? 17u : (WORDS == 2048 ? 16u : 13u);
    static_assert(WORDS == 4096 || WORDS == 2048 || WORDS == 256, "bit numbers are 17 / 16 / 13 bits of the hash");
    const uint32_t h = key * 2654435761u;
    GenNotePos p; p.a = h >> (32u - LOG2); p.b = (h ^ (h >> 11)) & ((1u << LOG2) - 1u);
    return p;
}
CG_DEVICE void gen_note_set(uint32_t *bits, const GenNotePos p)
{
    cg_atomic_or_u32(&bits[p.a >> 5], 1u << (p.a & 31u));
    cg_atomic_or_u32(&bits[p.b >> 5], 1u << (p.b & 31u));
}
CG_DEVICE uint32_t gen_note_get(const uint32_t *bits, const GenNotePos p)
{
    return (bits[p.a >> 5] >> (p.a & 31u)) & (bits[p.b >> 5] >> (p.b & 31u)) & 1u;
}

// ---- the populate-phase draws of one attempt (ProposalQueue.cpp:162-283: birth / death / move / exchange up to the conflict rules) ------
// What an attempt's lane knows once it has drawn: the proposal as it will be queued, the atoms and matrix entries it read, and -- for the
// chained launch, which draws a window AHEAD of the previous batch's decisions and must know which lanes to draw again -- what it read
// them from.
struct GenDraw {
    uint32_t go, flags; bool isB, pick;
    uint64_t rng, rngPick, pos, cpos;          // rngPick: the lane's generator behind the pick of its atom (uniform32 over the domain's size)
    uint32_t h1, h2, i1, hl, hr, r1, c1, r2, c2, bin;
    float nm1, nm2, amass, m2x, old1, old2; uint32_t gib1, gib2;
    uint64_t lposB, rposB; float rmassB;
    // drawn ahead only: the successor bin a birth found and the atom at its head; `redo`: the lane took (or would have taken) one of the
    // rare long ways -- the full gap search, a walk along a bin, front() as an exchange partner -- and draws again behind the decisions
    uint32_t headBin, v2, v3, xPick; bool redo;      // v3: the one further record a birth read along its bin; xPick: the 32 random bits the pick was made from
};
CG_DEVICE void gen_draw_clear(GenDraw &d)
{
    d.go = 0; d.flags = 0; d.isB = false; d.pick = false; d.rng = 0; d.rngPick = 0; d.pos = 0; d.cpos = 0;
    d.h1 = CG_NONE; d.h2 = CG_NONE; d.i1 = CG_NONE; d.hl = CG_NONE; d.hr = CG_NONE; d.r1 = 0; d.c1 = 0; d.r2 = 0; d.c2 = 0; d.bin = 0;
    d.nm1 = 0.f; d.nm2 = 0.f; d.amass = 0.f; d.m2x = 0.f; d.old1 = 0.f; d.old2 = 0.f; d.gib1 = 0; d.gib2 = 0; d.lposB = 0; d.rposB = 0; d.rmassB = 0.f;
    d.headBin = 0; d.v2 = CG_NONE; d.v3 = CG_NONE; d.xPick = 0; d.redo = false;
}
// first part: what needs only the window's scalars -- a birth's position and bin (SPEC: drawn with the classification, gen_spec_slot),
// a pick's index into the unsorted vector.  nR: the domain's size at the start of the round; an attempt sees nR + (births before it).
template <int WIN, bool SPEC, bool AHEAD = false>
CG_DEVICE void gen_draw_a(const SamplerDev &S, const GenRoundCtx &c, const GenSpec *spec, const bool go, const uint32_t type, const uint32_t bBefore, const uint64_t rng0, const uint32_t nR, GenDraw &d)
{
    gen_draw_clear(d);
    d.go = go ? 1u : 0u;
    d.isB = go && type == 'B';
    d.pick = go && type != 'B';                 // D/M/E: picks an existing atom
    d.rng = rng0;                                  // AtomicProposal ctor, ProposalQueue.cpp:12-15
    const uint32_t nT = nR + bBefore;              // domain size this attempt sees
    if (d.isB) {
        if (SPEC) { d.pos = spec->pos; d.bin = spec->bin; d.r1 = spec->r1; d.c1 = spec->c1; }      // (drawn ahead: gen_spec_slot)
        else {
            // uniform64(1, L) (Random.cpp:105-123) with the constant range's iPart precomputed
            uint64_t x = pcg_u64(d.rng);
            while (x >= S.limitL) x = pcg_u64(d.rng);
            d.pos = (S.iPartL == 1ull ? x : gm_udiv64(x, S.iPartL)) + 1ull;
            d.bin = gen_bin_of(S, d.pos); d.r1 = gen_div_k(S, d.bin); d.c1 = d.bin - d.r1 * c.K;
        }
        d.i1 = nT;
    } else if (d.pick) {
        if (AHEAD) {
            // (drawn ahead: the pick is checked later against the size the flush leaves -- gen_draw_valid -- from the 32 bits it was made from;
            // a pick that needed a second draw, one in ten thousand, is simply drawn again)
            uint64_t r2 = d.rng; d.xPick = pcg_u32(r2);
            if (d.xPick >= nT * (0xFFFFFFFFu / nT)) d.redo = true;
        }
        d.i1 = pcg_uniform32(d.rng, 0u, nT - 1u);
        if (d.i1 >= nR) { d.flags |= GEN_F_FAIL; d.pick = false; }   // an atom born earlier in this window: its row is in use
    }
    d.rngPick = d.rng;
}
// second part: the staged dependent loads (B: bitmap word -> bin head -> atom; D/M/E: vec -> atom record, which carries the neighbours'
// positions and the right neighbour's mass -> matrix entries) and what follows from them.  underTrip(): the caller's work for the
// first trip's shadow.  AHEAD: drawn before the previous batch's decisions are in -- the long ways are not taken, the lane is marked.
template <int WIN, bool AHEAD, class F>
CG_DEVICE void gen_draw_b(const SamplerDev &S, GenShared<WIN> &sh, const GenRoundCtx &c, const uint32_t type, GenDraw &d, F underTrip, const uint32_t keepH1 = CG_NONE)
{
    const uint32_t K = c.K;
    const bool isB = d.isB, pick = d.pick;
    uint32_t flags = d.flags;
    uint64_t rng = d.rng, pos = d.pos, cpos = 0, lbpos = 0, rbpos = 0;
    uint32_t h1 = CG_NONE, h2 = CG_NONE, hl = CG_NONE, hr = CG_NONE;
    uint32_t r1 = d.r1, c1 = d.c1, r2 = 0, c2 = 0; float nm1 = 0.f, nm2 = 0.f;
    uint32_t bin = d.bin, headBin = 0; unsigned long long w0 = 0;
    const uint32_t i1 = d.i1;
    // stage 1 ---------------------------------------------------------------------------------
    uint32_t v1 = CG_NONE;
    // (the word after the bin's own travels in the same trip: when the rest of the bin's word is empty -- one birth in twenty-five at the
    // headline shape's occupancy -- the successor bin is nearly always in the next 64, and the full search through the bitmap's upper
    // levels, half a dozen dependent trips that the whole wave waits for, stays for the domain's sparse stretches)
    unsigned long long w0n = 0ull;
    uint32_t v2 = CG_NONE;
    AtomRec b3; b3.pos = 0; b3.lpos = 0; b3.rpos = 0; b3.left = CG_NONE; b3.right = CG_NONE; b3.mass = 0.f; b3.rmass = 0.f; b3.idx = 0;
    if (isB) { w0 = S.bits0[bin >> 6]; w0n = ((bin >> 6) + 1u < S.nWords0) ? S.bits0[(bin >> 6) + 1u] : 0ull; }
    if (pick) v1 = keepH1 != CG_NONE ? keepH1 : S.vec[i1];      // (keepH1: a pick that stands -- gen_round, keepPick)
    underTrip();
    // stage 2 ---------------------------------------------------------------------------------
    bool slowB = false;
    if (isB) {
        const uint32_t bit = bin & 63u;
        if ((w0 >> bit) & 1ull) headBin = bin;
        else {
            flags |= GEN_F_BINEMPTY; if (w0 == 0ull) flags |= GEN_F_WORDZERO;
            const unsigned long long m = (bit == 63u) ? 0ull : (w0 & ~((2ull << bit) - 1ull));
            if (m) headBin = (bin & ~63u) + (uint32_t)cg_ctz64(m); else if (w0n) headBin = (bin & ~63u) + 64u + (uint32_t)cg_ctz64(w0n);
            else slowB = true;
        }
    }
    AtomRec a; a.pos = 0; a.lpos = 0; a.rpos = 0; a.left = CG_NONE; a.right = CG_NONE; a.mass = 0.f; a.rmass = 0.f; a.idx = 0;
    if (isB && !slowB) v2 = S.binHead[headBin];
    if (pick) { h1 = v1; a = S.atoms[h1]; }
    // stage 3 ---------------------------------------------------------------------------------
    // A picked atom's record carries its neighbours' positions and the right neighbour's mass (gaps_state.h): a move's bounds and
    // an exchange's partner need no trip to the neighbours' records -- every pick goes from its record straight to the matrix
    // entries.  (The one exception: the highest atom's exchange partner is front(), whose record is fetched.)
    uint64_t lp = 0, rp = 0;
    float m2x = 0.f;                        // exchange: the partner's mass
    bool frontE = false;                    // exchange of the highest atom: the partner is front()
    if (pick) {
        cpos = a.pos;
        const uint32_t b1 = gen_bin_of(S, cpos);
        r1 = gen_div_k(S, b1); c1 = b1 - r1 * K;
        hl = a.left;
        if (type == 'M') { hr = a.right; lp = a.lpos; rp = a.rpos; }
        else if (type == 'E') {
            hr = a.right;
            if (hr != CG_NONE) { h2 = hr; rbpos = a.rpos; m2x = a.rmass; }
            else { h2 = sh.g.front; frontE = true; }
        }
    }
    if (AHEAD && frontE) { d.redo = true; frontE = false; h2 = h1; }      // (front() may be another atom behind the decisions: drawn again)
    // the scalars the evaluation starts from travel in the queue record (consumed at commit)
    float old1 = 0.f, old2 = 0.f; uint32_t gib1 = 0, gib2 = 0;
    uint64_t lposB = 0, rposB = 0; float rmassB = 0.f;        // birth: what the new atom's record caches of its neighbours
    if (isB && !slowB) b3 = S.atoms[v2];
    if (frontE) b3 = S.atoms[h2];
    if (isB || pick) { old1 = c.sparse ? S.rows[(size_t)r1 * S.Kpad + c1] : S.mat[(size_t)c1 * S.Mpad + r1]; gib1 = S.otherColPos[c1]; }
    if (pick && type == 'M') {
        if (hl != CG_NONE) { flags |= GEN_F_HASLEFT; lbpos = lp; } else lbpos = 0;
        if (hr != CG_NONE) { flags |= GEN_F_HASRIGHT; rbpos = rp; } else rbpos = S.rboundNone;
        pos = pcg_uniform64(rng, lbpos + 1ull, rbpos - 1ull);
        const uint32_t bin2 = gen_bin_of(S, pos);
        r2 = gen_div_k(S, bin2); c2 = bin2 - r2 * K;
        if (r1 == r2 && c1 == c2) flags |= GEN_F_INLINE;
    }
    if (pick && type == 'E' && !frontE) {
        flags |= GEN_F_HASRIGHT;
        const uint32_t bin2 = gen_bin_of(S, rbpos);
        r2 = gen_div_k(S, bin2); c2 = bin2 - r2 * K;
    }
    if (pick && (type == 'M' || (type == 'E' && !frontE))) { old2 = c.sparse ? S.rows[(size_t)r2 * S.Kpad + c2] : S.mat[(size_t)c2 * S.Mpad + r2]; gib2 = S.otherColPos[c2]; }
    // finish ----------------------------------------------------------------------------------
    if (isB) {
        if (!slowB) {
            if ((flags & GEN_F_BINEMPTY) || b3.pos > pos) { hr = v2; hl = b3.left; lposB = b3.lpos; rposB = b3.pos; rmassB = b3.mass; flags |= GEN_F_NEWHEAD; }
            else if (b3.pos == pos) slowB = true;      // position already taken: the retry loop below
            else {
                // the bin's lowest atom lies below pos: go on to the right; the record in hand knows its right neighbour's
                // position, so the usual case (a bin holds 1.3 atoms on average) needs no further trip
                uint32_t cur = v2, nxt = b3.right; uint64_t curPos = b3.pos, nxtPos = b3.rpos; float nxtMass = b3.rmass;
                for (;;) {
                    if (nxt == CG_NONE) break;
                    if (nxtPos == pos) { slowB = true; break; }
                    if (nxtPos > pos) break;
                    if (AHEAD) { if (d.v3 != CG_NONE) { d.redo = true; break; } d.v3 = nxt; }      // (ahead: one further record, which the validation knows of; a longer walk is made again)
                    const AtomRec w = S.atoms[nxt];
                    cur = nxt; curPos = nxtPos; nxt = w.right; nxtPos = w.rpos; nxtMass = w.rmass;
                }
                hl = cur; hr = nxt; lposB = curPos; rposB = nxtPos; rmassB = nxtMass;
            }
        }
        if (AHEAD && slowB) { d.redo = true; slowB = false; }
#if defined(EXP_NO_SLOW)
        slowB = false;
#endif
        if (slowB) {
            bool occ, nh;
            gen_find_gap(S, pos, bin, &hl, &hr, &occ, &nh);
            while (occ) {           // randomFreePosition retry (ConcurrentAtomicDomain.cpp:46-54)
                pos = pcg_uniform64(rng, 1ull, S.domainLenU);
                bin = gen_bin_of(S, pos); r1 = gen_div_k(S, bin); c1 = bin - r1 * K;
                gen_find_gap(S, pos, bin, &hl, &hr, &occ, &nh);
            }
            flags &= ~(GEN_F_BINEMPTY | GEN_F_WORDZERO | GEN_F_NEWHEAD);
            if (nh) flags |= GEN_F_NEWHEAD;
            if (S.binHead[bin] == CG_NONE) { flags |= GEN_F_BINEMPTY; if (S.bits0[bin >> 6] == 0ull) flags |= GEN_F_WORDZERO; }
            old1 = c.sparse ? S.rows[(size_t)r1 * S.Kpad + c1] : S.mat[(size_t)c1 * S.Mpad + r1]; gib1 = S.otherColPos[c1];      // the retry may have moved the birth to another bin
            lposB = (hl != CG_NONE) ? S.atoms[hl].pos : 0ull;
            if (hr != CG_NONE) { rposB = S.atoms[hr].pos; rmassB = S.atoms[hr].mass; } else { rposB = 0ull; rmassB = 0.f; }
        }
    } else if (pick && type == 'E') {
        if (frontE) {
            rbpos = b3.pos; m2x = b3.mass;
            const uint32_t bin2 = gen_bin_of(S, rbpos);
            r2 = gen_div_k(S, bin2); c2 = bin2 - r2 * K;
            old2 = c.sparse ? S.rows[(size_t)r2 * S.Kpad + c2] : S.mat[(size_t)c2 * S.Mpad + r2]; gib2 = S.otherColPos[c2];
        }
        if (r1 == r2 && c1 == c2 && !(AHEAD && d.redo)) {
            flags |= GEN_F_INLINE;
            const float m1 = a.mass, m2 = m2x;
#if defined(EXP_NO_GAMMA)
            const float newMass = m1;
#else
            const float newMass = pcg_trunc_gamma_upper(rng, S.luts, m1 + m2, 1.f / S.lambda, S.mathMode);
#endif
            const float delta = (m1 > m2) ? newMass - m1 : m2 - newMass;
            if (m1 + delta > GAPS_EPSILON && m2 - delta > GAPS_EPSILON) { flags |= GEN_F_APPLY; nm1 = m1 + delta; nm2 = m2 - delta; }
        }
    }
    d.flags = flags; d.rng = rng; d.pos = pos; d.cpos = cpos; d.h1 = h1; d.h2 = h2; d.hl = hl; d.hr = hr; d.r1 = r1; d.c1 = c1; d.r2 = r2; d.c2 = c2; d.bin = bin;
    d.nm1 = nm1; d.nm2 = nm2; d.amass = a.mass; d.m2x = m2x; d.old1 = old1; d.old2 = old2; d.gib1 = gib1; d.gib2 = gib2; d.lposB = lposB; d.rposB = rposB; d.rmassB = rmassB;
    d.headBin = headBin; d.v2 = v2;      // (v3, xPick, redo: set where they arise)
}

// Did the lane, drawing ahead of the decisions, read only what they and the flush left alone?  gen_draw_check, ahead of the decisions:
// where the lane's keys sit in the note bit sets -- the matrix cells (a birth: its bin; a pick: its atom's bin and, for a move /
// exchange, the other site's), the atom record(s), the pick's slot in the unsorted vector.  gen_draw_valid, behind them: the bits, the
// bitmap words a birth read, and the pick itself from the size the flush leaves (nR; m atoms erased).  A pick is uniform32(0, size - 1)
// (Random.cpp:79-96): x / iPart with iPart = UINT32_MAX / size, x below size * iPart -- the same index from both sizes unless iPart or
// the rejection differs; iPart for the smaller size is the old one or the next (checked by multiplication, no division behind the wait).
struct GenCheck { GenNotePos atomA, atomB, slot, cellA, cellB, eraseA; uint32_t iPartS; };
CG_DEVICE GenCheck gen_draw_check(const GenSpec &sp, const GenDraw &d, const uint32_t nRs, const uint32_t K)
{
    const bool isB = (sp.info & 0xFFu) == 'B';
    GenCheck c;
    c.cellA = gen_note_pos<GEN_DIRTY_CELLS>(isB ? d.bin : d.r1 * K + d.c1); c.cellB = gen_note_pos<GEN_DIRTY_CELLS>(d.r2 * K + d.c2);
    c.atomA = gen_note_pos<GEN_DIRTY_ATOMS>(isB ? d.v2 : d.h1); c.atomB = gen_note_pos<GEN_DIRTY_ATOMS>(d.v3); c.slot = gen_note_pos<GEN_DIRTY_ATOMS>(~d.i1);
    c.eraseA = gen_note_pos<GEN_DIRTY_ERASE>(d.h1);
    c.iPartS = 0xFFFFFFFFu / (nRs + (sp.info >> 8));
    return c;
}
// Returns 0: the draw holds; 1: the lane draws again and reads nothing the flush changes -- a pick whose index and vector slot stand, whose
// record or matrix cells the DECISIONS rewrote (it keeps its pick and need not wait for the flush); 2: it draws again behind the flush.
template <int WIN>
CG_DEVICE uint32_t gen_draw_valid(const SamplerDev &S, GenShared<WIN> &sh, const GenSpec &sp, const GenDraw &d, const GenCheck &ck, const uint32_t nR, const uint32_t m)
{
    const uint32_t type = sp.info & 0xFFu, bBefore = sp.info >> 8;
    uint32_t bad = d.redo ? 1u : 0u, light = 0u;      // bad: behind the flush; light: the decisions' notes alone
#if defined(GEN_AHEAD_BAD_EVERY)
    if (((sp.ct + (uint32_t)sh.g.batchEpoch) % (uint32_t)GEN_AHEAD_BAD_EVERY) == 0u) { if (sp.ct & 1u) bad = 1u; else light = 1u; }      // test-only variant: lanes drawn again, regularly, either way
#endif
    const bool isB = type == 'B';
    const bool reads = isB || d.pick;       // (a lane without an attempt, or whose pick fell on an atom born in this window, read nothing)
    const uint32_t nA = gen_note_get(sh.dAtom, ck.atomA), nB = gen_note_get(sh.dAtom, ck.atomB), nS = gen_note_get(sh.dAtom, ck.slot);
    const uint32_t cA = gen_note_get(sh.dCell, ck.cellA), cB = gen_note_get(sh.dCell, ck.cellB);
    // (a birth reads the bitmap, bin heads and may walk along its bin: it always waits for the flush; a pick reads its atom's record
    // -- the flush rewrites the records of an erased atom's neighbours: dErase -- and matrix cells, which the flush never touches)
    if (isB) { if (reads) bad |= nA | cA; if (d.v3 != CG_NONE) bad |= nB; }
    else if (d.pick) {
        const uint32_t nE = gen_note_get(sh.dErase, ck.eraseA);
        light |= nA | cA | ((type == 'M' || type == 'E') ? cB : 0u);
        bad |= nA & nE;
    }
    if (isB) {
        // the bitmap words it read -- the bin's own, the next, and every further one up to the successor bin's
        const uint32_t wFirst = d.bin >> 6;
        uint32_t wLast = d.headBin >> 6; wLast = wLast > wFirst + 1u ? wLast : wFirst + 1u;
        uint32_t dd = (wLast - wFirst >= 16384u) ? 1u : 0u;
        for (uint32_t w = wFirst; !dd && w <= wLast; ) {
            const uint32_t wm = w & 16383u, n = 32u - (wm & 31u), left = wLast - w + 1u, take = n < left ? n : left;
            const uint32_t bits = sh.dirty[wm >> 5] >> (wm & 31u);
            dd = bits & (take >= 32u ? 0xFFFFFFFFu : ((1u << take) - 1u));
            w += take;
        }
        bad |= dd ? 1u : 0u;
    } else if (m != 0u && type != 0u) {
        // the pick again, from the size the flush leaves; its slot must not be one the flush refills from the vector's tail
        const uint32_t nT = nR + bBefore;
        uint32_t q = ck.iPartS, rem = 0xFFFFFFFFu - q * nT;             // (q * nT <= q * (the larger size) <= UINT32_MAX)
        const uint32_t up = (uint32_t)(rem >= nT);
        q += up; rem -= up ? nT : 0u;
        const uint32_t lo = d.i1 * q;                                   // (i1 < nT: no overflow)
        const uint32_t same = (uint32_t)(rem < nT) & (uint32_t)(d.xPick < 0xFFFFFFFFu - rem) & (uint32_t)(d.xPick >= lo) & (uint32_t)(d.xPick - lo < q) & (uint32_t)(!(d.pick && d.i1 >= nR));
        bad |= (same ^ 1u) | (d.pick ? nS : 0u);
    }
    const bool ok = !(d.go != 0u && (bad | light) != 0u);
    const uint32_t level = d.go == 0u ? 0u : (bad ? 2u : (light && d.pick ? 1u : (light ? 2u : 0u)));
#if defined(GEN_TIMELINE)
    // dev: why lanes draw again -- [0] lanes with an attempt, [1] drew again, [2] the pick moved (iPart / rejection / beyond the size), [3] its slot refilled,
    // [4] a noted atom record, [5] a noted matrix cell, [6] a birth's bitmap words, [7] one of the long ways (redo flag)
    if (d.go) {
        const uint32_t moved = (!isB && m != 0u && type != 0u) ? (uint32_t)(((bad & 1u) != 0u) && !(d.redo) && !((reads ? (nA | cA) : 0u) & 1u)) : 0u;
        cg_atomic_add_u64(&g_ahead_why[0], 1ull);
        if (!ok) cg_atomic_add_u64(&g_ahead_why[1], 1ull);
        if (moved) cg_atomic_add_u64(&g_ahead_why[2], 1ull);
        if (d.pick && m != 0u && nS) cg_atomic_add_u64(&g_ahead_why[3], 1ull);
        if (reads && (nA | ((isB && d.v3 != CG_NONE) ? nB : 0u))) cg_atomic_add_u64(&g_ahead_why[4], 1ull);
        if (reads && (cA | ((d.pick && (type == 'M' || type == 'E')) ? cB : 0u))) cg_atomic_add_u64(&g_ahead_why[5], 1ull);
        if (isB && !ok && !d.redo && !(nA | cA)) cg_atomic_add_u64(&g_ahead_why[6], 1ull);
        if (d.redo) cg_atomic_add_u64(&g_ahead_why[7], 1ull);
    }
#endif
#if defined(COGAPS_EMUL)
    if (d.go) cg_atomic_add_u64(&S.gs->prof[level == 0u ? 8 : (level == 1u ? 10 : 9)], 1ull);      // test-only build: lanes whose draw ahead held / that drew again behind the flush / keeping their pick
#endif
    (void)ok;
    return level;
}

template <int WIN, bool FIRST, bool SPEC = false, bool AHEAD = false>
CG_DEVICE bool gen_round(const SamplerDev &S, GenShared<WIN> &sh, const GenRoundCtx &c, const uint32_t roundNo, const GenSpec *spec = nullptr, const GenDraw *ahead = nullptr, const bool aheadValid = true, const bool keepPick = false)
{
    static_assert(FIRST || !SPEC, "only a batch's first window is classified ahead of the decisions");
    static_assert(SPEC || !AHEAD, "only a window classified ahead is drawn ahead");
    const unsigned t = c.t;
    const uint64_t jm0 = c.jm0, ji0 = c.ji0, jm1 = c.jm1, ji1 = c.ji1, seed1 = c.seed1, batchEpoch = c.batchEpoch;
    const uint32_t updBase = c.updBase, remaining = c.remaining, K = c.K;
    GenScalars *gs = c.gs;
    constexpr bool first = FIRST;
    GEN_TS_INIT(); GEN_TS_RESUME(FIRST ? (AHEAD ? 16u : 13u) : 40u);
    GEN_TS(4);
    const uint32_t nR = first ? c.n0 : sh.nR, minR = first ? c.n0 : sh.minAtoms, skip = first ? c.g_skip : sh.skip, processed = first ? 0u : sh.processed;
    const uint64_t qrngRound = first ? c.g_qrng : sh.qrngRound;
    const float u1c = first ? c.g_u1 : sh.u1c, u2c = first ? c.g_u2 : sh.u2c;
    const float dpLo0 = first ? c.dp0 : sh.dpLo[0], dpHi0 = first ? c.dp0 : sh.dpHi[0];
    const uint32_t left_ = remaining - processed;
    const uint32_t winN = left_ < (uint32_t)WIN ? left_ : (uint32_t)WIN;

    // ------------------------------------------------------------------ A1 (lane = attempt): (u1,u2), B/D/M/E
    uint32_t bBeforeA1 = 0, dBeforeA1 = 0, guessA1 = 0, activeA1 = 0; float u1A1 = 0.f, u2A1 = 0.f;      // the lane's OWN attempt, for its exact decision below
    if (SPEC) { bBeforeA1 = spec->bBefore; dBeforeA1 = spec->dBefore; guessA1 = spec->guess; activeA1 = spec->active; u1A1 = spec->u1; u2A1 = spec->u2; }
    else {
        // (0/1 words and selects instead of short-circuit logic: with one wave per SIMD a branch costs more
        // than the arithmetic it would skip)
        const uint32_t active = t < winN;
        const uint32_t tt = active ? t : 0u;
        const uint64_t mySeed = (processed == 0u) ? seed1 : S.seeds[updBase + processed + tt];      // round 1: prefetched
        uint64_t s = (skip ? jm1 : jm0) * qrngRound + (skip ? ji1 : ji0);
        float u1 = pcg_uniform(s), u2 = pcg_uniform(s);
        const uint32_t cached = (skip != 0u) & (uint32_t)(t == 0u);       // attempt 0 replays the cached pair
        u1 = cached ? u1c : u1; u2 = cached ? u2c : u2;
        uint32_t guess = gen_decide(u1, u2, minR, nR, dpLo0, dpHi0);
        guess = active ? guess : (uint32_t)GEN_T_NONE;
        GEN_PIN(guess); GEN_PIN(u1); GEN_PIN(u2);
        GEN_TS(5);
        sh.u1[t] = u1; sh.u2[t] = u2;
        if (first) { sh.dpHi[t] = c.tabHi; sh.dpLo[t] = c.tabLo; }      // read after the barrier inside the count (later rounds: gen_body)
        // ONE exchange for the whole classification (round 4; two until then): how many births / deaths / moves / exchanges -- by the first
        // guess -- precede this attempt.  The counts give the attempt's sorted slot (births+deaths | moves | exchanges: a wave runs one
        // code path) at once; the EXACT birth / death decision, which needs the birth / death counts, no longer stands between the two
        // counts: it is made by the attempt's own lane further down, under the draws' first memory trip (gen_a1_exact), and only feeds the
        // stop key.  An attempt whose exact decision will differ from its guess (a hazard: the window is cut there) is sorted and drawn like
        // the others -- it and everything behind it is never committed, and what it registers is only ever compared by later attempts.
        uint32_t eX[4], tX[4];
        gen_count4<WIN>(sh.wtot4, t, guess == 'B', guess == 'D', guess == 'M', guess == 'E', eX, tX);
        GEN_TS(6);
        bBeforeA1 = eX[0]; dBeforeA1 = eX[1]; u1A1 = u1; u2A1 = u2; guessA1 = guess; activeA1 = active;
        const uint32_t go = (uint32_t)(guess != GEN_T_NONE);
        const uint32_t k0 = (uint32_t)(guess == 'B') | (uint32_t)(guess == 'D'), k1 = (uint32_t)(guess == 'M');
        const uint32_t T0 = tX[0] + tX[1], T1 = tX[2], T2 = tX[3];
        if (go) {
            uint32_t slot = T0 + T1 + eX[3];
            slot = k1 ? T0 + eX[2] : slot;
            slot = k0 ? eX[0] + eX[1] : slot;
            sh.perm[slot] = (uint16_t)t;
            sh.info[t] = guess | (eX[0] << 8);
            sh.seed[t] = mySeed;                                     // consumed after the type sort
        }
        GEN_TS(8);
        if (t == 0) sh.nWork = T0 + T1 + T2;
    }
    if (!SPEC) cg_sync_lds();
    GEN_TS(9);

    // ------------------------------------------------------------------ A2 (lane = sorted slot): populate-phase draws (gen_draw_a / gen_draw_b)
    const bool go = SPEC ? spec->go != 0u : t < sh.nWork;
    const uint32_t ct = SPEC ? spec->ct : (go ? (uint32_t)sh.perm[t] : 0u);          // this lane's attempt ordinal in the window
    const uint32_t info = SPEC ? spec->info : (go ? sh.info[ct] : 0u);
    const uint32_t type = info & 0xFFu, bBefore = info >> 8;
    // the exact B/D/indeterminate decision of this lane's own attempt (ProposalQueue.cpp:129-160 with the atom bounds as the births /
    // deaths before it leave them): a guess that does not hold is a hazard (the window is cut there and redrawn with exact bounds), an
    // indeterminate attempt ends the batch -- the smallest such attempt is the stop key.  Made while the draws' first memory trip is on
    // its way (SPEC: the rows were never parked -- the table's window staged in LDS holds them: deathProb(n0 - d), deathProb(n0 + b))
    auto exactDecide = [&]() {
        const float dpLoX = SPEC ? (nR >= dBeforeA1 ? sh.dpWin[nR - dBeforeA1 - c.dpBase] : 0.f) : sh.dpLo[dBeforeA1];
        const float dpHiX = SPEC ? sh.dpWin[nR + bBeforeA1 - c.dpBase] : sh.dpHi[bBeforeA1];
        const uint32_t exact = gen_decide(u1A1, u2A1, (uint64_t)minR - dBeforeA1, (uint64_t)nR + bBeforeA1, dpLoX, dpHiX);
        const uint32_t hazA = activeA1 & (uint32_t)(exact != guessA1);
        const uint32_t failA = activeA1 & (hazA ^ 1u) & (uint32_t)(guessA1 == GEN_T_NONE);   // indeterminate: batch ends, no seed used
        if (hazA | failA) cg_atomic_min_u32(&sh.stopKey, 2u * t + (hazA ^ 1u));
        GEN_TS(7);
    };
    GenDraw d;
    if (AHEAD) {
        // (chained launch: the window was drawn ahead of the decisions -- gen_body -- against the domain as the previous batch's commit left
        // it; the lanes whose reads the decisions or the flush touched draw again, now, against the domain as it is: the same code, the
        // same results as if every lane had waited.  The join with the flush precedes both: gen_body.)
        d = *ahead;
        if (d.isB) d.i1 = nR + bBefore;      // (a birth's index in the unsorted vector: the domain's size, known now)
#if !defined(EXP_NO_REDO)
        const bool again = go && !aheadValid;
        if (cg_ballot(again) != 0ull) {      // (wave-uniform: the wave's other lanes walk through with nothing to draw, as lanes without an attempt do)
            // (keepPick -- wave-uniform: the whole window draws again without waiting for the flush -- the index vector is not read again: the
            // pick and its slot were validated, only the record and the matrix cells are read anew)
            GenDraw r; gen_draw_a<WIN, SPEC>(S, c, spec, again, type, bBefore, SPEC ? spec->rng : 0ull, nR, r);
            gen_draw_b<WIN, false>(S, sh, c, type, r, [&]() {}, keepPick ? ahead->h1 : CG_NONE);
            if (again) d = r;
        }
#endif
        exactDecide();
    } else {
        gen_draw_a<WIN, SPEC>(S, c, spec, go, type, bBefore, SPEC ? spec->rng : (go ? pcg_from_seed(sh.seed[ct]) : 0ull), nR, d);
        GEN_PIN(d.i1); GEN_PIN(d.bin); GEN_PIN(d.pos);
        GEN_TS(10);
        // Everything above needed only the window's scalars.  From here on the lanes read the domain (index vector, records, bitmap,
        // bin heads), which the helper wave's flush has been rewriting meanwhile: join it (its stores are acknowledged: cg_sync waits
        // for every wave's own outstanding memory operations).  Later rounds of a batch ended with such a barrier already.
        if (FIRST) cg_sync();
        GEN_TS(25);
        gen_draw_b<WIN, false>(S, sh, c, type, d, exactDecide);
    }
    uint32_t flags = d.flags;
    const bool isB = d.isB, pick = d.pick;
    const uint64_t rng = d.rng, pos = d.pos, cpos = d.cpos;
    uint32_t h1 = d.h1, h2 = d.h2, i1 = d.i1, i2 = CG_NONE; const uint32_t hl = d.hl, hr = d.hr;
    const uint32_t r1 = d.r1, c1 = d.c1, r2 = d.r2, c2 = d.c2, bin = d.bin; const float nm1 = d.nm1, nm2 = d.nm2;
    const float old1 = d.old1, old2 = d.old2, m2x = d.m2x; const uint32_t gib1 = d.gib1, gib2 = d.gib2;
    const uint64_t lposB = d.lposB, rposB = d.rposB; const float rmassB = d.rmassB;
    struct { float mass; } a; a.mass = d.amass;
    (void)isB; (void)pick;
    GEN_TS(14);

    // ------------------------------------------------------------------ B1: register rows / atoms / gaps
    // Round 1 of a batch (95 % of all rounds) keeps the conflict sets in an LDS hash table; later rounds,
    // which must also see what earlier rounds of the batch committed, use the stamp tables in HBM.
    const bool live = go && !(flags & GEN_F_FAIL);
    const bool queuedM = live && type == 'M' && !(flags & GEN_F_INLINE);
    const bool ldsRound = FIRST || roundNo <= (uint32_t)GEN_LDS_ROUNDS;
#if defined(COGAPS_EMUL)
    // test-only build: how many later rounds went through the LDS table / the stamp tables (tests check that both paths were taken)
    if (!FIRST && t == 0) cg_atomic_add_u64(&gs->prof[ldsRound ? 14 : 15], 1ull);
#endif
    // what an attempt registers under and compares with: its ordinal in the BATCH (window ordinal + attempts committed by earlier
    // rounds).  Round 1: the window ordinal itself.  Entries earlier rounds left behind belong to committed attempts and are smaller
    // than every ordinal of this window (the round's clean-up below removes everything else).
    const uint32_t gord = processed + ct;
    uint32_t rs0 = 0, rs1 = 0, rs2 = 0, rf0 = 0, rf1 = 0, rf2 = 0;      // the three (slot, field) registrations, for the clean-up
    uint64_t d9 = 0, d10 = 0;       // later rounds, birth: the destinations of the neighbours' committed queued moves (mProposedMoves)
    if (!FIRST && ldsRound && live && type == 'B') { d9 = (hl != CG_NONE) ? S.atomDest[hl] : 0ull; d10 = (hr != CG_NONE) ? S.atomDest[hr] : 0ull; }
    if (go) { sh.cpos[ct] = cpos; sh.pos[ct] = pos; sh.type[ct] = queuedM ? (uint8_t)'M' : (uint8_t)0; }
    if (live && ldsRound) {
        // up to three (key, field) registrations; an unused one repeats the first.  Predicates are 0/1 words
        // combined with bit operations: every short-circuit would be a branch, and a branch costs more
        // than the arithmetic it skips when one wave owns the SIMD
        const uint32_t inl = flags & GEN_F_INLINE, tB = type == 'B', tD = type == 'D', tM = type == 'M';
        const uint32_t k0 = inl ? h1 : (GEN_TAB_ROW | r1), f0 = inl << 1;
        const uint32_t use1 = 1u ^ (inl & tM), use2 = tM & (inl ^ 1u);
        const uint32_t hlKey = (hl == CG_NONE) ? GEN_TAB_FRONT : hl;
        uint32_t k1 = GEN_TAB_ROW | r2;              // queued move / exchange: the second row
        k1 = tD ? h1 : k1;                           // death: the atom
        k1 = tB ? hlKey : k1;                        // birth: the gap right of the left neighbour
        k1 = inl ? h2 : k1;                          // same-bin exchange: the partner
        k1 = use1 ? k1 : k0;
        uint32_t f1 = inl ? 2u : tB; f1 = use1 ? f1 : f0;
        const uint32_t k2 = use2 ? h1 : k0, f2 = use2 ? 0u : f0;
        // claim the three slots together: one compare-and-swap each per probe step (a placed key
        // repeats the swap on its own slot, which changes nothing)
        const uint32_t hh0 = gen_tab_hash(k0), hh1 = gen_tab_hash(k1), hh2 = gen_tab_hash(k2);
        uint32_t b0 = gen_tab_bucket<WIN>(hh0), b1_ = gen_tab_bucket<WIN>(hh1), b2_ = gen_tab_bucket<WIN>(hh2);
        const uint32_t j0 = gen_tab_start<WIN>(hh0), j1 = gen_tab_start<WIN>(hh1), j2 = gen_tab_start<WIN>(hh2);
        uint32_t s0 = 0, s1 = 0, s2 = 0, d0 = 0, d1 = 0, d2 = 0;
        for (uint32_t i = 0; ; ++i) {
            const uint32_t p0 = d0 ? s0 : 4u * b0 + ((j0 + i) & 3u), p1 = d1 ? s1 : 4u * b1_ + ((j1 + i) & 3u), p2 = d2 ? s2 : 4u * b2_ + ((j2 + i) & 3u);
            const uint32_t o0 = cg_atomic_cas_u32(&sh.bkey[p0], GEN_TAB_EMPTY, k0);
            const uint32_t o1 = cg_atomic_cas_u32(&sh.bkey[p1], GEN_TAB_EMPTY, k1);
            const uint32_t o2 = cg_atomic_cas_u32(&sh.bkey[p2], GEN_TAB_EMPTY, k2);
            s0 = p0; s1 = p1; s2 = p2;
            d0 |= (uint32_t)(o0 == GEN_TAB_EMPTY) | (uint32_t)(o0 == k0);
            d1 |= (uint32_t)(o1 == GEN_TAB_EMPTY) | (uint32_t)(o1 == k1);
            d2 |= (uint32_t)(o2 == GEN_TAB_EMPTY) | (uint32_t)(o2 == k2);
            if (d0 & d1 & d2) break;
            const uint32_t wrap = (i & 3u) == 3u;      // bucket exhausted: the next one
            b0 = (b0 + wrap) & (uint32_t)(GEN_TAB_NB - 1); b1_ = (b1_ + wrap) & (uint32_t)(GEN_TAB_NB - 1); b2_ = (b2_ + wrap) & (uint32_t)(GEN_TAB_NB - 1);
        }
        // every value word was set to "nobody" (all ones) at kernel entry by the helper wave, so the slot can be written at once:
        // the smallest registering ordinal wins, whoever opened the slot
        uint32_t *words = &sh.bval[0].used;       // word 0 = used, 1 = gap, 2 = inl
        cg_atomic_min_u32(&words[4u * s0 + f0], gord);
        cg_atomic_min_u32(&words[4u * s1 + f1], gord);
        cg_atomic_min_u32(&words[4u * s2 + f2], gord);
        rs0 = s0; rs1 = s1; rs2 = s2; rf0 = f0; rf1 = f1; rf2 = f2;
    } else if (live) {
        // up to three keys: (kind, id)
        uint32_t rk[3], rid[3]; int nk = 0;
        const bool inl = (flags & GEN_F_INLINE) != 0;
        if (type == 'B') { rk[0] = GEN_K_ROW; rid[0] = r1; rk[1] = GEN_K_GAP; rid[1] = (hl == CG_NONE) ? 0u : hl + 1u; nk = 2; }
        else if (type == 'D') { rk[0] = GEN_K_ROW; rid[0] = r1; rk[1] = GEN_K_ATOM; rid[1] = h1; nk = 2; }
        else if (type == 'M') {
            if (inl) { rk[0] = GEN_K_INL; rid[0] = h1; nk = 1; }
            else { rk[0] = GEN_K_ROW; rid[0] = r1; rk[1] = GEN_K_ROW; rid[1] = r2; rk[2] = GEN_K_ATOM; rid[2] = h1; nk = 3; }
        } else {
            if (inl) { rk[0] = GEN_K_INL; rid[0] = h1; rk[1] = GEN_K_INL; rid[1] = h2; nk = 2; }
            else { rk[0] = GEN_K_ROW; rid[0] = r1; rk[1] = GEN_K_ROW; rid[1] = r2; nk = 2; }
        }
        const unsigned long long st = gen_stamp(batchEpoch, roundNo, ct);
        for (int k = 0; k < nk; ++k) cg_atomic_max_u64(gen_stamp_ptr(S, rk[k], rid[k]), st);
    }
    GEN_TS(15);
    if (ldsRound) cg_sync_lds(); else cg_sync();
    GEN_TS(16);

    // ------------------------------------------------------------------ B2: probe the sets (all probes of a lane
    // are independent: issued together, then the per-type logic runs on registers)
    if (live && ldsRound) {
        // six bucket reads, then the six value reads of the matching slots; a key that is not in the table
        // reads "nobody".  0/1 words and bit operations again (see B1).
        const uint32_t tB = type == 'B', tM = type == 'M', tE = type == 'E', inl = flags & GEN_F_INLINE;
        const uint32_t hasL = hl != CG_NONE, hasR = hr != CG_NONE, noRight = (flags & GEN_F_HASRIGHT) == 0u;
        uint32_t key[6], use[6];
        key[0] = GEN_TAB_ROW | r1; use[0] = 1u;
        key[1] = GEN_TAB_ROW | r2; use[1] = tM | tE;
        key[2] = ((tM | tB) & hasL) ? hl : GEN_TAB_FRONT; use[2] = tM | tB | (tE & noRight);
        // (same-bin exchange: the gap LEFT of the centre -- a birth there earlier in this window is the holder of the centre's cached mass)
        const uint32_t eInl = tE & (uint32_t)(inl != 0u);
        key[3] = eInl ? (hasL ? hl : GEN_TAB_FRONT) : hr; use[3] = ((tM | tB) & hasR) | eInl;
        const uint32_t tD = type == 'D';
        key[4] = h1; use[4] = tM | tE | tD;
        key[5] = h2; use[5] = tE;
        uint32_t bk[6]; GenTabKeys kq[6];
        for (int k = 0; k < 6; ++k) { bk[k] = gen_tab_bucket<WIN>(gen_tab_hash(key[k])); kq[k] = *(const GenTabKeys *)&sh.bkey[4u * bk[k]]; }
        uint32_t sl[6], hit[6], over = 0;
        for (int k = 0; k < 6; ++k) {
            const uint32_t *q4 = kq[k].k;
            const uint32_t e1 = q4[1] == key[k], e2 = q4[2] == key[k], e3 = q4[3] == key[k];
            const uint32_t found = (uint32_t)(q4[0] == key[k]) | e1 | e2 | e3;
            const uint32_t hole = (uint32_t)(q4[0] == GEN_TAB_EMPTY) | (uint32_t)(q4[1] == GEN_TAB_EMPTY) | (uint32_t)(q4[2] == GEN_TAB_EMPTY) | (uint32_t)(q4[3] == GEN_TAB_EMPTY);
            sl[k] = 4u * bk[k] + e1 + 2u * e2 + 3u * e3;
            hit[k] = use[k] & found;
            over |= use[k] & (found ^ 1u) & (hole ^ 1u);             // the key may have spilled into the next bucket
        }
        if (over) {
            // Rare per key (a full bucket that does not hold it: 0.3 % of the lookups) but not per launch: with ~200 lookups per wave half
            // of the waves meet one, and the barrier behind this phase waits for the slowest wave.  So the spill is followed ONE bucket
            // on for exactly the keys that need it, all of them at once (the probe order of gen_tab_claim: the same start slot, next
            // bucket); only a key that finds a second full bucket without itself takes the serial search.
            uint32_t need[6], over2 = 0; GenTabKeys kq2[6];
            for (int k = 0; k < 6; ++k) {
                const uint32_t *q4 = kq[k].k;
                const uint32_t found = (uint32_t)(q4[0] == key[k]) | (uint32_t)(q4[1] == key[k]) | (uint32_t)(q4[2] == key[k]) | (uint32_t)(q4[3] == key[k]);
                const uint32_t hole = (uint32_t)(q4[0] == GEN_TAB_EMPTY) | (uint32_t)(q4[1] == GEN_TAB_EMPTY) | (uint32_t)(q4[2] == GEN_TAB_EMPTY) | (uint32_t)(q4[3] == GEN_TAB_EMPTY);
                need[k] = use[k] & (found ^ 1u) & (hole ^ 1u);
                kq2[k] = *(const GenTabKeys *)&sh.bkey[4u * ((bk[k] + 1u) & (uint32_t)(GEN_TAB_NB - 1))];
            }
            for (int k = 0; k < 6; ++k) {
                const uint32_t *q4 = kq2[k].k;
                const uint32_t e1 = q4[1] == key[k], e2 = q4[2] == key[k], e3 = q4[3] == key[k];
                const uint32_t found = (uint32_t)(q4[0] == key[k]) | e1 | e2 | e3;
                const uint32_t hole = (uint32_t)(q4[0] == GEN_TAB_EMPTY) | (uint32_t)(q4[1] == GEN_TAB_EMPTY) | (uint32_t)(q4[2] == GEN_TAB_EMPTY) | (uint32_t)(q4[3] == GEN_TAB_EMPTY);
                const uint32_t s2 = 4u * ((bk[k] + 1u) & (uint32_t)(GEN_TAB_NB - 1)) + e1 + 2u * e2 + 3u * e3;
                sl[k] = need[k] ? s2 : sl[k];
                hit[k] = need[k] ? found : hit[k];
                over2 |= need[k] & (found ^ 1u) & (hole ^ 1u);
            }
            if (over2) {                                              // two full buckets in a row: the serial search
                for (int k = 0; k < 6; ++k) if (use[k]) { const uint32_t f = gen_tab_find<WIN>(sh, key[k]); hit[k] = f != GEN_TAB_EMPTY; sl[k] = hit[k] ? f : 0u; }
            }
        }
        GenTabVal e[6];
        for (int k = 0; k < 6; ++k) e[k] = sh.bval[hit[k] ? sl[k] : 0u];
        // E(v) = 1 when an earlier attempt of this window registered under the word
        #define GEN_E(k, w) (hit[k] & (uint32_t)(e[k].w < gord))
        uint32_t fail = GEN_E(0, used) | GEN_E(1, used);                              // a row in use
        // move: a neighbour in use (mUsedAtoms), or a birth earlier in this window inside (left, right)
        fail |= tM & (GEN_E(2, used) | GEN_E(3, used) | GEN_E(2, gap) | GEN_E(4, gap));
        // exchange: an earlier birth right of the centre is the true partner (or, for the last atom, a new front())
        fail |= tE & (GEN_E(4, gap) | GEN_E(2, gap));
        // birth: an earlier birth in the same gap; move / birth / same-bin exchange: an earlier same-bin
        // move or exchange of this window touched an atom whose position this attempt relied on
        uint32_t haz = tB & (GEN_E(2, gap) | GEN_E(2, inl) | GEN_E(3, inl));
        haz |= tM & (GEN_E(4, inl) | GEN_E(2, inl) | GEN_E(3, inl));
        // death / exchange: the masses in the queue record were read before an earlier same-bin exchange of
        // this window rewrote them
        haz |= (tE | tD) & (GEN_E(4, inl) | GEN_E(5, inl));
        // same-bin exchange: it rewrites the copy of the centre's mass that the centre's left neighbour caches, and an earlier birth
        // of this window between the two has become that neighbour
        haz |= eInl & GEN_E(3, gap);
        if (tB) {
            // mProposedMoves.overlap(pos): a neighbour has a queued move whose interval covers pos
            const uint32_t uL = GEN_E(2, used), uR = GEN_E(3, used);
            // the registrant is an attempt of this window (its move, if it is one, sits in the window's arrays) or, in a later round,
            // one an earlier round committed (a queued move left its destination in atomDest; the atom itself has not moved yet)
            const uint32_t wL = uL & (uint32_t)(e[2].used >= processed), wR = uR & (uint32_t)(e[3].used >= processed);
            const uint32_t iL = wL ? e[2].used - processed : 0u, iR = wR ? e[3].used - processed : 0u;
            const uint64_t aL = sh.cpos[iL], bL = sh.pos[iL], aR = sh.cpos[iR], bR = sh.pos[iR];
            const uint32_t mL = wL & (uint32_t)(sh.type[iL] == 'M'), mR = wR & (uint32_t)(sh.type[iR] == 'M');
            const uint64_t loL = aL < bL ? aL : bL, hiL = aL < bL ? bL : aL, loR = aR < bR ? aR : bR, hiR = aR < bR ? bR : aR;
            fail |= mL & (uint32_t)(loL < pos) & (uint32_t)(pos < hiL);
            fail |= mR & (uint32_t)(loR < pos) & (uint32_t)(pos < hiR);
            if (!FIRST) {
                const uint32_t cL = uL & (wL ^ 1u) & (uint32_t)(d9 != 0ull), cR = uR & (wR ^ 1u) & (uint32_t)(d10 != 0ull);
                const uint64_t loCL = lposB < d9 ? lposB : d9, hiCL = lposB < d9 ? d9 : lposB, loCR = rposB < d10 ? rposB : d10, hiCR = rposB < d10 ? d10 : rposB;
                fail |= cL & (uint32_t)(loCL < pos) & (uint32_t)(pos < hiCL);
                fail |= cR & (uint32_t)(loCR < pos) & (uint32_t)(pos < hiCR);
            }
        }
        #undef GEN_E
        GEN_PIN(flags);
        GEN_TS(17);
        flags |= haz ? GEN_F_HAZARD : (fail ? GEN_F_FAIL : 0u);
    } else if (live) {
        const bool tB = type == 'B', tM = type == 'M', tE = type == 'E', inl = (flags & GEN_F_INLINE) != 0;
        const uint32_t keyL = (hl == CG_NONE) ? 0u : hl + 1u;
        uint32_t pk[10], pid[10]; bool pu[10];
        pk[0] = GEN_K_ROW; pid[0] = r1; pu[0] = true;
        pk[1] = GEN_K_ROW; pid[1] = r2; pu[1] = tM || tE;
        pk[2] = GEN_K_ATOM; pid[2] = hl; pu[2] = (tM || tB) && hl != CG_NONE;
        pk[3] = GEN_K_ATOM; pid[3] = hr; pu[3] = (tM || tB) && hr != CG_NONE;
        pk[4] = GEN_K_GAP; pid[4] = (tB || tM) ? keyL : h1 + 1u; pu[4] = tB || tM || tE;
        pk[5] = GEN_K_GAP; pid[5] = tM ? h1 + 1u : 0u; pu[5] = tM || (tE && !(flags & GEN_F_HASRIGHT));
        const bool tD = type == 'D';
        pk[6] = GEN_K_INL; pid[6] = tB ? hl : h1; pu[6] = tM || (tB && hl != CG_NONE) || tE || tD;
        pk[7] = GEN_K_INL; pid[7] = tM ? hl : (tB ? hr : h2); pu[7] = (tM && hl != CG_NONE) || (tB && hr != CG_NONE) || tE;
        pk[8] = GEN_K_INL; pid[8] = hr; pu[8] = tM && hr != CG_NONE;
        pk[9] = GEN_K_GAP; pid[9] = keyL; pu[9] = tE && inl;       // same-bin exchange: a birth of this window left of the centre (see the LDS round)
        int res[10]; uint32_t rix[10];
        {
            unsigned long long v[10];
            for (int k = 0; k < 10; ++k) v[k] = cg_load_l2_u64(pu[k] ? gen_stamp_ptr(S, pk[k], pid[k]) : &S.gapStamp[0]);
            d9 = (tB && hl != CG_NONE) ? S.atomDest[hl] : 0ull; d10 = (tB && hr != CG_NONE) ? S.atomDest[hr] : 0ull;
            for (int k = 0; k < 10; ++k) { rix[k] = 0; res[k] = pu[k] ? gen_probe(v[k], batchEpoch, roundNo, ct, &rix[k]) : 0; }
        }
        bool fail = res[0] != 0, haz = false;                                        // row r1 in use
        if (res[1] != 0) fail = true;                                                // row r2 in use
        if (tB) {
            if (res[4] == 2) haz = true;                                             // an earlier birth of this window in the same gap
            const uint32_t nb[2] = {hl, hr}; const uint64_t dest[2] = {d9, d10};
            for (int k = 0; k < 2; ++k) {
                if (nb[k] == CG_NONE) continue;
                // mProposedMoves.overlap(pos): the neighbour has a queued move whose interval covers pos
                const int u = res[2 + k]; const uint32_t ix = rix[2 + k];
                uint64_t ma = 0, mb = 0; bool mv = false;
                if (u == 1 && dest[k] != 0ull) { ma = S.atoms[nb[k]].pos; mb = dest[k]; mv = true; }
                else if (u == 2 && sh.type[ix] == 'M') { ma = sh.cpos[ix]; mb = sh.pos[ix]; mv = true; }
                if (mv) { const uint64_t lo = ma < mb ? ma : mb, hi = ma < mb ? mb : ma; if (lo < pos && pos < hi) fail = true; }
                // an earlier same-bin move of this window shifted the neighbour this gap search compared against
                if (res[6 + k] == 2) haz = true;
            }
        } else if (tM) {
            if (res[2] != 0 || res[3] != 0) fail = true;                             // mUsedAtoms: a neighbour is in use
            // a birth earlier in this window inside (left, right) is the true neighbour, and it is "used"
            if (res[4] == 2 || res[5] == 2) fail = true;
            // an earlier same-bin move/exchange of this window touched the centre or a neighbour: positions stale
            if (res[6] == 2 || res[7] == 2 || res[8] == 2) haz = true;
        } else if (tE) {
            // an earlier birth right of the centre is the true partner (or, for the last atom, a new front())
            if (res[4] == 2 || res[5] == 2) fail = true;
            // the masses in the queue record were read before an earlier same-bin exchange of this window rewrote them
            if (res[6] == 2 || res[7] == 2) haz = true;
            if (res[9] == 2) haz = true;
        } else if (tD) {
            if (res[6] == 2) haz = true;
        }
        if (haz) flags |= GEN_F_HAZARD; else if (fail) flags |= GEN_F_FAIL;
    }
    GEN_TS(18);
    if (go && (flags & (GEN_F_HAZARD | GEN_F_FAIL))) cg_atomic_min_u32(&sh.stopKey, 2u * ct + ((flags & GEN_F_HAZARD) ? 0u : 1u));
    if (ldsRound) cg_sync_lds(); else cg_sync();
    GEN_TS(19);

    // ------------------------------------------------------------------ C: commit attempts [0, stopT)
    const uint32_t stopKey = sh.stopKey;
    const uint32_t stopT = (stopKey == 0xFFFFFFFFu) ? winN : (stopKey >> 1);
    const bool stopFail = (stopKey != 0xFFFFFFFFu) && (stopKey & 1u);
    const bool commit = go && ct < stopT;            // every such attempt is live
    const bool queued = commit && (type == 'B' || type == 'D' || !(flags & GEN_F_INLINE));
    // what the commit reads of the round's scalars, taken BEFORE the barrier: behind it the helper wave's bookkeeping rewrites them
    // while the attempt lanes commit
    const uint32_t c_fc = sh.g.freeCount, c_handleHi = sh.g.handleHi, c_flushBase = sh.flushBase, c_flushM = sh.flushM, c_qlen = sh.qlen;
    const uint32_t c_traceOn = sh.g.traceOn, c_traceCount = sh.g.traceCount, c_traceCap = sh.g.traceCap, c_nBatches = sh.g.nBatches;
    const bool endB = stopFail || (processed + stopT >= remaining);      // the batch ends with this round (every lane knows)
    if (commit) {
        const unsigned long long bit = 1ull << (ct & 63u);
        if (queued) cg_atomic_or_u64(&sh.mq[ct >> 6], bit);
        if (type == 'B') { cg_atomic_or_u64(&sh.mb[ct >> 6], bit); if (hl == CG_NONE) sh.frontPending = 1u; }      // (at most one birth of a round lands before the front atom: two would share the gap)
        if (type == 'D') cg_atomic_or_u64(&sh.md[ct >> 6], bit);
    }
    cg_sync_lds();
    GEN_TS(20);
    if (commit) {
        uint32_t qBefore = 0, bRank = 0;
        {
            const uint32_t wq = ct >> 6; const unsigned long long lt = (1ull << (ct & 63u)) - 1ull;
            for (uint32_t w = 0; w < wq; ++w) { qBefore += (uint32_t)cg_popc64(sh.mq[w]); bRank += (uint32_t)cg_popc64(sh.mb[w]); }
            qBefore += (uint32_t)cg_popc64(sh.mq[wq] & lt); bRank += (uint32_t)cg_popc64(sh.mb[wq] & lt);
        }
        const unsigned long long done = (batchEpoch << 24) | GEN_STAMP_COMMITTED;
        const bool more = !endB;   // another round of this batch follows: it reads these
        if (type == 'B') {
            // handle allocation: free stack first (deterministic by rank), then bump
            const uint32_t fc = c_fc;
            // the top of the stack is what this launch's flush pushed, still in LDS
            uint32_t hb;
            if (bRank < fc) { const uint32_t fi = fc - 1u - bRank; hb = (fi >= c_flushBase && fi - c_flushBase < c_flushM) ? sh.fh[fi - c_flushBase] : ((fi < c_flushBase && c_flushBase - 1u - fi < 16u) ? sh.freeTop[c_flushBase - 1u - fi] : S.freeHandles[fi]); }
            else hb = c_handleHi + (bRank - fc);
            const uint32_t idx = nR + bRank;
            if (hb >= S.atomCap || idx >= S.atomCap) { gs->error = GAPS_ERR_ATOM_CAP; hb = 0; }
            S.vec[idx] = hb;
            AtomRec n; n.pos = pos; n.lpos = lposB; n.rpos = rposB; n.left = hl; n.right = hr; n.mass = 0.f; n.rmass = rmassB; n.idx = idx; n.pad0 = 0;
            S.atoms[hb] = n;
            h1 = hb;
            // splice: the neighbours' links and the copies they cache of the new atom (its mass is 0 until the evaluation sets it)
            if (hl != CG_NONE) { S.atoms[hl].right = hb; S.atoms[hl].rpos = pos; S.atoms[hl].rmass = 0.f; } else { sh.g.front = hb; if (endB) gs->front = hb; }      // (the helper's write-back leaves this word alone: frontPending)
            if (hr != CG_NONE) { S.atoms[hr].left = hb; S.atoms[hr].lpos = pos; }
            if (flags & GEN_F_NEWHEAD) S.binHead[bin] = hb;
            if (flags & GEN_F_BINEMPTY) {
                cg_atomic_or_u64(&S.bits0[bin >> 6], 1ull << (bin & 63u));
                if (flags & GEN_F_WORDZERO) { const uint32_t wa = bin >> 6, wb = wa >> 6, wc = wb >> 6; cg_atomic_or_u64(&S.bits1[wb], 1ull << (wa & 63u)); cg_atomic_or_u64(&S.bits2[wc], 1ull << (wb & 63u)); }
            }
            if (more) { S.rowStamp[r1] = done; S.atomStamp[hb] = done; S.atomDest[hb] = 0ull; }
        } else if (type == 'D') {
            if (more) { S.rowStamp[r1] = done; S.atomStamp[h1] = done; S.atomDest[h1] = 0ull; }
        } else if (type == 'M') {
            if (flags & GEN_F_INLINE) atom_set_pos(S, h1, hl, hr, pos);       // domain.move, same bin
            else if (more) { S.rowStamp[r1] = done; S.rowStamp[r2] = done; S.atomStamp[h1] = done; S.atomDest[h1] = pos; }
        } else {
            if (flags & GEN_F_INLINE) { if (flags & GEN_F_APPLY) { atom_set_mass(S, h1, hl, nm1); atom_set_mass(S, h2, (hr != CG_NONE) ? h1 : CG_NONE, nm2); } }
            else if (more) { S.rowStamp[r1] = done; S.rowStamp[r2] = done; }
        }
        if (queued) {
            const uint32_t slot = c_qlen + qBefore;
            if (slot >= S.queueCap) gs->error = GAPS_ERR_QUEUE_CAP;
            else {
                if (c_traceOn && type == 'E') i2 = S.atoms[h2].idx;         // the partner's index: traces only
                PropRec p; p.pos = (type == 'M') ? pos : 0ull; p.rng = rng; p.h1 = h1; p.h2 = h2; p.i1 = i1; p.i2 = i2;
                p.r1 = r1; p.c1 = c1; p.r2 = r2; p.c2 = c2; p.type = type; p.batch = 0; p.pad[0] = p.pad[1] = p.pad[2] = 0;
                const bool two = type == 'M' || type == 'E';
                p.gibbs = (gib1 > 0u ? 1u : 0u) | ((two && gib2 > 0u) ? 2u : 0u);
                p.m1 = (type == 'B') ? 0.f : a.mass; p.m2 = (type == 'E') ? m2x : 0.f;
                p.old1 = old1; p.old2 = two ? old2 : 0.f; p.curPos = (type == 'M') ? cpos : 0ull;
                c.queueOut[slot] = p;
                if (c_traceOn) { const uint32_t ti = c_traceCount + slot; if (ti < c_traceCap) { p.batch = c_nBatches; S.trace[ti] = p; } }
            }
        }
    }
    if (!endB && ldsRound) {
        // Another round of this batch follows and this one kept its conflict sets in the LDS table.  What the next round may find there
        // is what the stamp tables would show it: rows and atoms in use by COMMITTED attempts (their ordinals are smaller than every
        // ordinal of the next window) -- nothing of the attempts behind the cut, which are drawn again, and no gap / same-bin marks at
        // all (the domain the next round reads already holds the committed births and same-bin moves).  A value word holds the smallest
        // registrant, so whoever finds its own ordinal there empties the word; a committed attempt is smaller than every attempt behind
        // the cut, so its "in use" word survives whoever else registered under it.
        if (live) {
            uint32_t *words = &sh.bval[0].used;
            const bool behind = !(ct < stopT);
            if (rf0 != 0u || behind) cg_atomic_cas_u32(&words[4u * rs0 + rf0], gord, GEN_TAB_EMPTY);
            if (rf1 != 0u || behind) cg_atomic_cas_u32(&words[4u * rs1 + rf1], gord, GEN_TAB_EMPTY);
            if (rf2 != 0u || behind) cg_atomic_cas_u32(&words[4u * rs2 + rf2], gord, GEN_TAB_EMPTY);
        }
        // a committed birth's atom is in use (mUsedAtoms.insert, ProposalQueue.cpp:183): inside its own window the gap mark says so,
        // from the next round on the atom is an ordinary neighbour
        if (commit && type == 'B' && roundNo + 1u <= (uint32_t)GEN_LDS_ROUNDS) {
            const uint32_t sb = gen_tab_claim<WIN>(sh, h1);
            cg_atomic_min_u32(&sh.bval[sb].used, gord);
        }
    }
    GEN_TS(21);
    if (endB) { GEN_TS(22); GEN_RT(5); GEN_RT_DUMP(); { const bool ts_ok = c.e_prevQ >= 140u && c.remaining >= 512u && GEN_TS_ROUND_OK(roundNo); (void)ts_ok; GEN_TS_DUMP_WAVE(); } }
    return endB;
}

// ---- chained launch (chain_kernel.h): the generator applies the previous batch's decisions itself -----------------------------------------
// One launch evaluates batch n (its other workgroups) and generates batch n + 1 (this workgroup).  The evaluation workgroups write
// nothing the generator reads except one pair of tagged granules per proposal -- {code, traffic units} and one float -- and the
// generator's lane q carries the decision out on the atomic domain and the factor matrix: the stores the evaluation's writer thread
// makes in the two-launch form (eval_kernel.h: atom_set_mass, eval_store_matrix, eval_domain_move, eval_cache_erase), the same values
// from the same operations.  Everything those stores need that does not depend on the decision -- the queue record, the atom's record
// (its links name the holders of the cached copies, as the evaluation looks them up when it runs), the old bin's head and the bitmap's
// upper words of a move -- is fetched while the evaluation workgroups still run.
// The lane's part of the hand-over is split in two.  chain_fetch (before the wait) turns the record into ADDRESSES and old values: every
// word a decision can rewrite, as a pointer held in vector registers (null: nothing to write there).  chain_apply (behind the wait, on the
// decide -> generate chain) only computes the new values and stores -- no field of the sampler's record is read there: the compiler
// re-loads such fields through the scalar cache wherever they are used, and two dozen of those loads, each waited for, in the four
// type branches a wave walks through one after the other cost the first version 3 k cycles per launch.
struct ChainItem {
    uint32_t type; float m1, m2, old1, old2; uint64_t pos; unsigned long long eraseEntry;
    float *mass1, *rm1, *mass2, *rm2;        // atoms[h1].mass and the copy its left neighbour caches; exchange: the same for the partner
    float *mat1, *mat2; uint32_t *col1, *col2;      // mMatrix(r1,c1), mMatrix(r2,c2), the columns' counts of positive entries
    // move (ConcurrentAtomicDomain.cpp:126-132 across bins, as eval_domain_move decides it: the atom's own record is current -- births queued
    // after the move may have changed the links, nothing moves next to a moving atom, ProposalQueue.cpp:167,218)
    uint64_t *pos1, *rposL, *lposR;          // atoms[h1].pos, atoms[left].rpos, atoms[right].lpos
    uint32_t mb1, mb2;                       // the move's old and new bin (sh.dirty marks)
    uint32_t *head1, *head2; uint32_t head1Val, h1;      // old bin's head word (null: the atom is not the head) and what it becomes; new bin's head word (null: stays)
    unsigned long long *b0clr, *b0set, *b1set, *b2set; uint32_t bit1, bit2, bit1w, bit2w;      // bitmap words (null: nothing to do) and bit numbers
    // what the decision touches, by name (the notes for the window drawn ahead): the neighbours, the partner and its left neighbour, the
    // two matrix cells (bins), the atom's slot in the unsorted vector
    uint32_t hL, hR, h2, l2, cell1, cell2, idx;
    // sparse model (HybridMatrix: row copy, column copy with its epsilon rule and flag word, flagged count -- sp_change_matrix / sp_safely_change_matrix)
    uint32_t sparse; float *rows1, *rows2; unsigned long long *fl1, *fl2; unsigned long long fbit1, fbit2; float colv1, colv2; uint32_t flg1, flg2;
};
CG_DEVICE void chain_item_clear(ChainItem &it)
{
    it.type = 0; it.m1 = 0.f; it.m2 = 0.f; it.old1 = 0.f; it.old2 = 0.f; it.pos = 0; it.eraseEntry = 0ull;
    it.mass1 = nullptr; it.rm1 = nullptr; it.mass2 = nullptr; it.rm2 = nullptr; it.mat1 = nullptr; it.mat2 = nullptr; it.col1 = nullptr; it.col2 = nullptr;
    it.mb1 = 0; it.mb2 = 0;
    it.pos1 = nullptr; it.rposL = nullptr; it.lposR = nullptr; it.head1 = nullptr; it.head2 = nullptr; it.head1Val = CG_NONE; it.h1 = 0;
    it.b0clr = nullptr; it.b0set = nullptr; it.b1set = nullptr; it.b2set = nullptr; it.bit1 = 0; it.bit2 = 0; it.bit1w = 0; it.bit2w = 0;
    it.hL = CG_NONE; it.hR = CG_NONE; it.h2 = CG_NONE; it.l2 = CG_NONE; it.cell1 = 0; it.cell2 = 0; it.idx = 0;
    it.sparse = 0; it.rows1 = nullptr; it.rows2 = nullptr; it.fl1 = nullptr; it.fl2 = nullptr; it.fbit1 = 0ull; it.fbit2 = 0ull; it.colv1 = 0.f; it.colv2 = 0.f; it.flg1 = 0; it.flg2 = 0;
}
// what the second trip brings: the atom's record, the partner's left link, a move's old bin head and upper bitmap words
struct ChainMid { AtomRec a; uint32_t l2, head1, b1, b2; unsigned long long x1, x2;
                  float colv1, colv2; unsigned long long fw1, fw2; };      // sparse model: the column copy's entries and their flag words (sp_cell_load)
CG_DEVICE ChainMid chain_fetch_mid(const SamplerDev &S, const PropRec &p, const bool sparse)
{
    // every lane issues every load (a lane without a proposal, or of another type, reads harmless words: handle 0, bin 0): loads inside
    // divergent branches made the compiler wait for the whole trip where the branches join, before the work meant to run under it
    ChainMid m;
    const uint32_t hE = p.type == 'E' ? p.h2 : p.h1;
    m.b1 = gen_bin_of(S, p.curPos); m.b2 = gen_bin_of(S, p.pos);
    const uint32_t w0 = m.b2 >> 6, w1 = w0 >> 6, w2 = w1 >> 6;
    m.a = S.atoms[p.h1];
    m.l2 = S.atoms[hE].left;
    m.head1 = S.binHead[m.b1];
    m.x1 = S.bits1[w1]; m.x2 = S.bits2[w2];
    m.colv1 = 0.f; m.colv2 = 0.f; m.fw1 = 0ull; m.fw2 = 0ull;
    if (sparse) {      // (wave-uniform) the HybridMatrix column copy and its flags: rows are proposal-exclusive for the whole batch, so what is read here is what the decision finds
        m.colv1 = S.mat[(size_t)p.c1 * S.Mpad + p.r1]; m.fw1 = S.mflags[(size_t)p.c1 * S.Mw + (p.r1 >> 6)];
        m.colv2 = S.mat[(size_t)p.c2 * S.Mpad + p.r2]; m.fw2 = S.mflags[(size_t)p.c2 * S.Mw + (p.r2 >> 6)];
    }
    return m;
}
CG_DEVICE void chain_fetch_build(const SamplerDev &S, const PropRec &p, const ChainMid &m, ChainItem &it, const bool sparse);
CG_DEVICE void chain_fetch(const SamplerDev &S, const PropRec *queueRd, uint32_t q, ChainItem &it, const bool sparse)
{
    const PropRec p = queueRd[q];
    const ChainMid m = chain_fetch_mid(S, p, sparse);
    chain_fetch_build(S, p, m, it, sparse);
}
CG_DEVICE void chain_fetch_build(const SamplerDev &S, const PropRec &p, const ChainMid &m, ChainItem &it, const bool sparse)
{
    chain_item_clear(it);
    it.type = p.type; it.m1 = p.m1; it.m2 = p.m2; it.old1 = p.old1; it.old2 = p.old2; it.pos = p.pos; it.h1 = p.h1;
    it.eraseEntry = ((unsigned long long)(p.r1 * S.K + p.c1) << 32) | (unsigned long long)p.h1;
    const AtomRec a = m.a;
    it.hL = a.left; it.hR = a.right; it.idx = a.idx; it.cell1 = p.r1 * S.K + p.c1; it.cell2 = p.r2 * S.K + p.c2; it.h2 = p.h2; it.l2 = m.l2;
    it.mass1 = &S.atoms[p.h1].mass; it.rm1 = a.left != CG_NONE ? &S.atoms[a.left].rmass : nullptr;
    it.mat1 = &S.mat[(size_t)p.c1 * S.Mpad + p.r1]; it.col1 = &S.colPos[p.c1];
    const bool two = p.type == 'M' || p.type == 'E';
    if (two) { it.mat2 = &S.mat[(size_t)p.c2 * S.Mpad + p.r2]; it.col2 = &S.colPos[p.c2]; }
    if (sparse) {
        it.sparse = 1u;
        it.rows1 = &S.rows[(size_t)p.r1 * S.Kpad + p.c1]; it.fl1 = &S.mflags[(size_t)p.c1 * S.Mw + (p.r1 >> 6)]; it.fbit1 = 1ull << (p.r1 & 63u);
        it.colv1 = m.colv1; it.flg1 = (uint32_t)((m.fw1 >> (p.r1 & 63u)) & 1ull);
        if (two) {
            it.rows2 = &S.rows[(size_t)p.r2 * S.Kpad + p.c2]; it.fl2 = &S.mflags[(size_t)p.c2 * S.Mw + (p.r2 >> 6)]; it.fbit2 = 1ull << (p.r2 & 63u);
            it.colv2 = m.colv2; it.flg2 = (uint32_t)((m.fw2 >> (p.r2 & 63u)) & 1ull);
        }
    }
    if (p.type == 'E') { const uint32_t l2 = m.l2; it.mass2 = &S.atoms[p.h2].mass; it.rm2 = l2 != CG_NONE ? &S.atoms[l2].rmass : nullptr; }
    if (p.type == 'M') {
        const uint32_t b1 = m.b1, b2 = m.b2;
        it.mb1 = b1; it.mb2 = b2;
        const uint32_t head1 = m.head1;
        const uint32_t w0 = b2 >> 6, w1 = w0 >> 6, w2 = w1 >> 6;
        const unsigned long long x1 = m.x1, x2 = m.x2;
        it.pos1 = &S.atoms[p.h1].pos; it.rposL = a.left != CG_NONE ? &S.atoms[a.left].rpos : nullptr; it.lposR = a.right != CG_NONE ? &S.atoms[a.right].lpos : nullptr;
        if (head1 == p.h1) {      // the old bin loses its lowest atom: the right neighbour takes over if it lies in the same bin, else the bin is empty
            it.head1 = &S.binHead[b1];
            if (a.right != CG_NONE && gen_bin_of(S, a.rpos) == b1) it.head1Val = a.right; else { it.head1Val = CG_NONE; it.b0clr = &S.bits0[b1 >> 6]; it.bit1 = b1 & 63u; }
        }
        if (a.left == CG_NONE || gen_bin_of(S, a.lpos) != b2) it.head2 = &S.binHead[b2];
        it.b0set = &S.bits0[w0]; it.bit2 = b2 & 63u;      // bm_set, the upper levels' words read ahead
        if (!((x1 >> (w0 & 63u)) & 1ull)) { it.b1set = &S.bits1[w1]; it.bit1w = w0 & 63u; }
        if (!((x2 >> (w1 & 63u)) & 1ull)) { it.b2set = &S.bits2[w2]; it.bit2w = w1 & 63u; }
    }
}
// mMatrix entry = newv with the per-column count of positive entries (eval_store_matrix)
CG_DEVICE void chain_store_matrix(float *cell, uint32_t *col, float oldv, float newv)
{
    *cell = newv;
    const bool was = oldv > 0.f, is = newv > 0.f;
    if (was != is) { if (is) cg_atomic_add_u32(col, 1u); else cg_atomic_sub_u32(col, 1u); }
}
// sparse model: row copy = rowNew; column copy = colNew, or 0 with the flag cleared when colNew < epsilon (sp_store_col)
CG_DEVICE void chain_store_hybrid(float *rowCell, float *colCell, unsigned long long *flagWord, unsigned long long bit, uint32_t *colCount, float rowNew, float colNew, bool wasFlagged)
{
    *rowCell = rowNew;
    const bool zero = colNew < GAPS_EPSILON;
    if (zero) {
        if (wasFlagged) { (void)cg_atomic_and_u64(flagWord, ~bit); (void)cg_atomic_sub_u32(colCount, 1u); }
        *colCell = 0.f;
    } else {
        if (!wasFlagged) { (void)cg_atomic_or_u64(flagWord, bit); (void)cg_atomic_add_u32(colCount, 1u); }
        *colCell = colNew;
    }
}
// Carries the decision out (the stores of eval_kernel.h's writer thread: atom_set_mass, eval_store_matrix, eval_domain_move); returns
// whether the atom goes to the erase cache.  AsynchronousGibbsSampler.h:127-144 birth, :148-180 death / rebirth, :184-196 move, :201-219 exchange.
CG_DEVICE bool chain_apply(const ChainItem &it, uint32_t code, float val)
{
    const uint32_t tB = it.type == 'B', tD = it.type == 'D', tM = it.type == 'M', tE = it.type == 'E';
    const bool app = code == CHAIN_APPLY, era = code == CHAIN_ERASE;
    // new masses: B: val, D: the rebirth mass val, E: m1 + val and m2 - val
    const float n1 = tE ? it.m1 + val : val, n2 = it.m2 - val;
    // new matrix entries (safelyChangeMatrix: gm_max(old + delta, 0); changeMatrix for a birth and a move's destination)
    float d1 = tB ? val : (tD ? (val - it.m1) : (tM ? -it.m1 : (n1 - it.m1)));
    d1 = (era && tD) ? -1.f * it.m1 : d1;
    const float s1 = it.old1 + d1;
    const float nv1 = tB ? s1 : gm_max(s1, 0.f);
    const float s2 = it.old2 + (tM ? it.m1 : (n2 - it.m2));
    const float nv2 = tM ? s2 : gm_max(s2, 0.f);
    const bool doMat1 = app || (era && tD != 0u), doMat2 = app && (tM | tE) != 0u;
    const bool doMass1 = app && tM == 0u, doMass2 = app && tE != 0u;
    if (doMass1) { *it.mass1 = n1; if (it.rm1) *it.rm1 = n1; }
    if (doMass2) { *it.mass2 = n2; if (it.rm2) *it.rm2 = n2; }
    if (it.sparse) {
        // the HybridMatrix entries (sparse_kernels.h: sp_change_matrix for a birth and a move's destination, sp_safely_change_matrix elsewhere):
        // the row copy takes the new value; the column copy the new value -- for changeMatrix its OWN old value plus the change -- or zero
        // below epsilon, with its flag and the column's flagged count (HybridVector.cpp:55-86)
        if (doMat1) { const float colNew = tB ? it.colv1 + d1 : nv1; chain_store_hybrid(it.rows1, it.mat1, it.fl1, it.fbit1, it.col1, nv1, colNew, it.flg1 != 0u); }
        if (doMat2) { const float colNew = tM ? it.colv2 + it.m1 : nv2; chain_store_hybrid(it.rows2, it.mat2, it.fl2, it.fbit2, it.col2, nv2, colNew, it.flg2 != 0u); }
    } else {
        if (doMat1) chain_store_matrix(it.mat1, it.col1, it.old1, nv1);
        if (doMat2) chain_store_matrix(it.mat2, it.col2, it.old2, nv2);
    }
    if (app && tM != 0u) {
        *it.pos1 = it.pos; if (it.rposL) *it.rposL = it.pos; if (it.lposR) *it.lposR = it.pos;
        if (it.head1) *it.head1 = it.head1Val;
        if (it.b0clr) cg_atomic_and_u64(it.b0clr, ~(1ull << it.bit1));
        if (it.head2) *it.head2 = it.h1;
        cg_atomic_or_u64(it.b0set, 1ull << it.bit2);
        if (it.b1set) cg_atomic_or_u64(it.b1set, 1ull << it.bit1w);
        if (it.b2set) cg_atomic_or_u64(it.b2set, 1ull << it.bit2w);
    }
    return era;
}

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
#pragma unroll
        for (uint32_t k = 0; k < ROUNDS; ++k) { if (!spare && ((k + 1u) * TPB <= UNITS || t + k * TPB < UNITS)) tab[k * TPB] = none; }
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
        } else {
            // ---- the appliers: one proposal per lane and pass -- wait for its two granules (read past this workgroup's caches until both
            // carry the batch's tag), note an erased atom in the erase cache, carry the decision out, note what changed
            uint32_t unitAcc = 0;
            for (uint32_t base = 0; base < e_prevQ; base += NA) {
                const uint32_t q = base + al;
                bool have = applying && q < e_prevQ;
                if (base) { chain_item_clear(it); if (have) chain_fetch(S, hot.queueRd, q, it, isSparse); }      // (a queue longer than the applier lanes: the batch after a generator launch of two rounds)
                // where the notes of this proposal go, whatever is decided (the hashes ahead of the wait)
                const GenNotePos nH1 = gen_note_pos<GEN_DIRTY_ATOMS>(it.h1), nHL = gen_note_pos<GEN_DIRTY_ATOMS>(it.hL), nHR = gen_note_pos<GEN_DIRTY_ATOMS>(it.hR),
                                 nH2 = gen_note_pos<GEN_DIRTY_ATOMS>(it.h2), nL2 = gen_note_pos<GEN_DIRTY_ATOMS>(it.l2), nIdx = gen_note_pos<GEN_DIRTY_ATOMS>(~it.idx),
                                 nC1 = gen_note_pos<GEN_DIRTY_CELLS>(it.cell1), nC2 = gen_note_pos<GEN_DIRTY_CELLS>(it.cell2);
                const unsigned long long *gr = hot.grans + (size_t)q * CHAIN_GRAN_STRIDE;
                unsigned long long g0 = 0ull, g1 = 0ull; uint32_t spins = 0;
                for (;;) {
                    if (have) { g0 = cg_load_l2_u64(&gr[0]); g1 = cg_load_l2_u64(&gr[1]); }
                    const bool ok = !have || ((uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag);
                    if (cg_ballot(!ok) == 0ull) break;
                    // bounded (platform.h: two seconds at least), never a hang.  A bound that is hit applies NOTHING: a granule without this batch's
                    // tag is an older batch's decision -- the lane drops its proposal, the error word ends the update on the host (the session is
                    // then marked unusable: its domain lacks decisions) and the workgroup leaves behind the barrier below without generating
                    // (round 6: the dropped proposal is marked, the batch is completed by the host once the launch has ended -- chain_recover_kernel --
                    // and the update goes on with two launches per batch)
                    if (cg_poll_expired(++spins)) { if (!ok) { have = false; S.queueUnits[q] = CHAIN_DROPPED_MARK(tag); } if ((t & 63u) == 0u) { gs->error = GAPS_ERR_SPIN; sh.spinFail = 1u; } break; }
                    cg_poll_pause();
                }
#if defined(GEN_TEST_SPIN_FAIL_EPOCH)
                // test-only variant of the emulator build (whose workgroups run one after the other: nothing ever waits): at one batch every third
                // lane gives up as if its decision had not arrived -- the launch ends the way a lost hand-over ends it
                if (cg_ballot(have && tag == (uint32_t)GEN_TEST_SPIN_FAIL_EPOCH && (q % 3u) == 1u) != 0ull) {
                    if (have && (q % 3u) == 1u) { have = false; S.queueUnits[q] = CHAIN_DROPPED_MARK(tag); }
                    if ((t & 63u) == 0u) { gs->error = GAPS_ERR_SPIN; sh.spinFail = 1u; }
                }
#endif
                GEN_TS(33);
                if (base == 0u) GEN_RT_AT(3, WIN);
                const uint32_t code = have ? ((uint32_t)g0 & 0xFFu) : CHAIN_NONE;
                if (have) unitAcc += ((uint32_t)g0 >> 8) << (it.sparse ? 5u : 0u);      // (dense: units of 4N bytes; sparse: bytes / 32, GenScalars::evalBytes counts bytes there)
                // erase cache (ConcurrentAtomicDomain.cpp:62-69): one slot per erased atom, in any order -- the flush sorts by position.
                // (Before the stores: what the barrier below waits for is LDS traffic only.)
                const bool er = have && code == CHAIN_ERASE, ap = have && code == CHAIN_APPLY;
                // bitmap words whose bits or bins' heads this decision (or the flush, for an erased atom) changes: the births drawn ahead check them
                if (er) gen_mark_dirty(sh.dirty, it.cell1);
                if (ap && it.type == 'M') { gen_mark_dirty(sh.dirty, it.mb1); gen_mark_dirty(sh.dirty, it.mb2); }
                // atom records whose fields change: the atom's own (mass / position) and the neighbours that cache copies of them; an erased
                // atom's neighbours are relinked by the flush.  Matrix cells that are rewritten.
                if (er || ap) {
                    gen_note_set(sh.dAtom, nH1);
                    if (it.hL != CG_NONE) gen_note_set(sh.dAtom, nHL);
                    if ((er || it.type == 'M') && it.hR != CG_NONE) gen_note_set(sh.dAtom, nHR);
                    if (ap && it.type == 'E') { gen_note_set(sh.dAtom, nH2); if (it.l2 != CG_NONE) gen_note_set(sh.dAtom, nL2); }
                    if (er) {      // (the vector slot the flush refills from the tail; the records the flush rewrites: the erased atom's and its neighbours')
                        gen_note_set(sh.dAtom, nIdx);
                        gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.h1));
                        if (it.hL != CG_NONE) gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.hL));
                        if (it.hR != CG_NONE) gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.hR));
                    }
                    if (ap || it.type == 'D') gen_note_set(sh.dCell, nC1);
                    if (ap && (it.type == 'M' || it.type == 'E')) gen_note_set(sh.dCell, nC2);
                }
                const unsigned long long em = cg_ballot(er);
                if (em) {
                    const uint32_t cntE = (uint32_t)cg_popc64(em);
                    uint32_t b0 = 0;
                    if ((t & 63u) == 0u) b0 = cg_atomic_add_u32(&sh.eraseN, cntE);
                    b0 = cg_wave_bcast_u32(b0, 0);
                    if (er) {
                        const uint32_t k = b0 + (uint32_t)cg_popc64(em & ((1ull << (t & 63u)) - 1ull));
                        const unsigned long long e = it.eraseEntry;
                        if (k < (uint32_t)FLUSH_MAX) sh.eraseTmp[k] = e;
                        if (k < eraseCap) eraseList[k] = e; else gs->error = GAPS_ERR_ERASE_CAP;
                    }
                }
                if (have) chain_apply(it, code, gm_u2f((uint32_t)g1));
#if defined(COGAPS_EMUL)
                if (have && base) cg_atomic_add_u64(&gs->prof[6], 1ull);      // test-only build: decisions carried out in a pass beyond the first (a queue longer than the applier lanes)
#endif
            }
            const uint32_t waveUnits = cg_wave_sum_u32(unitAcc);
            if ((t & 63u) == 0u && waveUnits) cg_atomic_add_u32(&sh.unitSum, waveUnits);
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
