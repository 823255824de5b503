// gen_flush.h -- flushEraseCache by the helper wave, and the helper wave itself (table presets, round bookkeeping, write-back).  (part of the generator: included from gen_populate.h, which documents the method)
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
