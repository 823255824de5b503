// gen_populate.h -- the generator kernel body (included from gen_kernel.h, which documents the method).
//
// Lane assignment: attempts are classified in attempt order (one lane = one attempt), then SORTED BY
// TYPE so that a wavefront executes (mostly) one of the birth/death, move or exchange code paths instead
// of all four under divergence -- the attempt ordinal `ct` travels with the lane and is what the
// conflict stamps compare.  Per-attempt prefix counts (queue slot, birth rank) come from LDS bit masks.
#pragma once

template <int WIN>
CG_DEVICE void gen_body(const SamplerDev &S)
{
    CG_SHARED GenShared<WIN> sh;
    const unsigned t = cg_tid();
    GenScalars *gs = S.gs;

    unsigned long long prof_last = cg_clock(); (void)prof_last;
    // k-step PCG jumps for this lane's (u1,u2): k = 2t, or 2(t-1) when attempt 0 replays cached values
    const uint64_t jm0 = S.lcgMul[2u * t], ji0 = S.lcgInc[2u * t];
    const uint64_t jm1 = S.lcgMul[t ? 2u * (t - 1u) : 0u], ji1 = S.lcgInc[t ? 2u * (t - 1u) : 0u];
    // generator scalars the flush does not touch: read them while the flush runs
    uint64_t g_qrng = 0, g_epoch = 0; uint32_t g_nSteps = 0, g_nDone = 0, g_cached = 0; float g_u1 = 0.f, g_u2 = 0.f;
    if (t == 0) { g_qrng = gs->qrng; g_epoch = gs->batchEpoch; g_nSteps = gs->nSteps; g_nDone = gs->nDone; g_cached = gs->useCached; g_u1 = gs->u1; g_u2 = gs->u2; }
    {   // roofline bookkeeping: add up the traffic units the evaluation kernel left per queue slot
        const uint32_t prevQ = gs->qlen;
        uint32_t u = 0;
        for (uint32_t q = t; q < prevQ; q += WIN) u += S.queueUnits[q];
        if (u) cg_atomic_add_u64(&gs->evalBytes, (unsigned long long)u * 4ull * S.N);
        if (t == 0 && prevQ) cg_atomic_add_u64(&gs->evalProps, (unsigned long long)prevQ);
    }
    gen_flush<WIN>(S, sh);
    GEN_PROF(0);

    if (t == 0) {
        sh.done = (g_nDone >= g_nSteps) ? 1u : 0u;
        sh.batchEpoch = g_epoch + 1;
        sh.roundNo = 0;
        sh.qrngRound = g_qrng;
        const uint32_t n = gs->nAtoms;
        sh.nR = n; sh.minAtoms = n;
        sh.processed = 0; sh.qlen = 0; sh.skip = g_cached ? 1u : 0u;
        sh.remaining = g_nSteps - g_nDone;
        sh.u1c = g_u1; sh.u2c = g_u2; sh.updBase = g_nDone;
    }
    cg_sync();
    if (sh.done) { if (t == 0) { gs->qlen = 0; gs->batchNproc = 0; gs->updateFlushed = 1; } return; }

    const uint64_t batchEpoch = sh.batchEpoch;
    const uint32_t updBase = sh.updBase;        // attempts consumed by earlier batches of this update
    const uint32_t K = S.K;

    for (;;) {
        // ------------------------------------------------------------------ round set-up
        if (t == 0) { sh.roundNo += 1; sh.stopKey = 0xFFFFFFFFu; if (sh.roundNo >= 4094u) gs->error = GAPS_ERR_SPIN; }
        if (t < (unsigned)(WIN / 64)) { sh.mq[t] = 0ull; sh.mb[t] = 0ull; sh.md[t] = 0ull; }
        cg_sync();
        const uint32_t roundNo = sh.roundNo;
        const uint32_t nR = sh.nR, minR = sh.minAtoms, skip = sh.skip, processed = sh.processed;
        const uint32_t left_ = sh.remaining - processed;
        const uint32_t winN = left_ < (uint32_t)WIN ? left_ : (uint32_t)WIN;

        // ------------------------------------------------------------------ A1 (lane = attempt): (u1,u2), B/D/M/E
        {
            const bool active = t < winN;
            if (active) sh.seed[t] = S.seeds[updBase + processed + t];     // consumed after the type sort
            float u1 = 0.f, u2 = 0.f;
            uint32_t guess = GEN_T_NONE;
            if (active) {
                if (skip && t == 0) { u1 = sh.u1c; u2 = sh.u2c; }
                else {
                    uint64_t s = (skip ? jm1 : jm0) * sh.qrngRound + (skip ? ji1 : ji0);
                    u1 = pcg_uniform(s); u2 = pcg_uniform(s);
                }
                guess = gen_decide(S, u1, u2, minR, nR);
            }
            sh.u1[t] = u1; sh.u2[t] = u2;
            uint32_t bBefore, dBefore, e3, tB, tD, t3;
            gen_count3<WIN>(sh.wtotA, t, active && guess == 'B', active && guess == 'D', false, bBefore, dBefore, e3, tB, tD, t3);
            uint32_t aflags = 0;
            if (active && (bBefore | dBefore)) {
                // the exact B/D/indeterminate decision depends on how many births / deaths precede this attempt
                const uint32_t exact = (u1 < 0.5f || minR < 2u + dBefore || nR + bBefore < 2u) ? gen_decide(S, u1, u2, (uint64_t)minR - dBefore, (uint64_t)nR + bBefore) : guess;
                if (exact != guess) aflags |= GEN_F_HAZARD;
            }
            if (active && !(aflags & GEN_F_HAZARD) && guess == GEN_T_NONE) aflags |= GEN_F_FAIL;   // indeterminate: batch ends, no seed used
            if (active && aflags) cg_atomic_min_u32(&sh.stopKey, 2u * t + ((aflags & GEN_F_HAZARD) ? 0u : 1u));
            // sort the attempts that go on by code path: births+deaths | moves | exchanges
            const bool go = active && !aflags;
            const bool k0 = go && (guess == 'B' || guess == 'D'), k1 = go && guess == 'M', k2 = go && guess == 'E';
            uint32_t e0, e1, e2, T0, T1, T2;
            gen_count3<WIN>(sh.wtotB, t, k0, k1, k2, e0, e1, e2, T0, T1, T2);
            if (go) {
                const uint32_t slot = k0 ? e0 : (k1 ? T0 + e1 : T0 + T1 + e2);
                sh.perm[slot] = (uint16_t)t;
                sh.info[t] = guess | (bBefore << 8);
            }
            if (t == 0) sh.nWork = T0 + T1 + T2;
        }
        cg_sync();
        GEN_PROF(1);

        // ------------------------------------------------------------------ A2 (lane = sorted slot): populate-phase draws
        const bool go = t < sh.nWork;
        const uint32_t ct = go ? (uint32_t)sh.perm[t] : 0u;          // this lane's attempt ordinal in the window
        const uint32_t info = go ? sh.info[ct] : 0u;
        const uint32_t type = info & 0xFFu, bBefore = info >> 8;
        uint32_t flags = 0;
        const bool isB = go && type == 'B';
        bool pick = go && type != 'B';                 // D/M/E: picks an existing atom
        uint64_t rng = go ? pcg_from_seed(sh.seed[ct]) : 0ull;   // AtomicProposal ctor, ProposalQueue.cpp:12-15
        const uint32_t nT = nR + bBefore;              // domain size this attempt sees
        uint64_t pos = 0, cpos = 0, lbpos = 0, rbpos = 0;
        uint32_t h1 = CG_NONE, h2 = CG_NONE, i1 = CG_NONE, i2 = CG_NONE, hl = CG_NONE, hr = CG_NONE;
        uint32_t r1 = 0, c1 = 0, r2 = 0, c2 = 0; float nm1 = 0.f, nm2 = 0.f;
        uint32_t bin = 0, headBin = 0; unsigned long long w0 = 0;

        // stage 1 ---------------------------------------------------------------------------------
        if (isB) {
            // uniform64(1, L) (Random.cpp:105-123) with the constant range's iPart precomputed
            uint64_t x = pcg_u64(rng);
            while (x >= S.limitL) x = pcg_u64(rng);
            pos = (S.iPartL == 1ull ? x : x / S.iPartL) + 1ull;
            bin = gen_bin_of(S, pos); r1 = bin / K; c1 = bin - r1 * K;
            i1 = nT;
        } else if (pick) {
            i1 = pcg_uniform32(rng, 0u, nT - 1u);
            if (i1 >= nR) { flags |= GEN_F_FAIL; pick = false; }   // an atom born earlier in this window: its row is in use
        }
        uint32_t v1 = CG_NONE;
        if (isB) w0 = S.bits0[bin >> 6];
        if (pick) v1 = S.vec[i1];
        // stage 2 ---------------------------------------------------------------------------------
        bool slowB = false;
        if (isB) {
            const uint32_t bit = bin & 63u;
            if ((w0 >> bit) & 1ull) headBin = bin;
            else {
                flags |= GEN_F_BINEMPTY; if (w0 == 0ull) flags |= GEN_F_WORDZERO;
                const unsigned long long m = (bit == 63u) ? 0ull : (w0 & ~((2ull << bit) - 1ull));
                if (m) headBin = (bin & ~63u) + (uint32_t)cg_ctz64(m); else slowB = true;
            }
        }
        uint32_t v2 = CG_NONE; AtomRec a; a.pos = 0; a.left = CG_NONE; a.right = CG_NONE; a.mass = 0.f; a.idx = 0;
        if (isB && !slowB) v2 = S.binHead[headBin];
        if (pick) { h1 = v1; a = S.atoms[h1]; }
        // stage 3 ---------------------------------------------------------------------------------
        AtomRec b3; b3.pos = 0; b3.left = CG_NONE; b3.right = CG_NONE; b3.mass = 0.f; b3.idx = 0;
        uint64_t lp = 0, rp = 0;
        if (pick) {
            cpos = a.pos;
            const uint32_t b1 = gen_bin_of(S, cpos);
            r1 = b1 / K; c1 = b1 - r1 * K;
            if (type == 'M') { hl = a.left; hr = a.right; }
            else if (type == 'E') { hr = a.right; h2 = (hr != CG_NONE) ? hr : gs->front; }
        }
        if (isB && !slowB) b3 = S.atoms[v2];
        if (pick && type == 'M') { if (hl != CG_NONE) lp = S.atoms[hl].pos; if (hr != CG_NONE) rp = S.atoms[hr].pos; }
        if (pick && type == 'E') b3 = S.atoms[h2];
        // finish ----------------------------------------------------------------------------------
        if (isB) {
            if (!slowB) {
                if (flags & GEN_F_BINEMPTY) { hr = v2; hl = b3.left; flags |= GEN_F_NEWHEAD; }
                else if (b3.pos > pos) { hr = v2; hl = b3.left; flags |= GEN_F_NEWHEAD; }
                else slowB = true;      // walk inside the bin (or position already taken)
            }
            if (slowB) {
                bool occ, nh;
                gen_find_gap(S, pos, bin, &hl, &hr, &occ, &nh);
                while (occ) {           // randomFreePosition retry (ConcurrentAtomicDomain.cpp:46-54)
                    pos = pcg_uniform64(rng, 1ull, S.domainLenU);
                    bin = gen_bin_of(S, pos); r1 = bin / K; c1 = bin - r1 * K;
                    gen_find_gap(S, pos, bin, &hl, &hr, &occ, &nh);
                }
                flags &= ~(GEN_F_BINEMPTY | GEN_F_WORDZERO | GEN_F_NEWHEAD);
                if (nh) flags |= GEN_F_NEWHEAD;
                if (S.binHead[bin] == CG_NONE) { flags |= GEN_F_BINEMPTY; if (S.bits0[bin >> 6] == 0ull) flags |= GEN_F_WORDZERO; }
            }
        } else if (pick) {
            if (type == 'M') {
                if (hl != CG_NONE) { flags |= GEN_F_HASLEFT; lbpos = lp; } else lbpos = 0;
                if (hr != CG_NONE) { flags |= GEN_F_HASRIGHT; rbpos = rp; } else rbpos = S.rboundNone;
                pos = pcg_uniform64(rng, lbpos + 1ull, rbpos - 1ull);
                const uint32_t bin2 = gen_bin_of(S, pos);
                r2 = bin2 / K; c2 = bin2 - r2 * K;
                if (r1 == r2 && c1 == c2) flags |= GEN_F_INLINE;
            } else if (type == 'E') {
                if (hr != CG_NONE) flags |= GEN_F_HASRIGHT;
                rbpos = b3.pos; i2 = b3.idx;
                const uint32_t bin2 = gen_bin_of(S, rbpos);
                r2 = bin2 / K; c2 = bin2 - r2 * K;
                if (r1 == r2 && c1 == c2) {
                    flags |= GEN_F_INLINE;
                    const float m1 = a.mass, m2 = b3.mass;
                    const float newMass = pcg_trunc_gamma_upper(rng, S.luts, m1 + m2, 1.f / S.lambda);
                    const float delta = (m1 > m2) ? newMass - m1 : m2 - newMass;
                    if (m1 + delta > GAPS_EPSILON && m2 - delta > GAPS_EPSILON) { flags |= GEN_F_APPLY; nm1 = m1 + delta; nm2 = m2 - delta; }
                }
            }
        }
        GEN_PROF(2);

        // ------------------------------------------------------------------ B1: register rows / atoms / gaps
        const bool live = go && !(flags & GEN_F_FAIL);
        const bool queuedM = live && type == 'M' && !(flags & GEN_F_INLINE);
        if (go) { sh.cpos[ct] = cpos; sh.pos[ct] = pos; sh.type[ct] = queuedM ? (uint8_t)'M' : (uint8_t)0; }
        if (live) {
            const unsigned long long st = gen_stamp(batchEpoch, roundNo, ct);
            if (type == 'B') { cg_atomic_max_u64(&S.rowStamp[r1], st); cg_atomic_max_u64(&S.gapStamp[hl == CG_NONE ? 0u : hl + 1u], st); }
            else if (type == 'D') { cg_atomic_max_u64(&S.rowStamp[r1], st); cg_atomic_max_u64(&S.atomStamp[h1], st); }
            else if (type == 'M') {
                if (flags & GEN_F_INLINE) cg_atomic_max_u64(&S.inlineStamp[h1], st);
                else { cg_atomic_max_u64(&S.rowStamp[r1], st); cg_atomic_max_u64(&S.rowStamp[r2], st); cg_atomic_max_u64(&S.atomStamp[h1], st); }
            } else {
                if (flags & GEN_F_INLINE) { cg_atomic_max_u64(&S.inlineStamp[h1], st); cg_atomic_max_u64(&S.inlineStamp[h2], st); }
                else { cg_atomic_max_u64(&S.rowStamp[r1], st); cg_atomic_max_u64(&S.rowStamp[r2], st); }
            }
        }
        cg_sync();
        GEN_PROF(3);

        // ------------------------------------------------------------------ B2: probe -- every lane issues the same
        // eleven loads (unused slots read a harmless word), then the per-type logic runs on registers
        if (live) {
            const bool tB = type == 'B', tM = type == 'M', tE = type == 'E', inl = (flags & GEN_F_INLINE) != 0;
            const uint32_t keyL = (hl == CG_NONE) ? 0u : hl + 1u;
            const unsigned long long *p1 = &S.rowStamp[(tM || tE) ? r2 : r1];
            const unsigned long long *p2 = ((tM || tB) && hl != CG_NONE) ? &S.atomStamp[hl] : &S.gapStamp[0];
            const unsigned long long *p3 = ((tM || tB) && hr != CG_NONE) ? &S.atomStamp[hr] : &S.gapStamp[0];
            const unsigned long long *p4 = &S.gapStamp[(tB || tM) ? keyL : (tE ? h1 + 1u : 0u)];
            const unsigned long long *p5 = &S.gapStamp[tM ? h1 + 1u : 0u];
            const unsigned long long *p6 = tM ? &S.inlineStamp[h1] : ((tB && hl != CG_NONE) ? &S.inlineStamp[hl] : ((tE && inl) ? &S.inlineStamp[h1] : &S.gapStamp[0]));
            const unsigned long long *p7 = (tM && hl != CG_NONE) ? &S.inlineStamp[hl] : ((tB && hr != CG_NONE) ? &S.inlineStamp[hr] : ((tE && inl) ? &S.inlineStamp[h2] : &S.gapStamp[0]));
            const unsigned long long *p8 = (tM && hr != CG_NONE) ? &S.inlineStamp[hr] : &S.gapStamp[0];
            const uint64_t *p9 = (tB && hl != CG_NONE) ? &S.atomDest[hl] : &S.atomDest[0];
            const uint64_t *p10 = (tB && hr != CG_NONE) ? &S.atomDest[hr] : &S.atomDest[0];
            const unsigned long long v0 = cg_load_l2_u64(&S.rowStamp[r1]);
            const unsigned long long v1_ = cg_load_l2_u64(p1), v2_ = cg_load_l2_u64(p2), v3_ = cg_load_l2_u64(p3), v4_ = cg_load_l2_u64(p4);
            const unsigned long long v5_ = cg_load_l2_u64(p5), v6_ = cg_load_l2_u64(p6), v7_ = cg_load_l2_u64(p7), v8_ = cg_load_l2_u64(p8);
            const uint64_t d9 = *p9, d10 = *p10;
            bool fail = false, haz = false; uint32_t ix = 0;
            fail = gen_probe(v0, batchEpoch, roundNo, ct, &ix) != 0;                       // row r1 in use
            if (tM || tE) { if (gen_probe(v1_, batchEpoch, roundNo, ct, &ix) != 0) fail = true; }   // row r2 in use
            if (tB) {
                if (gen_probe(v4_, batchEpoch, roundNo, ct, &ix) == 2) haz = true;          // an earlier birth of this window in the same gap
                const uint32_t nb[2] = {hl, hr}; const unsigned long long sa[2] = {v2_, v3_}, si[2] = {v6_, v7_}; const uint64_t dest[2] = {d9, d10};
                for (int k = 0; k < 2; ++k) {
                    if (nb[k] == CG_NONE) continue;
                    // mProposedMoves.overlap(pos): the neighbour has a queued move whose interval covers pos
                    const int u = gen_probe(sa[k], batchEpoch, roundNo, ct, &ix);
                    uint64_t ma = 0, mb = 0; bool mv = false;
                    if (u == 1 && dest[k] != 0ull) { ma = S.atoms[nb[k]].pos; mb = dest[k]; mv = true; }
                    else if (u == 2 && sh.type[ix] == 'M') { ma = sh.cpos[ix]; mb = sh.pos[ix]; mv = true; }
                    if (mv) { const uint64_t lo = ma < mb ? ma : mb, hi = ma < mb ? mb : ma; if (lo < pos && pos < hi) fail = true; }
                    // an earlier same-bin move of this window shifted the neighbour this gap search compared against
                    if (gen_probe(si[k], batchEpoch, roundNo, ct, &ix) == 2) haz = true;
                }
            } else if (tM) {
                if ((hl != CG_NONE && gen_probe(v2_, batchEpoch, roundNo, ct, &ix) != 0) || (hr != CG_NONE && gen_probe(v3_, batchEpoch, roundNo, ct, &ix) != 0)) fail = true;   // mUsedAtoms
                // a birth earlier in this window inside (left, right) is the true neighbour, and it is "used"
                if (gen_probe(v4_, batchEpoch, roundNo, ct, &ix) == 2 || gen_probe(v5_, batchEpoch, roundNo, ct, &ix) == 2) fail = true;
                // an earlier same-bin move/exchange of this window touched the centre or a neighbour: positions stale
                if (gen_probe(v6_, batchEpoch, roundNo, ct, &ix) == 2) haz = true;
                if (hl != CG_NONE && gen_probe(v7_, batchEpoch, roundNo, ct, &ix) == 2) haz = true;
                if (hr != CG_NONE && gen_probe(v8_, batchEpoch, roundNo, ct, &ix) == 2) haz = true;
            } else if (tE) {
                // an earlier birth right of the centre is the true partner (or, for the last atom, a new front())
                if (gen_probe(v4_, batchEpoch, roundNo, ct, &ix) == 2) fail = true;
                if (!(flags & GEN_F_HASRIGHT) && gen_probe(v5_, batchEpoch, roundNo, ct, &ix) == 2) fail = true;
                if (inl) { if (gen_probe(v6_, batchEpoch, roundNo, ct, &ix) == 2 || gen_probe(v7_, batchEpoch, roundNo, ct, &ix) == 2) haz = true; }
            }
            if (haz) flags |= GEN_F_HAZARD; else if (fail) flags |= GEN_F_FAIL;
        }
        if (go && (flags & (GEN_F_HAZARD | GEN_F_FAIL))) cg_atomic_min_u32(&sh.stopKey, 2u * ct + ((flags & GEN_F_HAZARD) ? 0u : 1u));
        cg_sync();
        GEN_PROF(4);

        // ------------------------------------------------------------------ C: commit attempts [0, stopT)
        const uint32_t stopKey = sh.stopKey;
        const uint32_t stopT = (stopKey == 0xFFFFFFFFu) ? winN : (stopKey >> 1);
        const bool stopFail = (stopKey != 0xFFFFFFFFu) && (stopKey & 1u);
        const bool commit = go && ct < stopT;            // every such attempt is live
        const bool queued = commit && (type == 'B' || type == 'D' || !(flags & GEN_F_INLINE));
        if (commit) {
            const unsigned long long bit = 1ull << (ct & 63u);
            if (queued) cg_atomic_or_u64(&sh.mq[ct >> 6], bit);
            if (type == 'B') cg_atomic_or_u64(&sh.mb[ct >> 6], bit);
            if (type == 'D') cg_atomic_or_u64(&sh.md[ct >> 6], bit);
        }
        cg_sync();
        GEN_PROF(5);
        if (commit) {
            uint32_t qBefore = 0, bRank = 0;
            {
                const uint32_t wq = ct >> 6; const unsigned long long lt = (1ull << (ct & 63u)) - 1ull;
                for (uint32_t w = 0; w < wq; ++w) { qBefore += (uint32_t)cg_popc64(sh.mq[w]); bRank += (uint32_t)cg_popc64(sh.mb[w]); }
                qBefore += (uint32_t)cg_popc64(sh.mq[wq] & lt); bRank += (uint32_t)cg_popc64(sh.mb[wq] & lt);
            }
            const unsigned long long done = (batchEpoch << 24) | GEN_STAMP_COMMITTED;
            if (type == 'B') {
                // handle allocation: free stack first (deterministic by rank), then bump
                const uint32_t fc = gs->freeCount;
                uint32_t hb = (bRank < fc) ? S.freeHandles[fc - 1u - bRank] : gs->handleHi + (bRank - fc);
                const uint32_t idx = nR + bRank;
                if (hb >= S.atomCap || idx >= S.atomCap) { gs->error = GAPS_ERR_ATOM_CAP; hb = 0; }
                S.vec[idx] = hb;
                AtomRec n; n.pos = pos; n.left = hl; n.right = hr; n.mass = 0.f; n.idx = idx; n.pad0 = 0; n.pad1 = 0;
                S.atoms[hb] = n;
                h1 = hb;
                if (hl != CG_NONE) S.atoms[hl].right = hb; else gs->front = hb;
                if (hr != CG_NONE) S.atoms[hr].left = hb;
                if (flags & GEN_F_NEWHEAD) S.binHead[bin] = hb;
                if (flags & GEN_F_BINEMPTY) {
                    cg_atomic_or_u64(&S.bits0[bin >> 6], 1ull << (bin & 63u));
                    if (flags & GEN_F_WORDZERO) { const uint32_t wa = bin >> 6, wb = wa >> 6, wc = wb >> 6; cg_atomic_or_u64(&S.bits1[wb], 1ull << (wa & 63u)); cg_atomic_or_u64(&S.bits2[wc], 1ull << (wb & 63u)); }
                }
                S.rowStamp[r1] = done; S.atomStamp[hb] = done; S.atomDest[hb] = 0ull;
            } else if (type == 'D') {
                S.rowStamp[r1] = done; S.atomStamp[h1] = done; S.atomDest[h1] = 0ull;
            } else if (type == 'M') {
                if (flags & GEN_F_INLINE) S.atoms[h1].pos = pos;                  // domain.move, same bin
                else { S.rowStamp[r1] = done; S.rowStamp[r2] = done; S.atomStamp[h1] = done; S.atomDest[h1] = pos; }
            } else {
                if (flags & GEN_F_INLINE) { if (flags & GEN_F_APPLY) { S.atoms[h1].mass = nm1; S.atoms[h2].mass = nm2; } }
                else { S.rowStamp[r1] = done; S.rowStamp[r2] = done; }
            }
            if (queued) {
                const uint32_t slot = sh.qlen + qBefore;
                if (slot >= S.queueCap) gs->error = GAPS_ERR_QUEUE_CAP;
                else {
                    PropRec p; p.pos = (type == 'M') ? pos : 0ull; p.rng = rng; p.h1 = h1; p.h2 = h2; p.i1 = i1; p.i2 = i2;
                    p.r1 = r1; p.c1 = c1; p.r2 = r2; p.c2 = c2; p.type = type; p.pad[0] = p.pad[1] = p.pad[2] = 0;
                    S.queue[slot] = p;
                    if (gs->traceOn) { const uint32_t ti = gs->traceCount + slot; if (ti < gs->traceCap) { p.pad[0] = gs->nBatches; S.trace[ti] = p; } }
                }
            }
        }
        cg_sync();
        GEN_PROF(6);
        // ------------------------------------------------------------------ round bookkeeping
        if (t == 0) {
            uint32_t totQ = 0, totB = 0, totD = 0;
            for (uint32_t w = 0; w < (uint32_t)(WIN / 64); ++w) { totQ += (uint32_t)cg_popc64(sh.mq[w]); totB += (uint32_t)cg_popc64(sh.mb[w]); totD += (uint32_t)cg_popc64(sh.md[w]); }
            if (totB) { const uint32_t fc = gs->freeCount; if (totB <= fc) gs->freeCount = fc - totB; else { gs->freeCount = 0; gs->handleHi += totB - fc; } gs->nAtoms = nR + totB; }
            sh.nR = nR + totB; sh.minAtoms = minR - totD;
            sh.qlen += totQ; sh.processed = processed + stopT;
            const uint32_t attempted = stopT + (stopFail ? 1u : 0u);
            const uint32_t draws = 2u * (attempted - ((skip && attempted) ? 1u : 0u));
            uint64_t jm, ji; pcg_jump_coeffs(draws, jm, ji);
            sh.qrngRound = jm * sh.qrngRound + ji;
            if (attempted) sh.skip = 0;
            sh.stopT = stopT; sh.stopFail = stopFail ? 1u : 0u;
#if defined(GEN_PROFILE)
            cg_atomic_add_u64(&gs->prof[15], 1ull);
#endif
        }
        cg_sync();
        GEN_PROF(7);
        const bool endBatch = sh.stopFail || (sh.processed >= sh.remaining);
        if (endBatch) {
            if (t == 0) {
                gs->qrng = sh.qrngRound;
                if (sh.stopFail) { gs->useCached = 1; gs->u1 = sh.u1[sh.stopT]; gs->u2 = sh.u2[sh.stopT]; }
                else gs->useCached = 0;
                const uint32_t nDone = updBase + sh.processed;
                gs->nDone = nDone;
                gs->qlen = sh.qlen; gs->batchNproc = sh.processed;
                gs->batchEpoch = batchEpoch;
                if (nDone < sh.remaining + updBase) {           // n < nSteps: AsynchronousGibbsSampler.h:97-102
                    const float ns = gs->nQueueSamples + 1.f;
                    float avg = gs->avgQueue;
                    avg *= (ns - 1.f) / ns;
                    avg += (float)sh.qlen / ns;
                    gs->nQueueSamples = ns; gs->avgQueue = avg;
                }
                if (gs->traceOn) {
                    const uint32_t bi = gs->traceBatchCount;
                    if (bi < gs->traceCap) { S.traceBatchNproc[bi] = sh.processed; S.traceBatchQlen[bi] = sh.qlen; }
                    gs->traceBatchCount = bi + 1; gs->traceCount += sh.qlen;
                }
                gs->nBatches += 1;
            }
            return;
        }
    }
}

template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(WIN) gen_kernel(SamplerDev S) { gen_body<WIN>(S); }
