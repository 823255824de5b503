// gen_round.h -- one round of a batch for the attempt lanes: draws, registration, look-ups, commit.  (part of the generator: included from gen_populate.h, which documents the method)
#pragma once
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
