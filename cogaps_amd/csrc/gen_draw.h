// gen_draw.h -- the populate-phase draws of one attempt, the classification ahead of the decisions, the notes and the validation of a window drawn ahead.  (part of the generator: included from gen_populate.h, which documents the method)
#pragma once
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
    // (A window drawn AHEAD reads the domain while the applier waves of the same workgroup carry out the previous batch's decisions: a move
    // that empties a bin stores the bin's head and clears its bitmap bit in two stores, and a birth of this window can read the bit still
    // set and the head already gone.  Whatever such a lane read is noted and the lane draws again -- but its next load must not follow the
    // missing handle: found by the soak beside a foreign kernel, round 6, where the decisions arrive before the window is drawn -- a
    // memory fault of the round-5 build, profiles/r06_soak_beside_a_foreign_kernel.json)
    if (AHEAD && isB && !slowB && v2 == CG_NONE) { d.redo = true; slowB = true; }
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
