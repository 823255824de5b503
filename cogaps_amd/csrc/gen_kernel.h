// gen_kernel.h -- device-side proposal generator: ProposalQueue::populate + flushEraseCache.
//
// The reference builds a batch serially (atomic/ProposalQueue.cpp:53-76): attempt k of a batch draws
// (u1,u2) from the queue's PCG and one seed from the Xoroshiro seeder, picks atoms / positions, and
// the batch ends at the first conflict.  Every attempt's random inputs are a pure function of its
// ordinal k and of the generator states at batch start (a failed attempt rolls the seeder back one
// step and caches u1,u2), so one workgroup evaluates a WINDOW of attempts speculatively, one lane
// each, against the domain snapshot, then finds the first attempt that (a) genuinely conflicts with
// an earlier one -- the batch ends there, exactly as in the reference -- or (b) read state an earlier
// attempt of the same window modified (a "hazard": the window is cut there and re-run from that
// attempt, which then sees exact state).  Everything before the cut is committed.  The result is
// bit-identical to the serial procedure; tests/test_emul_populate.py checks that against the oracle.
//
// Conflict rules restated from ProposalQueue.cpp: birth :162-187, death :189-207, move :209-248,
// exchange :250-283, type choice :129-160; sets from data_structures/HashSets.cpp.
#pragma once
#include "gaps_state.h"

#if defined(GEN_PROFILE)
#define GEN_PROF(i) do { if (t == 0) { unsigned long long now_ = cg_clock(); gs->prof[i] += now_ - prof_last; prof_last = now_; } } while (0)
#else
#define GEN_PROF(i) do { } while (0)
#endif
#define GEN_T_NONE 0
#define GEN_F_INLINE 1u     // same-bin move / exchange: applied at populate time, not queued
#define GEN_F_FAIL 2u       // genuine conflict or indeterminate B/D: the batch ends here
#define GEN_F_HAZARD 4u     // speculative evaluation unreliable: cut the window here
#define GEN_F_HASRIGHT 8u
#define GEN_F_APPLY 16u     // inline exchange changes the two masses
#define GEN_F_HASLEFT 32u

template <int WIN>
struct GenShared {
    uint64_t rng[WIN], pos[WIN], cpos[WIN], lbpos[WIN], rbpos[WIN];
    uint32_t h1[WIN], h2[WIN], i1[WIN], i2[WIN], hl[WIN], hr[WIN];
    uint32_t r1[WIN], c1[WIN], r2[WIN], c2[WIN];
    float nm1[WIN], nm2[WIN], u1[WIN], u2[WIN];
    uint32_t scan[WIN];
    uint32_t bl_t[WIN], im_t[WIN], ie_t[WIN], qm_t[WIN];
    uint8_t type[WIN], flags[WIN];
    uint64_t fpos[WIN]; uint32_t fh[WIN];     // flush: positions / handles of the erase cache
    // batch / round scalars
    uint64_t qrngRound, batchEpoch, roundEpoch;
    uint32_t nBirths, nInlineM, nInlineE, nQueuedM;
    uint32_t stopKey, needSerialBirths;
    uint32_t nR, minAtoms, processed, qlen, skip, remaining, nBatchMoves, done, stopT, stopFail;
};

CG_DEVICE uint32_t gen_bin_of(const SamplerDev &S, uint64_t pos) { return (uint32_t)(pos / S.binLength); }

// ---- occupancy bitmap: level 0 exact, levels 1/2 monotone "maybe" hints -------------------------
CG_DEVICE void bm_set(const SamplerDev &S, uint32_t bin)
{
    uint32_t w0 = bin >> 6, w1 = w0 >> 6, w2 = w1 >> 6;
    cg_atomic_or_u64(&S.bits0[w0], 1ull << (bin & 63));
    if (!((S.bits1[w1] >> (w0 & 63)) & 1ull)) cg_atomic_or_u64(&S.bits1[w1], 1ull << (w0 & 63));
    if (!((S.bits2[w2] >> (w1 & 63)) & 1ull)) cg_atomic_or_u64(&S.bits2[w2], 1ull << (w1 & 63));
}
CG_DEVICE void bm_clear(const SamplerDev &S, uint32_t bin)
{
    cg_atomic_and_u64(&S.bits0[bin >> 6], ~(1ull << (bin & 63)));
}
// largest set index < i in a one-level bitmap `w` (nw words), ignoring hints; CG_NONE if none
CG_DEVICE uint32_t bm_prev_flat(const unsigned long long *w, uint32_t i)
{
    uint32_t wi = i >> 6, bit = i & 63;
    unsigned long long m = bit ? (w[wi] & ((1ull << bit) - 1ull)) : 0ull;
    for (;;) {
        if (m) return (wi << 6) + 63u - (uint32_t)cg_clz64(m);
        if (wi == 0) return CG_NONE;
        --wi; m = w[wi];
    }
}
CG_DEVICE uint32_t bm_next_flat(const unsigned long long *w, uint32_t nw, uint32_t i)
{
    uint32_t wi = i >> 6, bit = i & 63;
    unsigned long long m = (bit == 63) ? 0ull : (w[wi] & ~((2ull << bit) - 1ull));
    for (;;) {
        if (m) return (wi << 6) + (uint32_t)cg_ctz64(m);
        ++wi; if (wi >= nw) return CG_NONE;
        m = w[wi];
    }
}
// previous set index in level `lo` (words wlo) strictly below i, using level `hi` as a hint and
// `hi2` (flat scan) above that
CG_DEVICE uint32_t bm_prev_l1(const SamplerDev &S, uint32_t i)   // over bits1 (index = level-0 word)
{
    uint32_t wi = i >> 6, bit = i & 63;
    unsigned long long m = bit ? (S.bits1[wi] & ((1ull << bit) - 1ull)) : 0ull;
    if (m) return (wi << 6) + 63u - (uint32_t)cg_clz64(m);
    uint32_t cur = wi;
    for (;;) {
        uint32_t p = bm_prev_flat(S.bits2, cur);
        if (p == CG_NONE) return CG_NONE;
        unsigned long long x = S.bits1[p];
        if (x) return (p << 6) + 63u - (uint32_t)cg_clz64(x);
        cur = p;
    }
}
CG_DEVICE uint32_t bm_next_l1(const SamplerDev &S, uint32_t i)
{
    uint32_t wi = i >> 6, bit = i & 63;
    unsigned long long m = (bit == 63) ? 0ull : (S.bits1[wi] & ~((2ull << bit) - 1ull));
    if (m) return (wi << 6) + (uint32_t)cg_ctz64(m);
    uint32_t cur = wi;
    for (;;) {
        uint32_t p = bm_next_flat(S.bits2, S.nWords2, cur);
        if (p == CG_NONE) return CG_NONE;
        unsigned long long x = S.bits1[p];
        if (x) return (p << 6) + (uint32_t)cg_ctz64(x);
        cur = p;
    }
}
CG_DEVICE uint32_t bm_prev_bin(const SamplerDev &S, uint32_t bin)   // largest occupied bin < bin
{
    uint32_t wi = bin >> 6, bit = bin & 63;
    unsigned long long m = bit ? (S.bits0[wi] & ((1ull << bit) - 1ull)) : 0ull;
    if (m) return (wi << 6) + 63u - (uint32_t)cg_clz64(m);
    uint32_t cur = wi;
    for (;;) {
        uint32_t p = bm_prev_l1(S, cur);
        if (p == CG_NONE) return CG_NONE;
        unsigned long long x = S.bits0[p];
        if (x) return (p << 6) + 63u - (uint32_t)cg_clz64(x);
        cur = p;
    }
}
CG_DEVICE uint32_t bm_next_bin(const SamplerDev &S, uint32_t bin)   // smallest occupied bin > bin
{
    uint32_t wi = bin >> 6, bit = bin & 63;
    unsigned long long m = (bit == 63) ? 0ull : (S.bits0[wi] & ~((2ull << bit) - 1ull));
    if (m) return (wi << 6) + (uint32_t)cg_ctz64(m);
    uint32_t cur = wi;
    for (;;) {
        uint32_t p = bm_next_l1(S, cur);
        if (p == CG_NONE) return CG_NONE;
        unsigned long long x = S.bits0[p];
        if (x) return (p << 6) + (uint32_t)cg_ctz64(x);
        cur = p;
    }
}

// position-order neighbours a new atom at `p` (bin b) would get; *occupied = some atom already at p
// (std::map lookups of ConcurrentAtomicDomain.cpp:46-54 and :82-106)
CG_DEVICE void gen_find_gap(const SamplerDev &S, uint64_t p, uint32_t b, uint32_t *pred, uint32_t *succ, bool *occupied)
{
    *occupied = false;
    uint32_t head = S.binHead[b];
    if (head != CG_NONE) {
        uint32_t cur = head, pr = S.atoms[head].left;
        for (;;) {
            uint64_t cp = S.atoms[cur].pos;
            if (cp == p) *occupied = true;
            if (cp > p) break;
            pr = cur; cur = S.atoms[cur].right;
            if (cur == CG_NONE) break;
        }
        *pred = pr; *succ = cur;
        return;
    }
    uint32_t nb = bm_next_bin(S, b);
    if (nb != CG_NONE) { uint32_t s = S.binHead[nb]; *succ = s; *pred = S.atoms[s].left; return; }
    *succ = CG_NONE;
    uint32_t pb = bm_prev_bin(S, b);
    if (pb == CG_NONE) { *pred = CG_NONE; return; }
    uint32_t cur = S.binHead[pb];
    for (;;) { uint32_t r = S.atoms[cur].right; if (r == CG_NONE) break; cur = r; }
    *pred = cur;
}

// ---- ConcurrentAtomicDomain::erase (ConcurrentAtomicDomain.cpp:109-124), one atom ---------------
CG_DEVICE void gen_erase_one(const SamplerDev &S, uint32_t h, uint32_t &n, uint32_t &freeCount, uint32_t &front)
{
    AtomRec rec = S.atoms[h];
    if (rec.left != CG_NONE) S.atoms[rec.left].right = rec.right; else front = rec.right;
    if (rec.right != CG_NONE) S.atoms[rec.right].left = rec.left;
    uint32_t b = gen_bin_of(S, rec.pos);
    if (S.binHead[b] == h) {
        uint32_t nxt = rec.right;
        if (nxt != CG_NONE && gen_bin_of(S, S.atoms[nxt].pos) == b) S.binHead[b] = nxt;
        else { S.binHead[b] = CG_NONE; bm_clear(S, b); }
    }
    uint32_t last = S.vec[n - 1];
    S.vec[rec.idx] = last;
    S.atoms[last].idx = rec.idx;
    --n;
    S.freeHandles[freeCount++] = h;
}

// ---- flushEraseCache (ConcurrentAtomicDomain.cpp:71-79): sort by position, erase in that order --
template <int WIN>
CG_DEVICE void gen_flush(const SamplerDev &S, GenShared<WIN> &sh)
{
    const unsigned t = cg_tid();
    GenScalars *gs = S.gs;
    const uint32_t m = gs->eraseCount;
    if (m == 0) return;           // uniform across the block
    if (m <= (uint32_t)WIN) {
        // rank sort in LDS: rank = number of entries with a smaller position (positions are unique)
        if (t < m) { uint32_t h = S.eraseList[t]; sh.fh[t] = h; sh.fpos[t] = S.atoms[h].pos; }
        cg_sync();
        uint32_t myRank = 0, myH = 0;
        if (t < m) {
            const uint64_t p = sh.fpos[t]; myH = sh.fh[t];
            for (uint32_t j = 0; j < m; ++j) myRank += (sh.fpos[j] < p) ? 1u : 0u;
        }
        cg_sync();
        if (t < m) sh.scan[myRank] = myH;
        cg_sync();
        if (t == 0) {
            uint32_t n = gs->nAtoms, fc = gs->freeCount, fr = gs->front;
            for (uint32_t i = 0; i < m; ++i) gen_erase_one(S, sh.scan[i], n, fc, fr);
            gs->nAtoms = n; gs->freeCount = fc; gs->front = fr; gs->eraseCount = 0;
        }
    } else if (t == 0) {
        // rare: more erasures than lanes -- insertion sort in place
        for (uint32_t i = 1; i < m; ++i) {
            uint32_t h = S.eraseList[i]; uint64_t p = S.atoms[h].pos; uint32_t j = i;
            while (j > 0 && S.atoms[S.eraseList[j - 1]].pos > p) { S.eraseList[j] = S.eraseList[j - 1]; --j; }
            S.eraseList[j] = h;
        }
        uint32_t n = gs->nAtoms, fc = gs->freeCount, fr = gs->front;
        for (uint32_t i = 0; i < m; ++i) gen_erase_one(S, S.eraseList[i], n, fc, fr);
        gs->nAtoms = n; gs->freeCount = fc; gs->front = fr; gs->eraseCount = 0;
    }
    cg_sync();
}

// exclusive block scan (Hillis-Steele in LDS)
template <int WIN>
CG_DEVICE uint32_t gen_excl_scan(uint32_t *buf, unsigned t, uint32_t v)
{
    buf[t] = v; cg_sync();
    for (int off = 1; off < WIN; off <<= 1) {
        uint32_t x = (t >= (unsigned)off) ? buf[t - off] : 0u;
        cg_sync();
        buf[t] += x;
        cg_sync();
    }
    uint32_t incl = buf[t];
    cg_sync();
    return incl - v;
}

// ProposalQueue::makeProposal type choice (ProposalQueue.cpp:129-160); 0 = indeterminate
CG_DEVICE uint32_t gen_decide(const SamplerDev &S, float u1, float u2, uint64_t minAtoms, uint64_t maxAtoms)
{
    if (minAtoms < 2 && maxAtoms >= 2) return GEN_T_NONE;
    if (maxAtoms < 2) return 'B';
    if (u1 < 0.5f) {
        float lowerBound = gm_death_prob((double)minAtoms, S.domainLenD, S.alphaD, S.numBins);
        float upperBound = gm_death_prob((double)maxAtoms, S.domainLenD, S.alphaD, S.numBins);
        if (u2 < lowerBound) return 'D';
        if (u2 >= upperBound) return 'B';
        return GEN_T_NONE;
    }
    return (u1 < 0.75f) ? 'M' : 'E';
}

CG_DEVICE bool gen_row_used(const SamplerDev &S, uint32_t r, uint64_t batchEpoch, uint64_t roundEpoch, unsigned t)
{
    if (S.rowBatch[r] == batchEpoch) return true;
    unsigned long long v = S.rowRound[r];
    return (v >> 16) == roundEpoch && (65535u - (uint32_t)(v & 0xFFFFu)) < t + 1u;
}
CG_DEVICE bool gen_atom_used(const SamplerDev &S, uint32_t h, uint64_t batchEpoch, uint64_t roundEpoch, unsigned t)
{
    if (S.atomBatch[h] == batchEpoch) return true;
    unsigned long long v = S.atomRound[h];
    return (v >> 16) == roundEpoch && (65535u - (uint32_t)(v & 0xFFFFu)) < t + 1u;
}
CG_DEVICE unsigned long long gen_stamp(uint64_t roundEpoch, unsigned t) { return (roundEpoch << 16) | (unsigned long long)(65535u - (t + 1u)); }

// serial insert of a new atom whose snapshot gap is shared with other births of the same commit
CG_DEVICE void gen_link_birth_serial(const SamplerDev &S, uint32_t hb, uint64_t p, uint32_t predSnap, uint32_t &front)
{
    // walk right from the snapshot predecessor (or from the front) past atoms inserted meanwhile
    uint32_t pred = predSnap, succ;
    if (pred == CG_NONE) {
        succ = front;
        if (succ != CG_NONE && S.atoms[succ].pos < p) { pred = succ; succ = S.atoms[pred].right; }
    } else succ = S.atoms[pred].right;
    while (succ != CG_NONE && S.atoms[succ].pos < p) { pred = succ; succ = S.atoms[succ].right; }
    S.atoms[hb].left = pred; S.atoms[hb].right = succ;
    if (pred != CG_NONE) S.atoms[pred].right = hb; else front = hb;
    if (succ != CG_NONE) S.atoms[succ].left = hb;
    uint32_t b = gen_bin_of(S, p);
    if (pred == CG_NONE || gen_bin_of(S, S.atoms[pred].pos) != b) S.binHead[b] = hb;
    bm_set(S, b);
}

// =================================================================================================
template <int WIN>
CG_DEVICE void gen_body(const SamplerDev &S)
{
    CG_SHARED GenShared<WIN> sh;
    const unsigned t = cg_tid();
    GenScalars *gs = S.gs;

    unsigned long long prof_last = cg_clock(); (void)prof_last;
    gen_flush<WIN>(S, sh);
    GEN_PROF(0);

    if (t == 0) {
        sh.done = (gs->nDone >= gs->nSteps) ? 1u : 0u;
        sh.batchEpoch = gs->batchEpoch + 1;
        sh.roundEpoch = gs->roundEpoch;
        sh.qrngRound = gs->qrng;
        sh.nR = gs->nAtoms; sh.minAtoms = gs->nAtoms;
        sh.processed = 0; sh.qlen = 0; sh.skip = gs->useCached ? 1u : 0u;
        sh.remaining = gs->nSteps - gs->nDone;
        sh.nBatchMoves = 0;
    }
    cg_sync();
    if (sh.done) { if (t == 0) { gs->qlen = 0; gs->batchNproc = 0; gs->updateFlushed = 1; } return; }

    const uint64_t batchEpoch = sh.batchEpoch;
    const uint32_t updBase = gs->nDone;        // candidates consumed by earlier batches of this update

    for (;;) {
        // ------------------------------------------------------------------ round set-up
        if (t == 0) { sh.roundEpoch += 1; sh.nBirths = 0; sh.nInlineM = 0; sh.nInlineE = 0; sh.nQueuedM = 0; sh.stopKey = 0xFFFFFFFFu; sh.needSerialBirths = 0; }
        cg_sync();
        const uint64_t roundEpoch = sh.roundEpoch;
        const uint32_t nR = sh.nR, minR = sh.minAtoms, skip = sh.skip, processed = sh.processed;
        const uint32_t left_ = sh.remaining - processed;
        const uint32_t winN = left_ < (uint32_t)WIN ? left_ : (uint32_t)WIN;
        const bool active = t < winN;

        // ------------------------------------------------------------------ A1: (u1,u2), B/D/M/E
        float u1 = 0.f, u2 = 0.f;
        uint32_t guess = GEN_T_NONE;
        if (active) {
            if (skip && t == 0) { u1 = gs->u1; u2 = gs->u2; }
            else {
                const uint32_t k = 2u * (t - skip);
                uint64_t s = S.lcgMul[k] * sh.qrngRound + S.lcgInc[k];
                u1 = pcg_uniform(s); u2 = pcg_uniform(s);
            }
            guess = gen_decide(S, u1, u2, minR, nR);
        }
        sh.u1[t] = u1; sh.u2[t] = u2;
        const uint32_t packed = (guess == 'B' ? 1u : 0u) | (guess == 'D' ? 0x10000u : 0u);
        const uint32_t before = gen_excl_scan<WIN>(sh.scan, t, active ? packed : 0u);
        const uint32_t bBefore = before & 0xFFFFu, dBefore = before >> 16;
        GEN_PROF(1);
        uint32_t type = guess; uint32_t flags = 0;
        if (active) {
            const uint32_t exact = gen_decide(S, u1, u2, (uint64_t)minR - dBefore, (uint64_t)nR + bBefore);
            if (exact != guess) flags |= GEN_F_HAZARD;
            else if (guess == GEN_T_NONE) flags |= GEN_F_FAIL;       // indeterminate: batch ends, no seed used
        }

        // ------------------------------------------------------------------ A2: populate-phase draws
        uint64_t rng = 0, pos = 0, cpos = 0, lbpos = 0, rbpos = 0;
        uint32_t h1 = CG_NONE, h2 = CG_NONE, i1 = CG_NONE, i2 = CG_NONE, hl = CG_NONE, hr = CG_NONE;
        uint32_t r1 = 0, c1 = 0, r2 = 0, c2 = 0; float nm1 = 0.f, nm2 = 0.f;
        if (active && type != GEN_T_NONE && !(flags & GEN_F_HAZARD)) {
            rng = pcg_from_seed(S.seeds[updBase + processed + t]);   // AtomicProposal ctor, ProposalQueue.cpp:12-15
            const uint32_t nT = nR + bBefore;                            // domain size this attempt sees
            const uint32_t K = S.K;
            if (type == 'B') {
                bool occ;
                do {                                                     // randomFreePosition
                    pos = pcg_uniform64(rng, 1ull, S.domainLenU);
                    gen_find_gap(S, pos, gen_bin_of(S, pos), &hl, &hr, &occ);
                } while (occ);
                const uint64_t bin = pos / S.binLength;
                r1 = (uint32_t)(bin / K); c1 = (uint32_t)(bin % K);
                i1 = nT;
            } else {
                i1 = pcg_uniform32(rng, 0u, nT - 1u);
                if (i1 >= nR) {
                    flags |= GEN_F_FAIL;           // an atom born earlier in this window: its row is in use
                } else {
                    h1 = S.vec[i1];
                    const AtomRec a = S.atoms[h1];
                    cpos = a.pos;
                    const uint64_t bin = cpos / S.binLength;
                    r1 = (uint32_t)(bin / K); c1 = (uint32_t)(bin % K);
                    if (type == 'M') {
                        hl = a.left; hr = a.right;
                        if (hl != CG_NONE) { flags |= GEN_F_HASLEFT; lbpos = S.atoms[hl].pos; } else lbpos = 0;
                        if (hr != CG_NONE) { flags |= GEN_F_HASRIGHT; rbpos = S.atoms[hr].pos; } else rbpos = S.rboundNone;
                        pos = pcg_uniform64(rng, lbpos + 1ull, rbpos - 1ull);
                        const uint64_t bin2 = pos / S.binLength;
                        r2 = (uint32_t)(bin2 / K); c2 = (uint32_t)(bin2 % K);
                        if (r1 == r2 && c1 == c2) flags |= GEN_F_INLINE;
                    } else if (type == 'E') {
                        hr = a.right;
                        if (hr != CG_NONE) { flags |= GEN_F_HASRIGHT; h2 = hr; } else h2 = gs->front;
                        const AtomRec b = S.atoms[h2];
                        rbpos = b.pos; i2 = b.idx;
                        const uint64_t bin2 = rbpos / S.binLength;
                        r2 = (uint32_t)(bin2 / K); c2 = (uint32_t)(bin2 % K);
                        if (r1 == r2 && c1 == c2) {
                            flags |= GEN_F_INLINE;
                            const float m1 = a.mass, m2 = b.mass;
                            const float newMass = pcg_trunc_gamma_upper(rng, S.luts, m1 + m2, 1.f / S.lambda);
                            const float delta = (m1 > m2) ? newMass - m1 : m2 - newMass;
                            if (m1 + delta > GAPS_EPSILON && m2 - delta > GAPS_EPSILON) { flags |= GEN_F_APPLY; nm1 = m1 + delta; nm2 = m2 - delta; }
                        }
                    }
                }
            }
        }
        GEN_PROF(2);
        // publish the candidate
        sh.rng[t] = rng; sh.pos[t] = pos; sh.cpos[t] = cpos; sh.lbpos[t] = lbpos; sh.rbpos[t] = rbpos;
        sh.h1[t] = h1; sh.h2[t] = h2; sh.i1[t] = i1; sh.i2[t] = i2; sh.hl[t] = hl; sh.hr[t] = hr;
        sh.r1[t] = r1; sh.c1[t] = c1; sh.r2[t] = r2; sh.c2[t] = c2; sh.nm1[t] = nm1; sh.nm2[t] = nm2;
        sh.type[t] = (uint8_t)type;

        // ------------------------------------------------------------------ B1: register rows / atoms
        const bool live = active && type != GEN_T_NONE && !(flags & (GEN_F_HAZARD | GEN_F_FAIL));
        if (live) {
            const unsigned long long st = gen_stamp(roundEpoch, t);
            if (type == 'B') { cg_atomic_max_u64(&S.rowRound[r1], st); sh.bl_t[cg_atomic_add_u32(&sh.nBirths, 1u)] = t; }
            else if (type == 'D') { cg_atomic_max_u64(&S.rowRound[r1], st); cg_atomic_max_u64(&S.atomRound[h1], st); }
            else if (type == 'M') {
                if (flags & GEN_F_INLINE) sh.im_t[cg_atomic_add_u32(&sh.nInlineM, 1u)] = t;
                else {
                    cg_atomic_max_u64(&S.rowRound[r1], st); cg_atomic_max_u64(&S.rowRound[r2], st); cg_atomic_max_u64(&S.atomRound[h1], st);
                    sh.qm_t[cg_atomic_add_u32(&sh.nQueuedM, 1u)] = t;
                }
            } else {
                if (flags & GEN_F_INLINE) sh.ie_t[cg_atomic_add_u32(&sh.nInlineE, 1u)] = t;
                else { cg_atomic_max_u64(&S.rowRound[r1], st); cg_atomic_max_u64(&S.rowRound[r2], st); }
            }
        }
        cg_sync();
        GEN_PROF(3);

        // ------------------------------------------------------------------ B2: conflicts and hazards
        if (live) {
            bool fail = false, haz = false;
            if (type == 'B') {
                fail = gen_row_used(S, r1, batchEpoch, roundEpoch, t);
                // mProposedMoves.overlap(pos): queued moves of this window before t ...
                for (uint32_t k = 0; k < sh.nQueuedM && !fail; ++k) {
                    const uint32_t q = sh.qm_t[k];
                    if (q < t) { uint64_t a = sh.cpos[q], b = sh.pos[q]; uint64_t lo = a < b ? a : b, hi = a < b ? b : a; if (lo < pos && pos < hi) fail = true; }
                }
                // ... and of earlier rounds of this batch
                for (uint32_t k = 0; k < sh.nBatchMoves && !fail; ++k) { if (S.batchMoves[2 * k] < pos && pos < S.batchMoves[2 * k + 1]) fail = true; }
                // an earlier same-bin move of this window changed a position this birth's gap search compared against
                for (uint32_t k = 0; k < sh.nInlineM; ++k) {
                    const uint32_t q = sh.im_t[k];
                    if (q < t) { uint64_t a = sh.cpos[q], b = sh.pos[q]; uint64_t lo = a < b ? a : b, hi = a < b ? b : a; if (lo <= pos && pos <= hi) haz = true; }
                }
                for (uint32_t k = 0; k < sh.nBirths; ++k) { const uint32_t q = sh.bl_t[k]; if (q < t && sh.pos[q] == pos) haz = true; }
            } else if (type == 'D') {
                fail = gen_row_used(S, r1, batchEpoch, roundEpoch, t);
            } else if (type == 'M') {
                fail = gen_row_used(S, r1, batchEpoch, roundEpoch, t) || gen_row_used(S, r2, batchEpoch, roundEpoch, t);
                if ((flags & GEN_F_HASLEFT) && gen_atom_used(S, hl, batchEpoch, roundEpoch, t)) fail = true;
                if ((flags & GEN_F_HASRIGHT) && gen_atom_used(S, hr, batchEpoch, roundEpoch, t)) fail = true;
                // a birth earlier in this window inside (left, right) is the true neighbour, and is "used"
                for (uint32_t k = 0; k < sh.nBirths; ++k) {
                    const uint32_t q = sh.bl_t[k];
                    if (q < t) { const uint64_t p = sh.pos[q]; if (p > lbpos && ((flags & GEN_F_HASRIGHT) ? p < rbpos : true)) fail = true; }
                }
                // an earlier same-bin move of this window moved the centre or a neighbour: positions stale
                for (uint32_t k = 0; k < sh.nInlineM; ++k) {
                    const uint32_t q = sh.im_t[k];
                    if (q < t) { const uint32_t hq = sh.h1[q]; if (hq == h1 || hq == hl || hq == hr) haz = true; }
                }
            } else {
                fail = gen_row_used(S, r1, batchEpoch, roundEpoch, t) || gen_row_used(S, r2, batchEpoch, roundEpoch, t);
                for (uint32_t k = 0; k < sh.nBirths; ++k) {
                    const uint32_t q = sh.bl_t[k];
                    if (q < t) {
                        const uint64_t p = sh.pos[q];
                        if (flags & GEN_F_HASRIGHT) { if (p > cpos && p < rbpos) fail = true; }
                        else { if (p > cpos || p < rbpos) fail = true; }      // new right neighbour, or new front()
                    }
                }
                if (flags & GEN_F_INLINE) {
                    for (uint32_t k = 0; k < sh.nInlineE; ++k) {
                        const uint32_t q = sh.ie_t[k];
                        if (q < t) { const uint32_t a = sh.h1[q], b = sh.h2[q]; if (a == h1 || a == h2 || b == h1 || b == h2) haz = true; }
                    }
                }
            }
            if (haz) flags |= GEN_F_HAZARD; else if (fail) flags |= GEN_F_FAIL;
        }
        sh.flags[t] = (uint8_t)flags;
        if (active && (flags & (GEN_F_HAZARD | GEN_F_FAIL))) cg_atomic_min_u32(&sh.stopKey, 2u * t + ((flags & GEN_F_HAZARD) ? 0u : 1u));
        cg_sync();
        GEN_PROF(4);

        // ------------------------------------------------------------------ C: commit [0, stopT)
        const uint32_t stopKey = sh.stopKey;
        const uint32_t stopT = (stopKey == 0xFFFFFFFFu) ? winN : (stopKey >> 1);
        const bool stopFail = (stopKey != 0xFFFFFFFFu) && (stopKey & 1u);
        const bool commit = t < stopT;            // every such candidate is live
        const bool queued = commit && (type == 'B' || type == 'D' || !(flags & GEN_F_INLINE));
        const uint32_t packed2 = (queued ? 1u : 0u) | ((commit && type == 'B') ? 0x10000u : 0u) | 0u;
        const uint32_t before2 = gen_excl_scan<WIN>(sh.scan, t, packed2);
        const uint32_t qBefore = before2 & 0xFFFFu, bRank = before2 >> 16;
        GEN_PROF(5);
        if (commit) {
            uint32_t hb = CG_NONE;
            if (type == 'B') {
                // handle allocation: free stack first (deterministic by rank), then bump
                const uint32_t fc = gs->freeCount;
                hb = (bRank < fc) ? S.freeHandles[fc - 1u - bRank] : gs->handleHi + (bRank - fc);
                const uint32_t idx = nR + bRank;
                if (hb >= S.atomCap || idx >= S.atomCap) { gs->error = GAPS_ERR_ATOM_CAP; hb = 0; }
                S.vec[idx] = hb;
                AtomRec a; a.pos = pos; a.left = hl; a.right = hr; a.mass = 0.f; a.idx = idx; a.pad0 = 0; a.pad1 = 0;
                S.atoms[hb] = a;
                h1 = hb; sh.h1[t] = hb;
                bool shared = false;
                for (uint32_t k = 0; k < sh.nBirths; ++k) { const uint32_t q = sh.bl_t[k]; if (q != t && q < stopT && sh.hl[q] == hl && sh.hr[q] == hr) shared = true; }
                if (shared) { sh.flags[t] = (uint8_t)(flags | GEN_F_APPLY); sh.needSerialBirths = 1; }   // linked serially below
                else {
                    if (hl != CG_NONE) S.atoms[hl].right = hb; else gs->front = hb;
                    if (hr != CG_NONE) S.atoms[hr].left = hb;
                    const uint32_t b = gen_bin_of(S, pos);
                    if (hl == CG_NONE || gen_bin_of(S, S.atoms[hl].pos) != b) S.binHead[b] = hb;
                    bm_set(S, b);
                }
                S.rowBatch[r1] = batchEpoch; S.atomBatch[hb] = batchEpoch;
            } else if (type == 'D') {
                S.rowBatch[r1] = batchEpoch; S.atomBatch[h1] = batchEpoch;
            } else if (type == 'M') {
                if (flags & GEN_F_INLINE) S.atoms[h1].pos = pos;                  // domain.move, same bin
                else {
                    S.rowBatch[r1] = batchEpoch; S.rowBatch[r2] = batchEpoch; S.atomBatch[h1] = batchEpoch;
                    const uint32_t k = cg_atomic_add_u32(&sh.nBatchMoves, 1u);
                    S.batchMoves[2 * k] = cpos < pos ? cpos : pos; S.batchMoves[2 * k + 1] = cpos < pos ? pos : cpos;
                }
            } else {
                if (flags & GEN_F_INLINE) { if (flags & GEN_F_APPLY) { S.atoms[h1].mass = nm1; S.atoms[h2].mass = nm2; } }
                else { S.rowBatch[r1] = batchEpoch; S.rowBatch[r2] = batchEpoch; }
            }
            if (queued) {
                const uint32_t slot = sh.qlen + qBefore;
                if (slot >= S.queueCap) gs->error = GAPS_ERR_QUEUE_CAP;
                else {
                    PropRec p; p.pos = pos; p.rng = rng; p.h1 = h1; p.h2 = h2; p.i1 = i1; p.i2 = i2;
                    p.r1 = r1; p.c1 = c1; p.r2 = r2; p.c2 = c2; p.type = type; p.pad[0] = p.pad[1] = p.pad[2] = 0;
                    S.queue[slot] = p;
                    if (gs->traceOn) { const uint32_t ti = gs->traceCount + slot; if (ti < gs->traceCap) { p.pad[0] = gs->nBatches; S.trace[ti] = p; } }
                }
            }
        }
        cg_sync();
        GEN_PROF(6);
        // births that share a snapshot gap: link one by one in attempt order
        if (t == 0 && sh.needSerialBirths) {
            uint32_t fr = gs->front;
            for (uint32_t q = 0; q < stopT; ++q)
                if (sh.type[q] == 'B' && (sh.flags[q] & GEN_F_APPLY)) gen_link_birth_serial(S, sh.h1[q], sh.pos[q], sh.hl[q], fr);
            gs->front = fr;
        }
        // ------------------------------------------------------------------ round bookkeeping
        if (t == WIN - 1) {
            // totals of the committed prefix (t = WIN-1 holds the inclusive scan end)
            const uint32_t totQ = qBefore + (queued ? 1u : 0u), totB = bRank + ((commit && type == 'B') ? 1u : 0u);
            uint32_t totD = 0;
            // deaths: recount (cheap: only this lane, LDS reads)
            for (uint32_t q = 0; q < stopT; ++q) totD += (sh.type[q] == 'D') ? 1u : 0u;
            const uint32_t fc = gs->freeCount;
            if (totB) { if (totB <= fc) gs->freeCount = fc - totB; else { gs->freeCount = 0; gs->handleHi += totB - fc; } }
            gs->nAtoms = nR + totB;
            sh.nR = nR + totB; sh.minAtoms = minR - totD;
            sh.qlen += totQ; sh.processed = processed + stopT;
            const uint32_t attempted = stopT + (stopFail ? 1u : 0u);
            const uint32_t draws = 2u * (attempted - (skip && attempted ? 1u : 0u));
            sh.qrngRound = S.lcgMul[draws] * sh.qrngRound + S.lcgInc[draws];
            if (attempted) sh.skip = 0;
            sh.stopT = stopT; sh.stopFail = stopFail ? 1u : 0u;
        }
        cg_sync();
        GEN_PROF(7);
        if (t == 0) gs->prof[15] += 1;   // rounds
        const bool endBatch = sh.stopFail || (sh.processed >= sh.remaining);
        if (endBatch) {
            if (t == 0) {
                gs->qrng = sh.qrngRound;
                if (sh.stopFail) { gs->useCached = 1; gs->u1 = sh.u1[sh.stopT]; gs->u2 = sh.u2[sh.stopT]; }
                else gs->useCached = 0;
                gs->nDone = updBase + sh.processed;
                gs->qlen = sh.qlen; gs->batchNproc = sh.processed;
                gs->batchEpoch = batchEpoch; gs->roundEpoch = sh.roundEpoch;
                if (gs->nDone < gs->nSteps) {           // AsynchronousGibbsSampler.h:97-102
                    gs->nQueueSamples += 1.f;
                    gs->avgQueue *= (gs->nQueueSamples - 1.f) / gs->nQueueSamples;
                    gs->avgQueue += (float)sh.qlen / gs->nQueueSamples;
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
