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
// bit-identical to the serial procedure; tests/test_emul_parity.py checks that against the oracle.
//
// Conflict rules restated from ProposalQueue.cpp: birth :162-187, death :189-207, move :209-248,
// exchange :250-283, type choice :129-160; the three sets of data_structures/HashSets.cpp become
// stamp tables (gaps_state.h) probed with a handful of independent loads per lane.
//
// Latency structure (one lane = one attempt; lanes of all four types issue the same loads):
//   stage 1  B: occupancy word of its bin          D/M/E: vec[index]        (index -> handle)
//   stage 2  B: head handle of the successor bin   D/M/E: atom record
//   stage 3  B: that atom's record (-> pred/succ)  M: both neighbours   E: partner atom
//   register stamps (atomicMax) | barrier | probe stamps | barrier | commit (stores only)
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
#define GEN_F_NEWHEAD 64u   // birth becomes the lowest atom of its bin
#define GEN_F_BINEMPTY 128u // birth's bin had no atom
#define GEN_F_WORDZERO 256u // ... and its whole level-0 bitmap word was empty (hints must be set)

#define GEN_STAMP_COMMITTED 0xFFFFFFull

template <int WIN>
struct GenShared {
    uint64_t cpos[WIN], pos[WIN];        // centre position / destination, read by other lanes for queued moves
    float u1[WIN], u2[WIN];
    uint8_t type[WIN];                   // 'M' only when queued (birth-overlap test)
    uint32_t wtot[3][WIN / 64];
    uint64_t fpos[WIN]; uint32_t fh[WIN], sorted[WIN];   // flush: erase cache positions / handles
    uint64_t qrngRound, batchEpoch;
    uint32_t roundNo, stopKey;
    uint32_t nR, minAtoms, processed, qlen, skip, remaining, done, stopT, stopFail;
};

// bin index = pos / binLength, exact: double-precision reciprocal estimate (off by at most one), then a
// 64-bit multiply-back correction.  nBins < 2^32, so the quotient fits 32 bits and q*binLength <= L.
CG_DEVICE uint32_t gen_bin_of(const SamplerDev &S, uint64_t pos)
{
    double e = (double)pos * S.invBinLen;
    uint32_t q = (e >= S.numBins) ? (uint32_t)S.numBins : (uint32_t)e;
    const uint64_t prod = (uint64_t)q * S.binLength;
    if (prod > pos) --q;
    else if (pos - prod >= S.binLength) ++q;
    return q;
}

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
// largest set index < i in a one-level bitmap, CG_NONE if none
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

// full search for the position-order neighbours a new atom at `p` (bin b) would get
// (the std::map lookups of ConcurrentAtomicDomain.cpp:46-54 and :82-106); *newHead = it becomes the
// lowest atom of its bin; *occupied = some atom already sits at p
CG_DEVICE void gen_find_gap(const SamplerDev &S, uint64_t p, uint32_t b, uint32_t *pred, uint32_t *succ, bool *occupied, bool *newHead)
{
    *occupied = false; *newHead = true;
    uint32_t head = S.binHead[b];
    if (head != CG_NONE) {
        uint32_t cur = head, pr = S.atoms[head].left;
        for (;;) {
            uint64_t cp = S.atoms[cur].pos;
            if (cp == p) *occupied = true;
            if (cp > p) break;
            *newHead = false;
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
        if (t < m) sh.sorted[myRank] = myH;
        cg_sync();
        if (t == 0) {
            uint32_t n = gs->nAtoms, fc = gs->freeCount, fr = gs->front;
            for (uint32_t i = 0; i < m; ++i) gen_erase_one(S, sh.sorted[i], n, fc, fr);
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

// exclusive counts of flags a,b before this lane + block totals of a,b,c (wave ballots + one LDS hop)
template <int WIN>
CG_DEVICE void gen_count3(GenShared<WIN> &sh, unsigned t, bool a, bool b, bool c,
                          uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb, uint32_t &tc)
{
    const unsigned lane = t & 63u, wave = t >> 6;
    const unsigned long long ma = cg_ballot(a), mb = cg_ballot(b), mc = cg_ballot(c);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lane == 0) { sh.wtot[0][wave] = (uint32_t)cg_popc64(ma); sh.wtot[1][wave] = (uint32_t)cg_popc64(mb); sh.wtot[2][wave] = (uint32_t)cg_popc64(mc); }
    cg_sync();
    ea = (uint32_t)cg_popc64(ma & lt); eb = (uint32_t)cg_popc64(mb & lt); ta = 0; tb = 0; tc = 0;
    for (unsigned w = 0; w < (unsigned)(WIN / 64); ++w) {
        const uint32_t xa = sh.wtot[0][w], xb = sh.wtot[1][w], xc = sh.wtot[2][w];
        if (w < wave) { ea += xa; eb += xb; }
        ta += xa; tb += xb; tc += xc;
    }
    cg_sync();
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

// stamp helpers -----------------------------------------------------------------------------------
CG_DEVICE unsigned long long gen_stamp(uint64_t batchEpoch, uint32_t roundNo, unsigned t)
{
    return (batchEpoch << 24) | ((unsigned long long)roundNo << 12) | (unsigned long long)(4094u - t);
}
// 0 = not used; 1 = used by a committed attempt of this batch; 2 = used by attempt *idx < t of this window
CG_DEVICE int gen_probe(unsigned long long v, uint64_t batchEpoch, uint32_t roundNo, unsigned t, uint32_t *idx)
{
    if ((v >> 24) != batchEpoch) return 0;
    const uint32_t low = (uint32_t)(v & 0xFFFFFFull);
    if (low == GEN_STAMP_COMMITTED) return 1;
    if ((low >> 12) != roundNo) return 0;
    const uint32_t i = 4094u - (low & 0xFFFu);
    *idx = i;
    return i < t ? 2 : 0;
}

// =================================================================================================
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
    gen_flush<WIN>(S, sh);
    GEN_PROF(0);

    if (t == 0) {
        sh.done = (gs->nDone >= gs->nSteps) ? 1u : 0u;
        sh.batchEpoch = gs->batchEpoch + 1;
        sh.roundNo = 0;
        sh.qrngRound = gs->qrng;
        sh.nR = gs->nAtoms; sh.minAtoms = gs->nAtoms;
        sh.processed = 0; sh.qlen = 0; sh.skip = gs->useCached ? 1u : 0u;
        sh.remaining = gs->nSteps - gs->nDone;
    }
    cg_sync();
    if (sh.done) { if (t == 0) { gs->qlen = 0; gs->batchNproc = 0; gs->updateFlushed = 1; } return; }

    const uint64_t batchEpoch = sh.batchEpoch;
    const uint32_t updBase = gs->nDone;        // attempts consumed by earlier batches of this update
    const uint32_t K = S.K;

    for (;;) {
        // ------------------------------------------------------------------ round set-up
        if (t == 0) { sh.roundNo += 1; sh.stopKey = 0xFFFFFFFFu; if (sh.roundNo >= 4094u) gs->error = GAPS_ERR_SPIN; }
        cg_sync();
        const uint32_t roundNo = sh.roundNo;
        const uint32_t nR = sh.nR, minR = sh.minAtoms, skip = sh.skip, processed = sh.processed;
        const uint32_t left_ = sh.remaining - processed;
        const uint32_t winN = left_ < (uint32_t)WIN ? left_ : (uint32_t)WIN;
        const bool active = t < winN;
        const uint64_t seed = active ? S.seeds[updBase + processed + t] : 0ull;   // issued early, used in stage 1

        // ------------------------------------------------------------------ A1: (u1,u2), B/D/M/E
        float u1 = 0.f, u2 = 0.f;
        uint32_t guess = GEN_T_NONE;
        if (active) {
            if (skip && t == 0) { u1 = gs->u1; u2 = gs->u2; }
            else {
                uint64_t s = (skip ? jm1 : jm0) * sh.qrngRound + (skip ? ji1 : ji0);
                u1 = pcg_uniform(s); u2 = pcg_uniform(s);
            }
            guess = gen_decide(S, u1, u2, minR, nR);
        }
        sh.u1[t] = u1; sh.u2[t] = u2;
        uint32_t bBefore, dBefore, tB, tD, tUnused;
        gen_count3<WIN>(sh, t, active && guess == 'B', active && guess == 'D', false, bBefore, dBefore, tB, tD, tUnused);
        uint32_t type = guess; uint32_t flags = 0;
        if (active && (bBefore | dBefore)) {
            // the exact B/D/indeterminate decision depends on how many births / deaths precede this attempt
            const uint32_t exact = (u1 < 0.5f || minR < 2u + dBefore || nR + bBefore < 2u) ? gen_decide(S, u1, u2, (uint64_t)minR - dBefore, (uint64_t)nR + bBefore) : guess;
            if (exact != guess) flags |= GEN_F_HAZARD;
        }
        if (active && !(flags & GEN_F_HAZARD) && guess == GEN_T_NONE) flags |= GEN_F_FAIL;   // indeterminate: batch ends, no seed used
        GEN_PROF(1);

        // ------------------------------------------------------------------ A2: populate-phase draws
        const bool go = active && type != GEN_T_NONE && !(flags & GEN_F_HAZARD);
        const bool isB = go && type == 'B';
        bool pick = go && type != 'B';                 // D/M/E: picks an existing atom
        uint64_t rng = go ? pcg_from_seed(seed) : 0ull;   // AtomicProposal ctor, ProposalQueue.cpp:12-15
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
        const bool live = active && type != GEN_T_NONE && !(flags & (GEN_F_HAZARD | GEN_F_FAIL));
        const bool queuedM = live && type == 'M' && !(flags & GEN_F_INLINE);
        sh.cpos[t] = cpos; sh.pos[t] = pos; sh.type[t] = queuedM ? (uint8_t)'M' : (uint8_t)0;
        if (live) {
            const unsigned long long st = gen_stamp(batchEpoch, roundNo, t);
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
            fail = gen_probe(v0, batchEpoch, roundNo, t, &ix) != 0;                       // row r1 in use
            if (tM || tE) { if (gen_probe(v1_, batchEpoch, roundNo, t, &ix) != 0) fail = true; }   // row r2 in use
            if (tB) {
                if (gen_probe(v4_, batchEpoch, roundNo, t, &ix) == 2) haz = true;          // an earlier birth of this window in the same gap
                const uint32_t nb[2] = {hl, hr}; const unsigned long long sa[2] = {v2_, v3_}, si[2] = {v6_, v7_}; const uint64_t dest[2] = {d9, d10};
                for (int k = 0; k < 2; ++k) {
                    if (nb[k] == CG_NONE) continue;
                    // mProposedMoves.overlap(pos): the neighbour has a queued move whose interval covers pos
                    const int u = gen_probe(sa[k], batchEpoch, roundNo, t, &ix);
                    uint64_t ma = 0, mb = 0; bool mv = false;
                    if (u == 1 && dest[k] != 0ull) { ma = S.atoms[nb[k]].pos; mb = dest[k]; mv = true; }
                    else if (u == 2 && sh.type[ix] == 'M') { ma = sh.cpos[ix]; mb = sh.pos[ix]; mv = true; }
                    if (mv) { const uint64_t lo = ma < mb ? ma : mb, hi = ma < mb ? mb : ma; if (lo < pos && pos < hi) fail = true; }
                    // an earlier same-bin move of this window shifted the neighbour this gap search compared against
                    if (gen_probe(si[k], batchEpoch, roundNo, t, &ix) == 2) haz = true;
                }
            } else if (tM) {
                if ((hl != CG_NONE && gen_probe(v2_, batchEpoch, roundNo, t, &ix) != 0) || (hr != CG_NONE && gen_probe(v3_, batchEpoch, roundNo, t, &ix) != 0)) fail = true;   // mUsedAtoms
                // a birth earlier in this window inside (left, right) is the true neighbour, and it is "used"
                if (gen_probe(v4_, batchEpoch, roundNo, t, &ix) == 2 || gen_probe(v5_, batchEpoch, roundNo, t, &ix) == 2) fail = true;
                // an earlier same-bin move/exchange of this window touched the centre or a neighbour: positions stale
                if (gen_probe(v6_, batchEpoch, roundNo, t, &ix) == 2) haz = true;
                if (hl != CG_NONE && gen_probe(v7_, batchEpoch, roundNo, t, &ix) == 2) haz = true;
                if (hr != CG_NONE && gen_probe(v8_, batchEpoch, roundNo, t, &ix) == 2) haz = true;
            } else if (tE) {
                // an earlier birth right of the centre is the true partner (or, for the last atom, a new front())
                if (gen_probe(v4_, batchEpoch, roundNo, t, &ix) == 2) fail = true;
                if (!(flags & GEN_F_HASRIGHT) && gen_probe(v5_, batchEpoch, roundNo, t, &ix) == 2) fail = true;
                if (inl) { if (gen_probe(v6_, batchEpoch, roundNo, t, &ix) == 2 || gen_probe(v7_, batchEpoch, roundNo, t, &ix) == 2) haz = true; }
            }
            if (haz) flags |= GEN_F_HAZARD; else if (fail) flags |= GEN_F_FAIL;
        }
        if (active && (flags & (GEN_F_HAZARD | GEN_F_FAIL))) cg_atomic_min_u32(&sh.stopKey, 2u * t + ((flags & GEN_F_HAZARD) ? 0u : 1u));
        cg_sync();
        GEN_PROF(4);

        // ------------------------------------------------------------------ C: commit [0, stopT)
        const uint32_t stopKey = sh.stopKey;
        const uint32_t stopT = (stopKey == 0xFFFFFFFFu) ? winN : (stopKey >> 1);
        const bool stopFail = (stopKey != 0xFFFFFFFFu) && (stopKey & 1u);
        const bool commit = t < stopT;            // every such attempt is live
        const bool queued = commit && (type == 'B' || type == 'D' || !(flags & GEN_F_INLINE));
        uint32_t qBefore, bRank, totQ, totB, totD;
        gen_count3<WIN>(sh, t, queued, commit && type == 'B', commit && type == 'D', qBefore, bRank, totQ, totB, totD);
        GEN_PROF(5);
        if (commit) {
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
            const uint32_t fc = gs->freeCount;
            if (totB) { if (totB <= fc) gs->freeCount = fc - totB; else { gs->freeCount = 0; gs->handleHi += totB - fc; } }
            gs->nAtoms = nR + totB;
            sh.nR = nR + totB; sh.minAtoms = minR - totD;
            sh.qlen += totQ; sh.processed = processed + stopT;
            const uint32_t attempted = stopT + (stopFail ? 1u : 0u);
            const uint32_t draws = 2u * (attempted - ((skip && attempted) ? 1u : 0u));
            uint64_t jm, ji; pcg_jump_coeffs(draws, jm, ji);
            sh.qrngRound = jm * sh.qrngRound + ji;
            if (attempted) sh.skip = 0;
            sh.stopT = stopT; sh.stopFail = stopFail ? 1u : 0u;
#if defined(GEN_PROFILE)
            gs->prof[15] += 1;
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
                gs->nDone = updBase + sh.processed;
                gs->qlen = sh.qlen; gs->batchNproc = sh.processed;
                gs->batchEpoch = batchEpoch;
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
