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

#if defined(GEN_TIMELINE)
// per-wave timeline of one typical launch (lane 0 of every wave -- the attempt waves and the helper wave -- records (clock << 8 | id)); dev tool only
__device__ unsigned long long g_timeline[8 * 64];      // (up to eight waves: the chained launch's generator workgroup has the evaluation's size)
#define GEN_TS(id) do { if ((t & 63u) == 0u && ts_n < 64u) { sh.ts[(t & ~63u) + ts_n] = ((unsigned long long)cg_clock() << 8) | (unsigned long long)(id); ++ts_n; } } while (0)
// every wave leaves its own marks when it ends (a launch that found a well filled queue: the populated chain)
#define GEN_TS_DUMP_WAVE() do { if ((t & 63u) == 0u && ts_ok && (t >> 6) < 8u) { for (uint32_t i_ = 0; i_ < 64u; ++i_) g_timeline[(t & ~63u) + i_] = i_ < ts_n ? sh.ts[(t & ~63u) + i_] : 0ull; } } while (0)
#define GEN_TS_INIT() uint32_t ts_n = 0
#define GEN_TS_RESUME(k) ts_n = (k)
#define GEN_TS_ZERO(a, b) do { if ((t & 63u) == 0u) for (uint32_t i_ = (a); i_ < (b); ++i_) sh.ts[(t & ~63u) + i_] = 0ull; } while (0)
#define GEN_PIN(x) asm volatile("" : "+v"(x) :: "memory")      // the value is computed before the next timestamp
__device__ unsigned long long g_ahead_why[8];      // dev: why lanes of a window drawn ahead draw again (gen_draw_valid)
__device__ unsigned long long g_chain_gen[8];      // the chained launch's generator workgroup on the chip-wide 100 MHz clock (chain_kernel.h)
#define GEN_RT(i) do { if (t == 0u) sh.rt[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GEN_RT_AT(i, lane) do { if (t == (unsigned)(lane)) sh.rt[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GEN_LOG_N 65536
__device__ unsigned long long g_chain_log[GEN_LOG_N * 8]; __device__ unsigned int g_chain_log_n;      // one record per chained launch of a well filled queue
#define GEN_RT_DUMP() do { if (t == 0u && sh.rtOn) { for (int i_ = 0; i_ < 8; ++i_) g_chain_gen[i_] = sh.rt[i_]; } \
    if (t == 0u && sh.rtLog) { const unsigned int k_ = atomicAdd(&g_chain_log_n, 1u) % GEN_LOG_N; for (int i_ = 0; i_ < 6; ++i_) g_chain_log[k_ * 8 + i_] = sh.rt[i_]; \
        g_chain_log[k_ * 8 + 6] = sh.rtInfo; g_chain_log[k_ * 8 + 7] = (unsigned long long)roundNo; } } while (0)
#else
#define GEN_TS(id) do { } while (0)
#define GEN_TS_INIT() do { } while (0)
#define GEN_TS_RESUME(k) do { } while (0)
#define GEN_TS_ZERO(a, b) do { } while (0)
#define GEN_PIN(x) do { } while (0)
#define GEN_RT(i) do { } while (0)
#define GEN_RT_AT(i, lane) do { } while (0)
#define GEN_RT_DUMP() do { } while (0)
#define GEN_TS_DUMP_WAVE() do { } while (0)
#endif

// dev: which launch the timeline keeps -- any well filled one, or (GEN_TIMELINE_ROUND2) one whose batch needed exactly two rounds
#if defined(GEN_TIMELINE_ROUND2)
#define GEN_TS_ROUND_OK(r) ((r) == 2u)
#else
#define GEN_TS_ROUND_OK(r) true
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
// buckets of the LDS conflict table (round 1 of a batch), 4 slots each: 1024 for a window of 256 attempts (at most 768 registrations:
// 19 % of the slots), half of that for the 128-lane window -- the table is emptied at every launch (80 KB / 40 KB of LDS stores).
// (Both macros read the enclosing template's WIN.)
#define GEN_TAB_BBITS (WIN <= 128 ? 9 : 10)
#define GEN_TAB_NB (1 << GEN_TAB_BBITS)
// Rounds of a batch that keep their conflict sets in the LDS table (the first always does; gen_populate.h).  Nothing is ever removed
// from the table's keys, so a later round is admitted only while the keys of all admitted rounds (three registrations per lane and
// round, a committed birth's atom) stay below ~70 % of the slots; the stamp tables in HBM serve the rounds after them.
#ifndef GEN_LDS_ROUNDS_MAX
#define GEN_LDS_ROUNDS_MAX 3       // (test variants of the emulator build lower it: 1 = every later round through the stamp tables, 2 = the hand-over after two LDS rounds)
#endif
// (a round registers at most three keys per attempt, a committed birth's atom now and then: ~3.1 keys per attempt; 256 attempts: 3 rounds, 320 / 384: 2)
#define GEN_LDS_ROUNDS_FIT (((4 * GEN_TAB_NB) * 7 * 10) / (31 * WIN * 10))
#define GEN_LDS_ROUNDS (GEN_LDS_ROUNDS_FIT >= GEN_LDS_ROUNDS_MAX ? GEN_LDS_ROUNDS_MAX : (GEN_LDS_ROUNDS_FIT < 1 ? 1 : GEN_LDS_ROUNDS_FIT))
#define GEN_K_ROW 0u
#define GEN_K_ATOM 1u
#define GEN_K_GAP 2u
#define GEN_K_INL 3u
#define GEN_GS_WORDS ((uint32_t)(offsetof(GenScalars, evalProps) / 4u + 2u))      // words [0, GEN_GS_WORDS) of GenScalars are written back by the generator
#define GEN_GS_ERROR_WORD ((uint32_t)(offsetof(GenScalars, error) / 4u))
#ifndef FLUSH_MAX
#define FLUSH_MAX 64                 // erase caches up to this size are flushed in parallel
#endif
#define CG_KEEP 0xFFFFFFFEu          // "front unchanged" marker
#define GEN_CHAIN_THREADS 512        // workgroup size of the chained launches the generator body is part of (chain_kernel.h: CHAIN_MAX_THREADS)

#define GEN_DIRTY_ATOMS 4096      // 32-bit words of the note bit sets (a batch's decisions touch ~3 atom records and ~1.5 cells each: ~600 of 131072 bits)
#define GEN_DIRTY_CELLS 2048
#define GEN_DIRTY_ERASE 256       // (8192 bits for at most 3 x FLUSH_MAX keys)
struct GenTabVal { uint32_t used, gap, inl, pad; };
struct GenTabKeys { uint32_t k[4]; };

template <int WIN>
struct GenShared {
    uint64_t cpos[WIN], pos[WIN];        // centre position / destination, read by other lanes for queued moves
    float u1[WIN], u2[WIN];
    uint8_t type[WIN];                   // 'M' only when queued (birth-overlap test)
    uint64_t seed[WIN];                  // per attempt: seeder output (prefetched in attempt order)
    uint32_t info[WIN];                  // per attempt: type | bBefore << 8
    uint16_t perm[WIN];                  // lane -> attempt after sorting attempts by type
    unsigned long long mq[WIN / 64], mb[WIN / 64], md[WIN / 64];   // committed attempts: queued / birth / death bit masks
    uint32_t wtotA[3][WIN / 64], wtotB[3][WIN / 64], wtot4[4][WIN / 64];
    // flush: erase cache sorted by position (handles, links, vector indices, bins), the tail of the unsorted
    // vector and the net writes of the swap-with-last replay
    uint64_t fpos[FLUSH_MAX], flpos[FLUSH_MAX], frpos[FLUSH_MAX]; float frmass[FLUSH_MAX]; uint32_t fh[FLUSH_MAX], fl[FLUSH_MAX], fr[FLUSH_MAX], fidx[FLUSH_MAX], fbin[FLUSH_MAX], fhead[FLUSH_MAX], vt[FLUSH_MAX], lowSlot[FLUSH_MAX], lowH[FLUSH_MAX];
    uint32_t nLow, newFront, flushM, flushBase, unitSum, frontPending;
    uint32_t freeTop[16];                // the free-handle stack's top entries as the launch found them (below what its own flush pushes): a committing birth's handle without a memory trip
#if defined(GEN_TIMELINE)
    unsigned long long ts[8 * 64];
    unsigned long long rt[8]; uint32_t rtOn, rtLog; unsigned long long rtInfo;
#endif
    alignas(16) uint32_t bkey[4 * GEN_TAB_NB];      // conflict sets of round 1: keys, bucket-major
    alignas(16) GenTabVal bval[4 * GEN_TAB_NB];     // ... and the ordinals registered under each key
    float dpLo[WIN], dpHi[WIN];              // deathProb(minAtoms - k), deathProb(nAtoms + k) for this round
    uint64_t jmul[WIN + 1], jinc[WIN + 1];   // PCG jump by 2k steps, k = 0 .. WIN
    GenScalars g;                        // the generator's scalars, LDS-resident for the launch
    uint64_t qrngRound, batchEpoch;
    uint32_t roundNo, stopKey;
    uint32_t nR, minAtoms, processed, qlen, skip, remaining;
    uint32_t nWork, nBD, updBase; float u1c, u2c;      // nBD: births + deaths of the window by the first guess (their sorted slots come first)
    // chained launch (chain_kernel.h): the erase cache as the generator's own lanes fill it from the decisions they apply, and the window
    // of the death-probability table this launch can need (staged while the evaluation workgroups of the same launch still run)
    unsigned long long eraseTmp[FLUSH_MAX]; uint32_t eraseN, specBad, spinFail;
    float dpWin[4 * WIN];
    // `dirty`: one bit per level-0 bitmap word (mod 16384) that the decisions being applied or the flush change -- a birth drawn ahead whose
    // words are marked draws again
    uint32_t dirty[512];
    // ... and (round 5) the whole window DRAWN ahead: what the decisions being applied change, noted by the lanes that apply them --
    // atom records (handles), matrix cells (bins), the vector slots the flush refills (gen_populate.h, gen_draw_valid)
    // (bit sets, two hash positions per key; dErase: the atom records the FLUSH will rewrite -- an erased atom and its two neighbours;
    // anyRedo: 0 no lane of the window draws again, 1 some do and none of them reads what the flush changes, 2 some wait for the flush)
    alignas(16) uint32_t dAtom[GEN_DIRTY_ATOMS]; uint32_t dCell[GEN_DIRTY_CELLS]; uint32_t dErase[GEN_DIRTY_ERASE]; uint32_t anyRedo;
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

// bin / nPatterns, exact: the double product is below the true quotient by far less than 1/K, so the truncation
// is right or one short
CG_DEVICE uint32_t gen_div_k(const SamplerDev &S, uint32_t bin)
{
    uint32_t q = (uint32_t)((double)bin * S.invK);
    q += (uint32_t)(bin - q * S.K >= S.K);
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

// An atom's position / mass together with the copies its neighbours cache (gaps_state.h).  hl / hr: its neighbours' handles.
CG_DEVICE void atom_set_pos(const SamplerDev &S, uint32_t h, uint32_t hl, uint32_t hr, uint64_t p)
{
    S.atoms[h].pos = p;
    if (hl != CG_NONE) S.atoms[hl].rpos = p;
    if (hr != CG_NONE) S.atoms[hr].lpos = p;
}
CG_DEVICE void atom_set_mass(const SamplerDev &S, uint32_t h, uint32_t hl, float m)
{
    S.atoms[h].mass = m;
    if (hl != CG_NONE) S.atoms[hl].rmass = m;
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
    if (rec.left != CG_NONE) { S.atoms[rec.left].right = rec.right; S.atoms[rec.left].rpos = rec.rpos; S.atoms[rec.left].rmass = rec.rmass; } else front = rec.right;
    if (rec.right != CG_NONE) { S.atoms[rec.right].left = rec.left; S.atoms[rec.right].lpos = rec.lpos; }
    uint32_t b = gen_bin_of(S, rec.pos);
    if (S.binHead[b] == h) {
        uint32_t nxt = rec.right;
        if (nxt != CG_NONE && gen_bin_of(S, rec.rpos) == b) S.binHead[b] = nxt;
        else { S.binHead[b] = CG_NONE; bm_clear(S, b); }
    }
    uint32_t last = S.vec[n - 1];
    S.vec[rec.idx] = last;
    S.atoms[last].idx = rec.idx;
    --n;
    S.freeHandles[freeCount++] = h;
}

// exclusive counts of flags a,b,c before this lane + block totals (wave ballots + one LDS hop, one
// barrier; `w` must not be reused before the next barrier after the call)
template <int WIN>
CG_DEVICE void gen_count3(uint32_t (*w)[WIN / 64], unsigned t, bool a, bool b, bool c,
                          uint32_t &ea, uint32_t &eb, uint32_t &ec, uint32_t &ta, uint32_t &tb, uint32_t &tc)
{
    const unsigned lane = t & 63u, wave = t >> 6;
    const unsigned long long ma = cg_ballot(a), mb = cg_ballot(b), mc = cg_ballot(c);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lane == 0) { w[0][wave] = (uint32_t)cg_popc64(ma); w[1][wave] = (uint32_t)cg_popc64(mb); w[2][wave] = (uint32_t)cg_popc64(mc); }
    cg_sync_lds();
    ea = (uint32_t)cg_popc64(ma & lt); eb = (uint32_t)cg_popc64(mb & lt); ec = (uint32_t)cg_popc64(mc & lt); ta = 0; tb = 0; tc = 0;
    for (unsigned k = 0; k < (unsigned)(WIN / 64); ++k) {
        const uint32_t xa = w[0][k], xb = w[1][k], xc = w[2][k];
        const uint32_t before = k < wave ? 0xFFFFFFFFu : 0u;
        ea += xa & before; eb += xb & before; ec += xc & before;
        ta += xa; tb += xb; tc += xc;
    }
}

// the same for four flags (the classification's one exchange: births, deaths, moves, exchanges of the window by their first guess)
template <int WIN>
CG_DEVICE void gen_count4(uint32_t (*w)[WIN / 64], unsigned t, bool a, bool b, bool c, bool d, uint32_t (&e)[4], uint32_t (&tot)[4])
{
    const unsigned lane = t & 63u, wave = t >> 6;
    const unsigned long long m0 = cg_ballot(a), m1 = cg_ballot(b), m2 = cg_ballot(c), m3 = cg_ballot(d);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (lane == 0) { w[0][wave] = (uint32_t)cg_popc64(m0); w[1][wave] = (uint32_t)cg_popc64(m1); w[2][wave] = (uint32_t)cg_popc64(m2); w[3][wave] = (uint32_t)cg_popc64(m3); }
    cg_sync_lds();
    e[0] = (uint32_t)cg_popc64(m0 & lt); e[1] = (uint32_t)cg_popc64(m1 & lt); e[2] = (uint32_t)cg_popc64(m2 & lt); e[3] = (uint32_t)cg_popc64(m3 & lt);
    tot[0] = tot[1] = tot[2] = tot[3] = 0;
    for (unsigned k = 0; k < (unsigned)(WIN / 64); ++k) {
        const uint32_t before = k < wave ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int f = 0; f < 4; ++f) { const uint32_t x = w[f][k]; e[f] += x & before; tot[f] += x; }
    }
}

// ProposalQueue::makeProposal type choice (ProposalQueue.cpp:129-160); 0 = indeterminate
// lowerBound / upperBound = deathProb(minAtoms) / deathProb(maxAtoms), from the window's table.  Selects only.
CG_DEVICE uint32_t gen_decide(float u1, float u2, uint64_t minAtoms, uint64_t maxAtoms, float lowerBound, float upperBound)
{
    uint32_t r = (u2 >= upperBound) ? (uint32_t)'B' : (uint32_t)GEN_T_NONE;
    r = (u2 < lowerBound) ? (uint32_t)'D' : r;
    const uint32_t me = (u1 < 0.75f) ? (uint32_t)'M' : (uint32_t)'E';
    r = (u1 < 0.5f) ? r : me;
    r = (maxAtoms < 2) ? (uint32_t)'B' : r;
    r = ((uint32_t)(minAtoms < 2) & (uint32_t)(maxAtoms >= 2)) ? (uint32_t)GEN_T_NONE : r;
    return r;
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

// LDS conflict table of round 1 of a batch: GEN_TAB_NB buckets of four slots, keys and values in separate
// arrays so that one 16-byte read shows a whole bucket.  Keys are matrix rows (bit 31 set) and atom handles;
// the value words hold the smallest attempt ordinal that registered
//   used: the row / the atom as in use (mUsedMatrixIndices / mUsedAtoms)
//   gap : a birth landing right of the atom (before the front atom: pseudo-handle GEN_TAB_FRONT)
//   inl : a same-bin move / exchange touching the atom
// A key probes the slots of its bucket from a hash-chosen start, then the next bucket; nothing is ever removed,
// so a key sits in the first slot of its probe sequence that was empty when it arrived and a lookup may stop
// at the first empty slot.  Empty keys and values are all ones ("nobody").
#define GEN_TAB_ROW 0x80000000u
#define GEN_TAB_FRONT 0x7FFFFFFFu
#define GEN_TAB_EMPTY 0xFFFFFFFFu
CG_DEVICE uint32_t gen_tab_hash(uint32_t key) { return key * 2654435761u; }
template <int WIN> CG_DEVICE uint32_t gen_tab_bucket(uint32_t h) { return h >> (32 - GEN_TAB_BBITS); }
template <int WIN> CG_DEVICE uint32_t gen_tab_start(uint32_t h) { return (h >> (30 - GEN_TAB_BBITS)) & 3u; }
template <int WIN>
CG_DEVICE uint32_t gen_tab_claim(GenShared<WIN> &sh, uint32_t key)
{
    const uint32_t h = gen_tab_hash(key), j = gen_tab_start<WIN>(h);
    uint32_t b = gen_tab_bucket<WIN>(h);
    for (;;) {
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t s = 4u * b + ((j + i) & 3u);
            const uint32_t old = cg_atomic_cas_u32(&sh.bkey[s], GEN_TAB_EMPTY, key);
            if (old == GEN_TAB_EMPTY || old == key) return s;
        }
        b = (b + 1u) & (uint32_t)(GEN_TAB_NB - 1);
    }
}
// slot of `key`, or GEN_TAB_EMPTY when it was never registered
template <int WIN>
CG_DEVICE uint32_t gen_tab_find(const GenShared<WIN> &sh, uint32_t key)
{
    const uint32_t h = gen_tab_hash(key), j = gen_tab_start<WIN>(h);
    uint32_t b = gen_tab_bucket<WIN>(h);
    for (;;) {
        for (uint32_t i = 0; i < 4u; ++i) {
            const uint32_t s = 4u * b + ((j + i) & 3u);
            const uint32_t k = sh.bkey[s];
            if (k == key) return s;
            if (k == GEN_TAB_EMPTY) return GEN_TAB_EMPTY;
        }
        b = (b + 1u) & (uint32_t)(GEN_TAB_NB - 1);
    }
}
CG_DEVICE unsigned long long *gen_stamp_ptr(const SamplerDev &S, uint32_t kind, uint32_t id)
{
    return kind == GEN_K_ROW ? &S.rowStamp[id] : (kind == GEN_K_ATOM ? &S.atomStamp[id] : (kind == GEN_K_GAP ? &S.gapStamp[id] : &S.inlineStamp[id]));
}

// =================================================================================================

#include "gen_populate.h"
