// sparse_kernels.h -- SparseNormalModel (reference src/gibbs_sampler/SparseNormalModel.cpp) on the device.
//
// Data: vector r of the data matrix is a SparseVector (SparseVector.cpp:20-33): 64-bit flag words `dflags[r][*]`
// and the packed positive values `dvals[dptr[r] ..)`; `dprefix[r][w]` = number of values before word w.  The
// sampler's matrix is a HybridMatrix (HybridMatrix.cpp:25-39): a row copy `rows [M][Kpad]`, a column copy `mat
// [K][Mpad]` whose entries below epsilon are held at zero, and the column copy's flag words `mflags [K][Mw]`.  The two
// copies can disagree by less than epsilon and the reference reads both -- so does this file.
// No A*P cache exists in this model; an accepted proposal only changes matrix entries.
//
// Alpha parameters (SparseNormalModel.cpp:153-292): table terms Z1 / Z2 plus one term per index that is non-zero in
// both the data vector and the other matrix's column; every such term needs the K-length dot product of this
// sampler's matrix row with the other matrix's row at that index (gaps::dot in the scalar build's order).
// Lane order (the parity contract with the oracle): W = cogaps_sparse_width(N) virtual lanes = threads; lane L takes
// the flag words L, L+W, ... in increasing order, bits in increasing order, and accumulates its terms from +0; the
// lanes are folded by the ascending xor butterfly; the total is added to the table terms once; then beta.
#pragma once
#include "eval_kernel.h"
#include "aux_kernels.h"

#define SP_KMAX 512      // nPatterns limit of the sparse kernels (matrix rows staged in LDS)
// An evaluation workgroup has the model's width (cogaps_sparse_width: 256 threads at most); inside the chained launch, whose workgroups have 512
// threads, lanes 256.. are a second GROUP with a proposal and an LDS block of its own (chain_sparse_kernel).  sp_tid: the lane within its group.
#define SP_GROUP_THREADS 256u
#define SP_CHAIN_GROUPS(WIDE) ((WIDE) ? 1u : 2u)      // (the wide form's term list takes 80 KB: one group)
CG_DEVICE uint32_t sp_tid() { return cg_tid() & (SP_GROUP_THREADS - 1u); }
CG_DEVICE uint32_t sp_grp() { return cg_tid() / SP_GROUP_THREADS; }

// gaps::dot, scalar build (VectorMath.h:41-134): up to 25 elements are added last-to-first, more first-to-last
CG_DEVICE float sp_dot(const float *a, const float *b, uint32_t n)
{
    float d = 0.f;
    if (n <= 25u) { for (uint32_t i = n; i-- > 0u;) d = d + a[i] * b[i]; }
    else { for (uint32_t i = 0; i < n; ++i) d = d + a[i] * b[i]; }
    return d;
}

// A row of the other matrix's row copy (16-byte aligned, padded to a multiple of 4) held in registers: up to 64
// elements, all float4 loads issued together.
struct SpRow { cg_f4 r[16]; };
CG_DEVICE void sp_row_load(SpRow &R, const float *row, uint32_t n)
{
    const uint32_t nq = cg_fresh_u32((n + 3u) >> 2);
#pragma unroll
    for (uint32_t c = 0; c < 16u; ++c) R.r[c] = c < nq ? ld4(row, c) : f4_zero();
}
// gaps::dot(a, row) in the scalar build's order (sp_dot) with the row in registers, n <= 64.  `a` (LDS) and the row are
// zero-padded to a multiple of 4, so whole chunks are accumulated without per-element tests: a padding element adds
// a*0 = +-0 to the sum, which leaves its bits unchanged (the sum starts at +0), and one uniform test per chunk replaces
// four compare-and-branch pairs -- the kernel was issue-bound on exactly those.
CG_DEVICE float sp_row_dot(const float *a, const SpRow &R, uint32_t n)
{
    float d = 0.f;
    n = cg_fresh_u32(n);
    const uint32_t nq = (n + 3u) >> 2;
    if (n <= 25u) {
#pragma unroll
        for (uint32_t c = 7u; c-- > 0u;) {
            if (c < nq) {
                const uint32_t i = 4u * c;
                d = d + a[i + 3u] * R.r[c].w; d = d + a[i + 2u] * R.r[c].z; d = d + a[i + 1u] * R.r[c].y; d = d + a[i] * R.r[c].x;
            }
        }
        return d;
    }
#pragma unroll
    for (uint32_t c = 0; c < 16u; ++c) {
        if (c < nq) {
            const uint32_t i = 4u * c;
            d = d + a[i] * R.r[c].x; d = d + a[i + 1u] * R.r[c].y; d = d + a[i + 2u] * R.r[c].z; d = d + a[i + 3u] * R.r[c].w;
        }
    }
    return d;
}
// any length: straight from memory, 8 float4 at a time
CG_DEVICE float sp_dot_row(const float *a, const float *row, uint32_t n)
{
    float d = 0.f;
    const uint32_t nq = (n + 3u) >> 2;
    if (n <= 25u) {      // last-to-first: at most 7 chunks, all loaded first
        cg_f4 r[7];
#pragma unroll
        for (uint32_t c = 0; c < 7u; ++c) r[c] = c < nq ? ld4(row, c) : f4_zero();
#pragma unroll
        for (uint32_t c = 7u; c-- > 0u;) {
            const uint32_t i = 4u * c;
            if (i + 3u < n) d = d + a[i + 3u] * r[c].w;
            if (i + 2u < n) d = d + a[i + 2u] * r[c].z;
            if (i + 1u < n) d = d + a[i + 1u] * r[c].y;
            if (i < n) d = d + a[i] * r[c].x;
        }
        return d;
    }
    for (uint32_t c0 = 0; c0 < nq; c0 += 8u) {
        cg_f4 r[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) r[u] = (c0 + u) < nq ? ld4(row, c0 + u) : f4_zero();
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t i = 4u * (c0 + u);
            if (i < n) d = d + a[i] * r[u].x;
            if (i + 1u < n) d = d + a[i + 1u] * r[u].y;
            if (i + 2u < n) d = d + a[i + 2u] * r[u].z;
            if (i + 3u < n) d = d + a[i + 3u] * r[u].w;
        }
    }
    return d;
}

#define SP_MODE_ONE 0
#define SP_MODE_CH 1
#define SP_MODE_SAME 2

// one common non-zero's term added to the lane's partial sums (SparseNormalModel.cpp:176-186, 222-233, 274-285)
// ts is added to s, tm (then tm2, with a change) to s_mu.  (a - x is a + (-x) bit for bit.)
template <int MODE>
CG_DEVICE void sp_term_vals(float d_val, float v_val, float v2_val, float ex, float ap, float ch, float &ts, float &tm, float &tm2)
{
    tm2 = 0.f;
    if (MODE == SP_MODE_SAME) {
        const float d_recip = 1.f / d_val;
        const float term1 = 1.f - d_recip * d_recip;
        const float v_diff = v_val - v2_val;
        ts = -(v_diff * v_diff * term1);
        tm = v_diff * (ap * term1 + d_recip);
    } else {
        const float term1 = v_val / d_val;
        const float term2 = v_val - term1 / d_val;
        ts = term1 * term1 - v_val * v_val;
        tm = term1 + term2 * ap;
        if (MODE == SP_MODE_CH) tm2 = term2 * ex * ch;
    }
}
template <int MODE>
CG_DEVICE void sp_term(float d_val, float v_val, float v2_val, float ex, float ap, float ch, float &ps, float &pm)
{
    float ts, tm, tm2;
    sp_term_vals<MODE>(d_val, v_val, v2_val, ex, ap, ch, ts, tm, tm2);
    ps = ps + ts;
    pm = pm + tm;
    if (MODE == SP_MODE_CH) pm = pm + tm2;
}

// ---- per-lane partial sums of one alpha evaluation, lane-balanced ------------------------------------------------------------
// Lane order (the contract with the oracle): thread L owns the flag words L, L+W, ... and adds their common non-zeros' terms in
// increasing word and bit order from +0.  The common non-zeros are unevenly spread over the words: a lane that owns a word with
// five of them would do five K-length dots while its neighbours idle (data vectors of 2000 elements fill only 32 of the 64 lanes with a word at all).  Here the words' owners
// only LIST their common non-zeros (element index, position of the data value) in LDS -- slots handed out by one LDS atomic per
// word, the word's entries contiguous and in bit order --, every lane of the workgroup then takes listed entries round-robin and
// computes the entry's complete term (the loads, the dot, sp_term_vals), and the owner folds its own word's terms back in bit
// order: the same additions in the same order as if every owner had worked through its words alone.  A round (one word per thread)
// that lists more than SP_BAL_CAP entries is done that way.
#define SP_BAL_CAP 2048            // ... of the one-round form; the merged form (sp_partial_merged) lists up to SP_BAL_CAP_WIDE entries
#define SP_BAL_CAP_WIDE 4096
template <int CAP> struct SpBal { uint32_t n; uint32_t idx[CAP], dpos[CAP]; float ts[CAP], tm[CAP], tm2[CAP]; };

template <int MODE>
CG_DEVICE void sp_bal_term(const SamplerDev &S, uint32_t col, float ch, const float *arow, const float *data, const float *V, const float *V2,
                           uint32_t idx, uint32_t dpos, float &ts, float &tm, float &tm2)
{
    const float d = data[dpos];
    const float v = V[idx], v2 = (MODE == SP_MODE_SAME) ? V2[idx] : 0.f;
    const float ex = (MODE == SP_MODE_CH) ? S.orows[(size_t)idx * S.oKpad + col] : 0.f;
    float ap;
    if (S.K <= 64u) { SpRow R; sp_row_load(R, S.orows + (size_t)idx * S.oKpad, S.K); ap = sp_row_dot(arow, R, S.K); }
    else ap = sp_dot_row(arow, S.orows + (size_t)idx * S.oKpad, S.K);
    sp_term_vals<MODE>(d, v, v2, ex, ap, ch, ts, tm, tm2);
}

// the words of round 0 (word index = thread index), loaded by the caller together with the matrix rows so that they travel in the
// same memory trip: data flags, the (or-ed) column flags, the number of packed values before the word
struct SpPre { unsigned long long dfl, fv; uint32_t dbase; };
CG_DEVICE SpPre sp_preload(const SamplerDev &S, uint32_t row, uint32_t col, uint32_t col2, bool same)
{
    SpPre o; o.dfl = 0ull; o.fv = 0ull; o.dbase = 0u;
    const uint32_t w = sp_tid();
    if (w < S.Wn) {
        o.dfl = S.dflags[(size_t)row * S.Wn + w];
        o.fv = S.oflags[(size_t)col * S.oMw + w];
        if (same) o.fv |= S.oflags[(size_t)col2 * S.oMw + w];
        o.dbase = S.dprefix[(size_t)row * S.Wn + w];
    }
    return o;
}

// round by round (one word per thread and round): any vector length, any number of common non-zeros
template <int MODE, int CAP>
CG_DEVICE void sp_partial_rounds(const SamplerDev &S, uint32_t row, uint32_t col, uint32_t col2, float ch, const float *arow, SpBal<CAP> &bal, const SpPre &pre0, float &ps, float &pm, uint32_t &visited)
{
    const uint32_t BS = S.spW, t = sp_tid();
    const unsigned long long *fD = S.dflags + (size_t)row * S.Wn;
    const unsigned long long *fV = S.oflags + (size_t)col * S.oMw, *fV2 = S.oflags + (size_t)col2 * S.oMw;
    const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
    const float *data = S.dvals + S.dptr[row];
    const float *V = S.other + (size_t)col * S.Npad, *V2 = S.other + (size_t)col2 * S.Npad;
    ps = 0.f; pm = 0.f;
    for (uint32_t w0 = 0; w0 < S.Wn; w0 += BS) {
        const uint32_t w = w0 + t;
        if (t == 0) bal.n = 0u;
        cg_sync();
        unsigned long long dfl = 0ull, common = 0ull; uint32_t base = 0, cnt = 0, dbase = 0;
        if (w < S.Wn) {
            if (w0 == 0u) { dfl = pre0.dfl; common = dfl & pre0.fv; dbase = pre0.dbase; }
            else {
                dfl = fD[w];
                common = dfl & (MODE == SP_MODE_SAME ? (fV[w] | fV2[w]) : fV[w]);
                dbase = pre[w];
            }
            cnt = (uint32_t)cg_popc64(common);
            visited += cnt;
        }
        {   // list slots: a wave's words take one block of the list (one LDS atomic per wave), a word's offset inside it from a DPP
            // prefix sum over the wave's counts.  (An atomicAdd-with-return per lane at the one address is what the compiler turns
            // into a serial scan over the active lanes: 64 iterations, 2-3 k cycles per wave and call.)
            uint32_t waveTot;
            const uint32_t ex = cg_wave_excl_scan_u32(cnt, waveTot);
            uint32_t wbase = 0;
            if ((t & 63u) == 0u && waveTot) wbase = cg_atomic_add_u32(&bal.n, waveTot);
            base = cg_wave_bcast_u32(wbase, 0) + ex;
        }
        if (w < S.Wn) {
            if (cnt) {
                if (base + cnt <= (uint32_t)CAP) {
                    unsigned long long c = common; uint32_t j = 0;
                    while (c != 0ull) {
                        const uint32_t bit = (uint32_t)cg_ctz64(c); c &= c - 1ull;
                        bal.idx[base + j] = 64u * w + bit;
                        bal.dpos[base + j] = dbase + (uint32_t)cg_popc64(dfl & ((1ull << bit) - 1ull));
                        ++j;
                    }
                }
            }
        }
        cg_sync();
        const uint32_t total = bal.n;
        if (total <= (uint32_t)CAP) {
            // every lane: listed entries round-robin, two in flight
            for (uint32_t e = t; e < total; e += 2u * BS) {
                const uint32_t e1 = e + BS; const bool second = e1 < total;
                float a0, b0, c0, a1 = 0.f, b1 = 0.f, c1 = 0.f;
                const uint32_t i0 = bal.idx[e], p0 = bal.dpos[e], i1 = second ? bal.idx[e1] : i0, p1 = second ? bal.dpos[e1] : p0;
                sp_bal_term<MODE>(S, col, ch, arow, data, V, V2, i0, p0, a0, b0, c0);
                if (second) sp_bal_term<MODE>(S, col, ch, arow, data, V, V2, i1, p1, a1, b1, c1);
                bal.ts[e] = a0; bal.tm[e] = b0; if (MODE == SP_MODE_CH) bal.tm2[e] = c0;
                if (second) { bal.ts[e1] = a1; bal.tm[e1] = b1; if (MODE == SP_MODE_CH) bal.tm2[e1] = c1; }
            }
            cg_sync();
            for (uint32_t j = 0; j < cnt; ++j) {       // the word's owner: its terms in bit order
                ps = ps + bal.ts[base + j];
                pm = pm + bal.tm[base + j];
                if (MODE == SP_MODE_CH) pm = pm + bal.tm2[base + j];
            }
        } else {
            // a round with more common non-zeros than the list holds: every owner works through its own word
            while (common != 0ull) {
                const uint32_t bit = (uint32_t)cg_ctz64(common); common &= common - 1ull;
                float a0, b0, c0;
                sp_bal_term<MODE>(S, col, ch, arow, data, V, V2, 64u * w + bit, dbase + (uint32_t)cg_popc64(dfl & ((1ull << bit) - 1ull)), a0, b0, c0);
                ps = ps + a0; pm = pm + b0; if (MODE == SP_MODE_CH) pm = pm + c0;
            }
        }
        cg_sync();      // the list is reused by the next round / the next call
    }
}

// Up to SP_MERGE_ROUNDS rounds of words listed TOGETHER, then one pass over all listed entries, then the owners fold their words' terms
// in round order (= increasing word order, as the lane order prescribes).  A round by itself lists about one and a half entries per
// lane: its gather trip, three barriers and the fold were paid per round -- four times per alpha evaluation on BASELINE configs[4]'s
// 50000-element vectors -- for work that fits one trip.  Falls back to the round-by-round form when the vector has more rounds or the
// list would overflow.
#define SP_MERGE_ROUNDS 4
template <int MODE, int CAP>
CG_DEVICE void sp_partial_merged(const SamplerDev &S, uint32_t row, uint32_t col, uint32_t col2, float ch, const float *arow, SpBal<CAP> &bal, const SpPre &pre0, float &ps, float &pm, uint32_t &visited)
{
    const uint32_t BS = S.spW, t = sp_tid();
    if (S.Wn <= BS || S.Wn > (uint32_t)SP_MERGE_ROUNDS * BS) { sp_partial_rounds<MODE, CAP>(S, row, col, col2, ch, arow, bal, pre0, ps, pm, visited); return; }      // (one round: nothing to merge)
    const unsigned long long *fD = S.dflags + (size_t)row * S.Wn;
    const unsigned long long *fV = S.oflags + (size_t)col * S.oMw, *fV2 = S.oflags + (size_t)col2 * S.oMw;
    const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
    const float *data = S.dvals + S.dptr[row];
    const float *V = S.other + (size_t)col * S.Npad, *V2 = S.other + (size_t)col2 * S.Npad;
    if (t == 0) bal.n = 0u;
    cg_sync();
    // all the rounds' words in one trip
    unsigned long long dfl[SP_MERGE_ROUNDS], common[SP_MERGE_ROUNDS]; uint32_t dbase[SP_MERGE_ROUNDS], cnt[SP_MERGE_ROUNDS], base[SP_MERGE_ROUNDS];
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) {
        const uint32_t w = (uint32_t)r * BS + t;
        dfl[r] = 0ull; common[r] = 0ull; dbase[r] = 0u; cnt[r] = 0u; base[r] = 0u;
        if (w < S.Wn) {
            if (r == 0) { dfl[0] = pre0.dfl; common[0] = pre0.dfl & pre0.fv; dbase[0] = pre0.dbase; }
            else { dfl[r] = fD[w]; common[r] = dfl[r] & (MODE == SP_MODE_SAME ? (fV[w] | fV2[w]) : fV[w]); dbase[r] = pre[w]; }
        }
    }
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) { cnt[r] = (uint32_t)cg_popc64(common[r]); mine += cnt[r]; }
    {   // list slots: one block per wave (one LDS atomic), a thread's rounds contiguous inside it
        uint32_t waveTot;
        const uint32_t ex = cg_wave_excl_scan_u32(mine, waveTot);
        uint32_t wbase = 0;
        if ((t & 63u) == 0u && waveTot) wbase = cg_atomic_add_u32(&bal.n, waveTot);
        uint32_t b = cg_wave_bcast_u32(wbase, 0) + ex;
#pragma unroll
        for (int r = 0; r < SP_MERGE_ROUNDS; ++r) { base[r] = b; b += cnt[r]; }
    }
    cg_sync();
    const uint32_t total = bal.n;
    if (total > (uint32_t)CAP) {      // (uniform) more common non-zeros than the list holds: round by round
        cg_sync();
        sp_partial_rounds<MODE, CAP>(S, row, col, col2, ch, arow, bal, pre0, ps, pm, visited);
        return;
    }
    visited += mine;
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) {
        unsigned long long c = common[r]; uint32_t j = 0;
        const uint32_t w = (uint32_t)r * BS + t;
        while (c != 0ull) {
            const uint32_t bit = (uint32_t)cg_ctz64(c); c &= c - 1ull;
            bal.idx[base[r] + j] = 64u * w + bit;
            bal.dpos[base[r] + j] = dbase[r] + (uint32_t)cg_popc64(dfl[r] & ((1ull << bit) - 1ull));
            ++j;
        }
    }
    cg_sync();
    for (uint32_t e = t; e < total; e += 2u * BS) {      // every lane: listed entries round-robin, two in flight
        const uint32_t e1 = e + BS; const bool second = e1 < total;
        float a0, b0, c0, a1 = 0.f, b1 = 0.f, c1 = 0.f;
        const uint32_t i0 = bal.idx[e], p0 = bal.dpos[e], i1 = second ? bal.idx[e1] : i0, p1 = second ? bal.dpos[e1] : p0;
        sp_bal_term<MODE>(S, col, ch, arow, data, V, V2, i0, p0, a0, b0, c0);
        if (second) sp_bal_term<MODE>(S, col, ch, arow, data, V, V2, i1, p1, a1, b1, c1);
        bal.ts[e] = a0; bal.tm[e] = b0; if (MODE == SP_MODE_CH) bal.tm2[e] = c0;
        if (second) { bal.ts[e1] = a1; bal.tm[e1] = b1; if (MODE == SP_MODE_CH) bal.tm2[e1] = c1; }
    }
    cg_sync();
    ps = 0.f; pm = 0.f;
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) {          // the owner: its words in increasing order, a word's terms in bit order
        for (uint32_t j = 0; j < cnt[r]; ++j) {
            ps = ps + bal.ts[base[r] + j];
            pm = pm + bal.tm[base[r] + j];
            if (MODE == SP_MODE_CH) pm = pm + bal.tm2[base[r] + j];
        }
    }
    cg_sync();      // the list is reused by the next call
}

// The two alpha evaluations of a move / exchange across rows (SparseNormalModel.cpp:242-292: alphaParameters(r1, c1) + alphaParameters(r2, c2)),
// one-round vectors: both evaluations' common non-zeros in ONE list, one pass over the entries, each owner folds its word's terms per
// evaluation.  One after the other they paid the list, the gather trip and the barriers twice -- and an evaluation launch lasts as long
// as its slowest workgroup, which nearly always is one of the few two-row proposals of the batch.  (x: {s, s_mu} of the first
// evaluation, then of the second; bit 31 of a listed index marks the second.)
template <int CAP>
CG_DEVICE void sp_partial_pair(const SamplerDev &S, uint32_t rowA, uint32_t colA, const float *arowA, const SpPre &preA, uint32_t rowB, uint32_t colB, const float *arowB, const SpPre &preB,
                               SpBal<CAP> &bal, float (&x)[4], uint32_t &visited)
{
    const uint32_t BS = S.spW, t = sp_tid();
    const float *dataA = S.dvals + S.dptr[rowA], *dataB = S.dvals + S.dptr[rowB];
    const float *VA = S.other + (size_t)colA * S.Npad, *VB = S.other + (size_t)colB * S.Npad;
    if (t == 0) bal.n = 0u;
    cg_sync();
    const unsigned long long comA = preA.dfl & preA.fv, comB = preB.dfl & preB.fv;      // (words past the vector's last read as 0: sp_preload)
    const uint32_t cntA = (uint32_t)cg_popc64(comA), cntB = (uint32_t)cg_popc64(comB);
    uint32_t baseA, baseB;
    {
        uint32_t waveTot;
        const uint32_t ex = cg_wave_excl_scan_u32(cntA + cntB, waveTot);
        uint32_t wbase = 0;
        if ((t & 63u) == 0u && waveTot) wbase = cg_atomic_add_u32(&bal.n, waveTot);
        baseA = cg_wave_bcast_u32(wbase, 0) + ex; baseB = baseA + cntA;
    }
    cg_sync();
    const uint32_t total = bal.n;
    if (total > (uint32_t)CAP) {      // (uniform) rare: one after the other
        cg_sync();
        sp_partial_rounds<SP_MODE_ONE, CAP>(S, rowA, colA, 0u, 0.f, arowA, bal, preA, x[0], x[1], visited);
        sp_partial_rounds<SP_MODE_ONE, CAP>(S, rowB, colB, 0u, 0.f, arowB, bal, preB, x[2], x[3], visited);
        return;
    }
    visited += cntA + cntB;
    {
        unsigned long long c = comA; uint32_t j = 0;
        while (c != 0ull) { const uint32_t bit = (uint32_t)cg_ctz64(c); c &= c - 1ull; bal.idx[baseA + j] = 64u * t + bit; bal.dpos[baseA + j] = preA.dbase + (uint32_t)cg_popc64(preA.dfl & ((1ull << bit) - 1ull)); ++j; }
        c = comB; j = 0;
        while (c != 0ull) { const uint32_t bit = (uint32_t)cg_ctz64(c); c &= c - 1ull; bal.idx[baseB + j] = (64u * t + bit) | 0x80000000u; bal.dpos[baseB + j] = preB.dbase + (uint32_t)cg_popc64(preB.dfl & ((1ull << bit) - 1ull)); ++j; }
    }
    cg_sync();
    for (uint32_t e = t; e < total; e += 2u * BS) {      // every lane: listed entries round-robin, two in flight
        const uint32_t e1 = e + BS; const bool second = e1 < total;
        float a0, b0, c0, a1 = 0.f, b1 = 0.f, c1 = 0.f;
        const uint32_t i0 = bal.idx[e], p0 = bal.dpos[e], i1 = second ? bal.idx[e1] : i0, p1 = second ? bal.dpos[e1] : p0;
        const bool B0 = (i0 >> 31) != 0u, B1 = (i1 >> 31) != 0u;
        sp_bal_term<SP_MODE_ONE>(S, B0 ? colB : colA, 0.f, B0 ? arowB : arowA, B0 ? dataB : dataA, B0 ? VB : VA, VA, i0 & 0x7FFFFFFFu, p0, a0, b0, c0);
        if (second) sp_bal_term<SP_MODE_ONE>(S, B1 ? colB : colA, 0.f, B1 ? arowB : arowA, B1 ? dataB : dataA, B1 ? VB : VA, VA, i1 & 0x7FFFFFFFu, p1, a1, b1, c1);
        bal.ts[e] = a0; bal.tm[e] = b0;
        if (second) { bal.ts[e1] = a1; bal.tm[e1] = b1; }
    }
    cg_sync();
    x[0] = 0.f; x[1] = 0.f; x[2] = 0.f; x[3] = 0.f;
    for (uint32_t j = 0; j < cntA; ++j) { x[0] = x[0] + bal.ts[baseA + j]; x[1] = x[1] + bal.tm[baseA + j]; }
    for (uint32_t j = 0; j < cntB; ++j) { x[2] = x[2] + bal.ts[baseB + j]; x[3] = x[3] + bal.tm[baseB + j]; }
    cg_sync();      // the list is reused by the next call
}

// sp_partial_pair for the wide kernel: the words of both evaluations' rounds listed into the one list (first evaluation, then second, bit 31
// marking the second), one pass over the entries, the owners' folds per evaluation in round order.
template <int CAP>
CG_DEVICE void sp_list_rounds(const SamplerDev &S, uint32_t row, uint32_t col, const SpPre &pre0, SpBal<CAP> &bal, uint32_t tag,
                              uint32_t (&cnt)[SP_MERGE_ROUNDS], uint32_t (&base)[SP_MERGE_ROUNDS], uint32_t &mine)
{
    const uint32_t BS = S.spW, t = sp_tid();
    const unsigned long long *fD = S.dflags + (size_t)row * S.Wn, *fV = S.oflags + (size_t)col * S.oMw;
    const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
    unsigned long long dfl[SP_MERGE_ROUNDS], common[SP_MERGE_ROUNDS]; uint32_t dbase[SP_MERGE_ROUNDS];
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) {
        const uint32_t w = (uint32_t)r * BS + t;
        dfl[r] = 0ull; common[r] = 0ull; dbase[r] = 0u;
        if (w < S.Wn) {
            if (r == 0) { dfl[0] = pre0.dfl; common[0] = pre0.dfl & pre0.fv; dbase[0] = pre0.dbase; }
            else { dfl[r] = fD[w]; common[r] = dfl[r] & fV[w]; dbase[r] = pre[w]; }
        }
    }
    mine = 0;
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) { cnt[r] = (uint32_t)cg_popc64(common[r]); mine += cnt[r]; }
    {
        uint32_t waveTot;
        const uint32_t ex = cg_wave_excl_scan_u32(mine, waveTot);
        uint32_t wbase = 0;
        if ((t & 63u) == 0u && waveTot) wbase = cg_atomic_add_u32(&bal.n, waveTot);
        uint32_t b = cg_wave_bcast_u32(wbase, 0) + ex;
#pragma unroll
        for (int r = 0; r < SP_MERGE_ROUNDS; ++r) { base[r] = b; b += cnt[r]; }
    }
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) {
        if (base[r] + cnt[r] <= (uint32_t)CAP) {      // (a list that overflows is not used: the caller falls back)
            unsigned long long c = common[r]; uint32_t j = 0;
            const uint32_t w = (uint32_t)r * BS + t;
            while (c != 0ull) {
                const uint32_t bit = (uint32_t)cg_ctz64(c); c &= c - 1ull;
                bal.idx[base[r] + j] = (64u * w + bit) | tag;
                bal.dpos[base[r] + j] = dbase[r] + (uint32_t)cg_popc64(dfl[r] & ((1ull << bit) - 1ull));
                ++j;
            }
        }
    }
}
template <int CAP>
CG_DEVICE void sp_partial_merged_pair(const SamplerDev &S, uint32_t rowA, uint32_t colA, const float *arowA, const SpPre &preA, uint32_t rowB, uint32_t colB, const float *arowB, const SpPre &preB,
                                      SpBal<CAP> &bal, float (&x)[4], uint32_t &visited)
{
    const uint32_t BS = S.spW, t = sp_tid();
    if (S.Wn <= BS || S.Wn > (uint32_t)SP_MERGE_ROUNDS * BS || S.N >= 0x80000000u) {
        sp_partial_merged<SP_MODE_ONE, CAP>(S, rowA, colA, 0u, 0.f, arowA, bal, preA, x[0], x[1], visited);
        sp_partial_merged<SP_MODE_ONE, CAP>(S, rowB, colB, 0u, 0.f, arowB, bal, preB, x[2], x[3], visited);
        return;
    }
    const float *dataA = S.dvals + S.dptr[rowA], *dataB = S.dvals + S.dptr[rowB];
    const float *VA = S.other + (size_t)colA * S.Npad, *VB = S.other + (size_t)colB * S.Npad;
    if (t == 0) bal.n = 0u;
    cg_sync();
    uint32_t cntA[SP_MERGE_ROUNDS], baseA[SP_MERGE_ROUNDS], cntB[SP_MERGE_ROUNDS], baseB[SP_MERGE_ROUNDS], mineA, mineB;
    sp_list_rounds<CAP>(S, rowA, colA, preA, bal, 0u, cntA, baseA, mineA);
    sp_list_rounds<CAP>(S, rowB, colB, preB, bal, 0x80000000u, cntB, baseB, mineB);
    cg_sync();
    const uint32_t total = bal.n;
    if (total > (uint32_t)CAP) {      // (uniform) one evaluation after the other, each with the list to itself
        cg_sync();
        sp_partial_merged<SP_MODE_ONE, CAP>(S, rowA, colA, 0u, 0.f, arowA, bal, preA, x[0], x[1], visited);
        sp_partial_merged<SP_MODE_ONE, CAP>(S, rowB, colB, 0u, 0.f, arowB, bal, preB, x[2], x[3], visited);
        return;
    }
    visited += mineA + mineB;
    for (uint32_t e = t; e < total; e += 2u * BS) {      // every lane: listed entries round-robin, two in flight
        const uint32_t e1 = e + BS; const bool second = e1 < total;
        float a0, b0, c0, a1 = 0.f, b1 = 0.f, c1 = 0.f;
        const uint32_t i0 = bal.idx[e], p0 = bal.dpos[e], i1 = second ? bal.idx[e1] : i0, p1 = second ? bal.dpos[e1] : p0;
        const bool B0 = (i0 >> 31) != 0u, B1 = (i1 >> 31) != 0u;
        sp_bal_term<SP_MODE_ONE>(S, B0 ? colB : colA, 0.f, B0 ? arowB : arowA, B0 ? dataB : dataA, B0 ? VB : VA, VA, i0 & 0x7FFFFFFFu, p0, a0, b0, c0);
        if (second) sp_bal_term<SP_MODE_ONE>(S, B1 ? colB : colA, 0.f, B1 ? arowB : arowA, B1 ? dataB : dataA, B1 ? VB : VA, VA, i1 & 0x7FFFFFFFu, p1, a1, b1, c1);
        bal.ts[e] = a0; bal.tm[e] = b0;
        if (second) { bal.ts[e1] = a1; bal.tm[e1] = b1; }
    }
    cg_sync();
    x[0] = 0.f; x[1] = 0.f; x[2] = 0.f; x[3] = 0.f;
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) for (uint32_t j = 0; j < cntA[r]; ++j) { x[0] = x[0] + bal.ts[baseA[r] + j]; x[1] = x[1] + bal.tm[baseA[r] + j]; }
#pragma unroll
    for (int r = 0; r < SP_MERGE_ROUNDS; ++r) for (uint32_t j = 0; j < cntB[r]; ++j) { x[2] = x[2] + bal.ts[baseB[r] + j]; x[3] = x[3] + bal.tm[baseB[r] + j]; }
    cg_sync();      // the list is reused by the next call
}

// WIDE: the kernel instantiation for data vectors whose flag words take several rounds (launch_eval); the one-round kernel does not
// carry the merged form's registers (205 against 145 VGPRs) and its 90 KB list
template <int MODE, bool WIDE, int CAP>
CG_DEVICE void sp_partial_balanced(const SamplerDev &S, uint32_t row, uint32_t col, uint32_t col2, float ch, const float *arow, SpBal<CAP> &bal, const SpPre &pre0, float &ps, float &pm, uint32_t &visited)
{
    if (WIDE) sp_partial_merged<MODE, CAP>(S, row, col, col2, ch, arow, bal, pre0, ps, pm, visited);
    else sp_partial_rounds<MODE, CAP>(S, row, col, col2, ch, arow, bal, pre0, ps, pm, visited);
}

// ---- verification mode: SparseNormalModel.cpp:153-292 in the reference's own order -------------------------------------
// s and s_mu start from the table terms and take one term per common non-zero in ascending index (the `ch` form adds a
// second term to s_mu per element).  The terms are computed by the whole workgroup into its scratch rows in index order
// (word offsets from a serial scan of the per-word counts), then thread 0 folds them.  Returns (s, s_mu) BEFORE beta, in
// every thread.  wcnt: LDS [SP_SEQ_WORDS]; bc: LDS [2].
#define SP_SEQ_WORDS 4096
template <int MODE>
CG_DEVICE void sp_alpha_seq(const SamplerDev &S, uint32_t row, uint32_t col, uint32_t col2, float ch, const float *arow, float s0, float m0,
                            uint32_t *wcnt, float *bc, float &sOut, float &mOut, uint32_t &visited)
{
    const uint32_t BS = S.spW, t = sp_tid(), K = S.K;
    const unsigned long long *fD = S.dflags + (size_t)row * S.Wn;
    const unsigned long long *fV = S.oflags + (size_t)col * S.oMw, *fV2 = S.oflags + (size_t)col2 * S.oMw;
    const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
    const float *data = S.dvals + S.dptr[row];
    const float *V = S.other + (size_t)col * S.Npad, *V2 = S.other + (size_t)col2 * S.Npad;
    float *sc = S.seqScratch + (size_t)cg_bid() * 3u * S.Npad;
    for (uint32_t w = t; w < S.Wn; w += BS) wcnt[w] = (uint32_t)cg_popc64(fD[w] & (MODE == SP_MODE_SAME ? (fV[w] | fV2[w]) : fV[w]));
    cg_sync();
    if (t == 0) { uint32_t run = 0; for (uint32_t w = 0; w < S.Wn; ++w) { const uint32_t c = wcnt[w]; wcnt[w] = run; run += c; } wcnt[S.Wn] = run; }
    cg_sync();
    const uint32_t total = wcnt[S.Wn];
    for (uint32_t w = t; w < S.Wn; w += BS) {
        const unsigned long long dfl = fD[w];
        unsigned long long common = dfl & (MODE == SP_MODE_SAME ? (fV[w] | fV2[w]) : fV[w]);
        uint32_t o = wcnt[w];
        while (common != 0ull) {
            const uint32_t bit = (uint32_t)cg_ctz64(common);
            common &= common - 1ull;
            const uint32_t idx = 64u * w + bit;
            const float d = data[pre[w] + (uint32_t)cg_popc64(dfl & ((1ull << bit) - 1ull))];
            const float v = V[idx], v2 = (MODE == SP_MODE_SAME) ? V2[idx] : 0.f;
            const float ex = (MODE == SP_MODE_CH) ? S.orows[(size_t)idx * S.oKpad + col] : 0.f;
            const float ap = sp_dot(arow, S.orows + (size_t)idx * S.oKpad, K);
            float ts, tm, tm2;
            sp_term_vals<MODE>(d, v, v2, ex, ap, ch, ts, tm, tm2);
            sc[o] = ts; sc[S.Npad + o] = tm; if (MODE == SP_MODE_CH) sc[2u * S.Npad + o] = tm2;
            ++o;
        }
    }
    cg_sync();
    if (t == 0) {
        float s = s0, m = m0;
        for (uint32_t i = 0; i < total; ++i) { s = s + sc[i]; m = m + sc[S.Npad + i]; if (MODE == SP_MODE_CH) m = m + sc[2u * S.Npad + i]; }
        bc[0] = s; bc[1] = m;
        visited += total;
    }
    cg_sync();
    sOut = bc[0]; mOut = bc[1];
    cg_sync();
}

// HybridMatrix::add (HybridMatrix.cpp:25-31) / set (:33-39) on entry (row, col): the row copy, the column copy with
// its epsilon rule (HybridVector.cpp:55-86) and flag word, and the count of flagged entries per column (canUseGibbs =
// "the column has a flagged entry", VectorMath.cpp:125-128).  One thread per entry; rows are proposal-exclusive.
// The entry's current column-copy value and flag bit are read when the proposal's record arrives (SpCell), not when the decision is
// made: the row belongs to this proposal alone for the whole batch, so neither can change in between, and the update is then
// stores and non-returning atomics only -- no dependent memory trip after the decision.
struct SpCell { float colv; bool flagged; };
CG_DEVICE SpCell sp_cell_load(const SamplerDev &S, uint32_t row, uint32_t col)
{
    SpCell c;
    c.colv = S.mat[(size_t)col * S.Mpad + row];
    c.flagged = ((S.mflags[(size_t)col * S.Mw + (row >> 6)] >> (row & 63u)) & 1ull) != 0ull;
    return c;
}
CG_DEVICE void sp_store_col(const SamplerDev &S, uint32_t row, uint32_t col, float newCol, bool zero, bool wasFlagged)
{
    unsigned long long *f = S.mflags + (size_t)col * S.Mw + (row >> 6);
    const unsigned long long bit = 1ull << (row & 63u);
    if (zero) {
        if (wasFlagged) { (void)cg_atomic_and_u64(f, ~bit); (void)cg_atomic_sub_u32(&S.colPos[col], 1u); }
        S.mat[(size_t)col * S.Mpad + row] = 0.f;
    } else {
        if (!wasFlagged) { (void)cg_atomic_or_u64(f, bit); (void)cg_atomic_add_u32(&S.colPos[col], 1u); }
        S.mat[(size_t)col * S.Mpad + row] = newCol;
    }
}
CG_DEVICE void sp_change_matrix(const SamplerDev &S, uint32_t row, uint32_t col, float oldRow, float delta, const SpCell &cell)     // SparseNormalModel.cpp:110-114
{
    S.rows[(size_t)row * S.Kpad + col] = oldRow + delta;
    const bool zero = cell.colv + delta < GAPS_EPSILON;
    sp_store_col(S, row, col, cell.colv + delta, zero, cell.flagged);
}
CG_DEVICE void sp_safely_change_matrix(const SamplerDev &S, uint32_t row, uint32_t col, float oldRow, float delta, const SpCell &cell)   // :116-122
{
    const float newVal = gm_max(oldRow + delta, 0.f);
    S.rows[(size_t)row * S.Kpad + col] = newVal;
    sp_store_col(S, row, col, newVal, newVal < GAPS_EPSILON, cell.flagged);
}

// One workgroup of W = cogaps_sparse_width(N) threads per queued proposal (AsynchronousGibbsSampler.h:127-219 over the
// sparse model).
// the workgroup's LDS (one struct: the chained launch places it where its generator workgroup has the generator's, chain_kernel.h)
template <bool SEQ, bool WIDE>
struct SpShared {
    float lds[16 * 4];
    float arowA[SP_KMAX], arowB[SP_KMAX];
    float z2A[SP_KMAX], z2B[SP_KMAX];        // the Z2 columns of c1 / c2 (table terms)
    float decf; uint32_t deci;
    uint32_t nzShared;           // common non-zeros visited by this workgroup (roofline bookkeeping)
    uint32_t seqCnt[SEQ ? SP_SEQ_WORDS + 1 : 1]; float seqBc[2];     // verification mode (sp_alpha_seq)
    SpBal<(WIDE ? SP_BAL_CAP_WIDE : SP_BAL_CAP)> bal;                   // lane-balanced term list (sp_partial_balanced); unused in verification mode
};
// CHAIN: inside the chained launch (chain_kernel.h, round 5) -- the decision goes to the launch's generator workgroup as two tagged granules
// ({code, bytes of algorithmic traffic / 32}, {value}: gaps_state.h, CHAIN_*), which carries it out on the atomic domain and the
// HybridMatrix; nothing is written here.
template <bool SEQ, bool WIDE, bool CHAIN>
CG_DEVICE void eval_sparse_body_sh(const SamplerDev &S, const uint32_t vbid, const uint32_t vgdim, const EvalHot hot, const EvalFirst &first, SpShared<SEQ, WIDE> &sm)
{
    // (eval_kernel.h: the first record's trip starts from preloaded kernel arguments, the sampler's record comes in under it)
    PropRec pNext = first.p;
    const uint32_t qlen = first.qlen;
    const float T = first.T;
    float (&lds)[16 * 4] = sm.lds;
    float (&arowA)[SP_KMAX] = sm.arowA; float (&arowB)[SP_KMAX] = sm.arowB; float (&z2A)[SP_KMAX] = sm.z2A; float (&z2B)[SP_KMAX] = sm.z2B;
    float &decf = sm.decf; uint32_t &deci = sm.deci; uint32_t &nzShared = sm.nzShared;
    uint32_t (&seqCnt)[SEQ ? SP_SEQ_WORDS + 1 : 1] = sm.seqCnt; float (&seqBc)[2] = sm.seqBc;
    SpBal<(WIDE ? SP_BAL_CAP_WIDE : SP_BAL_CAP)> &bal = sm.bal;
    const uint32_t mm = SEQ ? S.mathMode : GM_MATH_PORTABLE;
    const uint32_t t = sp_tid(), BS = S.spW, K = S.K;      // (threads beyond the model's width never get here)
    const float lambda = S.lambda, beta = S.beta;
    const bool multiWave = BS > 64u;
    const bool scalarLane = !multiWave || t < 64u;
#if defined(GEN_TIMELINE)
    unsigned long long ets[11]; uint32_t ets_n = 0;
#endif
    EVAL_TS(0);
    for (uint32_t q = vbid; ; q += vgdim) {
        const PropRec p = pNext;
        if (q >= qlen) break;
        EVAL_TS(1);
#if defined(COGAPS_EMUL)
        if (CHAIN && t == 0u && sp_grp() != 0u) cg_atomic_add_u64(&S.gs->prof[4], 1ull);      // test-only build: proposals evaluated by a workgroup's second group
#endif
        uint64_t rng = p.rng;
        const EvalAtoms ea = eval_atoms_load(S, p, !CHAIN && t == 0u);      // (chained launch: the generator workgroup fetches what it rewrites)
        const bool two = (p.type == 'M' || p.type == 'E');
        const float m1 = p.m1, m2 = p.m2, old1 = p.old1, old2 = p.old2;     // old1/old2: the ROW copy (mMatrix(r,c))
        const bool gibbs1 = (p.gibbs & 1u) != 0u, gibbs2 = (p.gibbs & 2u) != 0u;
        const bool need = (p.type == 'B') ? gibbs1 : ((p.type == 'D' || p.type == 'M') ? true : (gibbs1 || gibbs2));
        const bool diff = two && p.r1 != p.r2;
        float s = 0.f, smu = 0.f;
        uint32_t nz = 0;
        if (t == 0) nzShared = 0u;
        // Everything the record addresses goes out in ONE memory trip: the cells the decision will rewrite (writer), the table terms'
        // Z1 entries, the first round of flag words, and below the matrix rows / Z2 columns for LDS.
        SpCell cell1, cell2; cell1.colv = 0.f; cell1.flagged = false; cell2 = cell1;
        if (!CHAIN && t == 0u) { cell1 = sp_cell_load(S, p.r1, p.c1); if (two) cell2 = sp_cell_load(S, p.r2, p.c2); }
        const float z1a = need ? S.Z1[p.c1] : 0.f, z1b = (need && two) ? S.Z1[p.c2] : 0.f;
        SpPre preA, preB; preA.dfl = preA.fv = 0ull; preA.dbase = 0u; preB = preA;
        if (!SEQ && need) { preA = sp_preload(S, p.r1, p.c1, p.c2, two && !diff); if (diff) preB = sp_preload(S, p.r2, p.c2, 0u, false); }
        if (need) {
            // this sampler's matrix row(s), read by every lane for the K-length dots
            // (the row copy is zero-padded to Kpad, a multiple of 4: sp_row_dot reads whole chunks)
            for (uint32_t k = t; k < S.Kpad; k += BS) {
                arowA[k] = S.rows[(size_t)p.r1 * S.Kpad + k];
                if (diff) arowB[k] = S.rows[(size_t)p.r2 * S.Kpad + k];
                if (k < K) { z2A[k] = S.Z2[(size_t)p.c1 * K + k]; if (two) z2B[k] = S.Z2[(size_t)p.c2 * K + k]; }
            }
            cg_sync();
            EVAL_TS(2);
            if (SEQ) {
                // table terms first, then the common non-zeros in index order (SparseNormalModel.cpp:160-161, 205-207, 256-258)
                uint32_t vis = 0;
                if (diff) {
                    float sa, ma, sb, mb;
                    sp_alpha_seq<SP_MODE_ONE>(S, p.r1, p.c1, 0u, 0.f, arowA, z1a, -1.f * sp_dot(arowA, z2A, K), seqCnt, seqBc, sa, ma, vis);
                    sp_alpha_seq<SP_MODE_ONE>(S, p.r2, p.c2, 0u, 0.f, arowB, z1b, -1.f * sp_dot(arowB, z2B, K), seqCnt, seqBc, sb, mb, vis);
                    sa = sa * beta; ma = ma * beta; sb = sb * beta; mb = mb * beta;
                    s = sa + sb; smu = ma - mb;
                } else if (two) {
                    const float s0 = z1a - 2.f * z2B[p.c1] + z1b;
                    float d0 = 0.f;
                    for (uint32_t k = 0; k < K; ++k) d0 += arowA[k] * (z2A[k] - z2B[k]);
                    sp_alpha_seq<SP_MODE_SAME>(S, p.r1, p.c1, p.c2, 0.f, arowA, s0, -1.f * d0, seqCnt, seqBc, s, smu, vis);
                    s = s * beta; smu = smu * beta;
                } else if (p.type == 'D') {
                    const float ch = -1.f * m1;
                    float m0 = -1.f * sp_dot(arowA, z2A, K);
                    m0 -= ch * z2A[p.c1];
                    sp_alpha_seq<SP_MODE_CH>(S, p.r1, p.c1, 0u, ch, arowA, z1a, m0, seqCnt, seqBc, s, smu, vis);
                    s = s * beta; smu = smu * beta;
                } else {
                    sp_alpha_seq<SP_MODE_ONE>(S, p.r1, p.c1, 0u, 0.f, arowA, z1a, -1.f * sp_dot(arowA, z2A, K), seqCnt, seqBc, s, smu, vis);
                    s = s * beta; smu = smu * beta;
                }
                if (t == 0) nzShared = vis;
            } else {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (diff) {
                if (!WIDE && S.Wn <= BS) sp_partial_pair(S, p.r1, p.c1, arowA, preA, p.r2, p.c2, arowB, preB, bal, x, nz);      // (one-round vectors: both evaluations in one pass)
                else if (WIDE) sp_partial_merged_pair(S, p.r1, p.c1, arowA, preA, p.r2, p.c2, arowB, preB, bal, x, nz);
                else { sp_partial_balanced<SP_MODE_ONE, WIDE>(S, p.r1, p.c1, 0u, 0.f, arowA, bal, preA, x[0], x[1], nz); sp_partial_balanced<SP_MODE_ONE, WIDE>(S, p.r2, p.c2, 0u, 0.f, arowB, bal, preB, x[2], x[3], nz); }
            }
            else if (p.type == 'D') sp_partial_balanced<SP_MODE_CH, WIDE>(S, p.r1, p.c1, 0u, -1.f * m1, arowA, bal, preA, x[0], x[1], nz);
            else if (two) sp_partial_balanced<SP_MODE_SAME, WIDE>(S, p.r1, p.c1, p.c2, 0.f, arowA, bal, preA, x[0], x[1], nz);
            else sp_partial_balanced<SP_MODE_ONE, WIDE>(S, p.r1, p.c1, 0u, 0.f, arowA, bal, preA, x[0], x[1], nz);
            EVAL_TS(3);
            { const uint32_t waveNz = cg_wave_sum_u32(nz); if ((t & 63u) == 0u && waveNz) cg_atomic_add_u32(&nzShared, waveNz); }
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = cg_wave_allsum_f32(x[c]);
            float tot[4] = {x[0], x[1], x[2], x[3]};
            if (multiWave) {
                if ((t & 63u) == 0) { for (int c = 0; c < 4; ++c) lds[(t >> 6) * 4 + c] = x[c]; }
                cg_sync();
                eval_vfinish<4, 1>(lds, tot, BS >> 6, t);
            }
            EVAL_TS(4);
            if (scalarLane) {
                // table terms (SparseNormalModel.cpp:160-161, 205-207, 256-258), then beta
                if (diff) {
                    const float sa = (z1a + tot[0]) * beta, ma = (-1.f * sp_dot(arowA, z2A, K) + tot[1]) * beta;
                    const float sb = (z1b + tot[2]) * beta, mb = (-1.f * sp_dot(arowB, z2B, K) + tot[3]) * beta;
                    s = sa + sb; smu = ma - mb;                                    // AlphaParameters.cpp:11-14
                } else if (two) {
                    float s0 = z1a - 2.f * z2B[p.c1] + z1b;
                    float d0 = 0.f;
                    for (uint32_t k = 0; k < K; ++k) d0 += arowA[k] * (z2A[k] - z2B[k]);   // dot_diff, VectorMath.h:137-155
                    float m0 = -1.f * d0;
                    s = (s0 + tot[0]) * beta; smu = (m0 + tot[1]) * beta;
                } else {
                    float m0 = -1.f * sp_dot(arowA, z2A, K);
                    if (p.type == 'D') m0 -= (-1.f * m1) * z2A[p.c1];
                    s = (z1a + tot[0]) * beta; smu = (m0 + tot[1]) * beta;
                }
            }
            }
        }
        EVAL_PIN(s); EVAL_TS(5);
        s = s * T; smu = smu * T;
        const bool writer = !CHAIN && t == 0u;
        // roofline bookkeeping in bytes (sparse): per alpha call the flag words of the data vector and of the
        // column(s), this matrix row and a Z2 column; per common non-zero the data value, the column entry and a row of
        // the other matrix.  (Every path above has passed a barrier since the lanes added their counts, or is one wave.)
        uint32_t bytes = 0;
        if (t == 0u && need) bytes = (diff ? 2u : 1u) * (16u * S.Wn + 8u * K) + ((two && !diff) ? 8u * S.Wn : 0u) + nzShared * (8u + 4u * K);
        // chained launch: thread 0 hands the decision to the launch's generator workgroup the moment it is made (eval_kernel.h, EVAL_PUBLISH)
#define SP_PUBLISH(CODE, VAL) do { if (CHAIN && t == 0u) { unsigned long long *gr_ = hot.grans + (size_t)q * CHAIN_GRAN_STRIDE; \
        cg_store_agent_u64(&gr_[0], ((unsigned long long)first.tag << 32) | (unsigned long long)((CODE) | (((bytes + 16u) >> 5) << 8))); \
        cg_store_agent_u64(&gr_[1], ((unsigned long long)first.tag << 32) | (unsigned long long)gm_f2u(VAL)); } } while (0)
#define SP_BCAST(F0, I0) do { if (multiWave && !CHAIN) { if (t == 0) { decf = (F0); deci = (I0); } cg_sync(); (F0) = decf; (I0) = deci; } } while (0)
        if (p.type == 'B') {
            float bv = 0.f; uint32_t bhas = 0;
            if (scalarLane) {
                if (gibbs1) { OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); bv = g.v; bhas = g.has ? 1u : 0u; }
                else { bv = pcg_exponential(rng, lambda, mm); bhas = 1u; }
                if (bhas != 0u && bv >= GAPS_EPSILON) SP_PUBLISH(CHAIN_APPLY, bv); else SP_PUBLISH(CHAIN_ERASE, 0.f);
            }
            SP_BCAST(bv, bhas);
            if (bhas != 0u && bv >= GAPS_EPSILON) {
                if (writer) { atom_set_mass(S, p.h1, ea.a1.left, bv); sp_change_matrix(S, p.r1, p.c1, old1, bv, cell1); }
            } else if (writer) eval_cache_erase(S, p.h1, p.r1, p.c1);
        } else if (p.type == 'D') {
            float rebirth = m1; uint32_t acc = 0;
            if (scalarLane) {
                if (gibbs1) { OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); if (g.has) rebirth = g.v; }
                const float deltaLL = rebirth * (smu - s * rebirth / 2.f);
                acc = (gm_logf_m(pcg_uniform(rng), mm) < deltaLL) ? 1u : 0u;
                if (acc != 0u) { if (rebirth != m1) SP_PUBLISH(CHAIN_APPLY, rebirth); else SP_PUBLISH(CHAIN_NONE, 0.f); } else SP_PUBLISH(CHAIN_ERASE, 0.f);
            }
            SP_BCAST(rebirth, acc);
            if (writer) {
                if (acc != 0u) { if (rebirth != m1) { sp_safely_change_matrix(S, p.r1, p.c1, old1, rebirth - m1, cell1); atom_set_mass(S, p.h1, ea.a1.left, rebirth); } }
                else { sp_safely_change_matrix(S, p.r1, p.c1, old1, -1.f * m1, cell1); eval_cache_erase(S, p.h1, p.r1, p.c1); }
            }
        } else if (p.type == 'M') {
            uint32_t acc = 0; float unused = 0.f;
            if (scalarLane) { const float deltaLL = -1.f * m1 * (smu + s * m1 / 2.f); acc = (gm_logf_m(pcg_uniform(rng), mm) < deltaLL) ? 1u : 0u;
                              if (acc) SP_PUBLISH(CHAIN_APPLY, 0.f); else SP_PUBLISH(CHAIN_NONE, 0.f); }
            SP_BCAST(unused, acc);
            if (acc && writer) {
                eval_domain_move(S, p, ea.a1);
                sp_safely_change_matrix(S, p.r1, p.c1, old1, -m1, cell1);
                sp_change_matrix(S, p.r2, p.c2, old2, m1, cell2);
            }
        } else if (need) {
            float gv = 0.f; uint32_t gh = 0;
            if (scalarLane) { OptF g0 = gm_gibbs_mass(s, smu, -m1, m2, rng, S.luts, false, 0.f); gv = g0.v; gh = g0.has ? 1u : 0u;
                              if (gh != 0u && m1 + gv > GAPS_EPSILON && m2 - gv > GAPS_EPSILON) SP_PUBLISH(CHAIN_APPLY, gv); else SP_PUBLISH(CHAIN_NONE, 0.f); }
            SP_BCAST(gv, gh);
            const float n1 = m1 + gv, n2 = m2 - gv;
            if (gh != 0u && n1 > GAPS_EPSILON && n2 > GAPS_EPSILON && writer) {
                sp_safely_change_matrix(S, p.r1, p.c1, old1, n1 - m1, cell1);
                sp_safely_change_matrix(S, p.r2, p.c2, old2, n2 - m2, cell2);
                atom_set_mass(S, p.h1, ea.a1.left, n1); atom_set_mass(S, p.h2, ea.left2, n2);
            }
        }
        else SP_PUBLISH(CHAIN_NONE, 0.f);      // an exchange that cannot use Gibbs: nothing happens, the generator still waits for its word
        EVAL_TS(6);
#if defined(GEN_TIMELINE)
        if (cg_bid() < 16u && (t & 63u) == 0u && ((t >> 6) == 0u || (t >> 6) == ((BS - 1u) >> 6)) && qlen >= 20u) {
            unsigned long long *o_ = &g_eval_timeline[((S.N > 4096u ? 16u : 0u) * 2u + cg_bid() * 2u + ((t >> 6) ? 1u : 0u)) * 12u];
            o_[0] = (unsigned long long)(p.type | ((need ? 1u : 0u) << 8) | ((p.r1 == p.r2 ? 1u : 0u) << 16));
            for (uint32_t i_ = 0; i_ < 11u; ++i_) o_[1 + i_] = i_ < ets_n ? ets[i_] : 0ull;
        }
#endif
        if (writer) S.queueUnits[q] = bytes;
        if (q + vgdim >= qlen) break;
        { const uint32_t qn_ = q + vgdim; pNext = hot.queue[qn_ < hot.queueCap ? qn_ : 0u]; }
        cg_sync();
    }
}
template <bool SEQ, bool WIDE>
CG_DEVICE void eval_sparse_body(const SamplerDev &S, const uint32_t vbid, const uint32_t vgdim, const EvalHot hot, const EvalFirst &first)
{
    CG_SHARED SpShared<SEQ, WIDE> sm;
    eval_sparse_body_sh<SEQ, WIDE, false>(S, vbid, vgdim, hot, first, sm);
}
CG_KERNEL void CG_LAUNCH_BOUNDS(256) eval_sparse_kernel(const PropRec *hotQueue, const GenScalars *hotGs, uint32_t hotCap, const SamplerDev CG_CONSTANT *sp)
{
    EvalHot hot; hot.queue = hotQueue; hot.gs = hotGs; hot.queueCap = hotCap; hot.slot = nullptr; hot.grans = nullptr;
    const EvalFirst first = eval_first<EVAL_FUSED>(hot, 1u, cg_bid());
    const SamplerDev &S = eval_record<EVAL_FUSED>(sp);
    eval_sparse_body<false, false>(S, cg_bid(), cg_gdim(), hot, first);
}
// data vectors of more than one round of flag words (more than 16384 elements): the rounds' common non-zeros listed together (sp_partial_merged)
CG_KERNEL void CG_LAUNCH_BOUNDS(256) eval_sparse_kernel_wide(const PropRec *hotQueue, const GenScalars *hotGs, uint32_t hotCap, const SamplerDev CG_CONSTANT *sp)
{
    EvalHot hot; hot.queue = hotQueue; hot.gs = hotGs; hot.queueCap = hotCap; hot.slot = nullptr; hot.grans = nullptr;
    const EvalFirst first = eval_first<EVAL_FUSED>(hot, 1u, cg_bid());
    const SamplerDev &S = eval_record<EVAL_FUSED>(sp);
    eval_sparse_body<false, true>(S, cg_bid(), cg_gdim(), hot, first);
}
CG_KERNEL void CG_LAUNCH_BOUNDS(256) eval_sparse_kernel_multi(const SamplerDev CG_CONSTANT *arr, uint32_t wgPerChain)
{
    const uint32_t chain = cg_bid() / wgPerChain;
    const SamplerDev CG_CONSTANT *sp = arr + chain;
    cg_const_warm<sizeof(SamplerDev)>(sp);
    const SamplerDev &S = *(const SamplerDev *)sp;
    EvalHot hot; hot.queue = S.queue; hot.gs = S.gs; hot.queueCap = S.queueCap; hot.slot = nullptr; hot.grans = nullptr;
    const uint32_t vbid = cg_bid() - chain * wgPerChain;
    eval_sparse_body<false, false>(S, vbid, wgPerChain, hot, eval_first<EVAL_FUSED>(hot, 1u, vbid));
}
CG_KERNEL void CG_LAUNCH_BOUNDS(256) eval_sparse_seq_kernel(SamplerDev S)
{
    cg_kernarg_warm<sizeof(SamplerDev)>();
    EvalHot hot; hot.queue = S.queue; hot.gs = S.gs; hot.queueCap = S.queueCap; hot.slot = nullptr; hot.grans = nullptr;
    eval_sparse_body<true, false>(S, cg_bid(), cg_gdim(), hot, eval_first<EVAL_FUSED>(hot, 1u, cg_bid()));
}

// SparseNormalModel::generateLookupTables (SparseNormalModel.cpp:294-311): Z1[i] = sum_k other(k,i)^2 through the
// other matrix's ROW copy, Z2(i,j) = dot of its column copies.  One workgroup per (i, j >= i) pair plus one per i;
// lane order of the dense reductions over the N elements (float4 chunks, V slots per thread).
template <int V>
CG_KERNEL void CG_LAUNCH_BOUNDS(1024) sparse_tables_kernel(SamplerDev S)
{
    CG_SHARED float lds[16 * V];
    const uint32_t K = S.K, t = cg_tid(), BS = cg_bdim(), W = (uint32_t)V * BS, nq = S.Npad >> 2, b = cg_bid();
    float tot[1] = {0.f};
    if (b < K) {
        const uint32_t i = b;
        for (int j = 0; j < V; ++j) {
            float acc = 0.f;
            for (uint32_t c = (uint32_t)j * BS + t; c < nq; c += W) {
                for (uint32_t e = 0; e < 4u; ++e) { const uint32_t k = 4u * c + e; if (k < S.N) { const float v = S.orows[(size_t)k * S.oKpad + i]; acc = acc + v * v; } }
            }
            eval_vpark<V>(acc, j, lds, tot);
        }
        if (BS > 64u) { cg_sync(); eval_vfinish<1, V>(lds, tot); }
        if (t == 0) S.Z1[i] = tot[0];
    } else {
        // pair index -> (i, j >= i)
        uint32_t r = b - K, i = 0;
        while (r >= K - i) { r -= K - i; ++i; }
        const uint32_t j2 = i + r;
        const float *ci = S.other + (size_t)i * S.Npad, *cj = S.other + (size_t)j2 * S.Npad;
        for (int j = 0; j < V; ++j) {
            float acc = 0.f;
            for (uint32_t c = (uint32_t)j * BS + t; c < nq; c += W) {
                const cg_f4 a = ld4(ci, c), bb = ld4(cj, c);
                acc = acc + a.x * bb.x; acc = acc + a.y * bb.y; acc = acc + a.z * bb.z; acc = acc + a.w * bb.w;
            }
            eval_vpark<V>(acc, j, lds, tot);
        }
        if (BS > 64u) { cg_sync(); eval_vfinish<1, V>(lds, tot); }
        if (t == 0) { S.Z2[(size_t)j2 * K + i] = tot[0]; S.Z2[(size_t)i * K + j2] = tot[0]; }
    }
}

// SparseNormalModel::chiSq (SparseNormalModel.cpp:40-62), per-vector partial in the dense lane order over the elements
template <int V>
CG_KERNEL void CG_LAUNCH_BOUNDS(1024) chisq_sparse_kernel(SamplerDev S, float *partial)
{
    CG_SHARED float lds[16 * V];
    CG_SHARED float arow[SP_KMAX];
    const uint32_t row = cg_bid(), t = cg_tid(), BS = cg_bdim(), W = (uint32_t)V * BS, nq = S.Npad >> 2, K = S.K;
    for (uint32_t k = t; k < K; k += BS) arow[k] = S.rows[(size_t)row * S.Kpad + k];
    cg_sync();
    const unsigned long long *fD = S.dflags + (size_t)row * S.Wn;
    const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
    const float *data = S.dvals + S.dptr[row];
    float tot[1] = {0.f};
    for (int j = 0; j < V; ++j) {
        float acc = 0.f;
        for (uint32_t c = (uint32_t)j * BS + t; c < nq; c += W) {
            for (uint32_t e = 0; e < 4u; ++e) {
                const uint32_t i = 4u * c + e;
                if (i < S.N) {
                    const float dot = sp_dot_row(arow, S.orows + (size_t)i * S.oKpad, K);
                    acc = acc + dot * dot;
                    const unsigned long long fl = fD[i >> 6];
                    if ((fl >> (i & 63u)) & 1ull) {
                        const float d = data[pre[i >> 6] + (uint32_t)cg_popc64(fl & ((1ull << (i & 63u)) - 1ull))];
                        const float dsq = d * d;
                        acc = acc + (1 + dot * (dot - 2 * d - dsq * dot) / dsq);
                    }
                }
            }
        }
        eval_vpark<V>(acc, j, lds, tot);
    }
    if (BS > 64u) { cg_sync(); eval_vfinish<1, V>(lds, tot); }
    if (t == 0) partial[row] = tot[0];
}

// The same for SP_CHI_ROWS vectors per workgroup (nPatterns <= 64: sixteen): a row of the other matrix, once in registers, serves all of them.  The
// one-vector kernel re-reads the whole other matrix (N x K floats) for every vector -- 125 GB at BASELINE configs[4]'s shard shape
// (12500 vectors of 50000 elements, K = 50), 13.4 ms per call; sixteen vectors per workgroup read a sixteenth of that.  Every vector's sum is
// the one the kernel above computes: the same virtual lane takes the same elements in the same order, the same dot products, the same
// butterfly -- the vectors of a workgroup only share the loads.
#define SP_CHI_ROWS 16
template <int V>
CG_KERNEL void CG_LAUNCH_BOUNDS(1024) chisq_sparse_tiled_kernel(SamplerDev S, float *partial)
{
    constexpr int R = SP_CHI_ROWS;
    CG_SHARED float lds[R][16 * V];
    CG_SHARED float arow[R][64];
    const uint32_t row0 = cg_bid() * (uint32_t)R, t = cg_tid(), BS = cg_bdim(), W = (uint32_t)V * BS, nq = S.Npad >> 2, K = S.K;
    const uint32_t nr = (S.M - row0) < (uint32_t)R ? (S.M - row0) : (uint32_t)R;
    for (uint32_t k = t; k < (uint32_t)R * 64u; k += BS) { const uint32_t r = k >> 6, c = k & 63u; arow[r][c] = (r < nr && c < K) ? S.rows[(size_t)(row0 + r) * S.Kpad + c] : 0.f; }
    cg_sync();
    float tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tot[r] = 0.f;
    for (int j = 0; j < V; ++j) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        for (uint32_t c = (uint32_t)j * BS + t; c < nq; c += W) {
            for (uint32_t e = 0; e < 4u; ++e) {
                const uint32_t i = 4u * c + e;
                if (i < S.N) {
                    SpRow O; sp_row_load(O, S.orows + (size_t)i * S.oKpad, K);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if ((uint32_t)r < nr) {
                            const float dot = sp_row_dot(arow[r], O, K);
                            acc[r] = acc[r] + dot * dot;
                            const unsigned long long fl = S.dflags[(size_t)(row0 + r) * S.Wn + (i >> 6)];
                            if ((fl >> (i & 63u)) & 1ull) {
                                const float d = S.dvals[S.dptr[row0 + r] + S.dprefix[(size_t)(row0 + r) * S.Wn + (i >> 6)] + (uint32_t)cg_popc64(fl & ((1ull << (i & 63u)) - 1ull))];
                                const float dsq = d * d;
                                acc[r] = acc[r] + (1 + dot * (dot - 2 * d - dsq * dot) / dsq);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { float t1[1] = {tot[r]}; eval_vpark<V>(acc[r], j, lds[r], t1); tot[r] = t1[0]; }
    }
    if (BS > 64u) {
        cg_sync();
#pragma unroll
        for (int r = 0; r < R; ++r) { float t1[1] = {0.f}; eval_vfinish<1, V>(lds[r], t1); tot[r] = t1[0]; }
    }
    if (t == 0) { for (uint32_t r = 0; r < nr; ++r) partial[row0 + r] = tot[r]; }
}

// ---- verification mode: the same three in the reference's order (aux_kernels.h: seq_sum) ---------------------------------
// generateLookupTables: Z1 front to back; Z2 through gaps::dot, i.e. back to front for vectors of at most 25 elements
CG_KERNEL void CG_LAUNCH_BOUNDS(256) sparse_tables_seq_kernel(SamplerDev S)
{
    CG_SHARED float lds[SEQ_CHUNK];
    const uint32_t K = S.K, N = S.N, b = cg_bid();
    if (b < K) {
        const uint32_t i = b;
        const float z = seq_sum(0.f, N, lds, [&](uint64_t k) { const float v = S.orows[(size_t)k * S.oKpad + i]; return v * v; });
        if (cg_tid() == 0) S.Z1[i] = z;
    } else {
        uint32_t r = b - K, i = 0;
        while (r >= K - i) { r -= K - i; ++i; }
        const uint32_t j2 = i + r;
        const float *ci = S.other + (size_t)i * S.Npad, *cj = S.other + (size_t)j2 * S.Npad;
        const bool rev = N <= 25u;
        const float d = seq_sum(0.f, N, lds, [&](uint64_t e) { const uint32_t k = rev ? N - 1u - (uint32_t)e : (uint32_t)e; return ci[k] * cj[k]; });
        if (cg_tid() == 0) { S.Z2[(size_t)j2 * K + i] = d; S.Z2[(size_t)i * K + j2] = d; }
    }
}
// SparseNormalModel::chiSq (SparseNormalModel.cpp:40-62): per vector all dense terms, then one correction per non-zero; one
// accumulator across the vectors.  One workgroup; out[0] is the sum before beta.
CG_KERNEL void CG_LAUNCH_BOUNDS(256) chisq_sparse_seq_kernel(SamplerDev S, float *out)
{
    CG_SHARED float lds[SEQ_CHUNK];
    CG_SHARED float arow[SP_KMAX];
    const uint32_t t = cg_tid(), BS = cg_bdim(), K = S.K;
    float acc = 0.f;
    for (uint32_t row = 0; row < S.M; ++row) {
        for (uint32_t k = t; k < K; k += BS) arow[k] = S.rows[(size_t)row * S.Kpad + k];
        cg_sync();
        acc = seq_sum(acc, S.N, lds, [&](uint64_t i) { const float dot = sp_dot_row(arow, S.orows + (size_t)i * S.oKpad, K); return dot * dot; });
        const unsigned long long *fD = S.dflags + (size_t)row * S.Wn;
        const uint32_t *pre = S.dprefix + (size_t)row * S.Wn;
        const float *data = S.dvals + S.dptr[row];
        const uint32_t nnz = S.dptr[row + 1] - S.dptr[row];
        for (uint32_t w = t; w < S.Wn; w += BS) {
            unsigned long long fl = fD[w];
            uint32_t o = pre[w];
            while (fl != 0ull) {
                const uint32_t bit = (uint32_t)cg_ctz64(fl); fl &= fl - 1ull;
                const float d = data[o];
                const float dot = sp_dot_row(arow, S.orows + (size_t)(64u * w + bit) * S.oKpad, K);
                const float dsq = d * d;
                S.seqScratch[o] = 1 + dot * (dot - 2 * d - dsq * dot) / dsq;
                ++o;
            }
        }
        cg_sync();
        acc = seq_sum(acc, nnz, lds, [&](uint64_t e) { return S.seqScratch[e]; });
    }
    if (t == 0) out[0] = acc;
}

// ---- the chained launch of the sparse model (rounds 5, 6; chain_kernel.h for the scheme) ---------------------------------------------------------
// Workgroups 0 .. n-2 evaluate batch b as eval_sparse_kernel[_wide] does and hand each decision to the LAST workgroup, the generator of
// batch b + 1, which carries it out on the atomic domain and the HybridMatrix (gen_populate.h: chain_apply, chain_store_hybrid).  The
// sparse evaluation is long (15-27 us per batch at BASELINE configs[4]'s shard shape against ~8 us of generator work that does not
// depend on the decisions), so the generator's prologue, classification and draws disappear behind it.  One static LDS block serves
// both roles -- a launch's workgroups all carry the kernel's static LDS, and the two roles' blocks side by side (142 + 49 / 90 KB) exceed
// a compute unit's 160 KB.  The launch has the generator's workgroup size (512 threads: window + helper wave + applier waves); an
// evaluation keeps the model's width (S.spW threads = virtual lanes: the parity contract) -- a workgroup's first and second four waves are two
// such evaluations side by side (one-round vectors; round 6, below), waves beyond the width leave at once.
template <int WIN, bool WIDE>
CG_KERNEL void CG_LAUNCH_BOUNDS(CHAIN_MAX_THREADS) chain_sparse_kernel(const uint64_t *lcgMul, const uint64_t *lcgInc, GenScalars *gs, PropRec *queue, unsigned long long *grans, ChainSlot *slots,
                                                                       uint32_t queueCap, uint32_t parity, const SamplerDev CG_CONSTANT *sp)
{
    constexpr size_t GROUP_LDS = (sizeof(SpShared<false, WIDE>) + 15u) & ~(size_t)15u, EVAL_LDS = SP_CHAIN_GROUPS(WIDE) * GROUP_LDS;
    constexpr size_t POOL = sizeof(GenShared<WIN>) > EVAL_LDS ? sizeof(GenShared<WIN>) : EVAL_LDS;
    CG_SHARED alignas(16) unsigned char pool[POOL];
    if (cg_bid() + 1u == cg_gdim()) {
        GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = nullptr; hot.queueUnits = nullptr; hot.eraseCap = 0; hot.queueCap = queueCap;
        hot.queueRd = queue + (size_t)parity * queueCap; hot.queueWr = queue + (size_t)(1u - parity) * queueCap; hot.grans = grans; hot.slotWr = &slots[1u - parity];
        gen_body_sh<WIN, true, true, 1>(sp, hot, *reinterpret_cast<GenShared<WIN> *>(pool));
        return;
    }
    EvalHot hot; hot.queue = queue + (size_t)parity * queueCap; hot.gs = gs; hot.queueCap = queueCap; hot.slot = &slots[parity]; hot.grans = grans;
    const unsigned long long clk0 = (cg_bid() == 0u && cg_tid() == 0u) ? cg_realtime() : 0ull;
    // (round 6) two proposals per evaluation workgroup: the launch's workgroups have 512 threads, the model's width is 256 at most -- the second
    // four waves, which used to leave at once, take the proposal one grid further on (queue slot bid + n: a queue longer than the grid --
    // 39 % of the A sampler's launches at BASELINE configs[4]'s shard shape -- no longer needs a second pass), with an LDS block of their own.
    // The two groups share the workgroup's barrier: every wave has passed the same number of barriers, so a group's waves always meet at
    // the group's own k-th barrier, whatever the other group's k-th one is; a group that is done ends and no longer counts.
    const uint32_t grp = sp_grp(), nEval = cg_gdim() - 1u, vb = cg_bid() + grp * nEval;
    if (grp >= SP_CHAIN_GROUPS(WIDE)) return;
    const EvalFirst first = eval_first<EVAL_CHAIN>(hot, 1u, vb);
    const SamplerDev &S = eval_record<EVAL_CHAIN>(sp);
    if (cg_bid() == 0u && cg_tid() == 0u && S.launchClock) S.launchClock[2u * (first.tag % GAPS_CLOCK_RING)] = clk0;      // (launch clock: gaps_state.h)
    if (sp_tid() >= S.spW) return;
    eval_sparse_body_sh<false, WIDE, true>(S, vb, nEval * SP_CHAIN_GROUPS(WIDE), hot, first, *reinterpret_cast<SpShared<false, WIDE> *>(pool + (size_t)grp * GROUP_LDS));
}
