// eval_kernel.h -- evaluation of one queued batch: the birth / death / move / exchange steps of
// AsynchronousGibbsSampler.h:127-219 over the DenseNormalModel reductions marked PERFORMANCE
// CRITICAL in DenseNormalModel.cpp:161-258.
//
// One workgroup of W = S.redW lanes per queued proposal (W = 64: one wavefront; wider for long data
// vectors).  A proposal touches one or two factor rows, the queue guarantees that no two proposals
// of a batch share a row, so workgroups never write the same AP row / matrix entry / atom.
//
// HBM traffic per proposal (N = data-vector length): alpha with or without change 16N bytes,
// 2-site same row 20N, different rows 32N, AP update 12N.  Rows are read as coalesced float4.
//
// Reduction order (the parity contract with oracle redW/redG=4): lane L accumulates float4 chunks
// j = L, L+W, L+2W ... in increasing j (x,y,z,w in order) from +0; then an ascending xor butterfly
// 1,2,4,...,W/2 (the reference's AVX hadd tree, SIMD.h:102-107, widened from 8 to W lanes).
#pragma once
#include "gaps_state.h"
#include "gen_kernel.h"   // gen_bin_of, bm_set, bm_clear

#if defined(COGAPS_EMUL)
struct cg_f4 { float x, y, z, w; };
#else
typedef float4 cg_f4;
#endif

CG_DEVICE cg_f4 ld4(const float *base, uint32_t j) { return reinterpret_cast<const cg_f4 *>(base)[j]; }
CG_DEVICE void st4(float *base, uint32_t j, cg_f4 v) { reinterpret_cast<cg_f4 *>(base)[j] = v; }

struct EvalAcc { float s, m; };

// DenseNormalModel.cpp:170-181 / :229-238, per element:  ratio = v/(S*S); s += v*ratio; s_mu += ratio*(D - AP[+ch*v])
#define EVAL_ELEM(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - (AA)); }
#define EVAL_ELEM_CH(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - ((AA) + ch * (V))); }

CG_DEVICE EvalAcc eval_partial_one(const SamplerDev &S, uint32_t row, uint32_t col, bool withCh, float ch)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    const float *D = S.D + (size_t)row * S.Npad, *S2 = S.S2 + (size_t)row * S.Npad, *AP = S.AP + (size_t)row * S.Npad;
    const float *V = S.other + (size_t)col * S.Npad;
    EvalAcc a; a.s = 0.f; a.m = 0.f;
    if (withCh) {
        for (uint32_t j = t; j < nq; j += W) {
            const cg_f4 v = ld4(V, j), d = ld4(D, j), s = ld4(S2, j), p = ld4(AP, j);
            EVAL_ELEM_CH(v.x, d.x, s.x, p.x) EVAL_ELEM_CH(v.y, d.y, s.y, p.y) EVAL_ELEM_CH(v.z, d.z, s.z, p.z) EVAL_ELEM_CH(v.w, d.w, s.w, p.w)
        }
    } else {
        for (uint32_t j = t; j < nq; j += W) {
            const cg_f4 v = ld4(V, j), d = ld4(D, j), s = ld4(S2, j), p = ld4(AP, j);
            EVAL_ELEM(v.x, d.x, s.x, p.x) EVAL_ELEM(v.y, d.y, s.y, p.y) EVAL_ELEM(v.z, d.z, s.z, p.z) EVAL_ELEM(v.w, d.w, s.w, p.w)
        }
    }
    return a;
}
// DenseNormalModel.cpp:200-212: same row, v = other[:,c1] - other[:,c2]
CG_DEVICE EvalAcc eval_partial_two_same(const SamplerDev &S, uint32_t row, uint32_t c1, uint32_t c2)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    const float *D = S.D + (size_t)row * S.Npad, *S2 = S.S2 + (size_t)row * S.Npad, *AP = S.AP + (size_t)row * S.Npad;
    const float *V1 = S.other + (size_t)c1 * S.Npad, *V2 = S.other + (size_t)c2 * S.Npad;
    EvalAcc a; a.s = 0.f; a.m = 0.f;
    for (uint32_t j = t; j < nq; j += W) {
        const cg_f4 v1 = ld4(V1, j), v2 = ld4(V2, j), d = ld4(D, j), s = ld4(S2, j), p = ld4(AP, j);
        { float v = v1.x - v2.x; EVAL_ELEM(v, d.x, s.x, p.x) }
        { float v = v1.y - v2.y; EVAL_ELEM(v, d.y, s.y, p.y) }
        { float v = v1.z - v2.z; EVAL_ELEM(v, d.z, s.z, p.z) }
        { float v = v1.w - v2.w; EVAL_ELEM(v, d.w, s.w, p.w) }
    }
    return a;
}

// ascending xor butterfly over the W lanes of the workgroup; every lane returns the same bits
CG_DEVICE EvalAcc eval_block_reduce(EvalAcc a, float *lds /* [32] */)
{
    for (int off = 1; off < 64; off <<= 1) {
        a.s = a.s + cg_shfl_xor_f32(a.s, off);
        a.m = a.m + cg_shfl_xor_f32(a.m, off);
    }
    const uint32_t nw = cg_bdim() >> 6;
    if (nw > 1) {
        const uint32_t t = cg_tid();
        if ((t & 63u) == 0) { lds[t >> 6] = a.s; lds[16 + (t >> 6)] = a.m; }
        cg_sync();
        float ws[16], wm[16];
        for (uint32_t i = 0; i < 16; ++i) { ws[i] = i < nw ? lds[i] : 0.f; wm[i] = i < nw ? lds[16 + i] : 0.f; }
        for (uint32_t stride = 1; stride < nw; stride <<= 1)
            for (uint32_t i = 0; i + stride < 16; i += 2 * stride) { ws[i] = ws[i] + ws[i + stride]; wm[i] = wm[i] + wm[i + stride]; }
        a.s = ws[0]; a.m = wm[0];
        cg_sync();
    }
    return a;
}

// DenseNormalModel.cpp:243-258: AP[:,row] += delta * other[:,col]
CG_DEVICE void eval_update_ap(const SamplerDev &S, uint32_t row, uint32_t col, float delta)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    float *AP = S.AP + (size_t)row * S.Npad;
    const float *V = S.other + (size_t)col * S.Npad;
    if (t == 0) cg_atomic_add_u64(&S.gs->evalBytes, 12ull * S.N);
    for (uint32_t j = t; j < nq; j += W) {
        const cg_f4 v = ld4(V, j); cg_f4 p = ld4(AP, j);
        p.x = p.x + delta * v.x; p.y = p.y + delta * v.y; p.z = p.z + delta * v.z; p.w = p.w + delta * v.w;
        st4(AP, j, p);
    }
}

// mMatrix(row,col) = newv, keeping the per-column count of positive entries (canUseGibbs) current
CG_DEVICE void eval_store_matrix(const SamplerDev &S, uint32_t row, uint32_t col, float oldv, float newv)
{
    S.mat[(size_t)col * S.Mpad + row] = newv;
    const bool was = oldv > 0.f, is = newv > 0.f;
    if (was != is) { if (is) cg_atomic_add_u32(&S.colPos[col], 1u); else cg_atomic_sub_u32(&S.colPos[col], 1u); }
}
CG_DEVICE void eval_cache_erase(const SamplerDev &S, uint32_t h)     // ConcurrentAtomicDomain.cpp:62-69
{
    const uint32_t k = cg_atomic_add_u32(&S.gs->eraseCount, 1u);
    if (k < S.eraseCap) S.eraseList[k] = h; else S.gs->error = GAPS_ERR_ERASE_CAP;
}
// ConcurrentAtomicDomain.cpp:126-132 move() across bins: position + the bin-head index
CG_DEVICE void eval_domain_move(const SamplerDev &S, uint32_t h, uint64_t oldPos, uint64_t newPos)
{
    const uint32_t b1 = gen_bin_of(S, oldPos), b2 = gen_bin_of(S, newPos);
    const uint32_t l = S.atoms[h].left, r = S.atoms[h].right;
    S.atoms[h].pos = newPos;
    if (S.binHead[b1] == h) {
        if (r != CG_NONE && gen_bin_of(S, S.atoms[r].pos) == b1) S.binHead[b1] = r;
        else { S.binHead[b1] = CG_NONE; bm_clear(S, b1); }
    }
    if (l == CG_NONE || gen_bin_of(S, S.atoms[l].pos) != b2) S.binHead[b2] = h;
    bm_set(S, b2);
}

CG_DEVICE void eval_body(const SamplerDev &S)
{
    CG_SHARED float lds[32];
    const uint32_t t = cg_tid();
    const uint32_t qlen = S.gs->qlen;
    const float T = S.annealTemp, lambda = S.lambda;
    for (uint32_t q = cg_bid(); q < qlen; q += cg_gdim()) {
        const PropRec p = S.queue[q];
        uint64_t rng = p.rng;
        // every lane reads the scalars this proposal depends on, then a barrier: lane 0 rewrites them
        // at the end of the step and must not overtake a slower wave's reads
        const bool two = (p.type == 'M' || p.type == 'E');
        const float m1 = (p.type == 'B') ? 0.f : S.atoms[p.h1].mass;
        const float m2 = (p.type == 'E') ? S.atoms[p.h2].mass : 0.f;
        const float old1 = S.mat[(size_t)p.c1 * S.Mpad + p.r1];
        const float old2 = two ? S.mat[(size_t)p.c2 * S.Mpad + p.r2] : 0.f;
        const uint64_t curPos = (p.type == 'M') ? S.atoms[p.h1].pos : 0ull;
        const bool gibbs1 = S.otherColPos[p.c1] > 0u;
        const bool gibbs2 = two ? (S.otherColPos[p.c2] > 0u) : false;
        cg_sync();
        if (p.type == 'B') {
            // ---------------------------------------------------------------- birth (:127-144)
            OptF mass;
            if (gibbs1) {
                EvalAcc a = eval_block_reduce(eval_partial_one(S, p.r1, p.c1, false, 0.f), lds);
                mass = gm_gibbs_mass(a.s * T, a.m * T, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda);
            } else { mass.v = pcg_exponential(rng, lambda); mass.has = true; }
            if (mass.has && mass.v >= GAPS_EPSILON) {
                eval_update_ap(S, p.r1, p.c1, mass.v);                              // changeMatrix
                if (t == 0) { S.atoms[p.h1].mass = mass.v; eval_store_matrix(S, p.r1, p.c1, old1, old1 + mass.v); }
            } else if (t == 0) eval_cache_erase(S, p.h1);
        } else if (p.type == 'D') {
            // ---------------------------------------------------------------- death / rebirth (:148-180)
            float rebirth = m1;
            EvalAcc a = eval_block_reduce(eval_partial_one(S, p.r1, p.c1, true, -1.f * m1), lds);
            const float s = a.s * T, smu = a.m * T;
            if (gibbs1) {
                OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda);
                if (g.has) rebirth = g.v;
            }
            const float deltaLL = rebirth * (smu - s * rebirth / 2.f);
            if (gm_logf(pcg_uniform(rng)) < deltaLL) {
                if (rebirth != m1) {
                    const float nv = gm_max(old1 + (rebirth - m1), 0.f);            // safelyChangeMatrix
                    eval_update_ap(S, p.r1, p.c1, nv - old1);
                    if (t == 0) { eval_store_matrix(S, p.r1, p.c1, old1, nv); S.atoms[p.h1].mass = rebirth; }
                }
            } else {
                const float nv = gm_max(old1 + (-1.f * m1), 0.f);
                eval_update_ap(S, p.r1, p.c1, nv - old1);
                if (t == 0) { eval_store_matrix(S, p.r1, p.c1, old1, nv); eval_cache_erase(S, p.h1); }
            }
        } else {
            // ---------------------------------------------------------------- 2-site alpha (:186-214)
            float s = 0.f, smu = 0.f;
            const bool need = (p.type == 'M') || gibbs1 || gibbs2;                  // exchange: canUseGibbs(c1,c2)
            if (need) {
                if (p.r1 == p.r2) {
                    EvalAcc a = eval_block_reduce(eval_partial_two_same(S, p.r1, p.c1, p.c2), lds);
                    s = a.s; smu = a.m;
                } else {
                    EvalAcc a = eval_block_reduce(eval_partial_one(S, p.r1, p.c1, false, 0.f), lds);
                    EvalAcc b = eval_block_reduce(eval_partial_one(S, p.r2, p.c2, false, 0.f), lds);
                    s = a.s + b.s; smu = a.m - b.m;                                 // AlphaParameters.cpp:11-14
                }
                s = s * T; smu = smu * T;
            }
            if (p.type == 'M') {
                // ------------------------------------------------------------ move (:184-196)
                const float deltaLL = -1.f * m1 * (smu + s * m1 / 2.f);
                if (gm_logf(pcg_uniform(rng)) < deltaLL) {
                    const float nv1 = gm_max(old1 + (-m1), 0.f);                    // safelyChangeMatrix(r1,c1,-m)
                    eval_update_ap(S, p.r1, p.c1, nv1 - old1);
                    eval_update_ap(S, p.r2, p.c2, m1);                              // changeMatrix(r2,c2,+m); same lane owns the same elements
                    if (t == 0) {
                        eval_domain_move(S, p.h1, curPos, p.pos);
                        eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                        eval_store_matrix(S, p.r2, p.c2, old2, old2 + m1);
                    }
                }
            } else if (need) {
                // ------------------------------------------------------------ exchange (:201-219)
                OptF g = gm_gibbs_mass(s, smu, -m1, m2, rng, S.luts, false, 0.f);
                const float n1 = m1 + g.v, n2 = m2 - g.v;
                if (g.has && n1 > GAPS_EPSILON && n2 > GAPS_EPSILON) {
                    const float nv1 = gm_max(old1 + (n1 - m1), 0.f);
                    eval_update_ap(S, p.r1, p.c1, nv1 - old1);
                    const float nv2 = gm_max(old2 + (n2 - m2), 0.f);
                    eval_update_ap(S, p.r2, p.c2, nv2 - old2);
                    if (t == 0) {
                        eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                        eval_store_matrix(S, p.r2, p.c2, old2, nv2);
                        S.atoms[p.h1].mass = n1; S.atoms[p.h2].mass = n2;
                    }
                }
            }
        }
        if (t == 0) {   // roofline bookkeeping: algorithmic bytes of this proposal (16N / 20N / 32N per alpha, 12N per AP update)
            const unsigned long long nb = 4ull * S.N;
            unsigned long long bytes = 0;
            if (p.type == 'B') bytes = gibbs1 ? 4 * nb : 0;
            else if (p.type == 'D') bytes = 4 * nb;
            else if (p.type == 'M' || gibbs1 || gibbs2) bytes = (p.r1 == p.r2) ? 5 * nb : 8 * nb;
            cg_atomic_add_u64(&S.gs->evalBytes, bytes);
            cg_atomic_add_u64(&S.gs->evalProps, 1ull);
        }
        cg_sync();   // lane 0's scalar writes are ordered before the next proposal's reads
    }
}

CG_KERNEL void eval_kernel(SamplerDev S) { eval_body(S); }
