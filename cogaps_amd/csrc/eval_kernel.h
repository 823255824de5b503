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
// The step is latency bound (a batch is only ~50-160 proposals), so the kernel is organised as few
// dependent memory round trips as possible: {queue record} -> {the proposal's scalars + every row
// chunk of the reduction, all in flight together} -> {two LUT reads} -> {AP update: re-read from L2,
// store}.
//
// Reduction order (the parity contract with the oracle's redW / redG=4): lane L accumulates float4
// chunks j = L, L+W, L+2W ... in increasing j (x,y,z,w in order) from +0; then an ascending xor
// butterfly 1,2,4,...,W/2 (the reference's AVX hadd tree, SIMD.h:102-107, widened from 8 to W lanes).
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
CG_DEVICE cg_f4 f4_zero() { cg_f4 z; z.x = 0.f; z.y = 0.f; z.z = 0.f; z.w = 0.f; return z; }
CG_DEVICE cg_f4 f4_one() { cg_f4 z; z.x = 1.f; z.y = 1.f; z.z = 1.f; z.w = 1.f; return z; }

struct EvalAcc { float s, m; };

// DenseNormalModel.cpp:170-181 / :229-238, per element:  ratio = v/(S*S); s += v*ratio; s_mu += ratio*(D - AP[+ch*v])
#define EVAL_ELEM(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - (AA)); }
#define EVAL_ELEM_CH(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - ((AA) + ch * (V))); }

// UN chunks per lane are loaded before any is consumed (4*UN independent float4 loads in flight), so a
// row costs one memory round trip when it has at most UN*W chunks.  A chunk index past the row reads
// nothing and contributes (v=0, S2=1, D=AP=0) -> +0 to both sums, which leaves them bit-unchanged.
template <int UN>
CG_DEVICE EvalAcc eval_partial_one(const SamplerDev &S, uint32_t row, uint32_t col, bool withCh, float ch)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    const float *D = S.D + (size_t)row * S.Npad, *S2 = S.S2 + (size_t)row * S.Npad, *AP = S.AP + (size_t)row * S.Npad;
    const float *V = S.other + (size_t)col * S.Npad;
    EvalAcc a; a.s = 0.f; a.m = 0.f;
#if defined(GEN_PROFILE)
    if (S.dbg & 4u) return a;
#endif
    for (uint32_t j0 = t; j0 < nq; j0 += UN * W) {
        cg_f4 v[UN], d[UN], s[UN], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * W;
            if (j < nq) { v[u] = ld4(V, j); d[u] = ld4(D, j); s[u] = ld4(S2, j); p[u] = ld4(AP, j); }
            else { v[u] = f4_zero(); d[u] = f4_zero(); s[u] = f4_one(); p[u] = f4_zero(); }
        }
        if (withCh) {
#pragma unroll
            for (int u = 0; u < UN; ++u) { EVAL_ELEM_CH(v[u].x, d[u].x, s[u].x, p[u].x) EVAL_ELEM_CH(v[u].y, d[u].y, s[u].y, p[u].y) EVAL_ELEM_CH(v[u].z, d[u].z, s[u].z, p[u].z) EVAL_ELEM_CH(v[u].w, d[u].w, s[u].w, p[u].w) }
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) { EVAL_ELEM(v[u].x, d[u].x, s[u].x, p[u].x) EVAL_ELEM(v[u].y, d[u].y, s[u].y, p[u].y) EVAL_ELEM(v[u].z, d[u].z, s[u].z, p[u].z) EVAL_ELEM(v[u].w, d[u].w, s[u].w, p[u].w) }
        }
    }
    return a;
}
// DenseNormalModel.cpp:200-212: same row, v = other[:,c1] - other[:,c2]
template <int UN>
CG_DEVICE EvalAcc eval_partial_two_same(const SamplerDev &S, uint32_t row, uint32_t c1, uint32_t c2)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    const float *D = S.D + (size_t)row * S.Npad, *S2 = S.S2 + (size_t)row * S.Npad, *AP = S.AP + (size_t)row * S.Npad;
    const float *V1 = S.other + (size_t)c1 * S.Npad, *V2 = S.other + (size_t)c2 * S.Npad;
    EvalAcc a; a.s = 0.f; a.m = 0.f;
#if defined(GEN_PROFILE)
    if (S.dbg & 4u) return a;
#endif
    for (uint32_t j0 = t; j0 < nq; j0 += UN * W) {
        cg_f4 v1[UN], v2[UN], d[UN], s[UN], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * W;
            if (j < nq) { v1[u] = ld4(V1, j); v2[u] = ld4(V2, j); d[u] = ld4(D, j); s[u] = ld4(S2, j); p[u] = ld4(AP, j); }
            else { v1[u] = f4_zero(); v2[u] = f4_zero(); d[u] = f4_zero(); s[u] = f4_one(); p[u] = f4_zero(); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            { float v = v1[u].x - v2[u].x; EVAL_ELEM(v, d[u].x, s[u].x, p[u].x) }
            { float v = v1[u].y - v2[u].y; EVAL_ELEM(v, d[u].y, s[u].y, p[u].y) }
            { float v = v1[u].z - v2[u].z; EVAL_ELEM(v, d[u].z, s[u].z, p[u].z) }
            { float v = v1[u].w - v2[u].w; EVAL_ELEM(v, d[u].w, s[u].w, p[u].w) }
        }
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
template <int UN>
CG_DEVICE void eval_update_ap(const SamplerDev &S, uint32_t row, uint32_t col, float delta)
{
    const uint32_t nq = S.Npad >> 2, W = cg_bdim(), t = cg_tid();
    float *AP = S.AP + (size_t)row * S.Npad;
    const float *V = S.other + (size_t)col * S.Npad;
#if defined(GEN_PROFILE)
    if (S.dbg & 2u) return;
#endif
    for (uint32_t j0 = t; j0 < nq; j0 += UN * W) {
        cg_f4 v[UN], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const uint32_t j = j0 + (uint32_t)u * W; if (j < nq) { v[u] = ld4(V, j); p[u] = ld4(AP, j); } else { v[u] = f4_zero(); p[u] = f4_zero(); } }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * W;
            if (j < nq) {
                cg_f4 q = p[u];
                q.x = q.x + delta * v[u].x; q.y = q.y + delta * v[u].y; q.z = q.z + delta * v[u].z; q.w = q.w + delta * v[u].w;
                st4(AP, j, q);
            }
        }
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

#if defined(GEN_PROFILE) && !defined(GEN_SUBMARKS) && !defined(GEN_ROUNDMARKS)
#define EVAL_PROF(i) do { if (t == 0 && cg_bid() == 0) { unsigned long long now_ = cg_clock(); cg_atomic_add_u64(&S.gs->prof[8 + (i)], now_ - eprof_last); eprof_last = now_; } } while (0)
#else
#define EVAL_PROF(i) do { } while (0)
#endif

#if defined(GEN_PROFILE) && !defined(COGAPS_EMUL)
// dev: timestamps of the first 16 workgroups of a launch (lane 0 of the first and of the last wave)
__device__ unsigned long long g_eval_timeline[16 * 2 * 12];
#define EVAL_TS(id) do { if ((t & 63u) == 0u && ets_n < 11u) { ets[ets_n++] = ((unsigned long long)cg_clock() << 8) | (unsigned long long)(id); } } while (0)
#define EVAL_TS_DUMP(ty) do { const uint32_t lastW_ = (cg_bdim() - 1u) >> 6; if (cg_bid() < 16u && (t & 63u) == 0u && ((t >> 6) == 0u || (t >> 6) == lastW_) && qlen >= 100u) { \
    unsigned long long *o_ = &g_eval_timeline[(cg_bid() * 2u + ((t >> 6) ? 1u : 0u)) * 12u]; o_[0] = (unsigned long long)(ty); for (uint32_t i_ = 0; i_ < 11u; ++i_) o_[1 + i_] = i_ < ets_n ? ets[i_] : 0ull; } } while (0)
#define EVAL_PIN(x) asm volatile("" : "+v"(x) :: "memory")
#else
#define EVAL_TS(id) do { } while (0)
#define EVAL_TS_DUMP(ty) do { } while (0)
#define EVAL_PIN(x) do { } while (0)
#endif

template <int UN>
CG_DEVICE void eval_body(const SamplerDev &S)
{
#if defined(GEN_PROFILE) && !defined(COGAPS_EMUL)
    unsigned long long ets[11]; uint32_t ets_n = 0;
#endif
    CG_SHARED float lds[32];
    CG_SHARED float decf; CG_SHARED uint32_t deci;     // decision of wave 0, broadcast to the other waves
    const uint32_t t = cg_tid();
    unsigned long long eprof_last = cg_clock(); (void)eprof_last;
    const float T = S.annealTemp, lambda = S.lambda;
    EVAL_TS(0);
    const bool multiWave = cg_bdim() > 64u;
    const bool scalarLane = !multiWave || t < 64u;       // the per-proposal scalar math (LUTs, fp64 log) runs in wave 0 only
#define EVAL_BCAST(F0, I0) do { if (multiWave) { if (t == 0) { decf = (F0); deci = (I0); } cg_sync(); (F0) = decf; (I0) = deci; } } while (0)
    for (uint32_t q = cg_bid(); ; q += cg_gdim()) {
        // the record is fetched together with the queue length (slot q always exists: q < queueCap)
        const PropRec p = S.queue[q < S.queueCap ? q : 0u];
        const uint32_t qlen = S.gs->qlen;
        if (q >= qlen) break;
        { uint32_t ty_ = p.type; EVAL_PIN(ty_); }
        EVAL_TS(1);
        uint64_t rng = p.rng; uint32_t nUpd = 0;
        // the scalars this proposal depends on: issued now, consumed after the row loads are in flight.
        // lane 0 rewrites them at the end of the step; the barriers inside the reduction (or the explicit
        // one on the paths without a reduction) keep it from overtaking a slower wave's reads.
        const bool two = (p.type == 'M' || p.type == 'E');
        const float m1 = (p.type == 'B') ? 0.f : S.atoms[p.h1].mass;
        const float m2 = (p.type == 'E') ? S.atoms[p.h2].mass : 0.f;
        const float old1 = S.mat[(size_t)p.c1 * S.Mpad + p.r1];
        const float old2 = two ? S.mat[(size_t)p.c2 * S.Mpad + p.r2] : 0.f;
        const uint64_t curPos = (p.type == 'M') ? S.atoms[p.h1].pos : 0ull;
        const bool gibbs1 = S.otherColPos[p.c1] > 0u;
        const bool gibbs2 = two ? (S.otherColPos[p.c2] > 0u) : false;
        EVAL_PROF(0);
        { float a_ = m1, b_ = m2, c_ = old1, d_ = old2; uint32_t g_ = (gibbs1 ? 1u : 0u) | (gibbs2 ? 2u : 0u); EVAL_PIN(a_); EVAL_PIN(b_); EVAL_PIN(c_); EVAL_PIN(d_); EVAL_PIN(g_); }
        EVAL_TS(2);
#if defined(GEN_PROFILE)
        if (S.dbg & 1u) { if (p.type == 0xFFu || m1 == -1.f) S.queueUnits[q] = (uint32_t)old1; break; }   // record + scalars only
#endif
        if (p.type == 'B') {
            // ---------------------------------------------------------------- birth (:127-144)
            OptF mass; mass.v = 0.f; mass.has = false;
            float bv = 0.f; uint32_t bhas = 0;
            if (gibbs1) {
                EvalAcc a = eval_block_reduce(eval_partial_one<UN>(S, p.r1, p.c1, false, 0.f), lds);
                EVAL_PIN(a.s); EVAL_TS(3);
                if (scalarLane) { OptF g = gm_gibbs_mass(a.s * T, a.m * T, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); bv = g.v; bhas = g.has ? 1u : 0u; }
            } else if (scalarLane) { bv = pcg_exponential(rng, lambda); bhas = 1u; }
            EVAL_PIN(bv); EVAL_TS(4);
            EVAL_BCAST(bv, bhas);
            EVAL_TS(5);
            mass.v = bv; mass.has = bhas != 0u;
            if (mass.has && mass.v >= GAPS_EPSILON) {
                eval_update_ap<UN>(S, p.r1, p.c1, mass.v); ++nUpd;                          // changeMatrix
                if (t == 0) { S.atoms[p.h1].mass = mass.v; eval_store_matrix(S, p.r1, p.c1, old1, old1 + mass.v); }
            } else if (t == 0) eval_cache_erase(S, p.h1);
        } else if (p.type == 'D') {
            // ---------------------------------------------------------------- death / rebirth (:148-180)
            float rebirth = m1;
            EvalAcc a = eval_block_reduce(eval_partial_one<UN>(S, p.r1, p.c1, true, -1.f * m1), lds);
            const float s = a.s * T, smu = a.m * T;
            EVAL_PROF(1);
            EVAL_PIN(a.s); EVAL_TS(3);
            uint32_t acc = 0;
            if (scalarLane) {
                if (gibbs1) {
                    OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda);
                    if (g.has) rebirth = g.v;
                }
                const float deltaLL = rebirth * (smu - s * rebirth / 2.f);
                acc = (gm_logf(pcg_uniform(rng)) < deltaLL) ? 1u : 0u;
            }
            EVAL_PIN(acc); EVAL_TS(4);
            EVAL_BCAST(rebirth, acc);
            EVAL_TS(5);
            const bool accept = acc != 0u;
            EVAL_PROF(2);
            if (accept) {
                if (rebirth != m1) {
                    const float nv = gm_max(old1 + (rebirth - m1), 0.f);            // safelyChangeMatrix
                    eval_update_ap<UN>(S, p.r1, p.c1, nv - old1); ++nUpd;
                    if (t == 0) { eval_store_matrix(S, p.r1, p.c1, old1, nv); S.atoms[p.h1].mass = rebirth; }
                }
            } else {
                const float nv = gm_max(old1 + (-1.f * m1), 0.f);
                eval_update_ap<UN>(S, p.r1, p.c1, nv - old1); ++nUpd;
                if (t == 0) { eval_store_matrix(S, p.r1, p.c1, old1, nv); eval_cache_erase(S, p.h1); }
            }
            EVAL_PROF(3);
        } else {
            // ---------------------------------------------------------------- 2-site alpha (:186-214)
            float s = 0.f, smu = 0.f;
            const bool need = (p.type == 'M') || gibbs1 || gibbs2;                  // exchange: canUseGibbs(c1,c2)
            if (need) {
                if (p.r1 == p.r2) {
                    EvalAcc a = eval_block_reduce(eval_partial_two_same<UN>(S, p.r1, p.c1, p.c2), lds);
                    s = a.s; smu = a.m;
                } else {
                    EvalAcc a = eval_block_reduce(eval_partial_one<UN>(S, p.r1, p.c1, false, 0.f), lds);
                    EvalAcc b = eval_block_reduce(eval_partial_one<UN>(S, p.r2, p.c2, false, 0.f), lds);
                    s = a.s + b.s; smu = a.m - b.m;                                 // AlphaParameters.cpp:11-14
                }
                s = s * T; smu = smu * T;
            } else if (multiWave) cg_sync();
            EVAL_PIN(s); EVAL_TS(3);
            if (p.type == 'M') {
                // ------------------------------------------------------------ move (:184-196)
                uint32_t acc = 0; float unused = 0.f;
                if (scalarLane) { const float deltaLL = -1.f * m1 * (smu + s * m1 / 2.f); acc = (gm_logf(pcg_uniform(rng)) < deltaLL) ? 1u : 0u; }
                EVAL_PIN(acc); EVAL_TS(4);
                EVAL_BCAST(unused, acc);
                EVAL_TS(5);
                if (acc) {
                    const float nv1 = gm_max(old1 + (-m1), 0.f);                    // safelyChangeMatrix(r1,c1,-m)
                    eval_update_ap<UN>(S, p.r1, p.c1, nv1 - old1); ++nUpd;
                    eval_update_ap<UN>(S, p.r2, p.c2, m1); ++nUpd;                          // changeMatrix(r2,c2,+m); same lane owns the same elements
                    if (t == 0) {
                        eval_domain_move(S, p.h1, curPos, p.pos);
                        eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                        eval_store_matrix(S, p.r2, p.c2, old2, old2 + m1);
                    }
                }
            } else if (need) {
                // ------------------------------------------------------------ exchange (:201-219)
                OptF g; g.v = 0.f; g.has = false;
                { float gv = 0.f; uint32_t gh = 0;
                  if (scalarLane) { OptF g0 = gm_gibbs_mass(s, smu, -m1, m2, rng, S.luts, false, 0.f); gv = g0.v; gh = g0.has ? 1u : 0u; }
                  EVAL_PIN(gv); EVAL_TS(4);
                  EVAL_BCAST(gv, gh); g.v = gv; g.has = gh != 0u; }
                EVAL_TS(5);
                const float n1 = m1 + g.v, n2 = m2 - g.v;
                if (g.has && n1 > GAPS_EPSILON && n2 > GAPS_EPSILON) {
                    const float nv1 = gm_max(old1 + (n1 - m1), 0.f);
                    eval_update_ap<UN>(S, p.r1, p.c1, nv1 - old1); ++nUpd;
                    const float nv2 = gm_max(old2 + (n2 - m2), 0.f);
                    eval_update_ap<UN>(S, p.r2, p.c2, nv2 - old2); ++nUpd;
                    if (t == 0) {
                        eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                        eval_store_matrix(S, p.r2, p.c2, old2, nv2);
                        S.atoms[p.h1].mass = n1; S.atoms[p.h2].mass = n2;
                    }
                }
            }
        }
        EVAL_TS(6);
        EVAL_TS_DUMP(p.type | (nUpd << 8) | ((p.r1 == p.r2 ? 1u : 0u) << 16));
        if (t == 0) {   // roofline bookkeeping: algorithmic traffic of this proposal in units of 4N bytes
            // (alpha: 4 one-site, 5 two-site same row, 8 different rows; 3 per AP update); the generator sums the slots
            uint32_t units = nUpd * 3u;
            if (p.type == 'B') units += gibbs1 ? 4u : 0u;
            else if (p.type == 'D') units += 4u;
            else if (p.type == 'M' || gibbs1 || gibbs2) units += (p.r1 == p.r2) ? 5u : 8u;
            S.queueUnits[q] = units;
        }
        if (q + cg_gdim() >= qlen) break;   // last proposal of this workgroup: nothing left to order
        cg_sync();   // lane 0's scalar writes are ordered before the next proposal's reads
    }
}

// UN = 8: workgroups of up to 256 lanes (512-VGPR budget); UN = 4: up to 1024 lanes (128 VGPRs per lane)
template <int UN>
CG_KERNEL void CG_LAUNCH_BOUNDS(UN == 8 ? 256 : 1024) eval_kernel(SamplerDev S) { eval_body<UN>(S); }
