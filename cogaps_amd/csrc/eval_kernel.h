// eval_kernel.h -- evaluation of one queued batch: the birth / death / move / exchange steps of
// AsynchronousGibbsSampler.h:127-219 over the DenseNormalModel reductions marked PERFORMANCE
// CRITICAL in DenseNormalModel.cpp:161-258.
//
// One workgroup per queued proposal.  A proposal touches one or two factor rows, the queue guarantees that
// no two proposals of a batch share a row, so workgroups never write the same AP row / matrix entry / atom.
//
// HBM traffic per proposal (N = data-vector length): alpha with or without change 16N bytes,
// 2-site same row 20N, different rows 32N, AP update 12N.  Rows are read as coalesced float4.
// The step is latency bound (a batch is only ~50-160 proposals), so the kernel is organised as few
// dependent memory round trips as possible: {queue record, which carries the proposal's scalars} ->
// {every row chunk of the reduction, both rows of a two-row step together} -> {two LUT reads} ->
// {AP update: re-read from L2, store}.
//
// Reduction order (the parity contract with the oracle's redW / redG=4): W = cogaps_reduction_width(N) virtual
// lanes, one float4 chunk each whenever N <= 4*W: virtual lane L accumulates chunks j = L, L+W, ... in
// increasing j (x,y,z,w in order) from +0; then an ascending xor butterfly 1,2,4,...,W/2 (the reference's AVX
// hadd tree, SIMD.h:102-107, widened from 8 to W lanes).  A workgroup has BS = min(W,1024) threads and thread t
// owns the V = W/BS virtual lanes t, t+BS, ...: butterfly bits 0-5 are wave shuffles, bits 6.. a tree over the
// waves' totals in LDS, the top log2(V) bits a tree over the thread's own V slots.
#pragma once
#include "gaps_state.h"
#include "gen_kernel.h"   // gen_bin_of, bm_set, bm_clear

// (cg_f4, cg_ld4_stream: platform.h)
CG_DEVICE cg_f4 ld4(const float *base, uint32_t j) { return reinterpret_cast<const cg_f4 *>(base)[j]; }
// streaming read (data / uncertainty rows are used once per batch): keeps them from displacing the lookup tables,
// the queue and the atom records in L2
CG_DEVICE cg_f4 ld4_stream(const float *base, uint32_t j) { return cg_ld4_stream(base, j); }
CG_DEVICE void st4(float *base, uint32_t j, cg_f4 v) { reinterpret_cast<cg_f4 *>(base)[j] = v; }
CG_DEVICE cg_f4 f4_zero() { cg_f4 z; z.x = 0.f; z.y = 0.f; z.z = 0.f; z.w = 0.f; return z; }
CG_DEVICE cg_f4 f4_one() { cg_f4 z; z.x = 1.f; z.y = 1.f; z.z = 1.f; z.w = 1.f; return z; }

struct EvalAcc { float s, m; };

// DenseNormalModel.cpp:170-181 / :229-238, per element:  ratio = v/(S*S); s += v*ratio; s_mu += ratio*(D - AP[+ch*v])
#define EVAL_ELEM(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - (AA)); }
#define EVAL_ELEM_CH(V, DD, SS, AA) { float ratio = (V) / (SS); a.s = a.s + (V) * ratio; a.m = a.m + ratio * ((DD) - ((AA) + ch * (V))); }

#define EVAL_MODE_ONE 0      // v = other[:,c1]
#define EVAL_MODE_CH 1       // ... with the change ch*v added to AP (death)
#define EVAL_MODE_SAME 2     // v = other[:,c1] - other[:,c2], one row (DenseNormalModel.cpp:200-212)

#if defined(GEN_TIMELINE)
extern __device__ unsigned long long g_chain_rt[256 * 4];      // (chain_kernel.h)
// dev: timestamps of the first 16 workgroups of a launch (lane 0 of the first and of the last wave)
__device__ unsigned long long g_eval_timeline[2 * 16 * 2 * 12];     // [narrow | wide workgroups]
#define EVAL_TS(id) do { if ((t & 63u) == 0u && ets_n < 11u) { ets[ets_n++] = ((unsigned long long)cg_clock() << 8) | (unsigned long long)(id); } } while (0)
#define EVAL_TS_DUMP(ty) do { const uint32_t lastW_ = (cg_bdim() - 1u) >> 6; if (cg_bid() < 16u && (t & 63u) == 0u && ((t >> 6) == 0u || (t >> 6) == lastW_) && qlen >= 40u && (PHASE != EVAL_CHAIN || (qlen >= 140u && hot.gs->nSteps - hot.gs->nDone >= 512u))) { \
    unsigned long long *o_ = &g_eval_timeline[((PHASE != EVAL_FUSED && PHASE != EVAL_CHAIN ? 16u : 0u) * 2u + cg_bid() * 2u + ((t >> 6) ? 1u : 0u)) * 12u]; o_[0] = (unsigned long long)(ty); for (uint32_t i_ = 0; i_ < 11u; ++i_) o_[1 + i_] = i_ < ets_n ? ets[i_] : 0ull; } } while (0)
#define EVAL_PIN(x) asm volatile("" : "+v"(x) :: "memory")
#define EVAL_TS_PARAMS , unsigned long long (&ets)[11], uint32_t &ets_n
#define EVAL_TS_ARGS , ets, ets_n
#else
#define EVAL_TS(id) do { } while (0)
#define EVAL_TS_DUMP(ty) do { } while (0)
#define EVAL_PIN(x) do { } while (0)
#define EVAL_TS_PARAMS
#define EVAL_TS_ARGS
#endif

// G: virtual lanes whose chunks are in flight together (registers: 16*G*NR floats per thread)
// lane-order sum of the NC components whose per-wave, per-slot partials sit in lds[wave][NC][V] (after a
// barrier): wave 0 folds them, lane i < NC*V over the waves (bits 6.. of the virtual lane index), then the
// thread-slot bits across lanes.  Totals are returned in wave 0.
// (nwaves: the workgroup's waves that took part -- all of them unless the caller says otherwise: an evaluation workgroup inside a chained
// launch of the sparse model has the launch's size and the model's width)
template <int NC, int V>
CG_DEVICE void eval_vfinish(const float *lds, float (&tot)[NC], const uint32_t nwaves = 0u, const uint32_t tid = 0xFFFFFFFFu)
{
    constexpr int NV = NC * V;
    const uint32_t t = tid == 0xFFFFFFFFu ? cg_tid() : tid, nw = cg_fresh_u32(nwaves ? nwaves : (cg_bdim() >> 6));
    if (t < 64u) {
        const uint32_t i = t < (uint32_t)NV ? t : 0u;
        // all sixteen slots are read at once (one wait instead of one per wave) and the ones past the last wave masked afterwards:
        // the scratch has 16 * NV floats whatever the workgroup size, stale words are never added
        float y[16];
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) y[w] = lds[w * NV + i];
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) y[w] = w < nw ? y[w] : 0.f;
        // the whole sixteen-slot tree, statically: above the last wave it adds +0 to sums that cannot be -0 (they start from +0), which
        // changes no bit, and the registers are indexed by constants (a run-time bound made every access an indexed move)
#pragma unroll
        for (uint32_t stride = 1; stride < 16u; stride <<= 1) {
#pragma unroll
            for (uint32_t k = 0; k + stride < 16u; k += 2u * stride) y[k] = y[k] + y[k + stride];
        }
        float z = y[0];
        for (int off = 1; off < V; off <<= 1) z = z + cg_shfl_xor_f32(z, off);     // lanes c*V .. c*V+V-1: the V slots of component c
#pragma unroll
        for (int c = 0; c < NC; ++c) tot[c] = cg_lane_read_f32(z, c * V);
    }
}

// one component: wave butterfly of slot `j`'s partial, parked for eval_vfinish<1, V> (one wave: the total)
template <int V>
CG_DEVICE void eval_vpark(float x, int j, float *lds, float (&tot)[1])
{
    x = cg_wave_allsum_f32(x);
    const uint32_t t = cg_tid();
    if (cg_bdim() > 64u) { if ((t & 63u) == 0) lds[(t >> 6) * V + j] = x; }
    else tot[0] = x;
}

// Alpha parameters of NR rows (NR = 2: the two rows of a move / exchange across rows, loaded together) over the
// chunks chunk0 + t, + stride, + 2*stride ...: thread t accumulates them in increasing order from +0 (one chunk
// unless the vector is longer than 4*W), the workgroup folds its BS partials by an ascending xor butterfly.
// tot = {s, s_mu} per row, valid in wave 0 (in every lane of a one-wave workgroup).  A chunk index past the
// row reads nothing and contributes (v=0, S2=1, D=AP=0) -> +0 to both sums, which leaves them bit-unchanged.
// lds: [16][2*NR].
// The accept test's logarithm, ahead of time (one-chain fused form).  log(uniform) is the last thing a death or a move computes, it is a
// chain of ~50 dependent fp64 operations, and its argument does not depend on the reduction: it is the proposal's first draw, or its
// second when the truncated normal of a death's rebirth consumed one (Random.cpp:178-191 draws only if it returns a value).  Wave 0
// computes the candidates right behind the request for its row chunks, while they travel, and the scalar step picks one.
struct EvalSpec { uint64_t rng; uint32_t n, mm; float l1, l2; };
CG_DEVICE void eval_spec_run(EvalSpec &sp)
{
    uint64_t r = sp.rng;
    sp.l1 = gm_logf_m(pcg_uniform(r), sp.mm);
    if (sp.n > 1u) sp.l2 = gm_logf_m(pcg_uniform(r), sp.mm);
    sp.n = 0u;
}
template <int NR, int MODE>
CG_DEVICE void eval_alpha(const SamplerDev &S, const uint32_t (&row)[NR], const uint32_t (&col)[NR], uint32_t col2, float ch, uint32_t chunk0, uint32_t stride, float *lds, float (&tot)[2 * NR], EvalSpec &spec EVAL_TS_PARAMS)
{
    constexpr int NC = 2 * NR;
    const uint32_t nq = S.Npad >> 2, BS = cg_bdim(), t = cg_tid(), nw = BS >> 6;
    float ps[NR], pm[NR];
    for (int r = 0; r < NR; ++r) { ps[r] = 0.f; pm[r] = 0.f; }
#if defined(GEN_PROFILE)
    if (!(S.dbg & 4u))
#endif
    for (uint32_t j = chunk0 + t; j < nq; j += stride) {
        cg_f4 v[NR], w2, d[NR], s[NR], p[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float *Dr = S.D + (size_t)row[r] * S.Npad, *Sr = S.S2 + (size_t)row[r] * S.Npad, *Ar = S.AP + (size_t)row[r] * S.Npad;
            const float *Vr = S.other + (size_t)col[r] * S.Npad, *V2 = S.other + (size_t)col2 * S.Npad;
            v[r] = ld4(Vr, j); d[r] = ld4_stream(Dr, j); if (!S.defaultS) s[r] = ld4_stream(Sr, j); p[r] = ld4(Ar, j); if (MODE == EVAL_MODE_SAME) w2 = ld4(V2, j);
        }
        if (spec.n) { cg_sched_fence(); eval_spec_run(spec); cg_sched_fence(); }      // (wave-uniform; the loads above are out)
        if (S.defaultS) {
            // default uncertainty S = max(0.1 D, 0.1) (MatrixMath.cpp:74-84), S*S: the same three fp32 operations the host
            // made when it filled S2, on the value just loaded -- one row less to bring in from HBM per proposal
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float sx = gm_max(d[r].x * 0.1f, 0.1f), sy = gm_max(d[r].y * 0.1f, 0.1f), sz = gm_max(d[r].z * 0.1f, 0.1f), sw = gm_max(d[r].w * 0.1f, 0.1f);
                s[r].x = sx * sx; s[r].y = sy * sy; s[r].z = sz * sz; s[r].w = sw * sw;
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            EvalAcc a; a.s = ps[r]; a.m = pm[r];
            if (MODE == EVAL_MODE_CH) { EVAL_ELEM_CH(v[r].x, d[r].x, s[r].x, p[r].x) EVAL_ELEM_CH(v[r].y, d[r].y, s[r].y, p[r].y) EVAL_ELEM_CH(v[r].z, d[r].z, s[r].z, p[r].z) EVAL_ELEM_CH(v[r].w, d[r].w, s[r].w, p[r].w) }
            else if (MODE == EVAL_MODE_SAME) {
                { float x = v[r].x - w2.x; EVAL_ELEM(x, d[r].x, s[r].x, p[r].x) }
                { float x = v[r].y - w2.y; EVAL_ELEM(x, d[r].y, s[r].y, p[r].y) }
                { float x = v[r].z - w2.z; EVAL_ELEM(x, d[r].z, s[r].z, p[r].z) }
                { float x = v[r].w - w2.w; EVAL_ELEM(x, d[r].w, s[r].w, p[r].w) }
            } else { EVAL_ELEM(v[r].x, d[r].x, s[r].x, p[r].x) EVAL_ELEM(v[r].y, d[r].y, s[r].y, p[r].y) EVAL_ELEM(v[r].z, d[r].z, s[r].z, p[r].z) EVAL_ELEM(v[r].w, d[r].w, s[r].w, p[r].w) }
            ps[r] = a.s; pm[r] = a.m;
        }
    }
    { float z_ = ps[0]; EVAL_PIN(z_); ps[0] = z_; } EVAL_TS(10);
    // bits 0-5 of the lane index: the wave butterfly
#pragma unroll
    for (int r = 0; r < NR; ++r) { ps[r] = cg_wave_allsum_f32(ps[r]); pm[r] = cg_wave_allsum_f32(pm[r]); }
    if (nw > 1) {
        if ((t & 63u) == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { lds[(t >> 6) * NC + 2 * r] = ps[r]; lds[(t >> 6) * NC + 2 * r + 1] = pm[r]; }
        }
        EVAL_TS(11);
        cg_sync();
        EVAL_TS(12);
        eval_vfinish<NC, 1>(lds, tot);
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) { tot[2 * r] = ps[r]; tot[2 * r + 1] = pm[r]; }
    }
}

// ---- verification mode: the reference's scalar order ------------------------------------------------------------------
// The reference's default build sums i = 0 .. N-1 into one accumulator (SIMD.h:36-47 with SIMD_INC = 1).  The element
// terms are computed by the whole workgroup (4 * BS at a time, parked in LDS) and folded by one thread per component in
// index order, so a sum costs N dependent additions: slow by design, bit-identical to the scalar reference build.
#define EVAL_SEQ_BS 256
// acc + term[0] + term[1] + ... + term[n-1], left to right
CG_DEVICE float eval_seq_fold(float acc, const float *term, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) acc = acc + term[i];
    return acc;
}
// tot = {s, s_mu} per row (every thread of the workgroup receives them).  term: LDS [2*NR][4*BS]; lds: [2*NR] broadcast
template <int NR, int MODE>
CG_DEVICE void eval_alpha_seq(const SamplerDev &S, const uint32_t (&row)[NR], const uint32_t (&col)[NR], uint32_t col2, float ch, float *term, float *lds, float (&tot)[2 * NR])
{
    constexpr int NC = 2 * NR;
    const uint32_t nq = S.Npad >> 2, BS = cg_bdim(), t = cg_tid();
    float acc = 0.f;                          // thread c < NC owns component c
    for (uint32_t base = 0; base < nq; base += BS) {
        const uint32_t j = base + t;
        if (j < nq) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float *Dr = S.D + (size_t)row[r] * S.Npad, *Ar = S.AP + (size_t)row[r] * S.Npad;
                const float *Vr = S.other + (size_t)col[r] * S.Npad, *V2 = S.other + (size_t)col2 * S.Npad;
                const cg_f4 v4 = ld4(Vr, j), d4 = ld4(Dr, j), p4 = ld4(Ar, j);
                cg_f4 s4, w4 = f4_zero();
                if (S.defaultS) {
                    const float sx = gm_max(d4.x * 0.1f, 0.1f), sy = gm_max(d4.y * 0.1f, 0.1f), sz = gm_max(d4.z * 0.1f, 0.1f), sw = gm_max(d4.w * 0.1f, 0.1f);
                    s4.x = sx * sx; s4.y = sy * sy; s4.z = sz * sz; s4.w = sw * sw;
                } else s4 = ld4(S.S2 + (size_t)row[r] * S.Npad, j);
                if (MODE == EVAL_MODE_SAME) w4 = ld4(V2, j);
                const float vv[4] = {v4.x, v4.y, v4.z, v4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (MODE == EVAL_MODE_SAME) ? vv[e] - ww[e] : vv[e];
                    const float ratio = v / ss[e];
                    const float ap = (MODE == EVAL_MODE_CH) ? pp[e] + ch * vv[e] : pp[e];
                    term[(2 * r) * 4u * BS + 4u * t + (uint32_t)e] = v * ratio;
                    term[(2 * r + 1) * 4u * BS + 4u * t + (uint32_t)e] = ratio * (dd[e] - ap);
                }
            }
        }
        cg_sync();
        const uint32_t first = 4u * base, n = (S.N - first) < 4u * BS ? (S.N - first) : 4u * BS;     // real elements only
        if (t < (uint32_t)NC) acc = eval_seq_fold(acc, term + t * 4u * BS, n);
        cg_sync();
    }
    if (t < (uint32_t)NC) lds[t] = acc;
    cg_sync();
#pragma unroll
    for (int c = 0; c < NC; ++c) tot[c] = lds[c];
    cg_sync();
}

// DenseNormalModel.cpp:243-258: AP[:,row] += delta * other[:,col]
// over the chunks chunk0 + t, + stride, ... (a whole row in the fused kernel, one slice in the split one)
CG_DEVICE void eval_update_ap(const SamplerDev &S, uint32_t row, uint32_t col, float delta, uint32_t chunk0, uint32_t stride)
{
    constexpr int UN = 4;
    const uint32_t nq = S.Npad >> 2, t = cg_tid();
    float *AP = S.AP + (size_t)row * S.Npad;
    const float *Vc = S.other + (size_t)col * S.Npad;
#if defined(GEN_PROFILE)
    if (S.dbg & 2u) return;
#endif
    for (uint32_t j0 = chunk0 + t; j0 < nq; j0 += UN * stride) {
        cg_f4 v[UN], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const uint32_t j = j0 + (uint32_t)u * stride; if (j < nq) { v[u] = ld4(Vc, j); p[u] = ld4(AP, j); } else { v[u] = f4_zero(); p[u] = f4_zero(); } }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * stride;
            if (j < nq) {
                cg_f4 q = p[u];
                q.x = q.x + delta * v[u].x; q.y = q.y + delta * v[u].y; q.z = q.z + delta * v[u].z; q.w = q.w + delta * v[u].w;
                st4(AP, j, q);
            }
        }
    }
}

// The two updates of an accepted move / exchange, AP[:,r1] += d1 * other[:,c1] then AP[:,r2] += d2 * other[:,c2], with all their loads in
// one trip (two calls of eval_update_ap pay the trip twice, one after the other, and the launch lasts as long as its slowest workgroup --
// a two-site one).  r1 == r2: the second update continues from the first one's result in registers -- the same additions in the same
// order as the stored-and-reloaded form.
CG_DEVICE void eval_update_ap2(const SamplerDev &S, uint32_t r1, uint32_t c1, float d1, uint32_t r2, uint32_t c2, float d2, uint32_t chunk0, uint32_t stride)
{
    constexpr int UN = 2;
    const uint32_t nq = S.Npad >> 2, t = cg_tid();
    float *AP1 = S.AP + (size_t)r1 * S.Npad, *AP2 = S.AP + (size_t)r2 * S.Npad;
    const float *V1 = S.other + (size_t)c1 * S.Npad, *V2 = S.other + (size_t)c2 * S.Npad;
    const bool same = r1 == r2;
#if defined(GEN_PROFILE)
    if (S.dbg & 2u) return;
#endif
    for (uint32_t j0 = chunk0 + t; j0 < nq; j0 += UN * stride) {
        cg_f4 v1[UN], v2[UN], p1[UN], p2[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * stride;
            if (j < nq) { v1[u] = ld4(V1, j); v2[u] = ld4(V2, j); p1[u] = ld4(AP1, j); p2[u] = same ? f4_zero() : ld4(AP2, j); }
            else { v1[u] = f4_zero(); v2[u] = f4_zero(); p1[u] = f4_zero(); p2[u] = f4_zero(); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t j = j0 + (uint32_t)u * stride;
            if (j < nq) {
                cg_f4 q = p1[u];
                q.x = q.x + d1 * v1[u].x; q.y = q.y + d1 * v1[u].y; q.z = q.z + d1 * v1[u].z; q.w = q.w + d1 * v1[u].w;
                if (same) {
                    q.x = q.x + d2 * v2[u].x; q.y = q.y + d2 * v2[u].y; q.z = q.z + d2 * v2[u].z; q.w = q.w + d2 * v2[u].w;
                    st4(AP1, j, q);
                } else {
                    st4(AP1, j, q);
                    cg_f4 w = p2[u];
                    w.x = w.x + d2 * v2[u].x; w.y = w.y + d2 * v2[u].y; w.z = w.z + d2 * v2[u].z; w.w = w.w + d2 * v2[u].w;
                    st4(AP2, j, w);
                }
            }
        }
    }
}

// The fused evaluation of ONE chain (a launch is as long as its slowest workgroup's dependent chain) asks for the chunk its thread
// will rewrite -- the A*P row(s) and the other matrix's column(s) the record names -- BEFORE the scalar step and uses it after the
// decision: the update's trip runs under the table lookups and the logarithm instead of after them.  (One chunk per thread there: the
// fused form has at least as many threads as chunks.)  A rejected proposal has read 16-32 KB for nothing, from L2 mostly.
struct EvalPre { cg_f4 v1, p1, v2, p2; };
// (j = chunk0 + t is the thread's first chunk; a vector longer than the reduction is wide has further chunks, done the usual way)
CG_DEVICE void eval_update_pre1(const SamplerDev &S, uint32_t row, uint32_t col, float delta, uint32_t chunk0, uint32_t stride, const cg_f4 &v, const cg_f4 &p)
{
    const uint32_t j = chunk0 + cg_tid(), nq = S.Npad >> 2;
    if (j < nq) {
        cg_f4 q = p;
        q.x = q.x + delta * v.x; q.y = q.y + delta * v.y; q.z = q.z + delta * v.z; q.w = q.w + delta * v.w;
        st4(S.AP + (size_t)row * S.Npad, j, q);
    }
    if (nq > stride) eval_update_ap(S, row, col, delta, chunk0 + stride, stride);
}
CG_DEVICE void eval_update_pre2(const SamplerDev &S, uint32_t r1, uint32_t c1, float d1, uint32_t r2, uint32_t c2, float d2, uint32_t chunk0, uint32_t stride, const EvalPre &e)
{
    const uint32_t j = chunk0 + cg_tid(), nq = S.Npad >> 2;
    if (j < nq) {
        cg_f4 q = e.p1;
        q.x = q.x + d1 * e.v1.x; q.y = q.y + d1 * e.v1.y; q.z = q.z + d1 * e.v1.z; q.w = q.w + d1 * e.v1.w;
        if (r1 == r2) {
            q.x = q.x + d2 * e.v2.x; q.y = q.y + d2 * e.v2.y; q.z = q.z + d2 * e.v2.z; q.w = q.w + d2 * e.v2.w;
            st4(S.AP + (size_t)r1 * S.Npad, j, q);
        } else {
            st4(S.AP + (size_t)r1 * S.Npad, j, q);
            cg_f4 w = e.p2;
            w.x = w.x + d2 * e.v2.x; w.y = w.y + d2 * e.v2.y; w.z = w.z + d2 * e.v2.z; w.w = w.w + d2 * e.v2.w;
            st4(S.AP + (size_t)r2 * S.Npad, j, w);
        }
    }
    if (nq > stride) eval_update_ap2(S, r1, c1, d1, r2, c2, d2, chunk0 + stride, stride);
}

// mMatrix(row,col) = newv, keeping the per-column count of positive entries (canUseGibbs) current
CG_DEVICE void eval_store_matrix(const SamplerDev &S, uint32_t row, uint32_t col, float oldv, float newv)
{
    S.mat[(size_t)col * S.Mpad + row] = newv;
    const bool was = oldv > 0.f, is = newv > 0.f;
    if (was != is) { if (is) cg_atomic_add_u32(&S.colPos[col], 1u); else cg_atomic_sub_u32(&S.colPos[col], 1u); }
}
CG_DEVICE void eval_cache_erase(const SamplerDev &S, uint32_t h, uint32_t row, uint32_t col)     // ConcurrentAtomicDomain.cpp:62-69
{
    const uint32_t k = cg_atomic_add_u32(&S.gs->eraseCount, 1u);
    if (k < S.eraseCap) S.eraseList[k] = ((unsigned long long)(row * S.K + col) << 32) | (unsigned long long)h; else S.gs->error = GAPS_ERR_ERASE_CAP;
}
// ConcurrentAtomicDomain.cpp:126-132 move() across bins: position (+ the copies the neighbours cache) + the bin-head index.  `a` is the
// atom's own record as the evaluation found it (fetched with the rows): its links and the neighbours' positions it caches are current --
// births queued after the move may have changed the links, nothing moves next to a moving atom (ProposalQueue.cpp:167,218) -- so the
// only dependent read left is the old bin's head.
CG_DEVICE void eval_domain_move(const SamplerDev &S, const PropRec &p, const AtomRec &a)
{
    const uint32_t h = p.h1, l = a.left, r = a.right;
    const uint32_t b1 = gen_bin_of(S, p.curPos), b2 = gen_bin_of(S, p.pos);
    const uint32_t head1 = S.binHead[b1];
    atom_set_pos(S, h, l, r, p.pos);
    if (head1 == h) {
        if (r != CG_NONE && gen_bin_of(S, a.rpos) == b1) S.binHead[b1] = r;
        else { S.binHead[b1] = CG_NONE; bm_clear(S, b1); }
    }
    if (l == CG_NONE || gen_bin_of(S, a.lpos) != b2) S.binHead[b2] = h;
    bm_set(S, b2);
}
// the writer thread's view of the atoms it may rewrite, requested when the queue record arrives and used after the decision
struct EvalAtoms { AtomRec a1; uint32_t left2; };
CG_DEVICE EvalAtoms eval_atoms_load(const SamplerDev &S, const PropRec &p, bool writer)
{
    EvalAtoms e; e.a1.pos = 0; e.a1.lpos = 0; e.a1.rpos = 0; e.a1.left = CG_NONE; e.a1.right = CG_NONE; e.a1.mass = 0.f; e.a1.rmass = 0.f; e.a1.idx = 0; e.a1.pad0 = 0; e.left2 = CG_NONE;
    if (writer) { e.a1 = S.atoms[p.h1]; if (p.type == 'E') e.left2 = S.atoms[p.h2].left; }
    return e;
}

// ---- chained launch: two proposals evaluated side by side by the halves of a workgroup ------------------------------------------------
// A chained launch has at most one evaluation workgroup per compute unit (chain_kernel.h), 240 of them; a batch that queued more
// proposals -- the batch after a generator launch of two rounds, 18 % of the headline chain's launches -- used to give some
// workgroups a second proposal AFTER their first, and the launch's generator workgroup, which needs the LAST decision, waited 4 us
// longer.  Here such a workgroup evaluates its two proposals at once: half h (threads h*H .. h*H+H-1, H = BS / 2) takes proposal h,
// thread u of a half holds the virtual lanes u and u + H of the W = BS lanes (one chunk each), the half's first wave makes the decision.
// The sums are the reduction contract's: each slot's lanes are folded by the wave butterfly and the tree over the half's waves --
// exactly the lower and the upper half of the whole workgroup's tree -- and the two slots are added last, which is the butterfly's
// last stage (x[L] + x[L ^ H]).  Both halves execute the same three barriers whatever their proposals' types.  Decisions, draws and
// the A*P updates are eval_body's, operation for operation.
template <int NC>
CG_DEVICE void eval_pair_fold(const float *lds, const uint32_t tv, const uint32_t nw, float (&tot)[NC])
{
    constexpr int NV = NC * 2;
    if (tv < 64u) {
        const uint32_t i = tv < (uint32_t)NV ? tv : 0u;
        float y[16];
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) y[w] = lds[w * NV + i];
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) y[w] = w < nw ? y[w] : 0.f;
#pragma unroll
        for (uint32_t stride = 1; stride < 16u; stride <<= 1) {
#pragma unroll
            for (uint32_t k = 0; k + stride < 16u; k += 2u * stride) y[k] = y[k] + y[k + stride];
        }
        float z = y[0];
        z = z + cg_shfl_xor_f32(z, 1);      // lanes 2c, 2c+1: the two slots of component c
#pragma unroll
        for (int c = 0; c < NC; ++c) tot[c] = cg_lane_read_f32(z, c * 2);
    }
}
// alpha parameters of NR rows over the half's two slots; tot valid in the half's first wave.  lds: this half's [16][2 * NR * 2] scratch.
// (The barrier between the parking and the fold is the caller's: both halves share it.)
template <int NR, int MODE>
CG_DEVICE void eval_pair_alpha(const SamplerDev &S, const uint32_t (&row)[NR], const uint32_t (&col)[NR], uint32_t col2, float ch, const uint32_t u, const uint32_t H, float *lds, EvalSpec &spec)
{
    constexpr int NC = 2 * NR;
    const uint32_t nq = S.Npad >> 2;
    float ps[NR][2], pm[NR][2];
    cg_f4 v[2][NR], w2[2], d[2][NR], sq[2][NR], pp[2][NR];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint32_t j = u + (uint32_t)sl * H, jj = j < nq ? j : nq - 1u;      // (a lane past the row reads its last chunk and adds nothing)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float *Dr = S.D + (size_t)row[r] * S.Npad, *Ar = S.AP + (size_t)row[r] * S.Npad, *Vr = S.other + (size_t)col[r] * S.Npad;
            v[sl][r] = ld4(Vr, jj); d[sl][r] = ld4_stream(Dr, jj); pp[sl][r] = ld4(Ar, jj);
            if (!S.defaultS) sq[sl][r] = ld4_stream(S.S2 + (size_t)row[r] * S.Npad, jj);
        }
        w2[sl] = (MODE == EVAL_MODE_SAME) ? ld4(S.other + (size_t)col2 * S.Npad, jj) : f4_zero();
    }
    if (spec.n) { cg_sched_fence(); eval_spec_run(spec); cg_sched_fence(); }      // (wave-uniform; the loads above are out)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint32_t j = u + (uint32_t)sl * H;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (S.defaultS) {
                const float sx = gm_max(d[sl][r].x * 0.1f, 0.1f), sy = gm_max(d[sl][r].y * 0.1f, 0.1f), sz = gm_max(d[sl][r].z * 0.1f, 0.1f), sw = gm_max(d[sl][r].w * 0.1f, 0.1f);
                sq[sl][r].x = sx * sx; sq[sl][r].y = sy * sy; sq[sl][r].z = sz * sz; sq[sl][r].w = sw * sw;
            }
            EvalAcc a; a.s = 0.f; a.m = 0.f;
            if (j < nq) {
                const cg_f4 vv = v[sl][r], dd = d[sl][r], ss = sq[sl][r], aa = pp[sl][r], ww = w2[sl];
                if (MODE == EVAL_MODE_CH) { EVAL_ELEM_CH(vv.x, dd.x, ss.x, aa.x) EVAL_ELEM_CH(vv.y, dd.y, ss.y, aa.y) EVAL_ELEM_CH(vv.z, dd.z, ss.z, aa.z) EVAL_ELEM_CH(vv.w, dd.w, ss.w, aa.w) }
                else if (MODE == EVAL_MODE_SAME) {
                    { float x = vv.x - ww.x; EVAL_ELEM(x, dd.x, ss.x, aa.x) }
                    { float x = vv.y - ww.y; EVAL_ELEM(x, dd.y, ss.y, aa.y) }
                    { float x = vv.z - ww.z; EVAL_ELEM(x, dd.z, ss.z, aa.z) }
                    { float x = vv.w - ww.w; EVAL_ELEM(x, dd.w, ss.w, aa.w) }
                } else { EVAL_ELEM(vv.x, dd.x, ss.x, aa.x) EVAL_ELEM(vv.y, dd.y, ss.y, aa.y) EVAL_ELEM(vv.z, dd.z, ss.z, aa.z) EVAL_ELEM(vv.w, dd.w, ss.w, aa.w) }
            }
            ps[r][sl] = a.s; pm[r][sl] = a.m;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) { ps[r][sl] = cg_wave_allsum_f32(ps[r][sl]); pm[r][sl] = cg_wave_allsum_f32(pm[r][sl]); }
    }
    if ((u & 63u) == 0u) {
        float *o = lds + (u >> 6) * (NC * 2);
#pragma unroll
        for (int r = 0; r < NR; ++r) { o[(2 * r) * 2 + 0] = ps[r][0]; o[(2 * r) * 2 + 1] = ps[r][1]; o[(2 * r + 1) * 2 + 0] = pm[r][0]; o[(2 * r + 1) * 2 + 1] = pm[r][1]; }
    }
}
// AP[:,row] += delta * other[:,col] over the half's two slots (eval_update_ap's operations); a second site continues from the first in
// registers when it is the same row (eval_update_ap2)
CG_DEVICE void eval_pair_update(const SamplerDev &S, const DecRec &d, const uint32_t u, const uint32_t H)
{
    const uint32_t nq = S.Npad >> 2;
    if (d.n == 0u) return;
    float *AP1 = S.AP + (size_t)d.r1 * S.Npad; const float *V1 = S.other + (size_t)d.c1 * S.Npad;
    const bool second = d.n == 2u, same = second && d.r1 == d.r2;
    float *AP2 = S.AP + (size_t)(second ? d.r2 : d.r1) * S.Npad; const float *V2 = S.other + (size_t)(second ? d.c2 : d.c1) * S.Npad;
    cg_f4 v1[2], p1[2], v2[2], p2[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint32_t j = u + (uint32_t)sl * H, jj = j < nq ? j : nq - 1u;
        v1[sl] = ld4(V1, jj); p1[sl] = ld4(AP1, jj); v2[sl] = ld4(V2, jj); p2[sl] = ld4(AP2, jj);
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint32_t j = u + (uint32_t)sl * H;
        if (j < nq) {
            cg_f4 q = p1[sl];
            q.x = q.x + d.d1 * v1[sl].x; q.y = q.y + d.d1 * v1[sl].y; q.z = q.z + d.d1 * v1[sl].z; q.w = q.w + d.d1 * v1[sl].w;
            if (same) { q.x = q.x + d.d2 * v2[sl].x; q.y = q.y + d.d2 * v2[sl].y; q.z = q.z + d.d2 * v2[sl].z; q.w = q.w + d.d2 * v2[sl].w; }
            st4(AP1, j, q);
            if (second && !same) {
                cg_f4 w = p2[sl];
                w.x = w.x + d.d2 * v2[sl].x; w.y = w.y + d.d2 * v2[sl].y; w.z = w.z + d.d2 * v2[sl].z; w.w = w.w + d.d2 * v2[sl].w;
                st4(AP2, j, w);
            }
        }
    }
}
// proposals qA (first half) and qB (second half) of the queue; every thread of the workgroup calls it
CG_DEVICE void eval_chain_pair(const SamplerDev &S, unsigned long long *grans, const uint32_t tag, const float T, const PropRec &pA, const PropRec &pB, const uint32_t qA, const uint32_t qB)
{
    CG_SHARED float ldsP[2][16 * 8];
    CG_SHARED DecRec decP[2];
    const uint32_t t = cg_tid(), H = cg_bdim() >> 1, h = t >= H ? 1u : 0u, u = t - h * H, nwH = H >> 6;
    const PropRec p = h ? pB : pA;
    const uint32_t q = h ? qB : qA;
    const uint32_t mm = GM_MATH_PORTABLE;
    const float lambda = S.lambda;
    const bool scalarLane = u < 64u;
    uint64_t rng = p.rng;
    const bool two = (p.type == 'M' || p.type == 'E');
    const float m1 = p.m1, m2 = p.m2, old1 = p.old1, old2 = p.old2;
    const bool gibbs1 = (p.gibbs & 1u) != 0u, gibbs2 = (p.gibbs & 2u) != 0u;
    const bool need = (p.type == 'B') ? gibbs1 : ((p.type == 'D' || p.type == 'M') ? true : (gibbs1 || gibbs2));
    const bool diff = two && p.r1 != p.r2;
    const uint32_t alphaUnits = need ? (!two ? 4u : (diff ? 8u : 5u)) : 0u;
    EvalSpec spec; spec.rng = rng; spec.mm = mm; spec.l1 = 0.f; spec.l2 = 0.f;
    spec.n = (scalarLane && need && (p.type == 'D' || p.type == 'M')) ? ((p.type == 'D' && gibbs1) ? 2u : 1u) : 0u;
    float *lds = &ldsP[h][0];
    if (need) {
        const uint32_t rowA[1] = {p.r1}, colA[1] = {p.c1};
        if (diff) { const uint32_t rowAB[2] = {p.r1, p.r2}, colAB[2] = {p.c1, p.c2}; eval_pair_alpha<2, EVAL_MODE_ONE>(S, rowAB, colAB, 0u, 0.f, u, H, lds, spec); }
        else if (p.type == 'D') eval_pair_alpha<1, EVAL_MODE_CH>(S, rowA, colA, 0u, -1.f * m1, u, H, lds, spec);
        else if (two) eval_pair_alpha<1, EVAL_MODE_SAME>(S, rowA, colA, p.c2, 0.f, u, H, lds, spec);
        else eval_pair_alpha<1, EVAL_MODE_ONE>(S, rowA, colA, 0u, 0.f, u, H, lds, spec);
    }
    cg_sync();                                  // barrier 1: the waves' partials are parked
    float s = 0.f, smu = 0.f;
    if (need) {
        if (diff) { float tot[4] = {0.f, 0.f, 0.f, 0.f}; eval_pair_fold<4>(lds, u, nwH, tot); s = tot[0] + tot[2]; smu = tot[1] - tot[3]; }      // AlphaParameters.cpp:11-14
        else { float t2[2] = {0.f, 0.f}; eval_pair_fold<2>(lds, u, nwH, t2); s = t2[0]; smu = t2[1]; }
        if (spec.n) eval_spec_run(spec);
    }
    s = s * T; smu = smu * T;
    DecRec owe; owe.n = 0; owe.r1 = 0; owe.c1 = 0; owe.d1 = 0.f; owe.r2 = 0; owe.c2 = 0; owe.d2 = 0.f; owe.pad = 0;
    if (scalarLane) {
        uint32_t code = CHAIN_NONE, nUpd = 0; float val = 0.f;
        if (p.type == 'B') {                                                   // AsynchronousGibbsSampler.h:127-144
            float bv = 0.f; uint32_t bhas = 0;
            if (gibbs1) { OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); bv = g.v; bhas = g.has ? 1u : 0u; }
            else { bv = pcg_exponential(rng, lambda, mm); bhas = 1u; }
            if (bhas != 0u && bv >= GAPS_EPSILON) { code = CHAIN_APPLY; val = bv; nUpd = 1u; owe.n = 1u; owe.r1 = p.r1; owe.c1 = p.c1; owe.d1 = bv; }
            else code = CHAIN_ERASE;
        } else if (p.type == 'D') {                                            // :148-180
            float rebirth = m1; bool drew = false;
            if (gibbs1) { OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); if (g.has) { rebirth = g.v; drew = true; } }
            const float deltaLL = rebirth * (smu - s * rebirth / 2.f);
            const float logU = drew ? spec.l2 : spec.l1;
            if (logU < deltaLL) {
                if (rebirth != m1) { const float nv = gm_max(old1 + (rebirth - m1), 0.f); code = CHAIN_APPLY; val = rebirth; nUpd = 1u; owe.n = 1u; owe.r1 = p.r1; owe.c1 = p.c1; owe.d1 = nv - old1; }
            } else { const float nv = gm_max(old1 + (-1.f * m1), 0.f); code = CHAIN_ERASE; nUpd = 1u; owe.n = 1u; owe.r1 = p.r1; owe.c1 = p.c1; owe.d1 = nv - old1; }
        } else if (p.type == 'M') {                                            // :184-196
            const float deltaLL = -1.f * m1 * (smu + s * m1 / 2.f);
            if (spec.l1 < deltaLL) { const float nv1 = gm_max(old1 + (-m1), 0.f); code = CHAIN_APPLY; nUpd = 2u; owe.n = 2u; owe.r1 = p.r1; owe.c1 = p.c1; owe.d1 = nv1 - old1; owe.r2 = p.r2; owe.c2 = p.c2; owe.d2 = m1; }
        } else if (need) {                                                     // exchange, :201-219
            OptF g0 = gm_gibbs_mass(s, smu, -m1, m2, rng, S.luts, false, 0.f);
            const float gv = g0.v, n1 = m1 + gv, n2 = m2 - gv;
            if (g0.has && n1 > GAPS_EPSILON && n2 > GAPS_EPSILON) {
                const float nv1 = gm_max(old1 + (n1 - m1), 0.f), nv2 = gm_max(old2 + (n2 - m2), 0.f);
                code = CHAIN_APPLY; val = gv; nUpd = 2u; owe.n = 2u; owe.r1 = p.r1; owe.c1 = p.c1; owe.d1 = nv1 - old1; owe.r2 = p.r2; owe.c2 = p.c2; owe.d2 = nv2 - old2;
            }
        }
        if (u == 0u) {
            unsigned long long *gr = grans + (size_t)q * CHAIN_GRAN_STRIDE; const uint32_t units = nUpd * 3u + alphaUnits;
            cg_store_agent_u64(&gr[0], ((unsigned long long)tag << 32) | (unsigned long long)(code | (units << 8)));
            cg_store_agent_u64(&gr[1], ((unsigned long long)tag << 32) | (unsigned long long)gm_f2u(val));
            decP[h] = owe;
        }
    }
    cg_sync();                                  // barrier 2: the decisions are in LDS
    const DecRec d = decP[h];
    eval_pair_update(S, d, u, H);
    cg_sync();                                  // barrier 3: the scratch is free again
}

#if defined(GEN_PROFILE) && !defined(GEN_SUBMARKS) && !defined(GEN_ROUNDMARKS)
#define EVAL_PROF(i) do { if (t == 0 && cg_bid() == 0) { unsigned long long now_ = cg_clock(); cg_atomic_add_u64(&S.gs->prof[8 + (i)], now_ - eprof_last); eprof_last = now_; } } while (0)
#else
#define EVAL_PROF(i) do { } while (0)
#endif

#define EVAL_FUSED 0     // one workgroup per proposal: alpha, decision, update
#define EVAL_ALPHA 1     // split evaluation, first kernel: `slices` workgroups per proposal, per-slice alpha partials
#define EVAL_APPLY 2     // split evaluation, second kernel: combine the partials, decide, update the slice
#define EVAL_SEQ 3       // verification mode: one workgroup per proposal, sums in the reference's scalar order, session math mode
#define EVAL_CHAIN 5     // the fused form inside the chained launch (chain_kernel.h): the decision goes to the next batch's generator workgroup of the SAME launch as
                         // two tagged granules and that workgroup applies it to the domain and the matrix; this workgroup only updates its A*P row(s)
#define EVAL_CHAIN_SPLIT 6   // the split evaluation inside the chained launch (round 5): slices as EVAL_DECIDE, the deciding workgroup hands the decision to the launch's
                             // generator workgroup as EVAL_CHAIN does, and every evaluation workgroup, done with its slices, takes a share of the A*P updates the
                             // decisions owe (eval_chain_updates) -- beside the generator's round, complete at the launch's end
#define EVAL_DECIDE 4    // split evaluation in ONE launch (one-chain form): the slices' workgroups hand their totals to the proposal's last slice
                         // workgroup, which decides and records what the A*P cache owes (DecRec); eval_apply_items carries that out beside the
                         // NEXT generator launch (gen_apply_kernel), off the generate -> decide -> generate chain

// The split form serves data vectors of more than 4096 elements (W > 1024 virtual lanes): one workgroup can pull a
// row no faster than its compute unit's ~64 B/clk, so the row is cut into slices of 1024 chunks, one workgroup
// each.  Slice j owns virtual lanes j*1024 .. j*1024+1023; the per-slice totals go through S.partials
// ([queueCap][4][16]) and every workgroup of the second kernel folds them in the same ascending order (the top
// bits of the butterfly), repeats the (deterministic) decision and updates its own slice of AP.
// vbid / vgdim: this workgroup's index and the number of workgroups that serve THIS sampler's queue (the grid itself, or
// one chain's share of a batched multi-chain launch)
// hot: the three values a workgroup's first memory trip needs, passed as leading scalar kernel arguments so that the dispatcher preloads
// them into SGPRs (-amdgpu-kernarg-preload-count): the queue record is requested at once, the lines of the sampler's record come in
// under the same trip (eval_first / eval_record).
struct EvalHot { const PropRec *queue; const GenScalars *gs; uint32_t queueCap; const ChainSlot *slot; unsigned long long *grans; };      // slot, grans: chained launch only
// The first memory trip of an evaluation workgroup: its first queue record, the queue length, the annealing temperature.  The
// addresses need only `hot` and the workgroup index, so the one-chain kernels issue it before they have seen the sampler's record.
struct EvalFirst { PropRec p; uint32_t qlen; float T; uint32_t tag; };      // tag: the batch's number (low word), what the in-launch hand-off marks its granules with
template <int PHASE>
CG_DEVICE EvalFirst eval_first(const EvalHot hot, uint32_t slices, uint32_t vbid)
{
    const uint32_t qFirst = (PHASE == EVAL_FUSED || PHASE == EVAL_SEQ || PHASE == EVAL_CHAIN) ? vbid : vbid / slices;
    EvalFirst f; f.p = hot.queue[qFirst < hot.queueCap ? qFirst : 0u]; f.T = hot.gs->annealTemp;
    if (PHASE == EVAL_CHAIN || PHASE == EVAL_CHAIN_SPLIT) { const ChainSlot cs = *hot.slot; f.qlen = cs.qlen; f.tag = cs.tag; }      // (this launch's parity: nothing in this launch writes it)
    else { f.qlen = hot.gs->qlen; f.tag = (PHASE == EVAL_DECIDE) ? (uint32_t)hot.gs->batchEpoch : 0u; }
    return f;
}
// The one-chain kernels read the sampler's record through a pointer in the constant address space (scalar loads), requested behind
// the first trip and fenced (platform.h, cg_const_warm_begin / _end) -- as the generator does.  Taken by value the 550-byte record was
// loaded at kernel entry, group by group, and with everything the compiler derives from it kept 40-280 scalar registers spilled in
// vector lanes for the whole kernel (the batched kernels, which always went through a pointer, spill 0-18).
template <int PHASE>
CG_DEVICE const SamplerDev &eval_record(const SamplerDev CG_CONSTANT *sp)
{
    cg_sched_fence();
    cg_const_lines lines;
    cg_const_warm_begin<sizeof(SamplerDev)>(sp, lines);
    sp = cg_const_warm_end(sp, lines);
    return *(const SamplerDev *)sp;
}
// ---- chained launch, split evaluation: the A*P updates the batch's decisions owe ------------------------------------------------------
// Every evaluation workgroup, done with its slices, takes items (proposal, share of its row(s)): it reads the proposal's decision where
// the generator workgroup reads it -- the two tagged granules its deciding workgroup published (waiting for them if need be: a deciding
// workgroup never waits for an update, so the wait ends) -- and derives what the A*P cache is owed from the queue record exactly as the
// decision did (AsynchronousGibbsSampler.h:127-219 with safelyChangeMatrix's clamp, DenseNormalModel.cpp:110-123); then
// AP += delta * other, element by element, eval_apply_items' operations in its order.  The launch's end completes them: the next
// launch's evaluation reads the rows.
CG_DEVICE void eval_chain_updates(const SamplerDev &S, const EvalHot hot, const uint32_t tag, const uint32_t qlen, const uint32_t wg, const uint32_t nwg)
{
    const uint32_t TPB = cg_bdim(), nq = S.Npad >> 2;
    uint32_t parts = (nq + 4u * TPB - 1u) / (4u * TPB);
    parts = parts < 1u ? 1u : (parts > 64u ? 64u : parts);
    for (uint32_t item = wg; item < qlen * parts; item += nwg) {
        const uint32_t q = item / parts, part = item - q * parts;
        const PropRec p = hot.queue[q < hot.queueCap ? q : 0u];
        const unsigned long long *gr = hot.grans + (size_t)q * CHAIN_GRAN_STRIDE;
        unsigned long long g0 = 0ull, g1 = 0ull; uint32_t spins = 0; bool have = true;
        for (;;) {
            g0 = cg_load_l2_u64(&gr[0]); g1 = cg_load_l2_u64(&gr[1]);
            const bool ok = (uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag;
            if (cg_ballot(!ok) == 0ull) break;
            if (cg_poll_expired(++spins)) { if (cg_tid() == 0u) S.gs->error = GAPS_ERR_SPIN; have = false; break; }      // (bounded; nothing is applied from a stale granule)
            cg_poll_pause();
        }
        if (!have) return;
        const uint32_t code = (uint32_t)g0 & 0xFFu; const float val = gm_u2f((uint32_t)g1);
        const float m1 = p.m1, m2 = p.m2, old1 = p.old1, old2 = p.old2;
        if (code == CHAIN_APPLY) {
            if (p.type == 'B') eval_update_ap(S, p.r1, p.c1, val, part * TPB, TPB * parts);
            else if (p.type == 'D') { const float nv = gm_max(old1 + (val - m1), 0.f); eval_update_ap(S, p.r1, p.c1, nv - old1, part * TPB, TPB * parts); }
            else if (p.type == 'M') { const float nv1 = gm_max(old1 + (-m1), 0.f); eval_update_ap2(S, p.r1, p.c1, nv1 - old1, p.r2, p.c2, m1, part * TPB, TPB * parts); }
            else if (p.type == 'E') {
                const float n1 = m1 + val, n2 = m2 - val;
                const float nv1 = gm_max(old1 + (n1 - m1), 0.f), nv2 = gm_max(old2 + (n2 - m2), 0.f);
                eval_update_ap2(S, p.r1, p.c1, nv1 - old1, p.r2, p.c2, nv2 - old2, part * TPB, TPB * parts);
            }
        } else if (code == CHAIN_ERASE && p.type == 'D') { const float nv = gm_max(old1 + (-1.f * m1), 0.f); eval_update_ap(S, p.r1, p.c1, nv - old1, part * TPB, TPB * parts); }
    }
}

// SINGLE: a one-chain launch (the fused form then asks for the chunk it will rewrite before the scalar step, see EvalPre)
template <int PHASE, bool SINGLE>
CG_DEVICE void eval_body(const SamplerDev &S, uint32_t slices, const uint32_t vbid, const uint32_t vgdim, const EvalHot hot, const EvalFirst &first)
{
#if defined(GEN_TIMELINE)
    unsigned long long ets[11]; uint32_t ets_n = 0;
#endif
    constexpr bool CHAINP = PHASE == EVAL_CHAIN_SPLIT;
    constexpr bool CHAIN = PHASE == EVAL_CHAIN || CHAINP;       // the decision goes to the launch's generator workgroup; nothing but A*P rows is written here
    constexpr bool FUSEDF = PHASE == EVAL_FUSED || PHASE == EVAL_CHAIN;      // the fused form: one workgroup reduces, decides and updates
    const uint32_t qFirst = (FUSEDF || PHASE == EVAL_SEQ) ? vbid : vbid / slices;
    PropRec pNext = first.p;
    const uint32_t qlen = first.qlen;
    const float T = first.T;
    CG_SHARED float lds[16 * 4];
    CG_SHARED float decf; CG_SHARED uint32_t deci;     // decision of wave 0, broadcast to the other waves

    CG_SHARED float seqTerm[PHASE == EVAL_SEQ ? 4 * 4 * EVAL_SEQ_BS : 1];
    constexpr bool WHOLE = FUSEDF || PHASE == EVAL_SEQ;      // one workgroup owns the whole proposal
    const uint32_t mm = (PHASE == EVAL_SEQ) ? S.mathMode : GM_MATH_PORTABLE;
    const uint32_t t = cg_tid(), BS = cg_bdim();
#if defined(GEN_PROFILE) && !defined(GEN_SUBMARKS) && !defined(GEN_ROUNDMARKS)
    unsigned long long eprof_last = cg_clock();      // (only the profile build reads the clock: the read is not free and cannot be dropped by the compiler)
#endif
    const float lambda = S.lambda;
    EVAL_TS(0);
    // (every slice workgroup of the split form repeats the scalar step and its two dependent table lookups; touching the tables' lines at
    // entry, which paid in round 1, costs 0.7 % since the launch prologues were shortened and is gone)
    const bool multiWave = BS > 64u;
    const bool scalarLane = !multiWave || t < 64u;       // the per-proposal scalar math (LUTs, fp64 log) runs in wave 0 only
    const uint32_t slice = WHOLE ? 0u : vbid % slices;
    const uint32_t qStep = WHOLE ? vgdim : vgdim / slices;
    const uint32_t chunk0 = slice * BS, stride = WHOLE ? BS : S.redW;
    constexpr bool DECIDE = PHASE == EVAL_DECIDE || CHAINP;
    const bool decider = DECIDE && slice + 1u == slices;      // the proposal's last slice: it is dispatched last, so its siblings are on the machine when it waits for their totals
    const bool writer = !CHAIN && (DECIDE ? decider : slice == 0u) && t == 0u;          // the one thread that stores the proposal's scalar results (chained launch: nobody here)
    // chained launch: thread 0 hands the decision to the generator workgroup of this launch the moment it is made, before the broadcast
    // and the A*P update -- {code | units << 8, tag} and {value, tag}, one write-through store each (gaps_state.h, CHAIN_*)
#if defined(GEN_TIMELINE)
#define EVAL_PUBLISH_RT() do { if (CHAIN && t == 0u && q < 255u && qlen >= 140u && hot.gs->nSteps - hot.gs->nDone >= 512u) g_chain_rt[q * 4u + 1u] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EVAL_PUBLISH_RT() do { } while (0)
#endif
#define EVAL_PUBLISH(CODE, VAL, NUPD) do { EVAL_TS(7); EVAL_PUBLISH_RT(); if (CHAIN && t == 0u) { unsigned long long *gr_ = hot.grans + (size_t)q * CHAIN_GRAN_STRIDE; const uint32_t units_ = (NUPD) * 3u + alphaUnits; \
        cg_store_agent_u64(&gr_[0], ((unsigned long long)first.tag << 32) | (unsigned long long)((CODE) | (units_ << 8))); \
        cg_store_agent_u64(&gr_[1], ((unsigned long long)first.tag << 32) | (unsigned long long)gm_f2u(VAL)); } } while (0)
    // (the deciding workgroup's other waves have nothing to do with the decision: no broadcast)
#define EVAL_BCAST(F0, I0) do { if (multiWave && !DECIDE) { if (t == 0) { decf = (F0); deci = (I0); } cg_sync(); (F0) = decf; (I0) = deci; } } while (0)
    if (PHASE == EVAL_DECIDE && vbid == 0u && t == 0u) S.gs->applyCount = qlen;      // what the next generator launch's update workgroups will find in S.dec (0: this launch found no queue)
    for (uint32_t q = qFirst; ; q += qStep) {
        // one trip: the record (slot q always exists: q < queueCap), the queue length, the annealing temperature
        const PropRec p = pNext;
        if (q >= qlen) break;
        if (PHASE == EVAL_CHAIN && q + qStep < qlen && (S.Npad >> 2) <= BS && BS >= 128u) {
            // a queue longer than the launch has evaluation workgroups: this workgroup's next two proposals side by side (eval_chain_pair)
            const uint32_t qB = q + qStep;
            const PropRec pB = hot.queue[qB < hot.queueCap ? qB : 0u];
            const uint32_t qN = qB + qStep;
            if (qN < qlen) pNext = hot.queue[qN < hot.queueCap ? qN : 0u];
#if defined(COGAPS_EMUL)
            if (t == 0u) cg_atomic_add_u64(&S.gs->prof[7], 1ull);      // test-only build: proposals evaluated in pairs
#endif
            eval_chain_pair(S, hot.grans, first.tag, T, p, pB, q, qB);
            if (qN >= qlen) break;
            q = qB;      // (the loop's step adds the second)
            continue;
        }
        { uint32_t ty_ = p.type; EVAL_PIN(ty_); }
        EVAL_TS(1);
        uint64_t rng = p.rng; uint32_t nUpd = 0;
#if defined(EVAL_ATOMS_LATE)
        EvalAtoms ea = eval_atoms_load(S, p, writer && PHASE == EVAL_SEQ);      // dev A/B: the fused form asks for the writer's atom record behind the reduction (below), with the update's chunks
#else
        EvalAtoms ea = eval_atoms_load(S, p, writer && (PHASE == EVAL_FUSED || PHASE == EVAL_SEQ));      // (chained launch: writer is false, the generator fetches them)
#endif      // (split evaluation: requested after the slices' totals, below -- asked for at once by the one-launch form's decider it cost 0.6 us: 7.2 -> 7.8 us per launch, the rows' wait then includes it)
        bool eaLoaded = false;
        const bool two = (p.type == 'M' || p.type == 'E');
        const float m1 = p.m1, m2 = p.m2, old1 = p.old1, old2 = p.old2;
        const bool gibbs1 = (p.gibbs & 1u) != 0u, gibbs2 = (p.gibbs & 2u) != 0u;
        EVAL_PROF(0);
        EVAL_TS(2);
#if defined(GEN_PROFILE)
        if (S.dbg & 1u) { if (p.type == 0xFFu || m1 == -1.f) S.queueUnits[q] = (uint32_t)old1; break; }   // record only
#endif
        // ---------------------------------------------------------------- alpha parameters (DenseNormalModel.cpp:161-240)
        // which reduction the step needs: birth with Gibbs, death, move, exchange with canUseGibbs(c1,c2)
        bool need = (p.type == 'B') ? gibbs1 : ((p.type == 'D' || p.type == 'M') ? true : (gibbs1 || gibbs2));
#if defined(GEN_PROFILE)
        if (S.dbg & 16u) need = false;     // timing experiment: no reduction at all
#endif
        const bool diff = two && p.r1 != p.r2;
        const uint32_t alphaUnits = need ? (!two ? 4u : (diff ? 8u : 5u)) : 0u;      // roofline bookkeeping, units of 4N bytes (below)
        float s = 0.f, smu = 0.f;          // un-annealed sums, valid in wave 0
        uint32_t typeX = p.type;           // the step that is decided below (0: none -- a hand-over inside the launch that never arrived)
        // the one-chain fused launch only: the batched one is throughput bound and at its register budget, and the split form's APPLY
        // launch got slower with it (10.3 -> 12 us: 80 KB rows fetched for every rejected proposal, registers at the launch bound)
        constexpr bool PRE = FUSEDF && SINGLE;
        EvalPre pre; pre.v1 = f4_zero(); pre.p1 = f4_zero(); pre.v2 = f4_zero(); pre.p2 = f4_zero();
        const uint32_t jPre = chunk0 + t;
#define EVAL_PREFETCH() do { if (PRE && jPre < (S.Npad >> 2)) { \
            pre.v1 = ld4(S.other + (size_t)p.c1 * S.Npad, jPre); pre.p1 = ld4(S.AP + (size_t)p.r1 * S.Npad, jPre); \
            if (two) { pre.v2 = ld4(S.other + (size_t)p.c2 * S.Npad, jPre); if (p.r1 != p.r2) pre.p2 = ld4(S.AP + (size_t)p.r2 * S.Npad, jPre); } } } while (0)
        // the A*P update(s) an accepted step owes: carried out here -- or, in the one-launch split form, recorded for eval_apply_items
        DecRec owe; owe.n = 0; owe.r1 = 0; owe.c1 = 0; owe.d1 = 0.f; owe.r2 = 0; owe.c2 = 0; owe.d2 = 0.f; owe.pad = 0;
#define EVAL_UPD1(ROW, COL, DELTA) do { if (DECIDE) { owe.n = 1u; owe.r1 = (ROW); owe.c1 = (COL); owe.d1 = (DELTA); } \
            else if (PRE) eval_update_pre1(S, (ROW), (COL), (DELTA), chunk0, stride, pre.v1, pre.p1); else eval_update_ap(S, (ROW), (COL), (DELTA), chunk0, stride); } while (0)
#define EVAL_UPD2(R1, C1, D1, R2, C2, D2) do { if (DECIDE) { owe.n = 2u; owe.r1 = (R1); owe.c1 = (C1); owe.d1 = (D1); owe.r2 = (R2); owe.c2 = (C2); owe.d2 = (D2); } \
            else if (PRE) eval_update_pre2(S, (R1), (C1), (D1), (R2), (C2), (D2), chunk0, stride, pre); else eval_update_ap2(S, (R1), (C1), (D1), (R2), (C2), (D2), chunk0, stride); } while (0)
        constexpr bool AHEAD = (FUSEDF && SINGLE) || DECIDE;       // (the batched fused kernel is at its register budget)
        EvalSpec spec; spec.rng = rng; spec.mm = mm; spec.l1 = 0.f; spec.l2 = 0.f;
        spec.n = (AHEAD && scalarLane && need && (p.type == 'D' || p.type == 'M')) ? ((p.type == 'D' && gibbs1) ? 2u : 1u) : 0u;
        if (need) {
            float tot[4] = {0.f, 0.f, 0.f, 0.f};
            if (PHASE == EVAL_SEQ) {
                const uint32_t rowA[1] = {p.r1}, colA[1] = {p.c1};
                if (diff) {
                    const uint32_t rowAB[2] = {p.r1, p.r2}, colAB[2] = {p.c1, p.c2};
                    eval_alpha_seq<2, EVAL_MODE_ONE>(S, rowAB, colAB, 0u, 0.f, seqTerm, lds, tot);
                } else {
                    float t2[2] = {0.f, 0.f};
                    if (p.type == 'D') eval_alpha_seq<1, EVAL_MODE_CH>(S, rowA, colA, 0u, -1.f * m1, seqTerm, lds, t2);
                    else if (two) eval_alpha_seq<1, EVAL_MODE_SAME>(S, rowA, colA, p.c2, 0.f, seqTerm, lds, t2);
                    else eval_alpha_seq<1, EVAL_MODE_ONE>(S, rowA, colA, 0u, 0.f, seqTerm, lds, t2);
                    tot[0] = t2[0]; tot[1] = t2[1];
                }
            } else if (PHASE != EVAL_APPLY) {
                const uint32_t rowA[1] = {p.r1}, colA[1] = {p.c1};
                if (diff) {
                    const uint32_t rowAB[2] = {p.r1, p.r2}, colAB[2] = {p.c1, p.c2};
                    eval_alpha<2, EVAL_MODE_ONE>(S, rowAB, colAB, 0u, 0.f, chunk0, stride, lds, tot, spec EVAL_TS_ARGS);
                } else {
                    float t2[2] = {0.f, 0.f};
                    if (p.type == 'D') eval_alpha<1, EVAL_MODE_CH>(S, rowA, colA, 0u, -1.f * m1, chunk0, stride, lds, t2, spec EVAL_TS_ARGS);
                    else if (two) eval_alpha<1, EVAL_MODE_SAME>(S, rowA, colA, p.c2, 0.f, chunk0, stride, lds, t2, spec EVAL_TS_ARGS);
                    else eval_alpha<1, EVAL_MODE_ONE>(S, rowA, colA, 0u, 0.f, chunk0, stride, lds, t2, spec EVAL_TS_ARGS);
                    tot[0] = t2[0]; tot[1] = t2[1];
                }
                if (PHASE == EVAL_ALPHA) { if (t == 0) { float *o = S.partials + (size_t)q * 64u + slice; o[0] = tot[0]; o[16] = tot[1]; o[32] = tot[2]; o[48] = tot[3]; } }
                if (DECIDE && t < 64u) {
                    // The slices' totals go to the proposal's deciding workgroup inside this launch: four {value, batch tag} granules per slice,
                    // one write-through store each; the decider (wave 0: lane i holds element (component i / 16, slice i % 16), as the two-launch
                    // form's record does) reads them past its caches until every tag is this batch's, then folds the slices in the same
                    // ascending order (the top bits of the butterfly; slots past the last slice hold +0).
                    unsigned long long *gr = S.grans + (size_t)q * 64u;
                    if (!decider) {
                        if (t < 4u) { const float v = t == 0u ? tot[0] : (t == 1u ? tot[1] : (t == 2u ? tot[2] : tot[3])); cg_store_agent_u64(&gr[slice * 4u + t], ((unsigned long long)first.tag << 32) | (unsigned long long)gm_f2u(v)); }
                    } else {
                        // (the writer's atom record: asked for here, it arrives under the wait for the siblings; asked for with the rows it
                        // held the reduction up -- 7.2 -> 7.8 us per launch --, asked for after the wait the decision waits for it)
                        ea = eval_atoms_load(S, p, writer); eaLoaded = true;
                        const uint32_t comp = t >> 4, sl = t & 15u;
                        const bool want = sl + 1u < slices;
                        unsigned long long g = 0ull; uint32_t spins = 0;
                        for (;;) {
                            if (want) g = cg_load_l2_u64(&gr[sl * 4u + comp]);
                            const bool ok = !want || (uint32_t)(g >> 32) == first.tag;
                            if (cg_ballot(!ok) == 0ull) break;
                            // bounded (platform.h: two seconds at least), never a hang; a bound that is hit decides NOTHING from stale totals: the
                            // proposal is dropped (typeX: no branch below matches, nothing is stored), the error word ends the update on the host
                            if (cg_poll_expired(++spins)) { if (t == 0u) S.gs->error = GAPS_ERR_SPIN; typeX = 0u; break; }
                            cg_poll_pause();
                        }
                        const float own = comp == 0u ? tot[0] : (comp == 1u ? tot[1] : (comp == 2u ? tot[2] : tot[3]));
                        float z = want ? gm_u2f((uint32_t)g) : (sl + 1u == slices ? own : 0.f);
                        const uint32_t slots = S.redW / BS;
                        for (uint32_t st = 1; st < slots; st <<= 1) z = z + cg_shfl_xor_f32(z, (int)st);
                        tot[0] = cg_lane_read_f32(z, 0); tot[1] = cg_lane_read_f32(z, 16); tot[2] = cg_lane_read_f32(z, 32); tot[3] = cg_lane_read_f32(z, 48);
                    }
                }
            } else {
                // fold the slices: the top bits of the butterfly (slots past the last slice hold +0, as the empty lanes
                // do).  Wave 0 only: lane i holds element (component i / 16, slice i % 16) of the 64-float record.
                if (t < 64u) {
                    const uint32_t slots = S.redW / BS;
                    float z = ((t & 15u) < slices) ? S.partials[(size_t)q * 64u + t] : 0.f;
                    for (uint32_t st = 1; st < slots; st <<= 1) z = z + cg_shfl_xor_f32(z, (int)st);
                    tot[0] = cg_lane_read_f32(z, 0); tot[1] = cg_lane_read_f32(z, 16); tot[2] = cg_lane_read_f32(z, 32); tot[3] = cg_lane_read_f32(z, 48);
                }
            }
            s = diff ? tot[0] + tot[2] : tot[0]; smu = diff ? tot[1] - tot[3] : tot[1];        // AlphaParameters.cpp:11-14
            if (AHEAD && spec.n) eval_spec_run(spec);      // (a lane of wave 0 without a chunk: vectors shorter than 256 elements)
        }
        EVAL_PIN(s); EVAL_TS(3);
        if (FUSEDF) EVAL_PREFETCH();                    // (after the reduction: the registers are free again)
#if defined(EVAL_ATOMS_LATE)
        if (PHASE == EVAL_FUSED) ea = eval_atoms_load(S, p, writer);
#endif
        if (PHASE == EVAL_APPLY || (DECIDE && !eaLoaded)) ea = eval_atoms_load(S, p, writer);
        if (PHASE == EVAL_ALPHA || (DECIDE && !decider)) { if (q + qStep >= qlen) break; { const uint32_t qn_ = q + qStep; pNext = hot.queue[qn_ < hot.queueCap ? qn_ : 0u]; } cg_sync(); continue; }
        s = s * T; smu = smu * T;
#if defined(GEN_PROFILE)
        if (S.dbg & 8u) { if (q + qStep >= qlen) break; { const uint32_t qn_ = q + qStep; pNext = hot.queue[qn_ < hot.queueCap ? qn_ : 0u]; } cg_sync(); continue; }    // timing experiment: stop before the scalar step
#endif
        if (typeX == 'B') {
            // ---------------------------------------------------------------- birth (:127-144)
            float bv = 0.f; uint32_t bhas = 0;
            if (scalarLane) {
                if (gibbs1) { OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda); bv = g.v; bhas = g.has ? 1u : 0u; }
                else { bv = pcg_exponential(rng, lambda, mm); bhas = 1u; }
                if (bhas != 0u && bv >= GAPS_EPSILON) EVAL_PUBLISH(CHAIN_APPLY, bv, 1u); else EVAL_PUBLISH(CHAIN_ERASE, 0.f, 0u);
            }
            EVAL_PIN(bv); EVAL_TS(4);
            EVAL_BCAST(bv, bhas);
            EVAL_TS(5);
            if (bhas != 0u && bv >= GAPS_EPSILON) {
                EVAL_UPD1(p.r1, p.c1, bv);
                ++nUpd;                          // changeMatrix
                if (writer) { atom_set_mass(S, p.h1, ea.a1.left, bv); eval_store_matrix(S, p.r1, p.c1, old1, old1 + bv); }
            } else if (writer) eval_cache_erase(S, p.h1, p.r1, p.c1);
        } else if (typeX == 'D') {
            // ---------------------------------------------------------------- death / rebirth (:148-180)
            float rebirth = m1;
            EVAL_PROF(1);
            uint32_t acc = 0;
            if (scalarLane) {
                bool drew = false;
                if (gibbs1) {
                    OptF g = gm_gibbs_mass(s, smu, 0.f, S.maxGibbsMass, rng, S.luts, true, lambda);
                    if (g.has) { rebirth = g.v; drew = true; }
                }
                const float deltaLL = rebirth * (smu - s * rebirth / 2.f);
                const float logU = (AHEAD && need) ? (drew ? spec.l2 : spec.l1) : gm_logf_m(pcg_uniform(rng), mm);      // (need is always true here; a dev build that switches the reduction off -- dbg & 16 -- never ran the look-ahead)
                acc = (logU < deltaLL) ? 1u : 0u;
                if (acc != 0u) { if (rebirth != m1) EVAL_PUBLISH(CHAIN_APPLY, rebirth, 1u); else EVAL_PUBLISH(CHAIN_NONE, 0.f, 0u); } else EVAL_PUBLISH(CHAIN_ERASE, 0.f, 1u);
            }
            EVAL_PIN(acc); EVAL_TS(4);
            EVAL_BCAST(rebirth, acc);
            EVAL_TS(5);
            EVAL_PROF(2);
            if (acc != 0u) {
                if (rebirth != m1) {
                    const float nv = gm_max(old1 + (rebirth - m1), 0.f);            // safelyChangeMatrix
                    EVAL_UPD1(p.r1, p.c1, nv - old1);
                    ++nUpd;
                    if (writer) { eval_store_matrix(S, p.r1, p.c1, old1, nv); atom_set_mass(S, p.h1, ea.a1.left, rebirth); }
                }
            } else {
                const float nv = gm_max(old1 + (-1.f * m1), 0.f);
                EVAL_UPD1(p.r1, p.c1, nv - old1);
                ++nUpd;
                if (writer) { eval_store_matrix(S, p.r1, p.c1, old1, nv); eval_cache_erase(S, p.h1, p.r1, p.c1); }
            }
            EVAL_PROF(3);
        } else if (typeX == 'M') {
            // ---------------------------------------------------------------- move (:184-196)
            uint32_t acc = 0; float unused = 0.f;
            if (scalarLane) { const float deltaLL = -1.f * m1 * (smu + s * m1 / 2.f); const float logU = (AHEAD && need) ? spec.l1 : gm_logf_m(pcg_uniform(rng), mm); acc = (logU < deltaLL) ? 1u : 0u;
                              if (acc) EVAL_PUBLISH(CHAIN_APPLY, 0.f, 2u); else EVAL_PUBLISH(CHAIN_NONE, 0.f, 0u); }
            EVAL_PIN(acc); EVAL_TS(4);
            EVAL_BCAST(unused, acc);
            EVAL_TS(5);
            if (acc) {
                const float nv1 = gm_max(old1 + (-m1), 0.f);                    // safelyChangeMatrix(r1,c1,-m)
                EVAL_UPD2(p.r1, p.c1, nv1 - old1, p.r2, p.c2, m1);
                nUpd += 2;      // ... then changeMatrix(r2,c2,+m); same thread owns the same elements
                if (writer) {
                    eval_domain_move(S, p, ea.a1);
                    eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                    eval_store_matrix(S, p.r2, p.c2, old2, old2 + m1);
                }
            }
        } else if (need && typeX == 'E') {
            // ---------------------------------------------------------------- exchange (:201-219)
            float gv = 0.f; uint32_t gh = 0;
            if (scalarLane) { OptF g0 = gm_gibbs_mass(s, smu, -m1, m2, rng, S.luts, false, 0.f); gv = g0.v; gh = g0.has ? 1u : 0u;
                              if (gh != 0u && m1 + gv > GAPS_EPSILON && m2 - gv > GAPS_EPSILON) EVAL_PUBLISH(CHAIN_APPLY, gv, 2u); else EVAL_PUBLISH(CHAIN_NONE, 0.f, 0u); }
            EVAL_PIN(gv); EVAL_TS(4);
            EVAL_BCAST(gv, gh);
            EVAL_TS(5);
            const float n1 = m1 + gv, n2 = m2 - gv;
            if (gh != 0u && n1 > GAPS_EPSILON && n2 > GAPS_EPSILON) {
                const float nv1 = gm_max(old1 + (n1 - m1), 0.f);
                const float nv2 = gm_max(old2 + (n2 - m2), 0.f);
                EVAL_UPD2(p.r1, p.c1, nv1 - old1, p.r2, p.c2, nv2 - old2);
                nUpd += 2;
                if (writer) {
                    eval_store_matrix(S, p.r1, p.c1, old1, nv1);
                    eval_store_matrix(S, p.r2, p.c2, old2, nv2);
                    atom_set_mass(S, p.h1, ea.a1.left, n1); atom_set_mass(S, p.h2, ea.left2, n2);
                }
            }
        }
        else EVAL_PUBLISH(CHAIN_NONE, 0.f, 0u);      // an exchange that cannot use Gibbs (:201-206): nothing happens, the generator still waits for its word
        EVAL_TS(6);
        EVAL_TS_DUMP(p.type | (nUpd << 8) | ((p.r1 == p.r2 ? 1u : 0u) << 16));
        if (DECIDE && writer) S.dec[q] = owe;
        if (writer) {   // roofline bookkeeping: algorithmic traffic of this proposal in units of 4N bytes
            // (alpha: 4 one-site, 5 two-site same row, 8 different rows; 3 per AP update); the generator sums the slots
            uint32_t units = nUpd * 3u;
            if (need) units += !two ? 4u : (diff ? 8u : 5u);
            S.queueUnits[q] = units;
        }
        if (q + qStep >= qlen) break;   // last proposal of this workgroup
        { const uint32_t qn_ = q + qStep; pNext = hot.queue[qn_ < hot.queueCap ? qn_ : 0u]; }
        cg_sync();   // the LDS scratch is reused by the next proposal
    }
#if !defined(COGAPS_EMUL)
    // (the test-only emulator runs workgroups one after the other: there the updates are a launch of their own behind this one, chain_updates_kernel)
    if (CHAINP) eval_chain_updates(S, hot, first.tag, qlen, vbid, vgdim);
#endif
}

// the split kernels are built for two resident 1024-thread workgroups per compute unit (<= 64 VGPRs)
template <int PHASE>
#ifndef EVAL_APPLY_WAVES
#define EVAL_APPLY_WAVES 6       // 70 VGPRs, nothing spilled (8 waves: 64 VGPRs and 12-20 bytes of scratch per lane; split evaluation 10.85 -> 10.40 us)
#endif
CG_KERNEL void CG_LAUNCH_BOUNDS2((PHASE == EVAL_SEQ ? EVAL_SEQ_BS : 1024), (PHASE == EVAL_FUSED || PHASE == EVAL_SEQ ? 4 : (PHASE == EVAL_APPLY || PHASE == EVAL_DECIDE ? EVAL_APPLY_WAVES : 8))) eval_kernel(const PropRec *hotQueue, const GenScalars *hotGs, uint32_t hotCap, uint32_t slices, const SamplerDev CG_CONSTANT *sp)
{
    EvalHot hot; hot.queue = hotQueue; hot.gs = hotGs; hot.queueCap = hotCap; hot.slot = nullptr; hot.grans = nullptr;
    const EvalFirst first = eval_first<PHASE>(hot, slices, cg_bid());
    const SamplerDev &S = eval_record<PHASE>(sp);
    eval_body<PHASE, true>(S, slices, cg_bid(), cg_gdim(), hot, first);
}

// Batched multi-chain launch: the samplers of C independent chains (GWCoGAPS / scCoGAPS shards on one GPU) stepped in lock-step by
// one stream.  `arr` is a device array of their SamplerDev records read through the constant address space (scalar loads, as the
// by-value kernel argument of the one-chain kernels is); workgroups [c * wgPerChain, (c + 1) * wgPerChain) serve chain c's queue.
template <int PHASE>
#ifndef EVAL_MULTI_FUSED_WAVES
#define EVAL_MULTI_FUSED_WAVES 6      // 80 VGPRs: three 512-thread workgroups per compute unit instead of two (8 chains: +3.5 %); 8 waves would spill
#endif
CG_KERNEL void CG_LAUNCH_BOUNDS2(1024, (PHASE == EVAL_FUSED ? EVAL_MULTI_FUSED_WAVES : (PHASE == EVAL_APPLY ? EVAL_APPLY_WAVES : 8))) eval_kernel_multi(const SamplerDev CG_CONSTANT *arr, uint32_t slices, uint32_t wgPerChain)
{
    const uint32_t chain = cg_bid() / wgPerChain;
    const SamplerDev CG_CONSTANT *sp = arr + chain;
    cg_const_warm<sizeof(SamplerDev)>(sp);
    const SamplerDev &S = *(const SamplerDev *)sp;
    EvalHot hot; hot.queue = S.queue; hot.gs = S.gs; hot.queueCap = S.queueCap; hot.slot = nullptr; hot.grans = nullptr;
    const uint32_t vbid = cg_bid() - chain * wgPerChain;
    eval_body<PHASE, false>(S, slices, vbid, wgPerChain, hot, eval_first<PHASE>(hot, slices, vbid));
}

// ---- the one-launch split evaluation's A*P updates, beside the NEXT generator launch ------------------------------------------------
// What eval_kernel<EVAL_DECIDE> recorded (S.dec[0 .. applyCount)) is carried out by the workgroups 1 .. of gen_apply_kernel while
// workgroup 0 generates the next batch: the generator needs the decisions (atoms, matrix entries, erase cache: written by the deciding
// workgroups) but never the A*P rows, and the next evaluation launch -- which does -- starts behind this launch's end.  One item = one
// proposal x one of `parts` interleaved shares of its row(s); AP += delta * other, element by element, the same operations in the same
// order as the two-launch form's apply kernel (a move's / exchange's second update continues from the first: eval_update_ap2).
CG_DEVICE void eval_apply_items(const SamplerDev &S, const uint32_t wg, const uint32_t nwg)
{
    const uint32_t count = S.gs->applyCount;
    const uint32_t TPB = cg_bdim(), nq = S.Npad >> 2;
    uint32_t parts = (nq + 4u * TPB - 1u) / (4u * TPB);
    parts = parts < 1u ? 1u : (parts > 64u ? 64u : parts);
    for (uint32_t item = wg; item < count * parts; item += nwg) {
        const uint32_t q = item / parts, part = item - q * parts;
        const DecRec d = S.dec[q];
        if (d.n == 1u) eval_update_ap(S, d.r1, d.c1, d.d1, part * TPB, TPB * parts);
        else if (d.n == 2u) eval_update_ap2(S, d.r1, d.c1, d.d1, d.r2, d.c2, d.d2, part * TPB, TPB * parts);
    }
}
// workgroup 0: the generator (gen_body, as gen_kernel); workgroups 1 .. : the previous batch's A*P updates
template <int WIN>
CG_KERNEL void CG_LAUNCH_BOUNDS(WIN + 64) gen_apply_kernel(const uint64_t *lcgMul, const uint64_t *lcgInc, GenScalars *gs, const unsigned long long *eraseList, const uint32_t *queueUnits,
                                                          uint32_t eraseCap, uint32_t queueCap, const SamplerDev CG_CONSTANT *sp)
{
    if (cg_bid() != 0u) {
        cg_const_warm<sizeof(SamplerDev)>(sp);
        eval_apply_items(*(const SamplerDev *)sp, cg_bid() - 1u, cg_gdim() - 1u);
        return;
    }
    GenHot hot; hot.lcgMul = lcgMul; hot.lcgInc = lcgInc; hot.gs = gs; hot.eraseList = eraseList; hot.queueUnits = queueUnits; hot.eraseCap = eraseCap; hot.queueCap = queueCap;
    gen_body<WIN, true, false, 0>(sp, hot);      // (the one-launch split evaluation is the dense model's)
}
