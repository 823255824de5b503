// aux_kernels.h -- the remaining DenseNormalModel / GapsStatistics pieces around the sampler loop:
// sync (AP transpose), extraInitialization, chiSq partials, running statistics, meanChiSq partials.
#pragma once
#include "gaps_state.h"
#include "eval_kernel.h"

// DenseNormalModel::sync (DenseNormalModel.cpp:20-36): dst.AP(j,i) = src.AP(i,j).
// src is [srcM][srcNpad] (vector r contiguous), dst is [dstM = srcN][dstNpad], dstN = srcM.
// 64x64 tiles through LDS (+1 padding), coalesced on both sides.
#define TR_TILE 64
CG_DEVICE void transpose_body(const float *src, float *dst, uint32_t srcM, uint32_t srcN, uint32_t srcNpad, uint32_t dstNpad, uint32_t tilesX)
{
    CG_SHARED float tile[TR_TILE][TR_TILE + 1];
    const uint32_t t = cg_tid();                 // 256 threads: 64 columns x 4 rows per pass
    const uint32_t bx = cg_bid() % tilesX, by = cg_bid() / tilesX;
    const uint32_t tx = t & 63u, ty = t >> 6;
    for (uint32_t k = ty; k < TR_TILE; k += 4) {
        const uint32_t r = by * TR_TILE + k, c = bx * TR_TILE + tx;   // src row r (vector), element c
        tile[k][tx] = (r < srcM && c < srcN) ? src[(size_t)r * srcNpad + c] : 0.f;
    }
    cg_sync();
    for (uint32_t k = ty; k < TR_TILE; k += 4) {
        const uint32_t r = bx * TR_TILE + k, c = by * TR_TILE + tx;   // dst row r = src element index, element c = src row
        if (r < srcN && c < srcM) dst[(size_t)r * dstNpad + c] = tile[tx][k];
    }
}
CG_KERNEL void transpose_kernel(const float *src, float *dst, uint32_t srcM, uint32_t srcN, uint32_t srcNpad, uint32_t dstNpad, uint32_t tilesX)
{
    transpose_body(src, dst, srcM, srcN, srcNpad, dstNpad, tilesX);
}

// DenseNormalModel::extraInitialization (DenseNormalModel.cpp:38-54): AP(i,j) = sum_k other(i,k)*mat(j,k), k ascending, one
// accumulator per entry starting from +0.  The one dense contraction of the path, run once per session (and it only ever multiplies
// by an all-zero factor: at most one of the two matrices is given, GapsRunner.cpp:329-350 -- the host skips the launch when neither
// is).  Tiled: a workgroup owns 256 consecutive elements i x INIT_JT vectors j; thread = element i (coalesced loads of other and
// stores of AP), INIT_JT accumulators in registers, the matrix tile [k][j] staged through LDS INIT_KT patterns at a time and read
// back as broadcasts.  Same products, same order per entry as the triple loop.
#define INIT_JT 32
#define INIT_KT 32
CG_KERNEL void CG_LAUNCH_BOUNDS(256) init_ap_kernel(SamplerDev S, uint32_t tilesI)
{
    CG_SHARED float mt[INIT_KT][INIT_JT];
    const uint32_t t = cg_tid();
    const uint32_t i = (cg_bid() % tilesI) * 256u + t, j0 = (cg_bid() / tilesI) * (uint32_t)INIT_JT;
    float acc[INIT_JT];
#pragma unroll
    for (int jj = 0; jj < INIT_JT; ++jj) acc[jj] = 0.f;
    for (uint32_t k0 = 0; k0 < S.K; k0 += INIT_KT) {
        cg_sync();
        for (uint32_t e = t; e < (uint32_t)(INIT_KT * INIT_JT); e += 256u) {
            const uint32_t k = k0 + e / (uint32_t)INIT_JT, j = j0 + e % (uint32_t)INIT_JT;
            mt[e / (uint32_t)INIT_JT][e % (uint32_t)INIT_JT] = (k < S.K && j < S.M) ? S.mat[(size_t)k * S.Mpad + j] : 0.f;
        }
        cg_sync();
        const uint32_t kn = (S.K - k0) < (uint32_t)INIT_KT ? (S.K - k0) : (uint32_t)INIT_KT;
        for (uint32_t kk = 0; kk < kn; ++kk) {
            const float o = (i < S.N) ? S.other[(size_t)(k0 + kk) * S.Npad + i] : 0.f;
#pragma unroll
            for (int jj = 0; jj < INIT_JT; ++jj) acc[jj] = acc[jj] + o * mt[kk][jj];
        }
    }
    if (i < S.N) {
#pragma unroll
        for (int jj = 0; jj < INIT_JT; ++jj) if (j0 + (uint32_t)jj < S.M) S.AP[(size_t)(j0 + (uint32_t)jj) * S.Npad + i] = acc[jj];
    }
}

// number of entries > 0 per column of mat (canUseGibbs counters), one workgroup per column
// ProposalQueue::deathProb (ProposalQueue.cpp:123-127) for every atom count the domain's arrays can hold (+ a window's worth): the
// generator's lanes read the entries of the counts their window can see (gen_populate.h) -- the same gm_death_prob, evaluated once
// per session instead of twice per lane and launch
CG_KERNEL void death_prob_table_kernel(float *tab, uint32_t n, double domainLen, double alpha, double numBins)
{
    const uint32_t i = cg_bid() * cg_bdim() + cg_tid();
    if (i < n) tab[i] = gm_death_prob((double)(uint64_t)i, domainLen, alpha, numBins);
}
CG_KERNEL void count_pos_kernel(SamplerDev S)
{
    CG_SHARED uint32_t cnt;
    const uint32_t k = cg_bid(), t = cg_tid();
    if (t == 0) cnt = 0;
    cg_sync();
    uint32_t c = 0;
    for (uint32_t r = t; r < S.M; r += cg_bdim()) c += (S.mat[(size_t)k * S.Mpad + r] > 0.f) ? 1u : 0u;
    if (c) cg_atomic_add_u32(&cnt, c);
    cg_sync();
    if (t == 0) S.colPos[k] = cnt;
}

// DenseNormalModel::chiSq (DenseNormalModel.cpp:56-68): per-vector partials in the (W, float4) lane order;
// the host adds the M partials sequentially.  One workgroup of redW lanes per vector.
// chiSq needs the un-squared uncertainty to reproduce ((D-AP)/S)^2 bit for bit: Sraw is [M][Npad]
template <int V>
CG_KERNEL void CG_LAUNCH_BOUNDS(1024) chisq_rows_kernel_s(SamplerDev S, const float *Sraw, float *partial)
{
    CG_SHARED float lds[16 * V];
    const uint32_t row = cg_bid(), t = cg_tid(), BS = cg_bdim(), W = (uint32_t)V * BS, nq = S.Npad >> 2;
    const float *D = S.D + (size_t)row * S.Npad, *SR = Sraw + (size_t)row * S.Npad, *AP = S.AP + (size_t)row * S.Npad;
    float tot[1] = {0.f};
    for (int j = 0; j < V; ++j) {
        float acc = 0.f;
        for (uint32_t c = (uint32_t)j * BS + t; c < nq; c += W) {
            const cg_f4 d = ld4(D, c), s = ld4(SR, c), p = ld4(AP, c);
            { float q = (d.x - p.x) / s.x; acc = acc + q * q; }
            { float q = (d.y - p.y) / s.y; acc = acc + q * q; }
            { float q = (d.z - p.z) / s.z; acc = acc + q * q; }
            { float q = (d.w - p.w) / s.w; acc = acc + q * q; }
        }
        eval_vpark<V>(acc, j, lds, tot);
    }
    if (BS > 64u) { cg_sync(); eval_vfinish<1, V>(lds, tot); }
    if (t == 0) partial[row] = tot[0];
}

// ---- verification mode (reference order) --------------------------------------------------------------------------------
// acc + f(0) + f(1) + ... + f(total-1), added left to right by thread 0 (the result is valid there); the terms are computed
// by the whole workgroup, SEQ_CHUNK at a time through LDS.
#define SEQ_CHUNK 4096
template <class F>
CG_DEVICE float seq_sum(float acc, uint64_t total, float *lds, F f)
{
    const uint32_t t = cg_tid(), BS = cg_bdim();
    for (uint64_t base = 0; base < total; base += SEQ_CHUNK) {
        const uint32_t n = (total - base) < (uint64_t)SEQ_CHUNK ? (uint32_t)(total - base) : (uint32_t)SEQ_CHUNK;
        for (uint32_t i = t; i < n; i += BS) lds[i] = f(base + i);
        cg_sync();
        if (t == 0) for (uint32_t i = 0; i < n; ++i) acc = acc + lds[i];
        cg_sync();
    }
    return acc;
}
// DenseNormalModel::chiSq (DenseNormalModel.cpp:56-68) in its own order: element index outer, vector inner, one accumulator
CG_KERNEL void CG_LAUNCH_BOUNDS(256) chisq_seq_kernel(SamplerDev S, const float *Sraw, float *out)
{
    CG_SHARED float lds[SEQ_CHUNK];
    const uint32_t M = S.M;
    const float c = seq_sum(0.f, (uint64_t)M * S.N, lds, [&](uint64_t e) {
        const uint32_t i = (uint32_t)(e / M), j = (uint32_t)(e % M);
        const size_t o = (size_t)j * S.Npad + i;
        const float q = (S.D[o] - S.AP[o]) / Sraw[o];
        return q * q; });
    if (cg_tid() == 0) out[0] = c;
}
// GapsStatistics::meanChiSq (GapsStatistics.cpp:63-86) in its own order: genes outer, samples inner
CG_KERNEL void CG_LAUNCH_BOUNDS(256) mean_chisq_seq_kernel(SamplerDev P, const float *Sraw, const float *Asum, const float *Psum, uint32_t AMpad, float n2, float *out)
{
    CG_SHARED float lds[SEQ_CHUNK];
    const uint32_t M = P.M;
    const float c = seq_sum(0.f, (uint64_t)M * P.N, lds, [&](uint64_t e) {
        const uint32_t i = (uint32_t)(e / M), j = (uint32_t)(e % M);
        float m = 0.f;
        for (uint32_t k = 0; k < P.K; ++k) m = m + Asum[(size_t)k * AMpad + i] * Psum[(size_t)k * P.Mpad + j];
        m = m / n2;
        const float d = P.D[(size_t)j * P.Npad + i], sd = Sraw[(size_t)j * P.Npad + i];
        return ((d - m) * (d - m)) / (sd * sd); });
    if (cg_tid() == 0) out[0] = c;
}

// GapsStatistics::update / updateA / updateP (GapsStatistics.h:130-185), one workgroup per pattern.
// sums are column-major like `mat`: [K][Mpad].  mode: 0 = both (norm = max P column), 1 = A only
// (norm 1), 2 = P only (norm 1).
CG_KERNEL void stats_kernel(SamplerDev A, SamplerDev P, float *Asum, float *Asq, float *Psum, float *Psq, uint32_t mode)
{
    CG_SHARED float red[256];
    const uint32_t k = cg_bid(), t = cg_tid(), B = cg_bdim();
    const float *pc = P.mat + (size_t)k * P.Mpad, *ac = A.mat + (size_t)k * A.Mpad;
    float norm = 1.f;
    if (mode == 0) {
        float mx = 0.f;                                   // gaps::max(Vector): starts at 0 (VectorMath.cpp:43-51)
        for (uint32_t i = t; i < P.M; i += B) { const float v = pc[i]; mx = (v > mx) ? v : mx; }
        red[t] = mx; cg_sync();
        for (uint32_t off = B >> 1; off > 0; off >>= 1) { if (t < off) { const float o = red[t + off]; if (o > red[t]) red[t] = o; } cg_sync(); }
        norm = red[0];
        norm = (norm == 0.f) ? 1.f : norm;
    }
    if (mode != 1) for (uint32_t i = t; i < P.M; i += B) { const float q = pc[i] / norm; Psum[(size_t)k * P.Mpad + i] += q; Psq[(size_t)k * P.Mpad + i] += q * q; }
    if (mode != 2) for (uint32_t i = t; i < A.M; i += B) { const float q = ac[i] * norm; Asum[(size_t)k * A.Mpad + i] += q; Asq[(size_t)k * A.Mpad + i] += q * q; }
}

// GapsStatistics::meanChiSq (GapsStatistics.cpp:63-86) per-vector partials over the P sampler's data
// (vector j = sample j, elements i = genes), lane order as chiSq.  Asum: [K][A.Mpad], Psum: [K][P.Mpad].
template <int V>
CG_KERNEL void CG_LAUNCH_BOUNDS(1024) mean_chisq_rows_kernel(SamplerDev P, const float *Sraw, const float *Asum, const float *Psum, uint32_t AMpad, float n2, float *partial)
{
    CG_SHARED float lds[16 * V];
    const uint32_t j = cg_bid(), t = cg_tid(), BS = cg_bdim(), W = (uint32_t)V * BS, nq = P.Npad >> 2;
    const float *D = P.D + (size_t)j * P.Npad, *SR = Sraw + (size_t)j * P.Npad;
    float tot[1] = {0.f};
    for (int slot = 0; slot < V; ++slot) {
        float acc = 0.f;
        for (uint32_t c = (uint32_t)slot * BS + t; c < nq; c += W) {
            const cg_f4 d = ld4(D, c), s = ld4(SR, c);
            float dd[4] = {d.x, d.y, d.z, d.w}, ss[4] = {s.x, s.y, s.z, s.w};
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t i = 4 * c + e;
                if (i < P.N) {
                    float m = 0.f;
                    for (uint32_t k = 0; k < P.K; ++k) m = m + Asum[(size_t)k * AMpad + i] * Psum[(size_t)k * P.Mpad + j];
                    m = m / n2;
                    acc = acc + ((dd[e] - m) * (dd[e] - m)) / (ss[e] * ss[e]);
                }
            }
        }
        eval_vpark<V>(acc, slot, lds, tot);
    }
    if (BS > 64u) { cg_sync(); eval_vfinish<1, V>(lds, tot); }
    if (t == 0) partial[j] = tot[0];
}

// GapsStatistics::updatePump (GapsStatistics.h:65-126): pumpMatrixUniqueThreshold and pumpMatrixCutThreshold are the
// same code -- per row of A the first column holding the row maximum (strictly greater than everything before it,
// starting from 0) gets +1.  A through operator(): the matrix itself, or the HybridMatrix row copy.
CG_KERNEL void pump_kernel(SamplerDev A, float *pump)
{
    const uint32_t i = cg_bid() * cg_bdim() + cg_tid();
    if (i >= A.M) return;
    float maxV = 0.f; uint32_t maxI = 0;
    for (uint32_t j = 0; j < A.K; ++j) {
        const float v = A.sparse ? A.rows[(size_t)i * A.Kpad + j] : A.mat[(size_t)j * A.Mpad + i];
        if (maxV < v) { maxV = v; maxI = j; }
    }
    pump[(size_t)i * A.K + maxI] += 1.f;
}
