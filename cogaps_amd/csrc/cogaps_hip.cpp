// cogaps_hip.cpp -- host side of libcogaps_hip.so: the runner around the HIP kernels and the C ABI
// of include/cogaps_hip.h.  Restates runCoGAPSAlgorithm / runOnePhase / updateSampler
// (reference src/GapsRunner.cpp:382-499, :272-327, :201-222), GapsStatistics (GapsStatistics.h/.cpp)
// and the DenseNormalModel constructor (gibbs_sampler/DenseNormalModel.h:66-88) with every matrix,
// the atomic domains and the proposal queues resident in HBM.  The host only draws the per-iteration
// Poisson step counts (GapsRunner.cpp:294-295), streams the Xoroshiro seed sequence the proposal
// generator consumes (math/Random.cpp:221-248) and enqueues kernels.
#include <atomic>
#include "../../include/cogaps_hip.h"
#include "rt.h"
#include "gaps_state.h"
#include "gen_kernel.h"
#include "file_reader.h"
#include "eval_kernel.h"
#include "chain_kernel.h"
#include "aux_kernels.h"
#include "sparse_kernels.h"

#include <math.h>
#include <cmath>
#include <stdio.h>
#include <time.h>
#include <algorithm>
#include <vector>

#ifndef GEN_WIN
#define GEN_WIN 256
#endif
// A sampler whose batches are short takes half the window: every wave with live lanes costs a generator launch 0.4-0.9 us, a second
// round of a batch that outgrows the window ~3.5 us (profiles/r02_ab_generator_window.txt: the headline shape's P sampler, ~94 attempts
// per batch, is 0.75 us per launch faster with 128 lanes; its A sampler, ~157, needs the 256).  Any window gives the same batches.
#ifndef GEN_WIN_HALF
#define GEN_WIN_HALF (GEN_WIN / 2 >= 64 ? GEN_WIN / 2 : GEN_WIN)      // the narrower instantiation, for a sampler with short batches
#endif
// A third, WIDER window for the sparse model's chained launch (round 6): there the generator workgroup's window drawn ahead and its hand-over
// disappear behind the evaluation, which is long, and a second round of the batch costs ~12 us -- with the widest window the launch's workgroup
// holds (seven attempt waves + the helper wave: 448 attempts; the attempt lanes carry out the queue behind the helper wave's 64 slots) a batch of
// ~250 proposals (BASELINE configs[4]'s shard shape: 34 % of the A sampler's launches took two rounds) nearly always ends in its first round:
// A 33.9 -> 31.5 us per launch at 384 attempts, 30.9 at 448: 6.24 -> 6.55 -> 6.65 M proposals/s (profiles/r06_ab_sparse_chained_launch.txt).
// The dense chain pays for every further attempt wave in every launch (profiles/r06_ab_windows_320_384.txt) and keeps two windows.  Only the
// chained sparse launch is instantiated at this window: a sampler that steps by two launches per batch goes back to GEN_WIN first (the window
// is free to change between batches: results do not depend on it).
#ifndef GEN_WIN_WIDE
#define GEN_WIN_WIDE (GEN_WIN == 256 ? GEN_CHAIN_THREADS - 64 : GEN_WIN)
#endif
static uint32_t gen_window_for(uint32_t current, float stepsPerBatch, bool wideOk = false)
{
    if (GEN_WIN_WIDE != GEN_WIN) {
        if (current == (uint32_t)GEN_WIN_WIDE) return (wideOk && (stepsPerBatch <= 1.f || stepsPerBatch > 0.75f * (float)GEN_WIN)) ? current : (uint32_t)GEN_WIN;
        if (wideOk && current == (uint32_t)GEN_WIN && stepsPerBatch > 0.9f * (float)GEN_WIN) return (uint32_t)GEN_WIN_WIDE;
    }
    if (GEN_WIN_HALF == GEN_WIN || stepsPerBatch <= 1.f) return current;
    if (current == (uint32_t)GEN_WIN && stepsPerBatch < 0.85f * (float)GEN_WIN_HALF) return (uint32_t)GEN_WIN_HALF;
    if (current == (uint32_t)GEN_WIN_HALF && stepsPerBatch > 0.95f * (float)GEN_WIN_HALF) return (uint32_t)GEN_WIN;
    return current;
}


static thread_local std::string g_last_error;
static thread_local int g_last_code = COGAPS_OK;
static int fail(const std::string &m, int code = COGAPS_ERR_GENERIC) { g_last_error = m; g_last_code = code; return 1; }
// a caught exception: device memory exhausted and host memory exhausted keep their own codes (cogaps_last_error_code)
static int fail_exc(const std::exception &e)
{
    if (dynamic_cast<const rt_out_of_memory *>(&e)) return fail(e.what(), COGAPS_ERR_OUT_OF_DEVICE_MEMORY);
    if (dynamic_cast<const std::bad_alloc *>(&e)) return fail(e.what(), COGAPS_ERR_OUT_OF_HOST_MEMORY);
    return fail(e.what());
}

// ------------------------------------------------------------------------------------------------
// host RNG pieces: Xoroshiro128+ seeder (Random.cpp:221-248) and the runner's PCG (GapsRunner.cpp:437)
// ------------------------------------------------------------------------------------------------
struct HostSeeder {
    uint64_t s0, s1;
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t a = s0; uint64_t b = s1; const uint64_t r = a + b;
        b ^= a; s0 = rotl(a, 24) ^ b ^ (b << 16); s1 = rotl(b, 37);
        return r;
    }
    void init(uint64_t seed) { s0 = seed | 1; s1 = seed | 1; for (int i = 0; i < 5000; ++i) next(); }
};

// Random.cpp:125-170 (host double math, as in the reference; gaps::lgamma = libm lgamma here)
static int host_poisson(uint64_t &st, double lambda)
{
    auto unifd = [&]() { return (double)pcg_u32(st) / 4294967295.0; };
    if (lambda <= 5.0) {
        int x = 0; double p = unifd(); const double cutoff = exp(-lambda);
        while (p >= cutoff) { p *= unifd(); ++x; }
        return x;
    }
    const double c = 0.767 - 3.36 / lambda;
    const double beta = 3.1415926535897932384626433832795 / sqrt(3.0 * lambda);
    const double alpha = beta * lambda;
    const double k = log(c) - lambda - log(beta);
    for (;;) {
        const double u = unifd();
        const double x = (alpha - log((1.0 - u) / u)) / beta;
        const double n = floor(x + 0.5);
        if (n < 0.0) continue;
        const double v = unifd();
        const double y = alpha - beta * x;
        const double w = 1.0 + exp(y);
        const double lhs = y + log(v / (w * w));
        const double rhs = k + n * log(lambda) - lgamma(n + 1);
        if (lhs <= rhs) return (int)n;
    }
}

// ------------------------------------------------------------------------------------------------
// lookup tables (Random.cpp:269-295 over Math.cpp:43-81): float argument, double-precision normal /
// gamma(2,1) cdf and quantile, rounded to float.  Boost.Math in the reference; libm + Newton here.
// ------------------------------------------------------------------------------------------------
static double lut_erfc_inv(double y)
{
    if (y == 1.0) return 0.0;
    double lo = -6.0, hi = 6.0;
    for (int i = 0; i < 60; ++i) { const double mid = 0.5 * (lo + hi); if (erfc(mid) > y) lo = mid; else hi = mid; }
    double x = 0.5 * (lo + hi);
    for (int i = 0; i < 4; ++i) {
        const double f = erfc(x) - y, fp = -2.0 / sqrt(3.1415926535897932384626433832795) * exp(-x * x), fpp = -2.0 * x * fp;
        const double dx = f / fp;
        x -= dx / (1.0 - 0.5 * dx * fpp / fp);
    }
    return x;
}
static double lut_gamma2_cdf(double x)
{
    if (x < 0.5) { double sum = 0.0, xk = x * x, fact = 2.0; for (int k = 2; k < 40; ++k) { const double t = xk * (double)(k - 1) / fact; sum += (k & 1) ? -t : t; xk *= x; fact *= (double)(k + 1); } return sum; }
    return 1.0 - exp(-x) * (1.0 + x);
}
static double lut_gamma2_quantile(double p)
{
    double lo = 0.0, hi = 60.0;
    for (int i = 0; i < 80; ++i) { const double mid = 0.5 * (lo + hi); if (lut_gamma2_cdf(mid) < p) lo = mid; else hi = mid; }
    double x = 0.5 * (lo + hi);
    for (int i = 0; i < 3; ++i) { const double f = lut_gamma2_cdf(x) - p, fp = x * exp(-x); if (fp > 0) x -= f / fp; }
    return x;
}
static void build_luts(std::vector<float> &e, std::vector<float> &ei, std::vector<float> &qg)
{
    e.resize(GAPS_ERF_N); ei.resize(GAPS_ERFINV_N); qg.resize(GAPS_QGAMMA_N);
    auto pnorm = [](float p) { return (float)(0.5 * erfc(-(double)p / 1.41421356237309504880)); };
    auto qnorm = [](float q) { double r = lut_erfc_inv(2.0 * (double)q); r = -r; r *= 1.41421356237309504880; return (float)(r + 0.0); };
    auto qgam = [](float q) { if (q < 0.000001f) return 0.f; return (float)lut_gamma2_quantile((double)q); };
    for (unsigned i = 0; i < GAPS_ERF_N; ++i) { const float x = (float)i / 1000.f; e[i] = 2.f * pnorm(x * GAPS_SQRT2F) - 1.f; }
    for (unsigned i = 0; i < GAPS_ERFINV_N - 1; ++i) { const float x = (float)i / (float)(GAPS_ERFINV_N - 1); ei[i] = qnorm((1.f + x) / 2.f) / GAPS_SQRT2F; }
    ei[GAPS_ERFINV_N - 1] = qnorm(1.9998f / 2.f) / GAPS_SQRT2F;
    qg[0] = 0.f;
    for (unsigned i = 1; i < GAPS_QGAMMA_N - 1; ++i) { const float x = (float)i / (float)(GAPS_QGAMMA_N - 1); qg[i] = qgam(x); }
    qg[GAPS_QGAMMA_N - 1] = qgam(0.9998f);
}

// Virtual lanes of the row reductions: one float4 chunk per lane while the vector has at most 16384 chunks
// (eval_kernel.h).  Workgroups have min(W, 1024) threads.
extern "C" uint32_t cogaps_reduction_width(uint32_t N)
{
    uint32_t need = (N + 3u) / 4u, w = 64;
    while (w < need && w < 16384u) w <<= 1;
    return w;
}
// threads (= virtual lanes) of the sparse evaluation: one per 64-bit flag word of a data vector, 64..256 (sparse_kernels.h)
extern "C" uint32_t cogaps_sparse_width(uint32_t N)
{
    uint32_t need = N / 64u + 1u, w = 64;
    while (w < need && w < 256u) w <<= 1;
    return w;
}
// launch a kernel template<int V> with V = W / threads virtual lanes per thread
#define LAUNCH_V(KERNEL, W, grid, stream, ...) do { const uint32_t bs_ = (W) < 1024u ? (W) : 1024u; \
    switch ((W) / bs_) { case 1: RT_LAUNCH(KERNEL<1>, grid, bs_, stream, __VA_ARGS__); break; case 2: RT_LAUNCH(KERNEL<2>, grid, bs_, stream, __VA_ARGS__); break; \
                         case 4: RT_LAUNCH(KERNEL<4>, grid, bs_, stream, __VA_ARGS__); break; case 8: RT_LAUNCH(KERNEL<8>, grid, bs_, stream, __VA_ARGS__); break; \
                         default: RT_LAUNCH(KERNEL<16>, grid, bs_, stream, __VA_ARGS__); break; } } while (0)

// Switches that change what is MEASURED and never a result (launch sizes, the two-launch split form, reading S instead of recomputing it)
// exist in development builds only (-DCOGAPS_DEV, -DGEN_PROFILE): the product library does not look at them.  What the product library
// does read from the environment, on purpose, is documented in include/cogaps_hip.h: COGAPS_NO_GRAPH (every launch as a plain call --
// counter collection hangs on replayed graphs), COGAPS_NO_CHAIN (two launches per batch: the A/B and the equality test of the chained launch) and
// COGAPS_FORCE_CHAIN (tests: the chained launch on a device with fewer compute units than the launch has workgroups), COGAPS_CHAIN_SPLIT (the
// split evaluation inside the chained launch: measured, not the default).
static const char *dev_env(const char *name)
{
#if defined(COGAPS_DEV) || defined(GEN_PROFILE) || defined(COGAPS_EMUL)
    return getenv(name);
#else
    (void)name; return nullptr;
#endif
}
static const uint32_t SEQ_SPARSE_GRID = 256;      // workgroups of the sparse model's verification-mode evaluation (one scratch region each)

// ------------------------------------------------------------------------------------------------
struct HostSampler {
    SamplerDev d;                 // device pointers + constants (passed by value to the kernels)
    float *Sraw = nullptr;        // un-squared uncertainty [M][Npad] (chiSq)
    uint64_t *seeds = nullptr; size_t seedCap = 0;
    uint64_t *hSeeds = nullptr; size_t hSeedCap = 0;
    float *partial = nullptr;     // [M] chi2 partials
    uint32_t nAtoms = 0;          // host copy after the last update
    float avgQueue = 0.f;
    float dataSparsity = 0.f;     // DenseNormalModel::dataSparsity
    SamplerDev *dRecord = nullptr; SamplerDev recordHeld;   // `d` in device memory (the generator reads it through a pointer) and what that copy holds
    bool recordValid = false;
    float stepsPerBatch = 0.f;
    uint32_t genWin = GEN_WIN;    // lanes of the generator launch (gen_window_for)
    float anneal = 1.f;           // annealing temperature of the next update
    rt_graph graph;               // GRAPH_PAIRS (generate, evaluate) pairs, replayed while the kernel parameters stay the same
    SamplerDev graphKey; bool graphValid = false;    // proposals per batch in the last update (chunk-size predictor)
    // chained launch (chain_kernel.h): one launch per batch; consecutive launches alternate the parity they carry as a kernel argument, so
    // a captured run of launches exists once per starting parity
    bool chain = false; uint32_t chainParity = 0;
    uint32_t chainParityStart = 0;      // parity of the current update's first chained launch (which copy a given launch of the update evaluated)
    bool chainOff = false; uint32_t chainRecoveries = 0;      // a hand-over inside a chained launch never arrived: the batch was completed by chain_recover, the sampler keeps two launches per batch from then on
    unsigned long long *chainGrans = nullptr;      // [queueCap][CHAIN_GRAN_STRIDE] the decisions' granules (the split form's per-slice totals keep SamplerDev::grans)
    rt_graph chainGraph[2]; bool chainGraphValid[2] = {false, false};
    size_t traceCap = 0;
    char name = 'A';
    // perf accounting
    uint64_t evalLaunches = 0, genLaunches = 0, batches = 0;
    // HIP-event samples, split into launches that processed a batch and launches past the end of an update
    double evalMs = 0, genMs = 0, evalNoopMs = 0, genNoopMs = 0; uint64_t evalTimed = 0, genTimed = 0, evalNoopTimed = 0, genNoopTimed = 0;
    uint64_t updLaunches = 0;    // (generator, evaluation) pairs enqueued in the current update
    uint64_t batchesAtTimingOn = 0;   // `batches` when event timing was switched on: the sampled times are scaled to the batches since then
    uint64_t plainRotor = 0;     // which graph replay of a chunk runs as plain, event-carrying launches while timing is on
    // launch clock of the chained launches (gaps_state.h): durations in 0.1 us bins since timing was switched on, their sum and count
    std::vector<unsigned long long> clockHost; uint64_t clockSeen = 0; std::vector<uint64_t> clockHist; double clockSumUs = 0; uint64_t clockN = 0;
    // ... and the launch-to-launch PERIOD (entry of one launch's first workgroup to the next launch's): the launch with everything the
    // dispatcher does around it -- what rocprofv3's dispatch duration plus the idle gap adds up to
    std::vector<uint64_t> periodHist; double periodSumUs = 0; uint64_t periodN = 0;
};

struct cogaps_session {
    double startTime = 0.0;                                          // gaps::run's startTime (GapsRunner.cpp:383): status lines, totalRunningTime
    float *pump = nullptr; uint32_t pumpUpdates = 0;                 // mPumpMatrix [nGenes][K] row-major (device), mPumpUpdates
    std::vector<float> snapA[2], snapP[2]; uint32_t nSnap[2] = {0, 0};   // [0] equilibration, [1] sampling snapshots, row-major

    cogaps_params p;
    std::vector<uint32_t> subset; std::vector<float> fixed;
    uint32_t nGenes = 0, nSamples = 0, K = 0;
    HostSampler A, P;
    HostSeeder seeder; uint64_t runnerRng = 0;
    // The seeder's output sequence does not depend on how it is consumed (update k takes the next nSteps values, Random.cpp:221-248), so it
    // is produced AHEAD of its use while the host waits for the GPU: seed_take() pops, seed_top_up() refills between launch and sync.
    std::vector<uint64_t> seedFifo; size_t seedHead = 0;
    rt_stream_t stream; bool ownsStream = true;      // a session that joined a cogaps_batch runs on the batch's stream
    float *dErf = nullptr, *dErfinv = nullptr, *dQgamma = nullptr; uint64_t *dLcgMul = nullptr, *dLcgInc = nullptr;
    float *Asum = nullptr, *Asq = nullptr, *Psum = nullptr, *Psq = nullptr;
    unsigned statUpdates = 0;
    std::vector<float> chisqHist; std::vector<uint32_t> atomHistA, atomHistP;
    uint64_t totalUpdates = 0; double samplerSeconds = 0; double syncMs = 0; uint64_t syncTimed = 0, syncBytes = 0;
    bool timing = false; bool evInit = false;
    bool noGraph = getenv("COGAPS_NO_GRAPH") != nullptr;     // diagnostics: every launch as a plain call (counter collection tools)
    bool noChain = getenv("COGAPS_NO_CHAIN") != nullptr;     // A/B and tests: two launches per batch (gen_kernel, eval_kernel<EVAL_FUSED>) where the chained launch would serve
    // The split evaluation inside the chained launch (EVAL_CHAIN_SPLIT) is built, tested on the hardware and NOT the default: measured on the
    // headline chain it loses 6 % (profiles/r05_ab_chained_split_evaluation_not_kept.txt -- the launch's workgroups carry the generator's 142 KB
    // of LDS, so one evaluation workgroup fits a compute unit and each takes 3-4 slices one after the other, where the two-launch form has
    // two per unit and all slices resident).  COGAPS_CHAIN_SPLIT=1 takes it (the A/B, the equality test).
    bool noChainSplit = getenv("COGAPS_CHAIN_SPLIT") == nullptr;
    bool testWideWindow = getenv("COGAPS_TEST_WIDE_WINDOW") != nullptr;      // tests: the sparse model's chained launch at GEN_WIN_WIDE from the first update on (otherwise once batches exceed 0.9 * GEN_WIN)
    bool forceChain = getenv("COGAPS_FORCE_CHAIN") != nullptr;      // tests: the chained launch also where the device shows fewer compute units than the launch has workgroups (they then run in turns, the generator last)
    unsigned computeUnits = 0;      // of the session's device: the chained launch wants all its workgroups resident at once, one per compute unit
    std::vector<rt_event_pair> evPool; std::vector<int> evKind; std::vector<HostSampler *> evOwner; std::vector<uint64_t> evOrd; size_t evUsed = 0;
    GenScalars *hGs = nullptr;    // pinned staging
    bool poisoned = false;        // a device error ended an update half way (capacity, a hand-over inside a launch that never arrived): the chain's state is not a state of the chain
};

static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

template <class T> static T *dalloc(size_t n) { return (T *)rt_malloc(n * sizeof(T)); }

// SamplerDev::deathProb for the atom arrays' current capacity (the sampler's creation, grow_atoms)
static void build_death_prob_table(cogaps_session *s, SamplerDev &d)
{
    const uint32_t n = d.atomCap + GAPS_DEATH_PROB_PAD;
    float *tab = dalloc<float>(n);
    RT_LAUNCH(death_prob_table_kernel, (n + 255u) / 256u, 256, s->stream, tab, n, d.domainLenD, d.alphaD, d.numBins);
    rt_sync(s->stream);
    d.deathProb = tab;
}

static void free_sampler(HostSampler &h)
{
    SamplerDev &d = h.d;
    rt_free(d.seqScratch); rt_free((void *)d.deathProb); rt_free(d.launchClock);
    rt_free((void *)d.D); rt_free((void *)d.S2); rt_free(d.AP); rt_free(d.mat); rt_free(d.colPos);
    rt_free(d.atoms); rt_free(d.vec); rt_free(d.freeHandles); rt_free(d.binHead);
    rt_free(d.bits0); rt_free(d.bits1); rt_free(d.bits2); rt_free(d.eraseList); rt_free(d.queue); rt_free(d.queueUnits); rt_free(d.chainSlots); rt_free(h.chainGrans); rt_free(d.partials); rt_free(d.grans); rt_free(d.dec);
    rt_free(d.rowStamp); rt_free(d.atomStamp); rt_free(d.gapStamp); rt_free(d.inlineStamp); rt_free(d.atomDest);
    rt_free(d.gs); rt_free(d.trace); rt_free(d.traceBatchNproc); rt_free(d.traceBatchQlen);
    rt_free(h.Sraw); rt_free(h.seeds); rt_free_host(h.hSeeds); rt_free(h.partial); rt_free(h.dRecord);
    rt_free((void *)d.dflags); rt_free((void *)d.dprefix); rt_free((void *)d.dptr); rt_free((void *)d.dvals); rt_free(d.rows); rt_free(d.mflags); rt_free(d.Z1); rt_free(d.Z2);
}

// Matrix(mat, genesInCols, subsetGenes, indices) (data_structures/Matrix.cpp:30-69) laid out as
// [vector j][element i] + the DenseNormalModel constructor (DenseNormalModel.h:66-88)
static void build_sampler(cogaps_session *s, HostSampler &h, char name, const float *data, uint32_t nrow, uint32_t ncol, const float *unc,
                          bool genesInCols, bool subsetGenes, float alpha, float maxGibbsMass)
{
    const cogaps_params &p = s->p;
    const bool subsetData = p.subsetData && !s->subset.empty();
    const uint32_t *indices = s->subset.data(); const uint32_t nIdx = (uint32_t)s->subset.size();
    const uint32_t nG = (subsetData && subsetGenes) ? nIdx : (genesInCols ? ncol : nrow);
    const uint32_t nS = (subsetData && !subsetGenes) ? nIdx : (genesInCols ? nrow : ncol);
    SamplerDev &d = h.d; memset(&d, 0, sizeof(d)); h.name = name;
    const bool sparse = p.useSparseOptimization != 0;
    d.N = nG; d.M = nS; d.K = p.nPatterns;
    d.seq = p.reductionMode == COGAPS_REDUCE_SEQ ? 1u : 0u; d.mathMode = (uint32_t)p.mathMode;
    d.Npad = (d.N + 3u) & ~3u; d.Mpad = (d.M + 3u) & ~3u;
    d.redW = cogaps_reduction_width(d.N);
    const size_t tot = (size_t)d.M * d.Npad;
    // With the default uncertainty the evaluation kernel recomputes S*S = max(0.1 D, 0.1)^2 from the D value it loads anyway
    // (bit-identical: the same three fp32 operations as the fill below): no S2 array, one row less per proposal from HBM.
    // (COGAPS_READ_S: diagnostics, keeps the array and the loads.)
    const bool defaultS = !sparse && unc == nullptr && !dev_env("COGAPS_READ_S");
    float *dD = dalloc<float>(tot), *dS2 = defaultS ? nullptr : dalloc<float>(tot); h.Sraw = dalloc<float>(tot);
    if (sparse) {
        if (d.K > SP_KMAX) throw std::runtime_error("useSparseOptimization supports at most 512 patterns");
        d.Wn = d.N / 64u + 1u;
    }
    // The vectors are staged through the host in blocks of at most 16 M elements (pad: D = 0, S = S2 = 1) -- never three dense
    // host copies of the matrix (BASELINE configs[4]'s shard is 2.5 GB per copy).  The sums run over the vectors in order, as
    // gaps::nonZeroMean does (MatrixMath.cpp:39-55).
    const uint32_t rowsPerBlock = (uint32_t)std::max<size_t>(1, std::min<size_t>(d.M, ((size_t)1 << 24) / std::max<uint32_t>(1u, d.Npad)));
    std::vector<float> D((size_t)rowsPerBlock * d.Npad), S2(dS2 ? D.size() : 0), SR(D.size());
    std::vector<unsigned long long> fl; std::vector<uint32_t> pre, ptr; std::vector<float> vals;
    if (sparse) { fl.assign((size_t)d.M * d.Wn, 0ull); pre.assign((size_t)d.M * d.Wn, 0u); ptr.assign((size_t)d.M + 1, 0u); }
    float sum = 0.f; unsigned nnz = 0;
    for (uint32_t j0 = 0; j0 < nS; j0 += rowsPerBlock) {
        const uint32_t j1 = std::min(nS, j0 + rowsPerBlock);
        std::fill(D.begin(), D.end(), 0.f); std::fill(SR.begin(), SR.end(), 1.f); if (dS2) std::fill(S2.begin(), S2.end(), 1.f);
        for (uint32_t j = j0; j < j1; ++j) {
            if (sparse) ptr[j] = (uint32_t)vals.size();
            for (uint32_t i = 0; i < nG; ++i) {
                const uint32_t dataRow = (subsetData && (subsetGenes != genesInCols)) ? indices[genesInCols ? j : i] - 1 : (genesInCols ? j : i);
                const uint32_t dataCol = (subsetData && (subsetGenes == genesInCols)) ? indices[genesInCols ? i : j] - 1 : (genesInCols ? i : j);
                float v = data[(size_t)dataRow * ncol + dataCol];
                if (sparse && !(v > 0.f)) v = 0.f;                                // SparseVector keeps v > 0 only (SparseVector.cpp:20-33)
                const size_t o = (size_t)(j - j0) * d.Npad + i;
                D[o] = v;
                const float sd = (unc && !sparse) ? unc[(size_t)dataRow * ncol + dataCol] : gm_max(v * 0.1f, 0.1f);   // gaps::pmax, MatrixMath.cpp:74-84; the sparse model always assumes the default (SparseNormalModel.h:90-96)
                SR[o] = sd; if (dS2) S2[o] = sd * sd;
                sum += v; if (v > 0.f) ++nnz;                                    // gaps::nonZeroMean, MatrixMath.cpp:39-55
                if (sparse) {      // SparseMatrix: flag words, number of packed values before each word, the values
                    const uint32_t w = i >> 6;
                    if ((i & 63u) == 0u) pre[(size_t)j * d.Wn + w] = (uint32_t)vals.size() - ptr[j];
                    if (v > 0.f) { fl[(size_t)j * d.Wn + w] |= 1ull << (i & 63u); vals.push_back(v); }
                }
            }
            if (sparse) for (uint32_t w = (nG + 63u) >> 6; w < d.Wn; ++w) pre[(size_t)j * d.Wn + w] = (uint32_t)vals.size() - ptr[j];   // the word past the last element (Wn = N/64 + 1)
        }
        const size_t off = (size_t)j0 * d.Npad, cnt = (size_t)(j1 - j0) * d.Npad;
        rt_h2d(dD + off, D.data(), cnt * 4, s->stream); if (dS2) rt_h2d(dS2 + off, S2.data(), cnt * 4, s->stream); rt_h2d(h.Sraw + off, SR.data(), cnt * 4, s->stream);
        rt_sync(s->stream);                                                     // the staging buffers are reused by the next block
    }
    const float meanD = sum / (float)nnz;
    h.dataSparsity = 1.f - (float)(uint32_t)nnz / (float)(d.M * d.N);       // gaps::sparsity, MatrixMath.cpp:6-21 (unsigned count, float product of the dimensions)
    d.alpha = alpha;
    d.lambda = alpha * sqrtf((float)(uint64_t)d.K / meanD);
    d.maxGibbsMass = maxGibbsMass / d.lambda;
    d.D = dD; d.S2 = dS2; d.defaultS = defaultS ? 1u : 0u;
    d.unitBytes = 4u * d.N;
    if (!sparse) d.AP = dalloc<float>(tot);
    else {
        // SparseMatrix (flag words + packed values per vector) and the HybridMatrix copies; D / Sraw stay for meanChiSq
        d.sparse = 1; d.beta = 100.f; d.unitBytes = 1u;
        d.Mw = d.M / 64u + 1u; d.Kpad = (d.K + 3u) & ~3u; d.spW = cogaps_sparse_width(d.N);
        ptr[d.M] = (uint32_t)vals.size();
        unsigned long long *dfl = dalloc<unsigned long long>(fl.size()); uint32_t *dpre = dalloc<uint32_t>(pre.size()), *dptr = dalloc<uint32_t>(ptr.size()); float *dv = dalloc<float>(vals.size() + 1);
        rt_h2d(dfl, fl.data(), fl.size() * 8, s->stream); rt_h2d(dpre, pre.data(), pre.size() * 4, s->stream); rt_h2d(dptr, ptr.data(), ptr.size() * 4, s->stream);
        if (!vals.empty()) rt_h2d(dv, vals.data(), vals.size() * 4, s->stream);
        rt_sync(s->stream);
        d.dflags = dfl; d.dprefix = dpre; d.dptr = dptr; d.dvals = dv;
        d.rows = dalloc<float>((size_t)d.M * d.Kpad);
        d.mflags = dalloc<unsigned long long>((size_t)d.K * d.Mw);
        d.Z1 = dalloc<float>(d.K); d.Z2 = dalloc<float>((size_t)d.K * d.K);
        if (d.seq) {
            if (d.Wn > (uint32_t)SP_SEQ_WORDS) throw std::runtime_error("reductionMode SEQ with the sparse model supports data vectors of up to 262080 elements");
            d.seqScratch = dalloc<float>((size_t)SEQ_SPARSE_GRID * 3u * d.Npad);
        }
    }
    d.mat = dalloc<float>((size_t)d.K * d.Mpad);
    d.colPos = dalloc<uint32_t>(d.K);
    d.luts.erf = s->dErf; d.luts.erfinv = s->dErfinv; d.luts.qgamma = s->dQgamma;
    // atomic domain
    const uint64_t nBins = (uint64_t)d.M * d.K;
    d.atomCap = (uint32_t)std::min<uint64_t>(nBins + 65536ull, 0x7FFFFFF0ull);
    if (const char *e = getenv("COGAPS_INITIAL_ATOM_CAP")) d.atomCap = (uint32_t)std::max(64l, atol(e));      // tests: exercises grow_atoms
    d.atoms = dalloc<AtomRec>(d.atomCap); d.vec = dalloc<uint32_t>(d.atomCap); d.freeHandles = dalloc<uint32_t>(d.atomCap);
    d.binHead = dalloc<uint32_t>(nBins); rt_memset(d.binHead, 0xFF, nBins * 4, s->stream);
    d.nWords0 = (uint32_t)((nBins + 63) / 64); d.nWords1 = (d.nWords0 + 63) / 64; d.nWords2 = (d.nWords1 + 63) / 64;
    d.bits0 = dalloc<unsigned long long>(d.nWords0); d.bits1 = dalloc<unsigned long long>(d.nWords1); d.bits2 = dalloc<unsigned long long>(d.nWords2);
    d.queueCap = d.M + 8; d.eraseCap = d.queueCap;
    d.eraseList = dalloc<unsigned long long>(d.eraseCap); d.queue = dalloc<PropRec>((size_t)2 * d.queueCap); d.chainSlots = dalloc<ChainSlot>(2); d.launchClock = dalloc<unsigned long long>(2u * GAPS_CLOCK_RING); h.chainGrans = dalloc<unsigned long long>((size_t)d.queueCap * CHAIN_GRAN_STRIDE); d.queueUnits = dalloc<uint32_t>(d.queueCap); d.partials = dalloc<float>((size_t)d.queueCap * 64); d.grans = dalloc<unsigned long long>((size_t)d.queueCap * 64); d.dec = dalloc<DecRec>(d.queueCap);
    d.rowStamp = dalloc<unsigned long long>(d.M);
    d.atomStamp = dalloc<unsigned long long>(d.atomCap); d.gapStamp = dalloc<unsigned long long>((size_t)d.atomCap + 1);
    d.inlineStamp = dalloc<unsigned long long>(d.atomCap); d.atomDest = dalloc<uint64_t>(d.atomCap);
    d.lcgMul = s->dLcgMul; d.lcgInc = s->dLcgInc;
    d.binLength = 0xFFFFFFFFFFFFFFFFull / nBins;               // ProposalQueue.cpp:27
    d.domainLenU = d.binLength * nBins;                         // ConcurrentAtomicDomain.cpp:16-18
    d.domainLenD = (double)d.domainLenU;                        // ProposalQueue.cpp:30
    d.numBins = (double)nBins; d.alphaD = (double)alpha;
    d.invBinLen = 1.0 / (double)d.binLength;
    d.invK = 1.0 / (double)d.K;
    if (nBins >= 0xFFFFFFF0ull) throw std::runtime_error("rows x nPatterns must stay below 2^32");
    d.rboundNone = gm_u64_from_double_x86(d.domainLenD);
    d.iPartL = 0xFFFFFFFFFFFFFFFFull / d.domainLenU; d.limitL = d.domainLenU * d.iPartL;   // uniform64(1, L)
    build_death_prob_table(s, d);
    d.gs = dalloc<GenScalars>(1);
    GenScalars g; memset(&g, 0, sizeof(g));
    g.front = CG_NONE;
    g.qrng = pcg_from_seed(s->seeder.next());                   // ProposalQueue::mRng(randState), ProposalQueue.cpp:24
    rt_h2d(d.gs, &g, sizeof(g), s->stream); rt_sync(s->stream);
    h.partial = dalloc<float>(d.M);
}

static void read_gs(cogaps_session *s, HostSampler &h)
{
    rt_d2h(s->hGs, h.d.gs, sizeof(GenScalars), s->stream);
    rt_sync(s->stream);
}

// the next n outputs of the session's seeder into dst, from the look-ahead buffer as far as it reaches
static void seed_take(cogaps_session *s, uint64_t *dst, size_t n)
{
    const size_t avail = s->seedFifo.size() - s->seedHead, k = std::min(avail, n);
    if (k) { memcpy(dst, s->seedFifo.data() + s->seedHead, k * 8); s->seedHead += k; }
    for (size_t i = k; i < n; ++i) dst[i] = s->seeder.next();
    if (s->seedHead == s->seedFifo.size()) { s->seedFifo.clear(); s->seedHead = 0; }
}
// look ahead until `target` outputs are waiting (called while the GPU works through a chunk of launches)
static void seed_top_up(cogaps_session *s, size_t target)
{
    size_t avail = s->seedFifo.size() - s->seedHead;
    if (avail >= target) return;
    if (s->seedHead) { s->seedFifo.erase(s->seedFifo.begin(), s->seedFifo.begin() + (ptrdiff_t)s->seedHead); s->seedHead = 0; }
    s->seedFifo.reserve(target);
    for (; avail < target; ++avail) s->seedFifo.push_back(s->seeder.next());
}

static void grow_atoms(cogaps_session *s, HostSampler &h, uint32_t need)
{
    SamplerDev &d = h.d;
    if (need <= d.atomCap) return;
    rt_alloc_scope allocOn(s->stream);
    uint32_t cap = d.atomCap;
    while (cap < need) cap = (uint32_t)std::min<uint64_t>((uint64_t)cap * 2, 0x7FFFFFF0ull);
    auto regrow = [&](auto *&ptr, size_t elt) {
        void *n = rt_malloc((size_t)cap * elt);
        rt_d2d(n, ptr, (size_t)d.atomCap * elt, s->stream); rt_sync(s->stream);
        rt_free(ptr); ptr = (decltype(ptr))n;
    };
    regrow(d.atoms, sizeof(AtomRec)); regrow(d.vec, 4); regrow(d.freeHandles, 4);
    regrow(d.atomStamp, 8); regrow(d.inlineStamp, 8); regrow(d.atomDest, 8);
    { void *n = rt_malloc(((size_t)cap + 1) * 8); rt_d2d(n, d.gapStamp, ((size_t)d.atomCap + 1) * 8, s->stream); rt_sync(s->stream); rt_free(d.gapStamp); d.gapStamp = (unsigned long long *)n; }
    d.atomCap = cap;
    rt_free((void *)d.deathProb); d.deathProb = nullptr;
    build_death_prob_table(s, d);
}

// HIP-event timing of a sample of the launches (every 8th): the start / stop events are attached to the kernel's
// own dispatch packet (hipExtLaunchKernelGGL), so their difference is the dispatch's begin-to-end time, the
// figure rocprofv3 --kernel-trace reports; no extra packets enter the stream.  Events are resolved after the
// chunk's synchronisation so that timing never stalls the queue.
static int timing_slot(cogaps_session *s, HostSampler &h, int kind, uint64_t ordinal)
{
    if (!s->timing || (kind != 4 && (ordinal % 8) != 0) || s->evUsed >= s->evPool.size()) return -1;
    const int i = (int)s->evUsed++;
    s->evKind[i] = kind; s->evOwner[i] = &h; s->evOrd[i] = h.updLaunches;
    return i;
}
// `realBatches` = batches the current update has generated so far: pair number k processed a batch iff k < realBatches
static void timing_resolve(cogaps_session *s, uint64_t realBatches)
{
    for (size_t i = 0; i < s->evUsed; ++i) {
        const float ms = rt_event_ms(s->evPool[i]);
        HostSampler *h = s->evOwner[i];
        const bool real = s->evOrd[i] < realBatches;
        if (s->evKind[i] == 0) { if (real) { h->genMs += ms; h->genTimed++; } else { h->genNoopMs += ms; h->genNoopTimed++; } }
        else if (s->evKind[i] == 1) { if (real) { h->evalMs += ms; h->evalTimed++; } else { h->evalNoopMs += ms; h->evalNoopTimed++; } }
        else if (s->evKind[i] == 4) { s->syncMs += ms; s->syncTimed++; }     // sync (AP transpose / lookup tables): every launch is timed
        else { if (real) h->evalMs += ms; else h->evalNoopMs += ms; }      // second kernel of a split evaluation: same sample as the first
    }
    s->evUsed = 0;
}
#define LAUNCH_MAYBE_TIMED(slot, KERNEL, grid, block, ...) do { if ((slot) >= 0) RT_LAUNCH_TIMED(KERNEL, grid, block, s->stream, s->evPool[slot], __VA_ARGS__); else RT_LAUNCH(KERNEL, grid, block, s->stream, __VA_ARGS__); } while (0)
// The generator reads the sampler's record from device memory (gen_populate.h: a by-value SamplerDev made the compiler open the kernel
// with seven serial scalar-cache misses).  The copy is refreshed, on the session's stream, whenever the host's record changed.
static void sync_record(cogaps_session *s, HostSampler &h)
{
    if (!h.dRecord) h.dRecord = dalloc<SamplerDev>(1);
    if (h.recordValid && memcmp(&h.recordHeld, &h.d, sizeof(SamplerDev)) == 0) return;
    h.recordHeld = h.d;
    rt_h2d(h.dRecord, &h.recordHeld, sizeof(SamplerDev), s->stream);
    rt_sync(s->stream);
    h.recordValid = true;
}
// The one-chain split evaluation (data vectors of more than 4096 elements, dense model, product arithmetic) is ONE launch that decides
// (eval_kernel<EVAL_DECIDE>); the A*P updates it owes are carried out by the further workgroups of the NEXT generator launch
// (gen_apply_kernel) -- see eval_kernel.h.  The batched multi-chain launches keep the two-launch form (alpha, apply).
static bool split_one_launch(const HostSampler &h)
{
    static const bool twoLaunches = dev_env("COGAPS_SPLIT_TWO_LAUNCHES") != nullptr;      // dev builds: A/B against the two-launch form (alpha kernel, apply kernel)
    return !twoLaunches && !h.d.seq && !h.d.sparse && h.d.redW > 1024u;
}
static uint32_t apply_grid()
{
#if defined(COGAPS_EMUL)
    static const uint32_t g = 5u;        // (test-only emulator: a workgroup is a set of fibers, few of them keep the tests quick; 5 does not divide the items evenly)
#else
    static const uint32_t g = dev_env("COGAPS_APPLY_GRID") ? (uint32_t)atoi(dev_env("COGAPS_APPLY_GRID")) : 127u;      // dev builds: A/B of the update workgroups' number
#endif
    return g < 1u ? 1u : g;
}
// Updates in flight in this process (sessions stepped from several host threads: shards in flight, bench --chains-mode threads).  A chained
// launch wants every workgroup of its launch resident at once -- one per compute unit: its kernel's LDS -- so two chains' chained launches
// take turns on the chip and the hand-over inside each waits for the other's workgroups to leave (correct, and slower than two launches
// per batch each).  The chained form is therefore taken only by an update that runs alone ON ITS GPU: one count per device ordinal (sessions
// on different GPUs of one process do not switch each other's chained launch off), taken by the one-chain update and by the batched update
// alike; the count is looked at when an update begins.
static const int MAX_DEVICES = 64;
static std::atomic<int> &g_updatesRunning(int device) { static std::atomic<int> n[MAX_DEVICES]; return n[device >= 0 && device < MAX_DEVICES ? device : 0]; }
struct UpdateInFlight { int dev; explicit UpdateInFlight(int device) : dev(device) { g_updatesRunning(dev).fetch_add(1); } ~UpdateInFlight() { g_updatesRunning(dev).fetch_sub(1); } };
// The chained launch serves the one-chain fused evaluation (dense model, product arithmetic) whose workgroups are at least as large as
// the generator's and small enough for the generator's register budget (chain_kernel.h); everything else keeps two launches per batch.
static bool chain_eligible(const cogaps_session *s, const HostSampler &h);
// launch geometry of the split evaluation (data vectors of more than 4096 elements): `slices` workgroups of `bs` threads per proposal
// (512 threads fill the machine a little better than 1024; at most 16 slices fit the partials record)
static void split_geometry(const HostSampler &h, uint32_t &bs, uint32_t &slices)
{
    bs = std::max<uint32_t>(512u, h.d.redW / 16u);
    slices = std::min<uint32_t>(h.d.redW / bs, ((h.d.Npad >> 2) + bs - 1u) / bs);
}
static void launch_chain(cogaps_session *s, HostSampler &h)
{
    const int slot = timing_slot(s, h, 1, h.evalLaunches);
    const SamplerDev CG_CONSTANT *rec = (const SamplerDev CG_CONSTANT *)h.dRecord;
    const uint32_t parity = h.chainParity; h.chainParity ^= 1u;
    if (h.d.sparse) {
        // sparse model (sparse_kernels.h, chain_sparse_kernel): the launch has 512 threads per workgroup whatever the model's width
        // (255 evaluation workgroups + the generator = the chip's 256 compute units; a workgroup evaluates two proposals side by side where the
        // model's vectors take one round of flag words -- sparse_kernels.h, sp_grp -- so queues of up to 510 fit one pass; the dense launch keeps
        // 240 workgroups, profiles/r04_ab_chained_launch_not_kept.txt)
#if defined(COGAPS_EMUL)
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, CHAIN_EVAL_GRID) + 1u;      // (test-only emulator: few workgroups, so that both groups of an evaluation workgroup get proposals)
#else
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, s->computeUnits >= 256u ? 255u : CHAIN_EVAL_GRID) + 1u;
#endif
        const bool wide = h.d.Wn > cogaps_sparse_width(h.d.N), big = h.genWin == (uint32_t)GEN_WIN;
        if (GEN_WIN_WIDE != GEN_WIN && h.genWin == (uint32_t)GEN_WIN_WIDE) {
            if (wide) LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN_WIDE, true>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
            else LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN_WIDE, false>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
        }
        else if (big && wide) LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN, true>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
        else if (big) LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN, false>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
        else if (wide) LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN_HALF, true>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
        else LAUNCH_MAYBE_TIMED(slot, (chain_sparse_kernel<GEN_WIN_HALF, false>), grid, CHAIN_MAX_THREADS, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);
    } else if (h.d.redW > 1024u) {
        // split evaluation: the evaluation workgroups are a multiple of the slices per proposal (chain_kernel.h)
        uint32_t bs, slices; split_geometry(h, bs, slices);
        const uint32_t groups = std::max<uint32_t>(1u, std::min<uint32_t>(h.d.queueCap, CHAIN_EVAL_GRID / slices));
        const uint32_t grid = groups * slices + 1u;
        if (h.genWin == (uint32_t)GEN_WIN) LAUNCH_MAYBE_TIMED(slot, (chain_kernel<GEN_WIN, true>), grid, bs, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, slices, rec);
        else LAUNCH_MAYBE_TIMED(slot, (chain_kernel<GEN_WIN_HALF, true>), grid, bs, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, slices, rec);
#if defined(COGAPS_EMUL)
        RT_LAUNCH(chain_updates_kernel, 5, bs, s->stream, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, rec);      // (test-only emulator: the updates behind the launch)
#endif
    } else {
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, CHAIN_EVAL_GRID) + 1u;
        if (h.genWin == (uint32_t)GEN_WIN) LAUNCH_MAYBE_TIMED(slot, (chain_kernel<GEN_WIN, false>), grid, h.d.redW, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, 1u, rec);
        else LAUNCH_MAYBE_TIMED(slot, (chain_kernel<GEN_WIN_HALF, false>), grid, h.d.redW, h.d.lcgMul, h.d.lcgInc, h.d.gs, h.d.queue, h.chainGrans, h.d.chainSlots, h.d.queueCap, parity, 1u, rec);
    }
    h.evalLaunches++;      // (one launch per batch: counted with the evaluation launches, as its time is)
}
static void launch_gen(cogaps_session *s, HostSampler &h)
{
    const int slot = timing_slot(s, h, 0, h.genLaunches);
    const SamplerDev CG_CONSTANT *rec = (const SamplerDev CG_CONSTANT *)h.dRecord;
    if (split_one_launch(h)) {
        const uint32_t grid = 1u + apply_grid();
        if (h.genWin == (uint32_t)GEN_WIN) LAUNCH_MAYBE_TIMED(slot, gen_apply_kernel<GEN_WIN>, grid, GEN_WIN + 64, h.d.lcgMul, h.d.lcgInc, h.d.gs, (const unsigned long long *)h.d.eraseList, (const uint32_t *)h.d.queueUnits, h.d.eraseCap, h.d.queueCap, rec);
        else LAUNCH_MAYBE_TIMED(slot, gen_apply_kernel<GEN_WIN_HALF>, grid, GEN_WIN_HALF + 64, h.d.lcgMul, h.d.lcgInc, h.d.gs, (const unsigned long long *)h.d.eraseList, (const uint32_t *)h.d.queueUnits, h.d.eraseCap, h.d.queueCap, rec);
        h.genLaunches++;
        return;
    }
    if (h.genWin == (uint32_t)GEN_WIN) LAUNCH_MAYBE_TIMED(slot, gen_kernel<GEN_WIN>, 1, GEN_WIN + 64, h.d.lcgMul, h.d.lcgInc, h.d.gs, (const unsigned long long *)h.d.eraseList, (const uint32_t *)h.d.queueUnits, h.d.eraseCap, h.d.queueCap, rec);
    else LAUNCH_MAYBE_TIMED(slot, gen_kernel<GEN_WIN_HALF>, 1, GEN_WIN_HALF + 64, h.d.lcgMul, h.d.lcgInc, h.d.gs, (const unsigned long long *)h.d.eraseList, (const uint32_t *)h.d.queueUnits, h.d.eraseCap, h.d.queueCap, rec);
    h.genLaunches++;
}
static void launch_eval(cogaps_session *s, HostSampler &h)
{
    const int slot = timing_slot(s, h, 1, h.evalLaunches);
    const SamplerDev CG_CONSTANT *rec = (const SamplerDev CG_CONSTANT *)h.dRecord;      // (kept current by sync_record, as for the generator)
    if (h.d.seq) {
        // verification mode: one workgroup per proposal whatever the vector length, sums in the reference's order
        if (h.d.sparse) LAUNCH_MAYBE_TIMED(slot, eval_sparse_seq_kernel, std::min<uint32_t>(h.d.queueCap, SEQ_SPARSE_GRID), cogaps_sparse_width(h.d.N), h.d);
        else LAUNCH_MAYBE_TIMED(slot, eval_kernel<EVAL_SEQ>, std::min<uint32_t>(h.d.queueCap, 512u), EVAL_SEQ_BS, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, 1u, rec);
    } else if (h.d.sparse) {
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, 1024u);
        const uint32_t W = cogaps_sparse_width(h.d.N);
        if (h.d.Wn > W) LAUNCH_MAYBE_TIMED(slot, eval_sparse_kernel_wide, grid, W, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, rec);      // several rounds of flag words per vector
        else LAUNCH_MAYBE_TIMED(slot, eval_sparse_kernel, grid, W, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, rec);
    } else if (h.d.redW <= 1024u) {
        // one workgroup of W threads per proposal
        static const uint32_t fusedGrid = dev_env("COGAPS_FUSED_GRID") ? (uint32_t)atoi(dev_env("COGAPS_FUSED_GRID")) : 512u;      // dev builds: A/B of the launch size
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, fusedGrid);
        LAUNCH_MAYBE_TIMED(slot, eval_kernel<EVAL_FUSED>, grid, h.d.redW, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, 1u, rec);
    } else {
        // long data vectors: `slices` workgroups of `bs` threads per proposal, alpha kernel then apply kernel
        // (512 threads fill the machine a little better than 1024; at most 16 slices fit the partials record)
        uint32_t bs, slices; split_geometry(h, bs, slices);
        const uint32_t perWave = std::max<uint32_t>(1u, (512u * (1024u / bs)) / slices);   // two resident 1024-thread workgroups per compute unit
        const uint32_t grid = std::min<uint32_t>(h.d.queueCap, perWave) * slices;
        if (split_one_launch(h)) {
            // one launch: the slices' totals reach the proposal's last slice workgroup inside it (eval_kernel.h, EVAL_DECIDE); the A*P updates
            // follow beside the next generator launch (launch_gen)
            LAUNCH_MAYBE_TIMED(slot, eval_kernel<EVAL_DECIDE>, grid, bs, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, slices, rec);
        } else {
            int slot2 = -1;
            if (slot >= 0 && s->evUsed < s->evPool.size()) { slot2 = (int)s->evUsed++; s->evKind[slot2] = 3; s->evOwner[slot2] = &h; s->evOrd[slot2] = h.updLaunches; }
            LAUNCH_MAYBE_TIMED(slot, eval_kernel<EVAL_ALPHA>, grid, bs, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, slices, rec);
            LAUNCH_MAYBE_TIMED(slot2, eval_kernel<EVAL_APPLY>, grid, bs, (const PropRec *)h.d.queue, (const GenScalars *)h.d.gs, h.d.queueCap, slices, rec);
        }
    }
    h.evalLaunches++;
}

// A dependent kernel pair costs ~3.5 us per launch on the host and leaves a ~5 us bubble on the GPU when it is
// launched call by call; replayed from a captured graph the same pair leaves ~1.6 us per kernel boundary.  The
// kernels take their whole state through SamplerDev (by value), so one graph serves until a pointer in it
// changes (atom arrays regrown, seed buffer reallocated).
static const uint32_t GRAPH_PAIRS = 64;
static bool chain_eligible(const cogaps_session *s, const HostSampler &h)
{
    // (a device with fewer compute units than the launch has workgroups -- a partitioned GPU -- would run them in turns, the generator
    // workgroup last: correct, and slower than two launches)
    if (s->noChain || h.d.seq || h.chainOff) return false;
    if (g_updatesRunning(s->p.device).load() > 1 && !s->forceChain) return false;      // (another update -- a session's or a batch's -- is in flight on this session's GPU: see g_updatesRunning)
    if (h.d.sparse)      // sparse model (round 5): a launch of 512-thread workgroups, the evaluation keeps the model's width inside it
        return CHAIN_MAX_THREADS >= h.genWin + 64u && (s->forceChain || s->computeUnits >= std::min<uint32_t>(h.d.queueCap, CHAIN_EVAL_GRID) + 1u);
    uint32_t block = h.d.redW;
    if (h.d.redW > 1024u) {      // split evaluation (round 5): workgroups of 512 threads, at most as many slices per proposal as the launch has evaluation workgroups
        uint32_t slices; split_geometry(h, block, slices);
        if (!split_one_launch(h) || slices > 16u || s->noChainSplit) return false;
    }
    // (the split form's update items wait for deciding workgroups of HIGHER index -- eval_chain_updates --: every workgroup must be resident,
    // so COGAPS_FORCE_CHAIN does not take it on a device with fewer compute units than the launch has workgroups; the fused form, whose
    // evaluation workgroups never wait, runs in turns there)
    const bool forced = s->forceChain && h.d.redW <= 1024u;
    return block <= (uint32_t)CHAIN_MAX_THREADS && block >= h.genWin + 64u
           && (forced || s->computeUnits >= std::min<uint32_t>(h.d.queueCap, CHAIN_EVAL_GRID) + 1u);
}
// one batch step: the chained launch, or a generator launch and an evaluation launch
static void launch_pair(cogaps_session *s, HostSampler &h)
{
    if (h.chain) launch_chain(s, h); else { launch_gen(s, h); launch_eval(s, h); }
}
static void drop_graphs(HostSampler &h)
{
    if (h.graphValid) { rt_graph_destroy(h.graph); h.graphValid = false; }
    for (int k = 0; k < 2; ++k) if (h.chainGraphValid[k]) { rt_graph_destroy(h.chainGraph[k]); h.chainGraphValid[k] = false; }
}
// the captured run of GRAPH_PAIRS batch steps that starts at the sampler's current parity (an even number of launches: the parity after
// a replay is the parity before it)
static rt_graph &ensure_chain_graph(cogaps_session *s, HostSampler &h)
{
    static_assert(GRAPH_PAIRS % 2u == 0u, "a replay must leave the parity as it found it");
    if ((h.chainGraphValid[0] || h.chainGraphValid[1]) && memcmp(&h.graphKey, &h.d, sizeof(SamplerDev)) != 0) drop_graphs(h);
    const uint32_t k = h.chainParity;
    if (!h.chainGraphValid[k]) {
        const bool timing = s->timing; s->timing = false;
        const uint64_t g0 = h.genLaunches, e0 = h.evalLaunches;
        rt_capture_begin(s->stream);
        for (uint32_t b = 0; b < GRAPH_PAIRS; ++b) launch_chain(s, h);
        rt_capture_end(s->stream, h.chainGraph[k]);
        h.genLaunches = g0; h.evalLaunches = e0; s->timing = timing;
        memcpy(&h.graphKey, &h.d, sizeof(SamplerDev)); h.chainGraphValid[k] = true;
    }
    return h.chainGraph[k];
}
static void ensure_graph(cogaps_session *s, HostSampler &h)
{
    if (h.graphValid && memcmp(&h.graphKey, &h.d, sizeof(SamplerDev)) == 0) return;
    drop_graphs(h);
    const bool timing = s->timing; s->timing = false;              // no event records inside a capture
    const uint64_t g0 = h.genLaunches, e0 = h.evalLaunches;
    rt_capture_begin(s->stream);
    for (uint32_t b = 0; b < GRAPH_PAIRS; ++b) { launch_gen(s, h); launch_eval(s, h); }
    rt_capture_end(s->stream, h.graph);
    h.genLaunches = g0; h.evalLaunches = e0; s->timing = timing;
    memcpy(&h.graphKey, &h.d, sizeof(SamplerDev)); h.graphValid = true;
}

// Launch clock of the chained launches (gaps_state.h): the ring holds {entry of the first workgroup, end of the generator workgroup} of
// every chained launch, slot = tag of the batch it evaluated.  Read back with the progress word of a chunk (at most 4096 launches, the ring
// holds 8192): the launches that evaluated batches (clockSeen, upTo] are complete -- the batch generated last is evaluated by the next
// launch, unless the update is over.
static void clock_collect(cogaps_session *s, HostSampler &h, uint64_t epoch, bool flushed)
{
    const uint64_t upTo = flushed ? epoch : (epoch ? epoch - 1u : 0u);
    if (!s->timing || !h.d.launchClock) { h.clockSeen = upTo; return; }
    if (upTo <= h.clockSeen) return;
    h.clockHost.resize(2u * GAPS_CLOCK_RING);
    rt_d2h(h.clockHost.data(), h.d.launchClock, sizeof(unsigned long long) * 2u * GAPS_CLOCK_RING, s->stream); rt_sync(s->stream);
    if (h.clockHist.empty()) h.clockHist.assign(2048, 0);
    if (h.periodHist.empty()) h.periodHist.assign(2048, 0);
    const uint64_t from = upTo - h.clockSeen > GAPS_CLOCK_RING ? upTo - GAPS_CLOCK_RING : h.clockSeen;
    for (uint64_t e = from + 1u; e <= upTo; ++e) {
        const unsigned long long b = h.clockHost[2u * (uint32_t)(e % GAPS_CLOCK_RING)], en = h.clockHost[2u * (uint32_t)(e % GAPS_CLOCK_RING) + 1u];
        if (!b || en <= b || en - b > 100000000ull) continue;      // (an entry whose halves belong to different launches: a stale slot)
        const double us = 0.01 * (double)(en - b);
        h.clockSumUs += us; h.clockN++;
        h.clockHist[std::min<size_t>(h.clockHist.size() - 1u, (size_t)(us * 10.0))]++;
        // the period to the next launch (the batch behind this one is evaluated by the very next launch on the stream; a chunk's last launch is
        // followed by the host's progress read-back: periods beyond 100 us are such seams and left out)
        if (e + 1u <= upTo) {
            const unsigned long long nb = h.clockHost[2u * (uint32_t)((e + 1u) % GAPS_CLOCK_RING)];
            if (nb > b && nb - b < 10000ull) {
                const double pu = 0.01 * (double)(nb - b);
                h.periodSumUs += pu; h.periodN++;
                h.periodHist[std::min<size_t>(h.periodHist.size() - 1u, (size_t)(pu * 10.0))]++;
            }
        }
    }
    h.clockSeen = upTo;
}

// A chained launch's generator gave up waiting for a decision (GAPS_ERR_SPIN; chain_kernel.h, chain_recover_kernel): the stream is idle
// (the error word was read behind a synchronisation), the evaluation workgroups of the failed launch have all ended.  The batch is
// completed on the device, the scalars are put back, and the sampler goes on with two launches per batch -- for the rest of the session:
// what kept a workgroup from being scheduled for two seconds (a foreign kernel holding compute units, a debugger) may well still be there.
// The split evaluation's chained form (not the default) is not recovered: its deciding workgroups wait as well.
static bool chain_recover(cogaps_session *s, HostSampler &h, uint32_t nSteps)
{
    if (!h.chain || h.d.seq || (!h.d.sparse && h.d.redW > 1024u)) return false;
    // launch k of the update (k = 0, 1, ...) has parity start ^ (k & 1), evaluates batch k and generates batch k + 1: the launch that gave up
    // had generated nothing, so it is launch number nBatches and the batch it was carrying out lies in the queue copy of its parity
    const uint32_t parityFail = (h.chainParityStart + s->hGs->nBatches) & 1u;
    const SamplerDev CG_CONSTANT *rec = (const SamplerDev CG_CONSTANT *)h.dRecord;
    RT_LAUNCH(chain_recover_kernel, 1, 256, s->stream, h.d.gs, (const PropRec *)(h.d.queue + (size_t)parityFail * h.d.queueCap), (const unsigned long long *)h.chainGrans, nSteps, rec);
    read_gs(s, h);
    if (s->hGs->error) return false;      // (a decision is still missing: the update cannot be completed)
    h.chain = false; h.chainOff = true; h.chainRecoveries++;
    if (h.genWin != gen_window_for(h.genWin, 0.f, false)) { h.genWin = gen_window_for(h.genWin, 0.f, false); drop_graphs(h); }      // (the wide window exists for the chained sparse launch only)
    return true;
}

// AsynchronousGibbsSampler::update (AsynchronousGibbsSampler.h:88-122): batches of generate + evaluate
// until nSteps proposals have been processed.  The number of batches is data dependent, so (generate,
// evaluate) pairs are enqueued in chunks and the generator's progress word is read back per chunk;
// pairs enqueued past the end are no-ops (the generator flushes the last erase cache and reports
// qlen = 0).
static int run_update(cogaps_session *s, HostSampler &h, uint32_t nSteps, bool trace, uint32_t traceCap)
{
    SamplerDev &d = h.d;
    if (s->poisoned) return fail("this session was ended by a device error in an earlier update; its chain cannot be continued");
    UpdateInFlight inFlight(s->p.device);
    read_gs(s, h);
    GenScalars g = *s->hGs;
    grow_atoms(s, h, g.nAtoms + nSteps + 1024u);
    // seed stream for this update: exactly nSteps seeder outputs are consumed (ProposalQueue.cpp:12-15;
    // a failed attempt rolls the seeder back, Random.cpp:244-248)
    if (h.seedCap < (size_t)nSteps + 1) { rt_free(h.seeds); h.seedCap = (size_t)nSteps * 5 / 4 + 1024; h.seeds = dalloc<uint64_t>(h.seedCap); }
    if (h.hSeedCap < (size_t)nSteps + 1) { rt_free_host(h.hSeeds); h.hSeedCap = (size_t)nSteps * 5 / 4 + 1024; h.hSeeds = (uint64_t *)rt_malloc_host(h.hSeedCap * 8); }
    seed_take(s, h.hSeeds, nSteps);
    rt_h2d(h.seeds, h.hSeeds, (size_t)nSteps * 8, s->stream);
    d.seeds = h.seeds;
    if (trace) {
        if (h.traceCap < traceCap) {
            rt_free(d.trace); rt_free(d.traceBatchNproc); rt_free(d.traceBatchQlen);
            d.trace = dalloc<PropRec>(traceCap); d.traceBatchNproc = dalloc<uint32_t>(traceCap); d.traceBatchQlen = dalloc<uint32_t>(traceCap);
            h.traceCap = traceCap;
        }
    }
    g.annealTemp = h.anneal;
    g.nSteps = nSteps; g.nDone = 0; g.nBatches = 0; g.updateFlushed = 0; g.qlen = 0;
    g.traceOn = trace ? 1u : 0u; g.traceCount = 0; g.traceCap = trace ? traceCap : 0; g.traceBatchCount = 0;
    *s->hGs = g;
    rt_h2d(d.gs, s->hGs, sizeof(GenScalars), s->stream);
    const ChainSlot emptySlots[2] = {{0u, 0u}, {0u, 0u}};
    rt_h2d(d.chainSlots, emptySlots, sizeof(emptySlots), s->stream);      // (chained launch: both parities start from an empty queue)
    rt_sync(s->stream);
    if (nSteps == 0) return 0;
    sync_record(s, h);
    h.chain = chain_eligible(s, h);
    {   // the wide window: the chained sparse launch only (gen_window_for); tests take it from the first update on
        const bool wideOk = h.chain && h.d.sparse != 0u;
        uint32_t win = gen_window_for(h.genWin, 0.f, wideOk);
        if (wideOk && s->testWideWindow) win = (uint32_t)GEN_WIN_WIDE;
        if (win != h.genWin) { h.genWin = win; drop_graphs(h); }
    }
    h.chainParityStart = h.chainParity;
    h.updLaunches = 0;
    h.clockSeen = g.batchEpoch;      // (launch clock: the batches of this update carry the tags behind this one)
    // proposals per batch: the previous update of this sampler is the best predictor
    float avgq = h.stepsPerBatch > 1.f ? h.stepsPerBatch : (g.avgQueue > 1.f ? g.avgQueue : 1.f);
    bool firstChunk = true, topped = false;
    for (;;) {
        const uint32_t remaining = nSteps - s->hGs->nDone;
        // a pair enqueued past the end of the update is two wasted launches, a progress read-back is one short
        // pipeline bubble: enqueue slightly fewer pairs than the estimate says and converge on the tail
        uint32_t chunk = (uint32_t)((double)remaining / avgq * (firstChunk ? 0.97 : 1.0)) + (firstChunk ? 0u : 2u);
        if (chunk < 6u) chunk = 6u;
        if (chunk > 4096u) chunk = 4096u;
        firstChunk = false;
        uint32_t plain = chunk;
        if (rt_graphs_supported() && !s->noGraph && !trace && plain >= GRAPH_PAIRS) {
            if (!h.chain) ensure_graph(s, h);
            // HIP events cannot ride on replayed launches.  While timing is on, one replay of every chunk -- its position moves
            // through the chunk from update to update -- is issued as plain launches that carry events, so that the sample covers
            // the whole population of batches and not only the tail of each chunk (the remainder below).
            const uint32_t nRep = plain / GRAPH_PAIRS;
            const uint64_t rot = s->timing ? h.plainRotor++ : 0;                       // (every fourth chunk: the plain launches leave longer gaps than a replay)
            const uint32_t timedRep = (s->timing && (rot & 3u) == 0u) ? (uint32_t)(((rot >> 2) * 7u) % nRep) : 0xFFFFFFFFu;
            for (uint32_t r = 0; plain >= GRAPH_PAIRS; plain -= GRAPH_PAIRS, ++r) {
                if (r == timedRep) { for (uint32_t b = 0; b < GRAPH_PAIRS; ++b) { launch_pair(s, h); h.updLaunches++; } continue; }
                rt_graph_launch(h.chain ? ensure_chain_graph(s, h) : h.graph, s->stream); if (!h.chain) h.genLaunches += GRAPH_PAIRS; h.evalLaunches += GRAPH_PAIRS; h.updLaunches += GRAPH_PAIRS;
            }
        }
        for (uint32_t b = 0; b < plain; ++b) { launch_pair(s, h); h.updLaunches++; }
        // while the GPU works: the seeds of the next update (the other sampler's, about one per atom it holds; Poisson spread + margin)
        if (!topped) { HostSampler &o = (&h == &s->A) ? s->P : s->A; seed_top_up(s, (size_t)std::max(o.nAtoms, 10u) + (size_t)(6.0 * sqrt((double)std::max(o.nAtoms, 10u))) + 64u); topped = true; }
        read_gs(s, h);
        timing_resolve(s, s->hGs->nBatches);
        if (h.chain) clock_collect(s, h, s->hGs->batchEpoch, s->hGs->updateFlushed != 0);
        if (s->hGs->error == GAPS_ERR_SPIN && chain_recover(s, h, nSteps)) continue;      // (the batch completed, the update goes on with two launches per batch)
        if (s->hGs->error) { s->poisoned = true; return fail(std::string("device error code ") + std::to_string(s->hGs->error) + " in sampler " + h.name); }
        if (s->hGs->updateFlushed) break;
        if (s->hGs->nBatches > 0) avgq = std::max(1.f, (float)s->hGs->nDone / (float)s->hGs->nBatches);
    }
    h.nAtoms = s->hGs->nAtoms; h.avgQueue = s->hGs->avgQueue; h.batches += s->hGs->nBatches;
    if (s->hGs->nBatches >= 8u) h.stepsPerBatch = (float)nSteps / (float)s->hGs->nBatches;
    uint32_t win = gen_window_for(h.genWin, h.stepsPerBatch, h.chain && h.d.sparse != 0u);
    if (h.chain && h.d.sparse != 0u && s->testWideWindow) win = (uint32_t)GEN_WIN_WIDE;
    if (win != h.genWin) { h.genWin = win; drop_graphs(h); }      // (the captured launches carry the window)
    return 0;
}

static void do_sync(cogaps_session *s, HostSampler &dst, HostSampler &src)
{
    if (dst.d.sparse) {     // SparseNormalModel::sync = generateLookupTables (SparseNormalModel.cpp:27-31, 294-311)
        const uint32_t K = dst.d.K;
        if (dst.d.seq) RT_LAUNCH(sparse_tables_seq_kernel, K + K * (K + 1u) / 2u, 256, s->stream, dst.d);
        else LAUNCH_V(sparse_tables_kernel, dst.d.redW, K + K * (K + 1u) / 2u, s->stream, dst.d);
        return;
    }
    const uint32_t tilesX = (src.d.N + TR_TILE - 1) / TR_TILE, tilesY = (src.d.M + TR_TILE - 1) / TR_TILE;
    const int slot = timing_slot(s, dst, 4, 0);
    if (slot >= 0) s->syncBytes += 8ull * src.d.M * src.d.N;         // algorithmic traffic of a sync: M x N floats read and written
    LAUNCH_MAYBE_TIMED(slot, transpose_kernel, tilesX * tilesY, 256, (const float *)src.d.AP, dst.d.AP, src.d.M, src.d.N, src.d.Npad, dst.d.Npad, tilesX);
}

static float chisq_of(cogaps_session *s, HostSampler &h)
{
    if (h.d.seq) {      // the reference's order: one accumulator over the whole matrix, on the device
        if (h.d.sparse) RT_LAUNCH(chisq_sparse_seq_kernel, 1, 256, s->stream, h.d, h.partial);
        else RT_LAUNCH(chisq_seq_kernel, 1, 256, s->stream, h.d, (const float *)h.Sraw, h.partial);
        float c = 0.f;
        rt_d2h(&c, h.partial, 4, s->stream); rt_sync(s->stream);
        return h.d.sparse ? c * h.d.beta : c;
    }
    if (h.d.sparse && h.d.K <= 64u) LAUNCH_V(chisq_sparse_tiled_kernel, h.d.redW, (h.d.M + (uint32_t)SP_CHI_ROWS - 1u) / (uint32_t)SP_CHI_ROWS, s->stream, h.d, h.partial);      // SP_CHI_ROWS vectors per workgroup share the other matrix's rows
    else if (h.d.sparse) LAUNCH_V(chisq_sparse_kernel, h.d.redW, h.d.M, s->stream, h.d, h.partial);
    else LAUNCH_V(chisq_rows_kernel_s, h.d.redW, h.d.M, s->stream, h.d, (const float *)h.Sraw, h.partial);
    std::vector<float> part(h.d.M);
    rt_d2h(part.data(), h.partial, (size_t)h.d.M * 4, s->stream); rt_sync(s->stream);
    float c = 0.f;
    for (uint32_t j = 0; j < h.d.M; ++j) c += part[j];
    return h.d.sparse ? c * h.d.beta : c;
}

CG_KERNEL void debug_math_kernel(int fn, uint32_t mode, const float *x, float *y, uint32_t n)
{
    const uint32_t i = cg_bid() * cg_bdim() + cg_tid();
    if (i < n) y[i] = fn ? gm_expf_m(x[i], mode) : gm_logf_m(x[i], mode);
}

extern "C" {

int cogaps_debug_math(int fn, int mathMode, const float *x, float *y, uint32_t n, int on_device)
{
    try {
        if (mathMode < COGAPS_MATH_PORTABLE || mathMode > COGAPS_MATH_GLIBC_SSE2 || (fn != 0 && fn != 1)) return fail("bad function or math mode");
        if (!on_device) { for (uint32_t i = 0; i < n; ++i) y[i] = fn ? gm_expf_m(x[i], (uint32_t)mathMode) : gm_logf_m(x[i], (uint32_t)mathMode); return 0; }
        rt_stream_t st = rt_stream_create();
        rt_alloc_scope allocOn(st);
        float *dx = dalloc<float>(n), *dy = dalloc<float>(n);
        rt_h2d(dx, x, (size_t)n * 4, st);
        RT_LAUNCH(debug_math_kernel, (n + 255u) / 256u, 256, st, fn, (uint32_t)mathMode, (const float *)dx, dy, n);
        rt_d2h(y, dy, (size_t)n * 4, st); rt_sync(st);
        rt_free(dx); rt_free(dy); rt_stream_destroy(st);
        return 0;
    } catch (const std::exception &e) { return fail_exc(e); }
}

int cogaps_current_device(int *device)
{
    try { if (!device) return fail("null argument"); *device = rt_get_device(); return 0; } catch (const std::exception &e) { return fail_exc(e); }
}

int cogaps_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes)
{
    try {
        if (!freeBytes || !totalBytes) return fail("null argument");
        const int before = rt_get_device();
        if (device >= 0) rt_set_device(device);
        size_t f = 0, t = 0; rt_mem_info(&f, &t);
        if (device >= 0) rt_set_device(before);
        *freeBytes = (uint64_t)f; *totalBytes = (uint64_t)t;
        return 0;
    } catch (const std::exception &e) { return fail_exc(e); }
}

void cogaps_default_params(cogaps_params *p)
{
    memset(p, 0, sizeof(*p));
    p->nPatterns = 3; p->nIterations = 1000; p->maxThreads = 1; p->outputFrequency = 500;   // GapsParameters.h:79-111
    p->alphaA = 0.01f; p->alphaP = 0.01f; p->maxGibbsMassA = 100.f; p->maxGibbsMassP = 100.f;
    p->printMessages = 0; p->asynchronousUpdates = 1; p->whichMatrixFixed = 'N'; p->workerID = 1; p->device = -1;
}
const char *cogaps_last_error(void) { return g_last_error.c_str(); }
int cogaps_last_error_code(void) { return g_last_code; }
#ifndef COGAPS_SOURCE_HASH
#define COGAPS_SOURCE_HASH "unknown"      // (builds that do not go through csrc/Makefile: the test-only emulator)
#endif
const char *cogaps_source_hash(void) { return COGAPS_SOURCE_HASH; }
const char *cogaps_build_report(void)
{
    static const std::string rep = std::string("cogaps-amd 0.2 | ") + CG_PLATFORM_NAME + " | asynchronous Gibbs sampler, dense and sparse normal model | checkpoints: no | sources " + COGAPS_SOURCE_HASH;
    return rep.c_str();
}
int cogaps_checkpoints_enabled(void) { return 0; }
int cogaps_compiled_with_openmp(void) { return 0; }

cogaps_session *cogaps_session_create(const float *data, uint32_t nrow, uint32_t ncol, const cogaps_params *params, const float *unc, int data_on_device)
{
    cogaps_session *s = nullptr;
    try {
        const cogaps_params &p = *params;
        // The reference's distributed caller forces asynchronousUpdates = FALSE on its workers (R/DistributedCogaps.R:28-29) -- there to keep
        // BiocParallel workers single-threaded, not for the sampler's sake.  Documented deviation: a
        // distributed worker call (runningDistributed, i.e. subsetDim > 0 in cogaps_cpp, Cogaps.cpp:82) runs the asynchronous sampler anyway,
        // so that GWCoGAPS / scCoGAPS through the real R package reach this library.  Everywhere else FALSE is refused.
        if (!p.asynchronousUpdates && !p.runningDistributed) { fail("asynchronousUpdates=FALSE (SingleThreadedGibbsSampler) is not part of this library"); return nullptr; }
        if (p.nPatterns == 0 || nrow == 0 || ncol == 0) { fail("empty problem"); return nullptr; }
        if (p.whichMatrixFixed != 'N' && p.whichMatrixFixed != 'A' && p.whichMatrixFixed != 'P') { fail("whichMatrixFixed must be 'N', 'A' or 'P'"); return nullptr; }
        if (p.reductionMode != COGAPS_REDUCE_LANES && p.reductionMode != COGAPS_REDUCE_SEQ) { fail("reductionMode must be COGAPS_REDUCE_LANES or COGAPS_REDUCE_SEQ"); return nullptr; }
        if (p.mathMode < COGAPS_MATH_PORTABLE || p.mathMode > COGAPS_MATH_GLIBC_SSE2) { fail("mathMode must be COGAPS_MATH_PORTABLE, _GLIBC_FMA or _GLIBC_SSE2"); return nullptr; }
        if (p.mathMode != COGAPS_MATH_PORTABLE && p.reductionMode != COGAPS_REDUCE_SEQ) { fail("mathMode other than COGAPS_MATH_PORTABLE needs reductionMode COGAPS_REDUCE_SEQ (the verification mode)"); return nullptr; }
        if (p.pumpThreshold != 0 && p.pumpThreshold != 1) { fail("pumpThreshold must be 0 (PUMP_UNIQUE) or 1 (PUMP_CUT)"); return nullptr; }
        if (p.whichMatrixFixed != 'N' && p.fixedCols != 0 && (uint32_t)p.fixedCols != p.nPatterns) { fail("fixedPatterns must have nPatterns columns"); return nullptr; }
        if (p.subsetData && p.dataIndicesSubset) {
            // 1-based indices into the subset dimension (Matrix.cpp:55-62): genes are the rows of the data unless transposeData
            const uint32_t dim = (p.subsetGenes != 0) == (p.transposeData == 0) ? nrow : ncol;
            if (p.nSubset == 0) { fail("dataIndicesSubset is empty"); return nullptr; }
            for (uint32_t i = 0; i < p.nSubset; ++i)
                if (p.dataIndicesSubset[i] < 1u || p.dataIndicesSubset[i] > dim) { fail("dataIndicesSubset holds an index outside 1 .. " + std::to_string(dim)); return nullptr; }
        }
        rt_set_device(p.device);
        s = new cogaps_session(); s->computeUnits = rt_compute_units();
        s->p = p;
        s->p.device = rt_get_device();            // (-1 resolved: later calls from other host threads select the same GPU)
        g_updatesRunning(s->p.device);            // (the counters exist before any session steps)
        s->startTime = now_s();
        if (p.printMessages) { printf("Loading Data..."); fflush(stdout); }                  // GapsRunner.cpp:399
        if (p.subsetData && p.dataIndicesSubset) s->subset.assign(p.dataIndicesSubset, p.dataIndicesSubset + p.nSubset);
        s->stream = rt_stream_create();
        rt_alloc_scope allocOn(s->stream);
        std::vector<float> hostData, hostUnc;
        if (data_on_device) {          // device-resident input: stage through the host once, outside any timed region
            hostData.resize((size_t)nrow * ncol); rt_d2h(hostData.data(), data, hostData.size() * 4, s->stream);
            if (unc) { hostUnc.resize((size_t)nrow * ncol); rt_d2h(hostUnc.data(), unc, hostUnc.size() * 4, s->stream); }
            rt_sync(s->stream); data = hostData.data(); if (unc) unc = hostUnc.data();
        }
        s->hGs = (GenScalars *)rt_malloc_host(sizeof(GenScalars));
        // GapsRandomState(seed): seeder + lookup tables (Cogaps.cpp:158, Random.cpp:264-267)
        s->seeder.init(p.seed);
        std::vector<float> e, ei, qg; build_luts(e, ei, qg);
        s->dErf = dalloc<float>(e.size() + 8); s->dErfinv = dalloc<float>(ei.size() + 8); s->dQgamma = dalloc<float>(qg.size() + 8);      // (read as whole float4 chunks by the evaluation kernel's LDS staging)
        rt_h2d(s->dErf, e.data(), e.size() * 4, s->stream); rt_h2d(s->dErfinv, ei.data(), ei.size() * 4, s->stream); rt_h2d(s->dQgamma, qg.data(), qg.size() * 4, s->stream);
        std::vector<uint64_t> lm(2 * GEN_WIN_WIDE + 2), li(2 * GEN_WIN_WIDE + 2);      // (jumps of up to 2 * window steps; GEN_WIN_WIDE >= GEN_WIN)
        for (uint32_t k = 0; k < lm.size(); ++k) pcg_jump_coeffs(k, lm[k], li[k]);
        s->dLcgMul = dalloc<uint64_t>(lm.size()); s->dLcgInc = dalloc<uint64_t>(li.size());
        rt_h2d(s->dLcgMul, lm.data(), lm.size() * 8, s->stream); rt_h2d(s->dLcgInc, li.data(), li.size() * 8, s->stream);
        rt_sync(s->stream);
        // samplers: A on the transposed data with the subset flag flipped (GapsRunner.cpp:402-406);
        // seed order: A queue, P queue, runner (AsynchronousGibbsSampler.h:68, GapsRunner.cpp:437)
        build_sampler(s, s->A, 'A', data, nrow, ncol, unc, !p.transposeData, !p.subsetGenes, p.alphaA, p.maxGibbsMassA);
        build_sampler(s, s->P, 'P', data, nrow, ncol, unc, p.transposeData != 0, p.subsetGenes != 0, p.alphaP, p.maxGibbsMassP);
        s->nGenes = s->A.d.M; s->nSamples = s->P.d.M; s->K = p.nPatterns;
        if (s->A.d.N != s->P.d.M || s->P.d.N != s->A.d.M) throw std::runtime_error("internal: sampler dimensions do not mirror");
        s->A.d.other = s->P.d.mat; s->A.d.otherColPos = s->P.d.colPos;
        s->P.d.other = s->A.d.mat; s->P.d.otherColPos = s->A.d.colPos;
        if (p.useSparseOptimization) {
            s->A.d.orows = s->P.d.rows; s->A.d.oflags = s->P.d.mflags; s->A.d.oMw = s->P.d.Mw; s->A.d.oKpad = s->P.d.Kpad;
            s->P.d.orows = s->A.d.rows; s->P.d.oflags = s->A.d.mflags; s->P.d.oMw = s->A.d.Mw; s->P.d.oKpad = s->A.d.Kpad;
        }
        // processFixedMatrix (GapsRunner.cpp:329-350)
        if (p.whichMatrixFixed != 'N') {
            HostSampler &f = (p.whichMatrixFixed == 'A') ? s->A : s->P;
            if (!p.fixedPatterns || p.fixedRows != f.d.M) throw std::runtime_error("fixedPatterns must have one row per row of the fixed matrix");
            std::vector<float> m((size_t)f.d.K * f.d.Mpad, 0.f);
            for (uint32_t r = 0; r < f.d.M; ++r) for (uint32_t k = 0; k < f.d.K; ++k) m[(size_t)k * f.d.Mpad + r] = p.fixedPatterns[(size_t)r * f.d.K + k];
            if (f.d.sparse) {
                // HybridMatrix::operator=(Matrix) (HybridMatrix.cpp:70-84): the row copy takes the value, the column copy
                // takes it through add(): entries below epsilon are held at zero and unflagged
                std::vector<float> rw((size_t)f.d.M * f.d.Kpad, 0.f); std::vector<unsigned long long> fl((size_t)f.d.K * f.d.Mw, 0ull); std::vector<uint32_t> cnt(f.d.K, 0u);
                for (uint32_t r = 0; r < f.d.M; ++r) for (uint32_t k = 0; k < f.d.K; ++k) {
                    const float v = p.fixedPatterns[(size_t)r * f.d.K + k];
                    rw[(size_t)r * f.d.Kpad + k] = v;
                    if (0.f + v < GAPS_EPSILON) m[(size_t)k * f.d.Mpad + r] = 0.f;
                    else { fl[(size_t)k * f.d.Mw + (r >> 6)] |= 1ull << (r & 63u); ++cnt[k]; }
                }
                rt_h2d(f.d.rows, rw.data(), rw.size() * 4, s->stream); rt_h2d(f.d.mflags, fl.data(), fl.size() * 8, s->stream); rt_h2d(f.d.colPos, cnt.data(), cnt.size() * 4, s->stream);
            }
            rt_h2d(f.d.mat, m.data(), m.size() * 4, s->stream); rt_sync(s->stream);
            if (!f.d.sparse) RT_LAUNCH(count_pos_kernel, f.d.K, 256, s->stream, f.d);
        }
        s->Asum = dalloc<float>((size_t)s->K * s->A.d.Mpad); s->Asq = dalloc<float>((size_t)s->K * s->A.d.Mpad);
        s->Psum = dalloc<float>((size_t)s->K * s->P.d.Mpad); s->Psq = dalloc<float>((size_t)s->K * s->P.d.Mpad);
        s->pump = dalloc<float>((size_t)s->nGenes * s->K);
        s->runnerRng = pcg_from_seed(s->seeder.next());
        // ASampler.sync(PSampler); PSampler.sync(ASampler); extraInitialization x2 (GapsRunner.cpp:444-447)
        if (p.useSparseOptimization) { do_sync(s, s->A, s->P); do_sync(s, s->P, s->A); }     // the lookup tables; extraInitialization is a no-op (SparseNormalModel.cpp:34-37)
        else if (p.whichMatrixFixed != 'N') {
            // (no fixed matrix: both factors are all zero and AP = 0 is what the allocation holds -- 0.11 s of multiplying zeros per
            // session at the headline shape otherwise)
            for (HostSampler *h : {&s->A, &s->P}) {
                const uint32_t tilesI = (h->d.N + 255u) / 256u, tilesJ = (h->d.M + (uint32_t)INIT_JT - 1u) / (uint32_t)INIT_JT;
                RT_LAUNCH(init_ap_kernel, tilesI * tilesJ, 256, s->stream, h->d, tilesI);
            }
        }
        rt_sync(s->stream);
        if (p.printMessages) {                                                   // GapsRunner.cpp:412-426
            const unsigned el = (unsigned)(now_s() - s->startTime);
            printf("Done! (%02u:%02u:%02u)\n", el / 3600u, (el % 3600u) / 60u, el % 60u);
            if (!p.useSparseOptimization && s->A.dataSparsity > 0.80f) printf("\nWarning: data is more than 80%% sparse and sparseOptimization is not enabled\n");
        }
        if (p.runningDistributed) printf("    worker %u is starting!\n", p.workerID);         // :428-433
        fflush(stdout);
        return s;
    } catch (const std::exception &e) {
        fail_exc(e);
        if (s) cogaps_session_destroy(s);
        return nullptr;
    }
}

void cogaps_session_destroy(cogaps_session *s)
{
    if (!s) return;
    rt_graph_destroy(s->A.graph); rt_graph_destroy(s->P.graph);
    for (int k = 0; k < 2; ++k) { rt_graph_destroy(s->A.chainGraph[k]); rt_graph_destroy(s->P.chainGraph[k]); }
    free_sampler(s->A); free_sampler(s->P);
    rt_free(s->dErf); rt_free(s->dErfinv); rt_free(s->dQgamma); rt_free(s->dLcgMul); rt_free(s->dLcgInc);
    rt_free(s->Asum); rt_free(s->Asq); rt_free(s->Psum); rt_free(s->Psq); rt_free(s->pump);
    rt_free_host(s->hGs);
    for (auto &e : s->evPool) rt_event_destroy(e);
    if (s->ownsStream) rt_stream_destroy(s->stream);
    delete s;
}

// (the HIP current device belongs to the calling host thread: a session used from another thread than its creator's selects its GPU again)
#define SESSION_TRY try { rt_set_device(s->p.device); rt_alloc_scope allocOn_(s->stream);      // allocations made on behalf of a session fill on its stream
#define SESSION_END } catch (const std::exception &e) { return fail_exc(e); } return 0;

static HostSampler &pick(cogaps_session *s, char w) { return w == 'A' ? s->A : s->P; }

int cogaps_session_set_annealing(cogaps_session *s, float temp) { s->A.anneal = temp; s->P.anneal = temp; return 0; }

int cogaps_session_draw_steps(cogaps_session *s, uint32_t *nA, uint32_t *nP)
{
    const unsigned a = std::max(s->A.nAtoms, 10u), b = std::max(s->P.nAtoms, 10u);
    *nA = (uint32_t)host_poisson(s->runnerRng, (double)a);
    *nP = (uint32_t)host_poisson(s->runnerRng, (double)b);
    return 0;
}

int cogaps_session_update(cogaps_session *s, char which, uint32_t nSteps, cogaps_trace_rec *trace, uint32_t traceCap, uint32_t *nTrace,
                          uint32_t *batchNproc, uint32_t *batchQlen, uint32_t batchCap, uint32_t *nBatches)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    const bool tr = trace != nullptr && traceCap > 0;
    const uint32_t cap = tr ? std::max(traceCap, batchCap) : 0;
    if (run_update(s, h, nSteps, tr, cap)) return 1;
    if (tr) {
        const uint32_t n = std::min(s->hGs->traceCount, cap), nb = std::min(s->hGs->traceBatchCount, cap);
        std::vector<PropRec> rec(n);
        if (n) rt_d2h(rec.data(), h.d.trace, (size_t)n * sizeof(PropRec), s->stream);
        std::vector<uint32_t> bn(nb), bq(nb);
        if (nb) { rt_d2h(bn.data(), h.d.traceBatchNproc, (size_t)nb * 4, s->stream); rt_d2h(bq.data(), h.d.traceBatchQlen, (size_t)nb * 4, s->stream); }
        rt_sync(s->stream);
        for (uint32_t i = 0; i < n && i < traceCap; ++i) {
            cogaps_trace_rec &o = trace[i]; const PropRec &r = rec[i];
            o.pos = r.pos; o.rng_state = r.rng; o.atom1 = r.i1; o.atom2 = r.i2; o.r1 = r.r1; o.c1 = r.c1; o.r2 = r.r2; o.c2 = r.c2; o.type = r.type; o.batch = r.batch;
        }
        for (uint32_t i = 0; i < nb && i < batchCap; ++i) { if (batchNproc) batchNproc[i] = bn[i]; if (batchQlen) batchQlen[i] = bq[i]; }
        if (nTrace) *nTrace = s->hGs->traceCount;
        if (nBatches) *nBatches = s->hGs->traceBatchCount;
    }
    SESSION_END
}

int cogaps_session_sync(cogaps_session *s, char which)
{
    SESSION_TRY
    if (which == 'A') do_sync(s, s->A, s->P); else do_sync(s, s->P, s->A);
    SESSION_END
}

// the part of an iteration after the two updates: the proposal counter and GapsStatistics::update* (GapsRunner.cpp:297-313)
static void iterate_tail(cogaps_session *s, uint32_t nA, uint32_t nP, int sampling)
{
    const char f = s->p.whichMatrixFixed;
    s->totalUpdates += (uint64_t)nA + nP;
    if (sampling) {
        const uint32_t mode = (f == 'N') ? 0u : (f == 'P' ? 1u : 2u);   // P fixed -> updateA ; A fixed -> updateP
        RT_LAUNCH(stats_kernel, s->K, 256, s->stream, s->A.d, s->P.d, s->Asum, s->Asq, s->Psum, s->Psq, mode);
        s->statUpdates++;
        if (f == 'N' && s->p.takePumpSamples) { RT_LAUNCH(pump_kernel, (s->A.d.M + 255u) / 256u, 256, s->stream, s->A.d, s->pump); s->pumpUpdates++; }   // GapsRunner.cpp:308-313
    }
}

// updateSampler (GapsRunner.cpp:201-222) + GapsStatistics::update* (GapsRunner.cpp:299-312)
int cogaps_session_iterate(cogaps_session *s, uint32_t nA, uint32_t nP, int sampling)
{
    SESSION_TRY
    const char f = s->p.whichMatrixFixed;
    if (f != 'A') { if (run_update(s, s->A, nA, false, 0)) return 1; if (f != 'P') do_sync(s, s->P, s->A); }
    if (f != 'P') { if (run_update(s, s->P, nP, false, 0)) return 1; if (f != 'A') do_sync(s, s->A, s->P); }
    iterate_tail(s, nA, nP, sampling);
    SESSION_END
}

// the head of one iteration of runOnePhase (GapsRunner.cpp:280-295): interrupt poll, annealing temperature, Poisson step counts
static int iteration_head(cogaps_session *s, int phase, uint32_t it, uint32_t *nA, uint32_t *nP)
{
    if (s->p.interrupt && s->p.interrupt(s->p.interruptArg)) return fail("interrupted");
    if (phase == 1) {
        const float temp = (float)(2 * it) / (float)s->p.nIterations;
        cogaps_session_set_annealing(s, gm_min(1.f, temp));
    }
    const int rc = cogaps_session_draw_steps(s, nA, nP);
#if defined(COGAPS_EMUL)
    // TEST-ONLY emulator build (never in the product library: a stray environment variable must not be able to change a chain):
    // COGAPS_TEST_ZERO_STEPS="<workerID>:<iteration>" turns that worker's A update of that equilibration iteration into update(0) -- what a
    // Poisson draw of 0 (probability e^-10 per draw while a chain holds at most ten atoms) does -- after the draw, so the generators'
    // sequences are unchanged
    if (phase == 1) if (const char *z = getenv("COGAPS_TEST_ZERO_STEPS")) { unsigned wk = 0, zi = 0; if (sscanf(z, "%u:%u", &wk, &zi) == 2 && wk == s->p.workerID && zi == it) *nA = 0; }
#endif
    return rc;
}
// ... and its tail (:314-325): snapshots, status line / histories
static int iteration_tail(cogaps_session *s, int phase, uint32_t it)
{
    if ((s->p.snapshotPhase == 0 || s->p.snapshotPhase == phase) && s->p.snapshotFrequency > 0 && ((it + 1) % s->p.snapshotFrequency) == 0) {
        // GapsStatistics::takeSnapshot (GapsStatistics.h:188-202): getMatrix() of both samplers
        const int w = phase - 1;
        const size_t na = (size_t)s->nGenes * s->K, np_ = (size_t)s->nSamples * s->K;
        s->snapA[w].resize((size_t)(s->nSnap[w] + 1) * na); s->snapP[w].resize((size_t)(s->nSnap[w] + 1) * np_);
        if (cogaps_session_get_rows(s, 'A', s->snapA[w].data() + (size_t)s->nSnap[w] * na)) return 1;
        if (cogaps_session_get_rows(s, 'P', s->snapP[w].data() + (size_t)s->nSnap[w] * np_)) return 1;
        s->nSnap[w]++;
    }
    if (s->p.outputFrequency > 0 && ((it + 1) % s->p.outputFrequency) == 0) {        // displayStatus, :162-199
        const float cs = (s->p.whichMatrixFixed == 'P') ? chisq_of(s, s->A) : chisq_of(s, s->P);
        s->chisqHist.push_back(cs); s->atomHistA.push_back(s->A.nAtoms); s->atomHistP.push_back(s->P.nAtoms);
        if (s->p.printMessages) {
            // elapsed / estimated total time (estimatedPercentComplete, GapsRunner.cpp:127-159)
            const double nIter = (double)it + (phase == 2 ? (double)s->p.nIterations : 0.0), totalIter = 2.0 * (double)s->p.nIterations;
            auto est = [](double current, double total, double nAtoms) {
                const double coef = nAtoms / std::log(current);
                return coef * std::log(std::sqrt(2.0 * total * 3.14159265358979323846)) + total * coef * std::log(total) - total * coef; };
            const double done = est(nIter, nIter, s->A.nAtoms) + est(nIter, nIter, s->P.nAtoms), all = est(nIter, totalIter, s->A.nAtoms) + est(nIter, totalIter, s->P.nAtoms);
            const unsigned el = (unsigned)(now_s() - s->startTime);
            const double frac = done / all;
            const unsigned tt = (frac > 0.0 && std::isfinite((double)el / frac)) ? (unsigned)((double)el / frac) : 0u;
            printf("%u of %u, Atoms: %u(A), %u(P), ChiSq: %.0f, Time: %02u:%02u:%02u / %02u:%02u:%02u\n", it + 1, s->p.nIterations, s->A.nAtoms, s->P.nAtoms, cs,
                   el / 3600u, (el % 3600u) / 60u, el % 60u, tt / 3600u, (tt % 3600u) / 60u, tt % 60u);
            fflush(stdout);
        }
    }
    return 0;
}

// runOnePhase (GapsRunner.cpp:272-327) for iterations [firstIter, firstIter+n)
int cogaps_session_run_iterations(cogaps_session *s, int phase, uint32_t firstIter, uint32_t n, uint64_t *updates)
{
    SESSION_TRY
    const double t0 = now_s();
    if (s->p.printMessages && firstIter == 0 && n > 0) { printf(phase == 1 ? "-- Equilibration Phase --\n" : "-- Sampling Phase --\n"); fflush(stdout); }   // GapsRunner.cpp:446-457
    for (uint32_t it = firstIter; it < firstIter + n; ++it) {
        uint32_t nA, nP;
        if (iteration_head(s, phase, it, &nA, &nP)) return 1;
        if (cogaps_session_iterate(s, nA, nP, phase == 2)) return 1;
        if (updates) *updates += (uint64_t)nA + nP;
        if (iteration_tail(s, phase, it)) return 1;
    }
    rt_sync(s->stream);
    s->samplerSeconds += now_s() - t0;
    SESSION_END
}

// ================================================================================================================================
// Batched multi-chain launches: C independent chains -- the subsets of a GWCoGAPS / scCoGAPS job that share one GPU (nSets > #GPUs),
// or replicas -- stepped in lock-step by ONE stream.  One chain alone alternates between a one-workgroup generator launch and an
// evaluation launch of a few hundred workgroups, both latency-bound; with C chains the generator grid is C workgroups and the
// evaluation grid C times as many, so a step of the batch costs about what a step of one chain costs and the evaluation kernels
// finally move enough rows per launch to approach the HBM roofline.  Every chain is the same chain it would be on its own, bit for
// bit (tests/test_gpu_parity.py::test_batched_chains_equal_single_sessions): the kernels are the one-chain kernels' bodies, fed from
// a device array of SamplerDev records instead of a by-value argument.
// ================================================================================================================================
struct cogaps_batch {
    std::vector<cogaps_session *> ss;
    rt_stream_t stream = rt_stream_t();
    SamplerDev *dev[2] = {nullptr, nullptr};            // [0] the A samplers' records, [1] the P samplers'
    std::vector<SamplerDev> host[2];                    // what the device arrays hold
    rt_graph graph[2]; bool graphValid[2] = {false, false};
    // round 6: a side whose evaluation is the fused one steps as ONE chained launch for all chains (chain_kernel_multi) where every workgroup
    // of it is resident at once; the captured run of launches exists per starting parity, as for the one-chain form
    bool chain[2] = {false, false}; uint32_t chainParity[2] = {0, 0}; uint32_t chainWg[2] = {0, 0};
    rt_graph chainGraph[2][2]; bool chainGraphValid[2][2] = {{false, false}, {false, false}};
    GenScalars *hGs = nullptr;                          // pinned, [C]
    bool sparse = false; char fixed = 'N';
    uint64_t launches[2] = {0, 0};
    // HIP-event samples of the plain-launch remainder of each chunk
    bool timing = false; std::vector<rt_event_pair> ev; std::vector<int> evKind; std::vector<uint64_t> evOrd; size_t evUsed = 0;
    double genMs[2] = {0, 0}, evalMs[2] = {0, 0}; uint64_t genTimed[2] = {0, 0}, evalTimed[2] = {0, 0};
    uint64_t ord = 0;
    uint32_t genWin[2] = {GEN_WIN, GEN_WIN};           // per side, from the chain with the longest batches
};

static HostSampler &bpick(cogaps_batch *b, uint32_t c, int w) { return w == 0 ? b->ss[c]->A : b->ss[c]->P; }

// launch geometry shared by the chains of a batch (checked at creation: equal reduction widths and slice counts)
struct MultiGeom { uint32_t block, slices, wgPerChain; bool fused; };
static MultiGeom multi_geom(cogaps_batch *b, int w)
{
    const SamplerDev &d = bpick(b, 0, w).d;
    const uint32_t C = (uint32_t)b->ss.size();
    uint32_t minCap = 0xFFFFFFFFu; for (uint32_t c = 0; c < C; ++c) minCap = std::min(minCap, bpick(b, c, w).d.queueCap);
    MultiGeom g;
    if (b->sparse) { g.block = cogaps_sparse_width(d.N); g.slices = 1; g.fused = true; g.wgPerChain = std::min<uint32_t>(minCap, std::max<uint32_t>(64u, 1024u / C)); return g; }
    if (d.redW <= 1024u) { g.block = d.redW; g.slices = 1; g.fused = true; g.wgPerChain = std::min<uint32_t>(minCap, std::max<uint32_t>(64u, 2048u / C)); return g; }
    g.fused = false;
    g.block = std::max<uint32_t>(512u, d.redW / 16u);
    g.slices = std::min<uint32_t>(d.redW / g.block, ((d.Npad >> 2) + g.block - 1u) / g.block);
    const uint32_t perWave = std::max<uint32_t>(1u, (512u * (1024u / g.block)) / g.slices);      // (launch_eval: two resident 1024-thread workgroups per compute unit)
    g.wgPerChain = std::min<uint32_t>(minCap, std::max<uint32_t>(4u, (2u * perWave) / C + 1u)) * g.slices;
    return g;
}
// The chained form for a batch (chain_kernel.h, chain_kernel_multi): the dense model's fused evaluation (workgroups as large as the generator's),
// every workgroup of the launch resident at once -- compute units / chains per chain -- and no other update in flight on the GPU.  Taken
// for up to FOUR chains (64 workgroups each on the MI355X): measured +9 % at two chains, +2 % at four, -7 % at eight, -20 % at sixteen
// (profiles/r06_ab_chained_batch.txt) -- a chain's evaluation workgroups are alone on their compute units (the generator's LDS), where the
// batched evaluation launch packs three per unit.  COGAPS_NO_CHAIN switches it off (A/B, equality tests).
static uint32_t multi_chain_wg(cogaps_batch *b, int w, const MultiGeom &g)
{
    const cogaps_session *s0 = b->ss[0];
    const uint32_t C = (uint32_t)b->ss.size();
    if (s0->noChain || b->sparse || !g.fused || g.block > (uint32_t)CHAIN_MAX_THREADS || g.block < b->genWin[w] + 64u) return 0u;
    if (g_updatesRunning(s0->p.device).load() > 1) return 0u;
    for (cogaps_session *s : b->ss) if ((w == 0 ? s->A : s->P).d.seq) return 0u;
    const uint32_t perChain = std::min<uint32_t>(s0->computeUnits / C, CHAIN_EVAL_GRID + 1u);
    return perChain >= (CHAIN_EVAL_GRID >= 16u ? 64u : 3u) ? perChain : 0u;      // (the test-only emulator's launches have seven evaluation workgroups)
}
static void multi_launch_pair(cogaps_batch *b, int w, const MultiGeom &g, int slotGen, int slotEval, int slotEval2)
{
    const uint32_t C = (uint32_t)b->ss.size();
    const SamplerDev CG_CONSTANT *arr = (const SamplerDev CG_CONSTANT *)b->dev[w];
#define MLAUNCH(slot, KERNEL, grid, block, ...) do { if ((slot) >= 0) RT_LAUNCH_TIMED(KERNEL, grid, block, b->stream, b->ev[slot], __VA_ARGS__); else RT_LAUNCH(KERNEL, grid, block, b->stream, __VA_ARGS__); } while (0)
    if (b->chain[w]) {
        const uint32_t parity = b->chainParity[w]; b->chainParity[w] ^= 1u;
        if (b->genWin[w] == (uint32_t)GEN_WIN) MLAUNCH(slotEval, chain_kernel_multi<GEN_WIN>, C * b->chainWg[w], g.block, arr, parity, b->chainWg[w]);
        else MLAUNCH(slotEval, chain_kernel_multi<GEN_WIN_HALF>, C * b->chainWg[w], g.block, arr, parity, b->chainWg[w]);
        b->launches[w]++;
        return;
    }
    if (b->genWin[w] == (uint32_t)GEN_WIN) MLAUNCH(slotGen, gen_kernel_multi<GEN_WIN>, C, GEN_WIN + 64, arr);
    else MLAUNCH(slotGen, gen_kernel_multi<GEN_WIN_HALF>, C, GEN_WIN_HALF + 64, arr);
    if (b->sparse) MLAUNCH(slotEval, eval_sparse_kernel_multi, C * g.wgPerChain, g.block, arr, g.wgPerChain);
    else if (g.fused) MLAUNCH(slotEval, eval_kernel_multi<EVAL_FUSED>, C * g.wgPerChain, g.block, arr, 1u, g.wgPerChain);
    else {
        MLAUNCH(slotEval, eval_kernel_multi<EVAL_ALPHA>, C * g.wgPerChain, g.block, arr, g.slices, g.wgPerChain);
        MLAUNCH(slotEval2, eval_kernel_multi<EVAL_APPLY>, C * g.wgPerChain, g.block, arr, g.slices, g.wgPerChain);
    }
#undef MLAUNCH
    b->launches[w]++;
}

// AsynchronousGibbsSampler::update of sampler `w` (0 = A, 1 = P) of every chain, nSteps[c] proposals each
static int run_update_multi(cogaps_batch *b, int w, const std::vector<uint32_t> &nSteps)
{
    const uint32_t C = (uint32_t)b->ss.size();
    for (uint32_t c = 0; c < C; ++c)
        if (b->ss[c]->poisoned) return fail("chain " + std::to_string(c) + " of this batch was ended by a device error in an earlier update; the batch cannot be continued");
    UpdateInFlight inFlight(b->ss[0]->p.device);      // (a one-chain session stepped beside the batch on the same GPU keeps two launches per batch meanwhile)
    rt_alloc_scope allocOn(b->stream);
    for (uint32_t c = 0; c < C; ++c) rt_d2h(&b->hGs[c], bpick(b, c, w).d.gs, sizeof(GenScalars), b->stream);
    rt_sync(b->stream);
    std::vector<float> avgq(C); std::vector<char> done(C, 0);
    for (uint32_t c = 0; c < C; ++c) {
        cogaps_session *s = b->ss[c]; HostSampler &h = bpick(b, c, w);
        GenScalars &g = b->hGs[c];
        grow_atoms(s, h, g.nAtoms + nSteps[c] + 1024u);
        const uint32_t n = nSteps[c];
        if (h.seedCap < (size_t)n + 1) { rt_free(h.seeds); h.seedCap = (size_t)n * 5 / 4 + 1024; h.seeds = dalloc<uint64_t>(h.seedCap); }
        if (h.hSeedCap < (size_t)n + 1) { rt_free_host(h.hSeeds); h.hSeedCap = (size_t)n * 5 / 4 + 1024; h.hSeeds = (uint64_t *)rt_malloc_host(h.hSeedCap * 8); }
        seed_take(s, h.hSeeds, n);
        rt_h2d(h.seeds, h.hSeeds, (size_t)n * 8, b->stream);
        h.d.seeds = h.seeds;
        g.annealTemp = h.anneal;
        g.nSteps = n; g.nDone = 0; g.nBatches = 0; g.updateFlushed = 0; g.qlen = 0;
        g.traceOn = 0; g.traceCount = 0; g.traceCap = 0; g.traceBatchCount = 0;
        // update(0) -- a Poisson draw of 0 has probability e^-10 while a chain holds at most 10 atoms -- is a no-op in the reference
        // (AsynchronousGibbsSampler.h:94: the loop body never runs): the chain is done before the first launch and the lock-stepped
        // batch never waits for it (its generator workgroup sees nDone >= nSteps and leaves at once)
        if (n == 0) { g.updateFlushed = 1; done[c] = 1; }
        rt_h2d(h.d.gs, &g, sizeof(GenScalars), b->stream);
        avgq[c] = h.stepsPerBatch > 1.f ? h.stepsPerBatch : (g.avgQueue > 1.f ? g.avgQueue : 1.f);
        h.updLaunches = 0;
    }
    if (std::all_of(done.begin(), done.end(), [](char d) { return d != 0; })) { rt_sync(b->stream); return 0; }
    // the records the kernels read: re-uploaded when a pointer in one of them changed (atom tables regrown, seed buffer moved);
    // the captured graph stays valid -- its kernels' arguments are the array's address and the launch geometry
    bool changed = false;
    for (uint32_t c = 0; c < C; ++c) if (memcmp(&b->host[w][c], &bpick(b, c, w).d, sizeof(SamplerDev)) != 0) { b->host[w][c] = bpick(b, c, w).d; changed = true; }
    if (changed) rt_h2d(b->dev[w], b->host[w].data(), (size_t)C * sizeof(SamplerDev), b->stream);
    rt_sync(b->stream);
    const MultiGeom geo = multi_geom(b, w);
    b->chainWg[w] = multi_chain_wg(b, w, geo);
    b->chain[w] = b->chainWg[w] != 0u;
    for (uint32_t c = 0; c < C; ++c) bpick(b, c, w).chain = b->chain[w];      // (cogaps_session_chained reports what runs)
    if (b->chain[w]) {      // both parities of every chain start from an empty queue
        const ChainSlot emptySlots[2] = {{0u, 0u}, {0u, 0u}};
        for (uint32_t c = 0; c < C; ++c) rt_h2d(bpick(b, c, w).d.chainSlots, emptySlots, sizeof(emptySlots), b->stream);
        rt_sync(b->stream);
    }
    bool first = true, topped = false;
    for (;;) {
        // pairs to enqueue: what the slowest unfinished chain still needs (launches past the end of a chain's update are no-ops for it)
        uint32_t chunk = 0;
        for (uint32_t c = 0; c < C; ++c) if (!done[c]) {
            const uint32_t remaining = nSteps[c] - b->hGs[c].nDone;
            chunk = std::max(chunk, (uint32_t)((double)remaining / avgq[c] * (first ? 0.97 : 1.0)) + (first ? 0u : 2u));
        }
        chunk = std::min(std::max(chunk, 6u), 4096u);
        first = false;
        uint32_t plain = chunk;
        const bool noGraph = b->ss[0]->noGraph;
        if (rt_graphs_supported() && !noGraph && plain >= GRAPH_PAIRS && b->chain[w]) {
            // (an even number of launches per replay: the parity behind a replay is the parity before it)
            for (; plain >= GRAPH_PAIRS; plain -= GRAPH_PAIRS) {
                const uint32_t k = b->chainParity[w];
                if (!b->chainGraphValid[w][k]) {
                    const uint64_t l0 = b->launches[w];
                    rt_capture_begin(b->stream);
                    for (uint32_t i = 0; i < GRAPH_PAIRS; ++i) multi_launch_pair(b, w, geo, -1, -1, -1);
                    rt_capture_end(b->stream, b->chainGraph[w][k]);
                    b->launches[w] = l0; b->chainGraphValid[w][k] = true;
                }
                rt_graph_launch(b->chainGraph[w][k], b->stream); b->launches[w] += GRAPH_PAIRS; b->ord += GRAPH_PAIRS;
            }
        } else if (rt_graphs_supported() && !noGraph && plain >= GRAPH_PAIRS) {
            if (!b->graphValid[w]) {
                const uint64_t l0 = b->launches[w];
                rt_capture_begin(b->stream);
                for (uint32_t k = 0; k < GRAPH_PAIRS; ++k) multi_launch_pair(b, w, geo, -1, -1, -1);
                rt_capture_end(b->stream, b->graph[w]);
                b->launches[w] = l0; b->graphValid[w] = true;
            }
            for (; plain >= GRAPH_PAIRS; plain -= GRAPH_PAIRS) { rt_graph_launch(b->graph[w], b->stream); b->launches[w] += GRAPH_PAIRS; b->ord += GRAPH_PAIRS; }
        }
        for (uint32_t k = 0; k < plain; ++k) {
            int sg = -1, se = -1, se2 = -1;
            if (b->timing && (b->ord % 4u) == 0u && b->evUsed + 3 <= b->ev.size()) {
                if (b->chain[w]) { se = (int)b->evUsed++; b->evKind[se] = 1; }      // (one launch per step: timed as the evaluation launch, as the one-chain form's is)
                else {
                    sg = (int)b->evUsed++; b->evKind[sg] = 0; se = (int)b->evUsed++; b->evKind[se] = 1;
                    if (!geo.fused) { se2 = (int)b->evUsed++; b->evKind[se2] = 2; }
                }
            }
            multi_launch_pair(b, w, geo, sg, se, se2);
            b->ord++;
        }
        if (!topped) {      // while the GPU works: every chain's seeds for its next update (the other sampler's)
            for (uint32_t c = 0; c < C; ++c) { const uint32_t na = std::max(bpick(b, c, 1 - w).nAtoms, 10u); seed_top_up(b->ss[c], (size_t)na + (size_t)(6.0 * sqrt((double)na)) + 64u); }
            topped = true;
        }
        for (uint32_t c = 0; c < C; ++c) rt_d2h(&b->hGs[c], bpick(b, c, w).d.gs, sizeof(GenScalars), b->stream);
        rt_sync(b->stream);
        for (size_t i = 0; i < b->evUsed; ++i) {          // (sampled launches near the end of a chunk: most chains still have work there)
            const float ms = rt_event_ms(b->ev[i]);
            if (b->evKind[i] == 0) { b->genMs[w] += ms; b->genTimed[w]++; } else { b->evalMs[w] += ms; if (b->evKind[i] == 1) b->evalTimed[w]++; }
        }
        b->evUsed = 0;
        bool all = true;
        for (uint32_t c = 0; c < C; ++c) {
            const GenScalars &g = b->hGs[c];
            if (g.error) { b->ss[c]->poisoned = true; return fail(std::string("device error code ") + std::to_string(g.error) + " in sampler " + (w ? 'P' : 'A') + " of chain " + std::to_string(c)); }
            if (g.updateFlushed) done[c] = 1; else all = false;
            if (g.nBatches > 0) avgq[c] = std::max(1.f, (float)g.nDone / (float)g.nBatches);
        }
        if (all) break;
    }
    for (uint32_t c = 0; c < C; ++c) {
        HostSampler &h = bpick(b, c, w); const GenScalars &g = b->hGs[c];
        h.nAtoms = g.nAtoms; h.avgQueue = g.avgQueue; h.batches += g.nBatches;
        if (g.nBatches >= 8u) h.stepsPerBatch = (float)nSteps[c] / (float)g.nBatches;
    }
    float spb = 0.f; for (uint32_t c = 0; c < C; ++c) spb = std::max(spb, bpick(b, c, w).stepsPerBatch);
    const uint32_t win = gen_window_for(b->genWin[w], spb);
    if (win != b->genWin[w]) {
        b->genWin[w] = win;
        if (b->graphValid[w]) { rt_graph_destroy(b->graph[w]); b->graphValid[w] = false; }
        for (int k = 0; k < 2; ++k) if (b->chainGraphValid[w][k]) { rt_graph_destroy(b->chainGraph[w][k]); b->chainGraphValid[w][k] = false; }
    }
    return 0;
}

cogaps_batch *cogaps_batch_create(cogaps_session **sessions, uint32_t n)
{
    cogaps_batch *b = nullptr;
    try {
        if (!sessions || n == 0) { fail("no sessions"); return nullptr; }
        const cogaps_session *s0 = sessions[0];
        for (uint32_t c = 0; c < n; ++c) {
            const cogaps_session *s = sessions[c];
            if (!s) { fail("null session"); return nullptr; }
            if (!s->ownsStream) { fail("a session can be in one batch only"); return nullptr; }
            if (s->A.d.seq || s0->A.d.seq) { fail("the verification mode runs one chain at a time"); return nullptr; }
            // one launch geometry for all chains: the same model, reduction widths and slice counts (subsets of one job have them)
            if (s->p.useSparseOptimization != s0->p.useSparseOptimization || s->p.whichMatrixFixed != s0->p.whichMatrixFixed || s->p.device != s0->p.device
                || s->A.d.redW != s0->A.d.redW || s->P.d.redW != s0->P.d.redW || ((s->A.d.Npad >> 2) + 511u) / 512u != ((s0->A.d.Npad >> 2) + 511u) / 512u
                || ((s->P.d.Npad >> 2) + 511u) / 512u != ((s0->P.d.Npad >> 2) + 511u) / 512u
                || (s->p.useSparseOptimization && (cogaps_sparse_width(s->A.d.N) != cogaps_sparse_width(s0->A.d.N) || cogaps_sparse_width(s->P.d.N) != cogaps_sparse_width(s0->P.d.N))))      // (the dense kernels never see the sparse model's workgroup width: subsets of 4095 and 4096 rows share a batch)
            { fail("the sessions of a batch must share the model, the fixed matrix, the device and the evaluation launch shape (equal reduction widths)"); return nullptr; }
        }
        rt_set_device(s0->p.device);
        b = new cogaps_batch();
        b->ss.assign(sessions, sessions + n);
        b->sparse = s0->p.useSparseOptimization != 0; b->fixed = s0->p.whichMatrixFixed;
        b->stream = rt_stream_create();
        rt_alloc_scope allocOn(b->stream);
        // everything that can fail is done BEFORE a session is touched: a failed creation leaves every session as it was
        b->hGs = (GenScalars *)rt_malloc_host(sizeof(GenScalars) * n);
        for (int w = 0; w < 2; ++w) { b->dev[w] = dalloc<SamplerDev>(n); b->host[w].resize(n); memset(b->host[w].data(), 0, sizeof(SamplerDev) * n); }
        for (cogaps_session *s : b->ss) rt_sync(s->stream);
        for (cogaps_session *s : b->ss) {      // from here on the sessions run on the batch's stream, one after the other (nothing below throws)
            drop_graphs(s->A); drop_graphs(s->P);
            rt_stream_destroy(s->stream); s->stream = b->stream; s->ownsStream = false; s->A.chain = false; s->P.chain = false;      // (the batched launches keep two launches per step: cogaps_session_chained reports what runs)
        }
        return b;
    } catch (const std::exception &e) {
        fail_exc(e);
        if (b) {
            for (int w = 0; w < 2; ++w) rt_free(b->dev[w]);
            rt_free_host(b->hGs);
            if (b->stream) rt_stream_destroy(b->stream);
            delete b;
        }
        return nullptr;
    }
}

void cogaps_batch_destroy(cogaps_batch *b)
{
    if (!b) return;
    try { rt_sync(b->stream); } catch (...) { }
    for (cogaps_session *s : b->ss) { try { s->stream = rt_stream_create(); s->ownsStream = true; } catch (...) { } }      // the sessions outlive the batch
    for (int w = 0; w < 2; ++w) { rt_graph_destroy(b->graph[w]); rt_free(b->dev[w]); for (int k = 0; k < 2; ++k) if (b->chainGraphValid[w][k]) rt_graph_destroy(b->chainGraph[w][k]); }
    for (auto &e : b->ev) rt_event_destroy(e);
    rt_free_host(b->hGs);
    rt_stream_destroy(b->stream);
    delete b;
}

// runOnePhase for every chain of the batch: iterations [firstIter, firstIter + n) of `phase`; updates[c] += proposals of chain c
int cogaps_batch_run_iterations(cogaps_batch *b, int phase, uint32_t firstIter, uint32_t n, uint64_t *updates)
{
    try {
        rt_set_device(b->ss[0]->p.device);
        rt_alloc_scope allocOn(b->stream);
        const uint32_t C = (uint32_t)b->ss.size();
        for (uint32_t c = 0; c < C; ++c)
            if (b->ss[c]->poisoned) return fail("chain " + std::to_string(c) + " of this batch was ended by a device error in an earlier update; the batch cannot be continued");
        const double t0 = now_s();
        for (cogaps_session *s : b->ss)
            if (s->p.printMessages && firstIter == 0 && n > 0) { printf(phase == 1 ? "-- Equilibration Phase --\n" : "-- Sampling Phase --\n"); fflush(stdout); }
        std::vector<uint32_t> nA(C), nP(C);
        const char f = b->fixed;
        for (uint32_t it = firstIter; it < firstIter + n; ++it) {
            for (uint32_t c = 0; c < C; ++c)
                if (iteration_head(b->ss[c], phase, it, &nA[c], &nP[c])) return 1;
            // updateSampler (GapsRunner.cpp:201-222), every chain at once
            if (f != 'A') { if (run_update_multi(b, 0, nA)) return 1; if (f != 'P') for (cogaps_session *s : b->ss) do_sync(s, s->P, s->A); }
            if (f != 'P') { if (run_update_multi(b, 1, nP)) return 1; if (f != 'A') for (cogaps_session *s : b->ss) do_sync(s, s->A, s->P); }
            for (uint32_t c = 0; c < C; ++c) {
                iterate_tail(b->ss[c], nA[c], nP[c], phase == 2);
                if (updates) updates[c] += (uint64_t)nA[c] + nP[c];
                if (iteration_tail(b->ss[c], phase, it)) return 1;
            }
        }
        rt_sync(b->stream);
        const double dt = now_s() - t0;
        for (cogaps_session *s : b->ss) s->samplerSeconds += dt;
        return 0;
    } catch (const std::exception &e) { return fail_exc(e); }
}

int cogaps_batch_set_timing(cogaps_batch *b, int on)
{
    try {
        if (on && b->ev.empty()) { b->ev.resize(1536); b->evKind.resize(1536); for (auto &e : b->ev) rt_event_create(e); }
        if (on && !b->timing) for (int w = 0; w < 2; ++w) { b->genMs[w] = b->evalMs[w] = 0; b->genTimed[w] = b->evalTimed[w] = 0; }
        b->timing = on != 0;
        return 0;
    } catch (const std::exception &e) { return fail_exc(e); }
}

// mean HIP-event time of the sampled batched launches since cogaps_batch_set_timing(1), per sampler side (0 = A, 1 = P); the
// algorithmic bytes and proposal counts are the sessions' own (cogaps_session_perf_sampler)
int cogaps_batch_perf(cogaps_batch *b, int side, double *genUs, double *evalUs, uint64_t *sampled, uint64_t *launches)
{
    if (side < 0 || side > 1) return fail("side must be 0 (A) or 1 (P)");
    if (genUs) *genUs = b->genTimed[side] ? 1e3 * b->genMs[side] / (double)b->genTimed[side] : 0.0;
    if (evalUs) *evalUs = b->evalTimed[side] ? 1e3 * b->evalMs[side] / (double)b->evalTimed[side] : 0.0;
    if (sampled) *sampled = b->evalTimed[side];
    if (launches) *launches = b->launches[side];
    return 0;
}

int cogaps_session_natoms(cogaps_session *s, char which, uint32_t *n) { *n = pick(s, which).nAtoms; return 0; }
int cogaps_session_chisq(cogaps_session *s, char which, float *c) { SESSION_TRY *c = chisq_of(s, pick(s, which)); SESSION_END }
int cogaps_session_dims(cogaps_session *s, char which, uint32_t *M, uint32_t *N, uint32_t *K) { HostSampler &h = pick(s, which); *M = h.d.M; *N = h.d.N; *K = h.d.K; return 0; }
int cogaps_session_avg_queue(cogaps_session *s, char which, float *a) { *a = pick(s, which).avgQueue; return 0; }

int cogaps_session_get_matrix(cogaps_session *s, char which, float *out)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    std::vector<float> m((size_t)h.d.K * h.d.Mpad);
    rt_d2h(m.data(), h.d.mat, m.size() * 4, s->stream); rt_sync(s->stream);
    for (uint32_t r = 0; r < h.d.M; ++r) for (uint32_t k = 0; k < h.d.K; ++k) out[(size_t)r * h.d.K + k] = m[(size_t)k * h.d.Mpad + r];
    SESSION_END
}
int cogaps_session_get_rows(cogaps_session *s, char which, float *out)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    if (!h.d.sparse) return cogaps_session_get_matrix(s, which, out);
    std::vector<float> m((size_t)h.d.M * h.d.Kpad);
    rt_d2h(m.data(), h.d.rows, m.size() * 4, s->stream); rt_sync(s->stream);
    for (uint32_t r = 0; r < h.d.M; ++r) memcpy(out + (size_t)r * h.d.K, m.data() + (size_t)r * h.d.Kpad, (size_t)h.d.K * 4);
    SESSION_END
}
int cogaps_session_get_ap(cogaps_session *s, char which, float *out)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    if (h.d.sparse) { memset(out, 0, (size_t)h.d.M * h.d.N * 4); return 0; }     // the sparse model keeps no A*P cache
    std::vector<float> m((size_t)h.d.M * h.d.Npad);
    rt_d2h(m.data(), h.d.AP, m.size() * 4, s->stream); rt_sync(s->stream);
    for (uint32_t r = 0; r < h.d.M; ++r) memcpy(out + (size_t)r * h.d.N, m.data() + (size_t)r * h.d.Npad, (size_t)h.d.N * 4);
    SESSION_END
}
int cogaps_session_get_atoms(cogaps_session *s, char which, uint64_t *pos, float *mass, uint32_t *left, uint32_t *right)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    read_gs(s, h);
    const uint32_t n = s->hGs->nAtoms;
    std::vector<uint32_t> vec(n); std::vector<AtomRec> atoms(h.d.atomCap);
    if (n) rt_d2h(vec.data(), h.d.vec, (size_t)n * 4, s->stream);
    rt_d2h(atoms.data(), h.d.atoms, (size_t)h.d.atomCap * sizeof(AtomRec), s->stream); rt_sync(s->stream);
    for (uint32_t i = 0; i < n; ++i) {
        const AtomRec &a = atoms[vec[i]];
        if (pos) pos[i] = a.pos;
        if (mass) mass[i] = a.mass;
        if (left) left[i] = a.left != CG_NONE ? atoms[a.left].idx : CG_NONE;
        if (right) right[i] = a.right != CG_NONE ? atoms[a.right].idx : CG_NONE;
    }
    SESSION_END
}

// test hook: the atomic domain's redundant state -- links symmetric, vec / idx inverse of each other, every record's cached neighbour
// positions and right-neighbour mass equal to the neighbours' own -- *violations = number of broken invariants
int cogaps_session_debug_check_domain(cogaps_session *s, char which, uint32_t *violations)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    read_gs(s, h);
    const uint32_t n = s->hGs->nAtoms;
    std::vector<uint32_t> vec(n); std::vector<AtomRec> atoms(h.d.atomCap);
    if (n) rt_d2h(vec.data(), h.d.vec, (size_t)n * 4, s->stream);
    rt_d2h(atoms.data(), h.d.atoms, (size_t)h.d.atomCap * sizeof(AtomRec), s->stream); rt_sync(s->stream);
    uint32_t bad = 0, fronts = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t hd = vec[i];
        if (hd >= h.d.atomCap) { ++bad; continue; }
        const AtomRec &a = atoms[hd];
        if (a.idx != i) ++bad;
        if (a.left != CG_NONE) { const AtomRec &l = atoms[a.left]; if (l.right != hd || a.lpos != l.pos || !(l.pos < a.pos)) ++bad; } else { ++fronts; if (s->hGs->front != hd) ++bad; }
        if (a.right != CG_NONE) { const AtomRec &r = atoms[a.right]; if (r.left != hd || a.rpos != r.pos || gm_f2u(a.rmass) != gm_f2u(r.mass)) ++bad; }
    }
    if (n && fronts != 1) ++bad;
    *violations = bad;
    SESSION_END
}

int cogaps_session_finish(cogaps_session *s, cogaps_result *out)
{
    SESSION_TRY
    memset(out, 0, sizeof(*out));
    if (s->poisoned) return fail("this session was ended by a device error; it holds no result (the state getters still read the half-applied state, for diagnosis)");
    out->nGenes = s->nGenes; out->nSamples = s->nSamples; out->nPatterns = s->K;
    const uint32_t K = s->K;
    auto fetch = [&](float *dptr, size_t n) { std::vector<float> v(n); rt_d2h(v.data(), dptr, n * 4, s->stream); rt_sync(s->stream); return v; };
    std::vector<float> As = fetch(s->Asum, (size_t)K * s->A.d.Mpad), Aq = fetch(s->Asq, (size_t)K * s->A.d.Mpad);
    std::vector<float> Ps = fetch(s->Psum, (size_t)K * s->P.d.Mpad), Pq = fetch(s->Psq, (size_t)K * s->P.d.Mpad);
    const float n = (float)s->statUpdates;
    auto fill = [&](uint32_t rows, uint32_t Mpad, const std::vector<float> &sum, const std::vector<float> &sq, float *&mean, float *&sd) {
        mean = (float *)malloc((size_t)rows * K * 4 + 4); sd = (float *)malloc((size_t)rows * K * 4 + 4);
        for (uint32_t i = 0; i < rows; ++i) for (uint32_t k = 0; k < K; ++k) {       // GapsStatistics.cpp:13-59
            const float a = sum[(size_t)k * Mpad + i], q = sq[(size_t)k * Mpad + i];
            mean[(size_t)i * K + k] = a / n;
            const float meanTerm = (a * a) / n, numer = gm_max(0.f, q - meanTerm);
            sd[(size_t)i * K + k] = sqrtf(numer / (n - 1.f));
        }
    };
    fill(s->nGenes, s->A.d.Mpad, As, Aq, out->Amean, out->Asd);
    fill(s->nSamples, s->P.d.Mpad, Ps, Pq, out->Pmean, out->Psd);
    out->nHistory = (uint32_t)s->chisqHist.size();
    out->chisqHistory = (float *)malloc(out->nHistory * 4 + 4); out->atomHistoryA = (uint32_t *)malloc(out->nHistory * 4 + 4); out->atomHistoryP = (uint32_t *)malloc(out->nHistory * 4 + 4);
    memcpy(out->chisqHistory, s->chisqHist.data(), out->nHistory * 4); memcpy(out->atomHistoryA, s->atomHistA.data(), out->nHistory * 4); memcpy(out->atomHistoryP, s->atomHistP.data(), out->nHistory * 4);
    out->totalUpdates = s->totalUpdates; out->seed = s->p.seed;
    out->averageQueueLengthA = s->A.avgQueue; out->averageQueueLengthP = s->P.avgQueue;
    out->samplerSeconds = s->samplerSeconds; out->totalRunningTime = (uint32_t)(now_s() - s->startTime);   // GapsRunner.cpp:463
    out->meanChiSq = 0.f;                                                             // GapsRunner.cpp:478-484
    if (s->p.whichMatrixFixed == 'N' && s->statUpdates > 0) {
        const float n2 = (float)s->statUpdates * (float)s->statUpdates;
        if (s->P.d.seq) {
            RT_LAUNCH(mean_chisq_seq_kernel, 1, 256, s->stream, s->P.d, (const float *)s->P.Sraw, (const float *)s->Asum, (const float *)s->Psum, s->A.d.Mpad, n2, s->P.partial);
            rt_d2h(&out->meanChiSq, s->P.partial, 4, s->stream); rt_sync(s->stream);
        } else {
        LAUNCH_V(mean_chisq_rows_kernel, s->P.d.redW, s->P.d.M, s->stream, s->P.d, (const float *)s->P.Sraw, (const float *)s->Asum, (const float *)s->Psum, s->A.d.Mpad, n2, s->P.partial);
        std::vector<float> part(s->P.d.M);
        rt_d2h(part.data(), s->P.partial, (size_t)s->P.d.M * 4, s->stream); rt_sync(s->stream);
        float c = 0.f; for (uint32_t j = 0; j < s->P.d.M; ++j) c += part[j];
        out->meanChiSq = c;
        }
    }
    if (s->p.takePumpSamples) {                                                       // GapsRunner.cpp:487-492, GapsStatistics.cpp:113-131
        const size_t na = (size_t)s->nGenes * K;
        std::vector<float> pm = fetch(s->pump, na);
        const float denom = s->pumpUpdates != 0 ? (float)s->pumpUpdates : 1.f;
        out->pumpMatrix = (float *)malloc(na * 4 + 4); out->meanPatternAssignment = (float *)calloc(na + 1, 4);
        for (size_t t = 0; t < na; ++t) out->pumpMatrix[t] = pm[t] / denom;
        for (uint32_t i = 0; i < s->nGenes; ++i) {                                    // meanPattern(): the same rule on Amean
            float maxV = 0.f; uint32_t maxI = 0;
            for (uint32_t j = 0; j < K; ++j) { const float v = out->Amean[(size_t)i * K + j]; if (maxV < v) { maxV = v; maxI = j; } }
            out->meanPatternAssignment[(size_t)i * K + maxI] += 1.f;
        }
    }
    out->nEquilibrationSnapshots = s->nSnap[0]; out->nSamplingSnapshots = s->nSnap[1];
    auto dup = [](const std::vector<float> &v) { float *p = (float *)malloc(v.size() * 4 + 4); if (!v.empty()) memcpy(p, v.data(), v.size() * 4); return p; };
    out->equilibrationSnapshotsA = dup(s->snapA[0]); out->equilibrationSnapshotsP = dup(s->snapP[0]);
    out->samplingSnapshotsA = dup(s->snapA[1]); out->samplingSnapshotsP = dup(s->snapP[1]);
        if (s->p.runningDistributed) {                                                     // GapsRunner.cpp:494-500
        const unsigned el = (unsigned)(now_s() - s->startTime);
        printf("    worker %u is finished! Time: %02u:%02u:%02u\n", s->p.workerID, el / 3600u, (el % 3600u) / 60u, el % 60u); fflush(stdout);
    }
    SESSION_END
}

// development aid: relaunch the evaluation (kind 1) or generator (kind 0) kernel `n` times on the current
// device state and return the mean wall time per launch in microseconds (the chain state is garbage afterwards)
int cogaps_session_debug_replay(cogaps_session *s, char which, int kind, uint32_t n, uint32_t dbgFlags, double *usPerLaunch)
{
    SESSION_TRY
    HostSampler &h = pick(s, which);
    {   // arm a fresh update of 4096 steps and generate one batch so that the queue is populated
        read_gs(s, h);
        GenScalars g = *s->hGs;
        const uint32_t steps = kind >= 2 ? 65536u : 4096u;       // kinds 2 (generator alone) / 3 (pairs) run many real batches
        if (h.seedCap < steps) { rt_free(h.seeds); h.seedCap = 2u * steps; h.seeds = dalloc<uint64_t>(h.seedCap); }
        std::vector<uint64_t> sd(steps); seed_take(s, sd.data(), sd.size());
        rt_h2d(h.seeds, sd.data(), sd.size() * 8, s->stream); h.d.seeds = h.seeds;
        g.nSteps = steps; g.nDone = 0; g.updateFlushed = 0; g.qlen = 0; g.traceOn = 0;
        *s->hGs = g; rt_h2d(h.d.gs, s->hGs, sizeof(GenScalars), s->stream);
        sync_record(s, h);
        launch_gen(s, h);
        if (kind == 0 || kind == 3) launch_eval(s, h);
    }
    rt_sync(s->stream);
    h.d.dbg = dbgFlags;
    sync_record(s, h);
    const double t0 = now_s();
    for (uint32_t i = 0; i < n; ++i) { if (kind == 0 || kind == 3) { launch_gen(s, h); launch_eval(s, h); } else if (kind == 2) launch_gen(s, h); else launch_eval(s, h); }
    rt_sync(s->stream);
    *usPerLaunch = 1e6 * (now_s() - t0) / (double)n;
    h.d.dbg = 0;
    SESSION_END
}
int cogaps_session_debug_prof(cogaps_session *s, char which, uint64_t *out16)
{
    SESSION_TRY
    read_gs(s, pick(s, which));
    for (int i = 0; i < 16; ++i) out16[i] = s->hGs->prof[i];
    SESSION_END
}
#if defined(GEN_TIMELINE)
extern "C" int cogaps_debug_timeline(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int cogaps_debug_chain_timeline(unsigned long long *wgs, unsigned long long *gen)
{
    if (hipMemcpyFromSymbol(wgs, HIP_SYMBOL(g_chain_rt), sizeof(unsigned long long) * 1024) != hipSuccess) return 1;
    return (int)hipMemcpyFromSymbol(gen, HIP_SYMBOL(g_chain_gen), sizeof(unsigned long long) * 8);
}
extern "C" int cogaps_debug_ahead_why(unsigned long long *out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ahead_why), sizeof(unsigned long long) * 8); }
extern "C" int cogaps_debug_chain_log(unsigned long long *out, unsigned int *n)
{
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_chain_log_n), sizeof(unsigned int)) != hipSuccess) return 1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_log), sizeof(unsigned long long) * (size_t)GEN_LOG_N * 8);
}
extern "C" int cogaps_debug_eval_timeline(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eval_timeline), sizeof(unsigned long long) * (size_t)n);
}
#endif
int cogaps_session_set_timing(cogaps_session *s, int on)
{
    SESSION_TRY
    if (on && !s->evInit) {
        s->evPool.resize(2048); s->evKind.resize(2048); s->evOwner.resize(2048); s->evOrd.resize(2048);
        for (auto &e : s->evPool) rt_event_create(e);
        s->evInit = true;
    }
    if (on && !s->timing) {      // a new window: the sampled times are scaled to the batches processed from here on
        for (HostSampler *h : {&s->A, &s->P}) {
            h->batchesAtTimingOn = h->batches;
            h->evalMs = h->genMs = h->evalNoopMs = h->genNoopMs = 0; h->evalTimed = h->genTimed = h->evalNoopTimed = h->genNoopTimed = 0;
            h->clockHist.assign(2048, 0); h->clockSumUs = 0; h->clockN = 0;
            h->periodHist.assign(2048, 0); h->periodSumUs = 0; h->periodN = 0;
        }
        s->syncMs = 0; s->syncTimed = 0; s->syncBytes = 0;
    }
    s->timing = on != 0;
    SESSION_END
}
static void add_perf(cogaps_session *s, HostSampler *h, cogaps_perf *out)
{
    read_gs(s, *h);
    out->evalBytes += s->hGs->evalBytes; out->proposalsQueued += s->hGs->evalProps;
    out->evalLaunches += h->evalLaunches; out->genLaunches += h->genLaunches; out->batches += h->batches;
    // sampled event timing scaled to the batches processed since timing was switched on
    const double win = (double)(h->batches - h->batchesAtTimingOn);
    out->timedBatches += h->batches - h->batchesAtTimingOn; out->evalTimed += h->evalTimed; out->genTimed += h->genTimed;
    if (h->evalTimed) out->evalMs += h->evalMs * win / (double)h->evalTimed;
    if (h->genTimed) out->genMs += h->genMs * win / (double)h->genTimed;
    out->evalNoopMs += h->evalNoopMs; out->evalNoopTimed += h->evalNoopTimed;
    out->genNoopMs += h->genNoopMs; out->genNoopTimed += h->genNoopTimed;
}
int cogaps_session_perf(cogaps_session *s, cogaps_perf *out)
{
    SESSION_TRY
    memset(out, 0, sizeof(*out));
    rt_sync(s->stream); timing_resolve(s, 0);        // sync launches timed since the last update (generator / evaluation events are resolved per chunk)
    for (HostSampler *h : {&s->A, &s->P}) add_perf(s, h, out);
    out->syncMs = s->syncMs; out->syncTimed = s->syncTimed; out->syncBytes = s->syncBytes;
    SESSION_END
}
int cogaps_session_perf_sampler(cogaps_session *s, char which, cogaps_perf *out)
{
    SESSION_TRY
    memset(out, 0, sizeof(*out));
    add_perf(s, &pick(s, which), out);
    SESSION_END
}

int cogaps_session_launch_clock(cogaps_session *s, char which, double *meanUs, double *percentilesUs, uint64_t *launches)
{
    SESSION_TRY
    if (!meanUs || !percentilesUs || !launches) return fail("null argument");
    HostSampler &h = pick(s, which);
    *launches = h.clockN; *meanUs = h.clockN ? h.clockSumUs / (double)h.clockN : 0.0;
    static const double q[5] = {0.10, 0.50, 0.75, 0.90, 0.99};
    for (int k = 0; k < 5; ++k) {
        percentilesUs[k] = 0.0;
        if (!h.clockN) continue;
        const uint64_t want = (uint64_t)(q[k] * (double)h.clockN); uint64_t acc = 0;
        for (size_t b = 0; b < h.clockHist.size(); ++b) { acc += h.clockHist[b]; if (acc > want) { percentilesUs[k] = 0.1 * ((double)b + 0.5); break; } }
    }
    SESSION_END
}
int cogaps_session_launch_period(cogaps_session *s, char which, double *meanUs, double *percentilesUs, uint64_t *launches)
{
    SESSION_TRY
    if (!meanUs || !percentilesUs || !launches) return fail("null argument");
    HostSampler &h = pick(s, which);
    *launches = h.periodN; *meanUs = h.periodN ? h.periodSumUs / (double)h.periodN : 0.0;
    static const double q[5] = {0.10, 0.50, 0.75, 0.90, 0.99};
    for (int k = 0; k < 5; ++k) {
        percentilesUs[k] = 0.0;
        if (!h.periodN) continue;
        const uint64_t want = (uint64_t)(q[k] * (double)h.periodN); uint64_t acc = 0;
        for (size_t b = 0; b < h.periodHist.size(); ++b) { acc += h.periodHist[b]; if (acc > want) { percentilesUs[k] = 0.1 * ((double)b + 0.5); break; } }
    }
    SESSION_END
}
int cogaps_session_chain_recoveries(cogaps_session *s, char which, uint32_t *n) { *n = pick(s, which).chainRecoveries; return 0; }
int cogaps_session_generator_window(cogaps_session *s, char which, uint32_t *attempts) { *attempts = pick(s, which).genWin; return 0; }
int cogaps_session_chained(cogaps_session *s, char which, int *chained)
{
    SESSION_TRY
    if (!chained) return fail("null argument");
    *chained = pick(s, which).chain ? 1 : 0;
    SESSION_END
}

int cogaps_run(const float *data, uint32_t nrow, uint32_t ncol, const cogaps_params *params, const float *unc, cogaps_result *out)
{
    cogaps_session *s = cogaps_session_create(data, nrow, ncol, params, unc, 0);
    if (!s) return 1;
    int rc = cogaps_session_run_iterations(s, 1, 0, params->nIterations, nullptr);
    if (!rc) rc = cogaps_session_run_iterations(s, 2, 0, params->nIterations, nullptr);
    if (!rc) rc = cogaps_session_finish(s, out);
    cogaps_session_destroy(s);
    return rc;
}

// ---- the file entry point (gaps::run(const std::string&...), GapsRunner.h:24-29; cogaps_from_file_cpp, Cogaps.cpp:217-227) ----
static int table_out(const cgio::Table &t, uint32_t *nrow, uint32_t *ncol, float **data)
{
    float *v = (float *)malloc(std::max<size_t>(1, t.v.size()) * sizeof(float));
    if (!v) return fail("out of memory");
    memcpy(v, t.v.data(), t.v.size() * sizeof(float));
    *nrow = t.nrow; *ncol = t.ncol; *data = v;
    return 0;
}
int cogaps_read_matrix_file(const char *path, uint32_t *nrow, uint32_t *ncol, float **data)
{
    try {
        if (!path || !nrow || !ncol || !data) return fail("null argument");
        return table_out(cgio::read_matrix_file(path), nrow, ncol, data);
    } catch (const std::exception &e) { return fail_exc(e); }
}
// the rows (byRows != 0) or columns of the file named by the 1-based `indices`, as the reference's workers read their subset of a
// file (Matrix(path, genesInCols, subsetGenes, indices), data_structures/Matrix.cpp:70-134: sorted indices, lower_bound placement);
// the rest of the matrix is never materialised
int cogaps_read_matrix_file_subset(const char *path, int byRows, const uint32_t *indices, uint32_t nIndices, uint32_t *nrow, uint32_t *ncol, float **data)
{
    try {
        if (!path || !nrow || !ncol || !data || !indices || nIndices == 0) return fail("null argument or empty subset");
        cgio::ReadOpts o; o.sub = cgio::Subset(byRows != 0, indices, nIndices);
        return table_out(cgio::read_matrix_file(path, o), nrow, ncol, data);
    } catch (const std::exception &e) { return fail_exc(e); }
}

void cogaps_matrix_free(float *data) { free(data); }

// getFileInfo_cpp (Cogaps.cpp:229-246): dimensions and the names the file carries, '\n'-joined into caller buffers
// (a NULL buffer or zero capacity skips the names; *needed reports the bytes a complete copy takes, terminator included)
int cogaps_file_info(const char *path, uint32_t *nrow, uint32_t *ncol, char *rowNames, size_t rowCap, size_t *rowNeeded,
                     char *colNames, size_t colCap, size_t *colNeeded)
{
    try {
        if (!path || !nrow || !ncol) return fail("null argument");
        cgio::ReadOpts o; o.values = false;                      // dimensions and names: no value is parsed, no matrix is built
        cgio::Table t = cgio::read_matrix_file(path, o);
        *nrow = t.nrow; *ncol = t.ncol;
        auto join = [](const std::vector<std::string> &v, char *out, size_t cap, size_t *needed) {
            std::string s; for (size_t i = 0; i < v.size(); ++i) { if (i) s += '\n'; s += v[i]; }
            if (needed) *needed = s.size() + 1;
            if (out && cap) { const size_t n = std::min(cap - 1, s.size()); memcpy(out, s.data(), n); out[n] = 0; }
        };
        join(t.rowNames, rowNames, rowCap, rowNeeded); join(t.colNames, colNames, colCap, colNeeded);
        return 0;
    } catch (const std::exception &e) { return fail_exc(e); }
}

int cogaps_run_from_file(const char *dataPath, const cogaps_params *params, const char *uncertaintyPath, cogaps_result *out)
{
    try {
        if (!dataPath || !params || !out) return fail("null argument");
        // A worker of a distributed run reads ITS subset of the file (Matrix.cpp:70-134): the rows or columns the indices name, in
        // sorted order -- never the whole matrix.  The run then sees an ordinary matrix with no subset left to take.
        cgio::ReadOpts o; cogaps_params p = *params;
        if (p.subsetData && p.dataIndicesSubset && p.nSubset) {
            const bool byRows = (p.subsetGenes != 0) == (p.transposeData == 0);      // genes are the file's rows unless transposeData
            o.sub = cgio::Subset(byRows, p.dataIndicesSubset, p.nSubset);
            p.subsetData = 0; p.dataIndicesSubset = nullptr; p.nSubset = 0;
        }
        cgio::Table d = cgio::read_matrix_file(dataPath, o), u;
        const bool haveUnc = uncertaintyPath && uncertaintyPath[0];
        if (haveUnc) {
            u = cgio::read_matrix_file(uncertaintyPath, o);
            if (u.nrow != d.nrow || u.ncol != d.ncol || u.fileRows != d.fileRows || u.fileCols != d.fileCols) return fail("uncertainty matrix has different dimensions than the data");
        }
        return cogaps_run(d.v.data(), d.nrow, d.ncol, &p, haveUnc ? u.v.data() : nullptr, out);
    } catch (const std::exception &e) { return fail_exc(e); }
}

void cogaps_result_free(cogaps_result *r)
{
    if (!r) return;
    free(r->Amean); free(r->Asd); free(r->Pmean); free(r->Psd); free(r->chisqHistory); free(r->atomHistoryA); free(r->atomHistoryP);
    free(r->pumpMatrix); free(r->meanPatternAssignment);
    free(r->equilibrationSnapshotsA); free(r->equilibrationSnapshotsP); free(r->samplingSnapshotsA); free(r->samplingSnapshotsP);
    memset(r, 0, sizeof(*r));
}

} // extern "C"
