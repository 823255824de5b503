// gaps_state.h -- HBM-resident state of one Gibbs sampler (the A or the P instance) as the kernels
// see it.  One `SamplerDev` mirrors AsynchronousGibbsSampler<DenseNormalModel> =
// DenseNormalModel matrices + ConcurrentAtomicDomain + ProposalQueue
// (reference: gibbs_sampler/AsynchronousGibbsSampler.h:32-84, DenseNormalModel.h:56-64,
//  atomic/ConcurrentAtomicDomain.h:43-47, atomic/ProposalQueue.h:55-73).
#pragma once
#include "platform.h"
#include "gaps_math.h"

// One atom.  Addressed by HANDLE (stable for the atom's life, like the reference's heap pointer);
// `vec` (index -> handle) restates the unsorted mAtoms vector used for uniform random picks
// (ConcurrentAtomicDomain.cpp:32-44) and `idx` is the back pointer (ConcurrentAtom::mIndex).
// The record also CACHES what the generator would otherwise fetch from the neighbours' records in a further dependent memory trip:
// their positions (a move's bounds, an exchange partner's bin, the bin-head decisions of erase / move / insert) and the right
// neighbour's mass (an exchange's partner).  Whoever changes an atom's position or mass rewrites the copies its neighbours hold
// (atom_set_pos / atom_set_mass in gen_kernel.h; insert and erase splice them like the links): one proposal owns one atom, adjacent
// atoms never move in the same batch (ProposalQueue.cpp:218), and the copies are separate words, so the writers never collide.  The
// evaluation looks the holders up when it runs (the atom's own record, fetched with the rows): a birth queued later in the same batch
// may have become the atom's neighbour after the proposal was drawn.
struct alignas(16) AtomRec {
    uint64_t pos;
    uint64_t lpos, rpos;    // cached: positions of the left / right neighbour (meaningless where left / right is CG_NONE)
    uint32_t left, right;   // neighbour handles in position order, CG_NONE at the ends
    float mass;
    float rmass;            // cached: mass of the right neighbour
    uint32_t idx;
    uint32_t pad0;
};

// One queued proposal (ProposalQueue.h:15-28) plus every scalar its evaluation starts from, so that the
// evaluation kernel's first memory trip (this record) is also its last dependent one before the row loads.
// 96 bytes.  The copied values cannot go stale: a batch never holds two proposals on one row or one atom, the
// matrix only changes in evaluation kernels, and an attempt that follows a same-bin exchange of its atoms in
// the same window is cut off as a hazard and re-drawn in the next round.
struct alignas(16) PropRec {
    uint64_t pos;        // move destination
    uint64_t rng;        // PCG state after the populate-phase draws
    uint32_t h1, h2;     // atom handles
    uint32_t i1, i2;     // their indices in `vec` at populate time (trace / parity only)
    uint32_t r1, c1, r2, c2;
    uint32_t type;       // 'B','D','M','E'
    uint32_t gibbs;      // bit 0 / 1: canUseGibbs(c1) / (c2), i.e. the other matrix has a positive entry in that pattern
    float m1, m2;        // atom masses (death, move: m1; exchange: both)
    float old1, old2;    // mMatrix(r1,c1), mMatrix(r2,c2)
    uint64_t curPos;     // move: the atom's position
    uint32_t batch;      // trace only
    uint32_t pad[3];
};

// What the evaluation of one queued proposal decided about the A*P cache, for the launch that carries the updates out (split
// evaluation, eval_kernel.h): n = 0 nothing (rejected, or accepted without a matrix change), 1: AP[:,r1] += d1 * other[:,c1],
// 2: ... then AP[:,r2] += d2 * other[:,c2] (an accepted move / exchange; r1 == r2 allowed: the second continues from the first).
struct alignas(16) DecRec { uint32_t n, r1, c1; float d1; uint32_t r2, c2; float d2; uint32_t pad; };

// The chained launch (chain_kernel.h: one launch evaluates batch n and generates batch n + 1) keeps the queue and its length in two
// copies, one per launch parity: a launch of parity p evaluates queue copy p / slot p and its generator workgroup writes copy 1 - p, so
// nothing an evaluation workgroup reads at its start can change while its launch runs.  tag: the batch's number (low word), what the
// evaluation marks its decision granules with.
struct alignas(8) ChainSlot { uint32_t qlen, tag; };
// decision of one evaluated proposal, handed to the next batch's generator inside the launch as two {value, tag} granules
// (grans[q * CHAIN_GRAN_STRIDE + 0] = code | units << 8, + 1 = one float): what the generator's lane applies to the atomic domain
#define CHAIN_GRAN_STRIDE 2u  // 64-bit words per proposal in the granule array: the two granules side by side, four proposals per 64-byte line (a wave's poll
                             // touches 16 lines; at the split evaluation's stride of 64 words it touched 128 and took that much longer to return)
#define CHAIN_NONE 0u        // nothing to change (rejected move / exchange, a death's rebirth with the old mass)
#define CHAIN_APPLY 1u       // B: mass = value; D: rebirth mass = value; M: the move; E: delta = value
#define CHAIN_ERASE 2u       // B: rejected, D: the atom dies -- the atom goes to the erase cache
// what a generator lane leaves in SamplerDev::queueUnits[q] when it gave up waiting for proposal q's decision (GAPS_ERR_SPIN): the decision
// was NOT carried out -- the host's recovery (chain_recover_kernel) carries it out once the launch has ended
#define CHAIN_DROPPED_MARK(tag) (0xD0000000u | ((tag) & 0x0FFFFFFFu))

// Mutable scalars of the proposal generator, one cache line region in HBM.
struct GenScalars {
    uint64_t qrng;            // ProposalQueue::mRng state
    uint64_t batchEpoch;      // stamps for the conflict tables
    uint64_t roundEpoch;
    uint32_t nAtoms;          // domain size (mAtoms.size())
    uint32_t front;           // handle of the lowest-position atom (ConcurrentAtomicDomain::front)
    uint32_t freeCount;       // free-handle stack depth
    uint32_t handleHi;        // bump allocator high-water mark
    uint32_t nSteps, nDone;   // update(nSteps) progress
    uint32_t qlen;            // queue size of the current batch
    uint32_t batchNproc;      // mNumProcessed of the current batch
    uint32_t eraseCount;      // erase cache fill
    uint32_t useCached;       // mUseCachedRng
    float u1, u2;             // mU1, mU2
    float avgQueue, nQueueSamples;   // AsynchronousGibbsSampler::mAvgQueueLength / mNumQueueSamples
    uint32_t nBatches;        // batches generated in this update
    uint32_t error;           // sticky error code (capacity overflow ...)
    uint32_t traceOn, traceCount, traceCap, traceBatchCount;
    uint32_t updateFlushed;   // set by the generator once nDone == nSteps and the last erase cache is flushed
    float annealTemp;         // DenseNormalModel::mAnnealingTemp for this update (kernel parameters stay constant so that launches can be replayed from a graph)
    unsigned long long evalBytes;   // algorithmic HBM bytes of the evaluation kernel (roofline numerator)
    unsigned long long evalProps;   // proposals evaluated
    uint32_t applyCount;      // split evaluation: decision records (DecRec) the next generator launch's update workgroups have to carry out; written by every evaluation launch (0 when its queue was empty)
                              // (a chained launch whose hand-over never arrived parks the batch's queue length here: chain_recover_kernel)
    uint32_t savedErase;      // ... and here the erase cache's fill at that moment (the entries are in SamplerDev::eraseList)
    unsigned long long prof[16];    // GEN_PROFILE builds: cycles per generator phase
};

enum GapsError { GAPS_OK = 0, GAPS_ERR_ATOM_CAP = 1, GAPS_ERR_QUEUE_CAP = 2, GAPS_ERR_ERASE_CAP = 3, GAPS_ERR_SPIN = 4 };

#define GAPS_DEATH_PROB_PAD 1024u
struct SamplerDev {
    // ---- dimensions -------------------------------------------------------------------------
    uint32_t M;        // rows of this sampler's factor matrix (genes for A, samples for P)
    uint32_t N;        // length of every data vector (samples for A, genes for P)
    uint32_t K;        // nPatterns
    uint32_t Npad;     // row stride of D/S2/AP (N rounded up to a multiple of 4; pad: D=0,S2=1,AP=0)
    uint32_t Mpad;     // column stride of `mat` (== the other sampler's Npad)
    uint32_t redW;     // reduction lanes (= evaluation workgroup size), power of two, 64..1024
    // ---- DenseNormalModel ----------------------------------------------------------------------
    const float *D;    // [M][Npad]
    const float *S2;   // [M][Npad]  S*S precomputed (v/(S*S) is evaluated as v/S2: same rounding)
    float *AP;         // [M][Npad]
    float *mat;        // column-major [K][Mpad]: mMatrix(r,k) = mat[k*Mpad + r]
    const float *other;// the other sampler's mat: [K][Npad]
    uint32_t *colPos;  // [K] number of entries > 0 in each column of `mat`
    const uint32_t *otherColPos; // the other sampler's colPos (canUseGibbs, DenseNormalModel.cpp:100-108)
    float lambda, maxGibbsMass, alpha;   // (the annealing temperature changes per iteration: GenScalars::annealTemp)
    GapsLuts luts;
    // ---- ConcurrentAtomicDomain ---------------------------------------------------------------
    AtomRec *atoms;    // [atomCap] by handle
    uint32_t *vec;     // [atomCap] index -> handle
    uint32_t *freeHandles; // [atomCap] stack
    uint32_t atomCap;
    uint32_t *binHead; // [M*K] lowest-position atom of each bin or CG_NONE
    unsigned long long *bits0, *bits1, *bits2;  // occupancy bitmap: exact level 0, monotone hints above
    uint32_t nWords0, nWords1, nWords2;
    unsigned long long *eraseList; // [eraseCap] mEraseCache: handle in the low word, the atom's bin (row * nPatterns + column) in the high word --
                                   // so that the flush can ask for the bin's head in the same memory trip as for the atom's record
    uint32_t eraseCap;
    // ---- ProposalQueue -------------------------------------------------------------------------
    PropRec *queue;    // [queueCap]
    float *partials;      // [queueCap][4][16] per-slice alpha totals of the split evaluation (eval_kernel.h), batched two-launch form
    unsigned long long *grans;   // [queueCap][16][4] the same totals as {value, batch tag} granules handed from the slices' workgroups to the proposal's deciding one inside ONE launch (one-chain form)
    DecRec *dec;          // [queueCap] what each evaluated proposal does to the A*P cache (one-chain split form: carried out beside the next generator launch)
    uint32_t *queueUnits; // [queueCap] algorithmic traffic of each evaluated proposal, in units of 4N bytes
    uint32_t queueCap;
    ChainSlot *chainSlots;  // [2] chained launch: queue length and batch tag per launch parity (the queue then holds 2 * queueCap records)
    const uint64_t *seeds;  // seeder outputs for this update(): candidate k of the update uses seeds[k]
    // conflict stamps, one 64-bit word per key: [batch epoch:40][round:12][priority:12], written with
    // atomicMax.  priority 4094-t for attempt t of the current window (earliest attempt wins), 0xFFFFFF
    // in the low 24 bits once the registering attempt is committed.
    unsigned long long *rowStamp;     // [M]         mUsedMatrixIndices (FixedHashSetU32)
    unsigned long long *atomStamp;    // [atomCap]   mUsedAtoms (SmallHashSetU64), keyed by handle
    unsigned long long *gapStamp;     // [atomCap+1] a birth landed right of atom h (key h+1) / before the front (key 0)
    unsigned long long *inlineStamp;  // [atomCap]   atom touched by a same-bin move/exchange of the current window
    uint64_t *atomDest;               // [atomCap]   destination of the committed queued move of this atom, else 0 (mProposedMoves)
    const uint64_t *lcgMul, *lcgInc; // k-step PCG jump coefficients, k = 0 .. 2*GEN_WIN+1
    uint64_t binLength;    // mBinLength
    uint64_t domainLenU;   // ConcurrentAtomicDomain::mDomainLength (exact)
    double domainLenD;     // ProposalQueue::mDomainLength
    double numBins, alphaD;
    const float *deathProb; // [atomCap + GAPS_DEATH_PROB_PAD] ProposalQueue::deathProb(n) for every atom count n (death_prob_table_kernel): a window's
                            // attempts see counts within +-GEN_WIN of the domain's, the generator reads its two rows of the table instead of dividing
    double invBinLen;      // 1.0 / binLength (quotient estimate of gen_bin_of)
    double invK;           // 1.0 / nPatterns (gen_div_k)
    uint64_t rboundNone;   // static_cast<uint64_t>(mDomainLength), ProposalQueue.cpp:216
    uint64_t iPartL, limitL; // uniform64(1, domainLenU): iPart = UINT64_MAX / L, limit = L * iPart (Random.cpp:112-117)
    // ---- SparseNormalModel (sparse_kernels.h); unused (null / 0) with the dense model -------------------------------
    uint32_t sparse;       // 1: useSparseOptimization
    uint32_t Wn, Mw, oMw;  // flag words per data vector (N/64+1), per column of this matrix (M/64+1) and of the other one
    uint32_t Kpad, oKpad;  // row stride of the row copies
    uint32_t spW;          // cogaps_sparse_width(N): threads = virtual lanes of an evaluation workgroup (the launch's block size, except inside a chained launch, which has the generator's)
    const unsigned long long *dflags; const uint32_t *dprefix, *dptr; const float *dvals;
    float *rows;           // [M][Kpad] HybridMatrix row copy (mMatrix(r,c))
    unsigned long long *mflags;        // [K][Mw] flags of the column copy `mat`
    const float *orows;    // the other sampler's rows
    const unsigned long long *oflags;  // ... and mflags
    float *Z1, *Z2;        // [K], column-major [K][K]
    float beta;
    uint32_t unitBytes;    // bytes per unit of queueUnits (4N with the dense model, 1 with the sparse one)
    uint32_t defaultS;     // no uncertainty matrix was given: S2 = max(0.1 D, 0.1)^2 is recomputed from D instead of read
    // ---- verification mode (cogaps_params.reductionMode / mathMode) ---------------------------------------------------
    uint32_t seq;          // 1: every floating-point sum in the reference's scalar order (SIMD.h:36-47: one accumulator, i = 0 .. N-1)
    uint32_t mathMode;     // GM_MATH_*: logf / expf of the accept tests and draws (honoured by the seq kernels and the generator)
    float *seqScratch;     // [seqGrid][3][Npad] per-workgroup term scratch of the sparse model's sequential sums
    GenScalars *gs;
    // ---- optional trace (parity tests) -------------------------------------------------------
    PropRec *trace;        // [traceCap] copies of queued proposals
    uint32_t *traceBatchNproc, *traceBatchQlen; // [traceCap]
    uint32_t dbg;          // GEN_PROFILE builds: skip parts of the evaluation kernel (timing experiments)
    // ---- launch clock (chained launch) ---------------------------------------------------------
    // [GAPS_CLOCK_RING][2] chip-wide 100 MHz clock (s_memrealtime) at the entry of a chained launch's first workgroup and at the end of
    // its generator workgroup (the launch's last to finish), slot = the evaluated batch's tag mod the ring: the duration of EVERY launch
    // of the timed region, replayed graphs included, where HIP events can only ride on plain launches.  Null: not recorded.
    unsigned long long *launchClock;
};
#define GAPS_CLOCK_RING 8192u
