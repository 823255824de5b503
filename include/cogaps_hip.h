/*
 * cogaps_hip.h -- C ABI of libcogaps_hip.so, the MI355X (gfx950) implementation of the CoGAPS
 * asynchronous Gibbs sampler hot path.
 *
 * The boundary it replaces is the reference's language-neutral core entry
 *     GapsResult gaps::run(const Matrix &data, GapsParameters &params,
 *                          const Matrix &uncertainty, GapsRandomState *randState)
 * (reference src/GapsRunner.h:17-22, src/GapsRunner.cpp:113-123), which Rcpp's cogaps_cpp
 * (src/Cogaps.cpp:205-215, R/RcppExports.R:8-10) reaches through cogapsRun (src/Cogaps.cpp:148-186).
 * INTEGRATION.md shows the Rcpp stub that binds these entry points in place of gaps::run.
 *
 * Plain pointers and sizes only; the callee copies its inputs; results are callee-allocated and
 * released with cogaps_result_free.  All functions return 0 on success and a non-zero code plus a
 * message (cogaps_last_error) on failure; nothing calls exit() (reference: utils/GapsAssert.h:19-25).
 * Threading: a session (or a cogaps_run call) is driven by one host thread at a time; different sessions may run
 * concurrently from different host threads, on the same GPU or on different ones (each owns one non-blocking
 * stream and the library creates no other; no legacy-stream operation is issued; graph capture is thread-local).
 * Up to four sessions per process and GPU run truly side by side (HIP's four hardware queues per process).  The library keeps no global
 * mutable state besides the per-thread last error and, per device, a count of the updates in flight on it (a chained launch, which wants
 * the whole chip, is taken only by an update -- a session's or a batch's -- that runs alone on its GPU).
 * Environment, all optional, none changes a result; each is read when a session is created and has its test (tests/test_gpu_parity.py):
 *   COGAPS_NO_GRAPH (any value)  every kernel as a plain launch instead of replaying captured graphs -- for counter-collection tools
 *                                (tools/pmc_pass.sh); the whole -m gpu suite passes with it
 *   COGAPS_NO_CHAIN              two launches per batch instead of the chained launch: A/B runs, test_chained_equals_two_launches_on_the_gpu
 *   COGAPS_FORCE_CHAIN           the fused evaluation's chained launch also where the device shows fewer compute units than the launch has
 *                                workgroups, or another update is in flight: test_chained_launch_with_half_the_compute_units.  (Not the split
 *                                evaluation's: its update items wait for deciding workgroups of higher index, every workgroup must be resident.)
 *   COGAPS_CHAIN_SPLIT           the split evaluation (data vectors of more than 4096 elements) inside the chained launch -- built, measured
 *                                6 % slower on the headline chain (profiles/r05_ab_chained_split_evaluation_not_kept.txt), not the default:
 *                                test_chained_split_evaluation_equals_two_launches_on_the_gpu
 *   COGAPS_TEST_WIDE_WINDOW      the sparse model's chained launch with its widest generator window (448 attempts) from the first update on; the
 *                                library takes it by itself once a sampler's batches exceed 230 proposals:
 *                                test_sparse_chained_launch_with_long_queues_equals_two_launches_on_the_gpu
 * Development builds (-DCOGAPS_DEV, never shipped) read further switches that only change what is measured.
 */
#ifndef COGAPS_HIP_H
#define COGAPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of GapsParameters (reference src/GapsParameters.h:35-66; defaults :79-111) */
typedef struct cogaps_params {
    uint32_t seed;
    uint32_t nPatterns;          /* default 3 */
    uint32_t nIterations;        /* default 1000, per phase */
    uint32_t maxThreads;         /* accepted for API parity; the GPU path ignores it */
    uint32_t outputFrequency;    /* default 500 */
    uint32_t checkpointInterval; /* accepted, must be 0 (checkpoints are disabled, Cogaps.cpp:224-231) */
    uint32_t snapshotFrequency;  /* GapsRunner.cpp:316-322: a copy of A and P every so many iterations of the snapshot phase(s); 0 = none.
                                    cogaps_cpp derives it as nIterations / nSnapshots (Cogaps.cpp:104-109) */
    float alphaA, alphaP;        /* default 0.01 */
    float maxGibbsMassA, maxGibbsMassP; /* default 100 */
    int32_t transposeData;
    int32_t printMessages;
    int32_t subsetData;          /* dataIndicesSubset in use */
    int32_t subsetGenes;         /* subsetDim == 1 (rows of A) else samples */
    const uint32_t *dataIndicesSubset; /* 1-based indices, as R passes them (Matrix.cpp:55-62) */
    uint32_t nSubset;
    int32_t useSparseOptimization; /* SparseNormalModel (default uncertainty only) instead of DenseNormalModel */
    int32_t takePumpSamples;       /* GapsStatistics::updatePump per sampling iteration (GapsRunner.cpp:310-313) */
    int32_t asynchronousUpdates;   /* must be 1: this library IS the asynchronous sampler.  One exception: 0 is accepted, and ignored, when
                                      runningDistributed is set -- R's distributed caller forces FALSE on its workers (R/DistributedCogaps.R:28-29) */
    char whichMatrixFixed;         /* 'N', 'A' or 'P' */
    const float *fixedPatterns;    /* row-major [fixedRows][nPatterns] when whichMatrixFixed != 'N' */
    uint32_t fixedRows;
    uint32_t workerID;
    int32_t runningDistributed;
    int32_t device;                /* HIP device ordinal, -1 = the creating thread's current one (resolved at session creation; every later call on the session selects that GPU for its calling thread) */
    int (*interrupt)(void *);      /* polled once per iteration (GapsRunner.cpp:280); non-zero aborts */
    void *interruptArg;
    int32_t snapshotPhase;         /* 0 = all phases (GAPS_ALL_PHASES, the default of GapsParameters.h:98), 1 = equilibration, 2 = sampling */
    int32_t pumpThreshold;         /* PumpThreshold (GapsParameters.h:52; GapsStatistics.h:58-63): 0 = PUMP_UNIQUE (default), 1 = PUMP_CUT.
                                      The reference's two rules are the same code (GapsStatistics.h:65-126) and so are they here */
    int32_t fixedCols;             /* columns of fixedPatterns; 0 = nPatterns.  Anything else is rejected (GapsRunner.cpp:329-350 copies
                                      nPatterns columns) */
    /* ---- verification mode: no counterpart in GapsParameters -------------------------------------------------------------
     * reductionMode COGAPS_REDUCE_LANES (default): the kernels' lane order (cogaps_reduction_width).  COGAPS_REDUCE_SEQ: every
     * floating-point sum runs in the order of the reference's default scalar build (src/math/SIMD.h:36-47: one accumulator,
     * i = 0 .. N-1; chiSq DenseNormalModel.cpp:56-68; gaps::dot VectorMath.h:41-134) -- slow, and bit-identical to that build.
     * mathMode (honoured with COGAPS_REDUCE_SEQ): the logf / expf of the accept tests and draws.  COGAPS_MATH_PORTABLE
     * (default): the kernels' own correctly rounded algorithm.  COGAPS_MATH_GLIBC_FMA / _SSE2: GNU libc 2.35's logf / expf
     * restated (its -mfma ifunc variant, which x86-64 glibc selects on FMA-capable hosts / its generic variant) -- what the
     * reference binary computes when it is linked against that C library.  With SEQ + GLIBC the HIP library reproduces the
     * reference's atom histories, totalUpdates and statistics digit for digit (tests/test_gpu_parity.py). */
    int32_t reductionMode;
    int32_t mathMode;
} cogaps_params;
#define COGAPS_REDUCE_LANES 0
#define COGAPS_REDUCE_SEQ 1
#define COGAPS_MATH_PORTABLE 0
#define COGAPS_MATH_GLIBC_FMA 1
#define COGAPS_MATH_GLIBC_SSE2 2

/* POD mirror of GapsResult (reference src/GapsResult.h:17-36) + the names cogapsRun returns */
typedef struct cogaps_result {
    uint32_t nGenes, nSamples, nPatterns;
    float *Amean, *Asd;          /* row-major [nGenes][nPatterns] */
    float *Pmean, *Psd;          /* row-major [nSamples][nPatterns] */
    uint32_t nHistory;
    float *chisqHistory;         /* diagnostics$chisq */
    uint32_t *atomHistoryA;      /* diagnostics$atomsA */
    uint32_t *atomHistoryP;      /* diagnostics$atomsP */
    uint64_t totalUpdates;
    uint32_t seed;
    uint32_t totalRunningTime;   /* seconds */
    float meanChiSq;
    float averageQueueLengthA, averageQueueLengthP;
    double samplerSeconds;       /* wall time of the two phases, for proposals/s */
    float *pumpMatrix;           /* diagnostics$pumpStat, row-major [nGenes][nPatterns]; NULL unless takePumpSamples */
    float *meanPatternAssignment;/* diagnostics$meanPatternAssignment, same shape */
    uint32_t nEquilibrationSnapshots, nSamplingSnapshots;
    float *equilibrationSnapshotsA, *equilibrationSnapshotsP;   /* [n][rows][nPatterns] row-major */
    float *samplingSnapshotsA, *samplingSnapshotsP;
} cogaps_result;

void cogaps_default_params(cogaps_params *p);

/* gaps::run for an in-memory matrix: data row-major [nrow][ncol] fp32 (genes x samples unless
 * transposeData), uncertainty the same shape or NULL.  Host pointers. */
int cogaps_run(const float *data, uint32_t nrow, uint32_t ncol, const cogaps_params *params,
               const float *uncertainty, cogaps_result *out);
/* gaps::run for a matrix file (src/GapsRunner.h:24-29; Rcpp cogaps_from_file_cpp, src/Cogaps.cpp:217-227): .mtx, .csv, .tsv
 * or .gct, read as the reference's parsers read them (src/file_parser/, incl. the text -> fp32 rule of
 * MatrixElement.cpp:10-47); uncertaintyPath NULL or "" for the default uncertainty. */
int cogaps_run_from_file(const char *dataPath, const cogaps_params *params, const char *uncertaintyPath, cogaps_result *out);
/* the file as a dense row-major fp32 matrix (callee-allocated; release with cogaps_matrix_free).  Host only: no GPU needed. */
int cogaps_read_matrix_file(const char *path, uint32_t *nrow, uint32_t *ncol, float **data);
/* ... of a SUBSET of the file: the rows (byRows != 0) or columns named by the 1-based `indices`, read the way the reference's workers
 * read their subset of a file (Matrix(path, genesInCols, subsetGenes, indices), src/data_structures/Matrix.cpp:70-134: the indices
 * are sorted first, a duplicated index fills its first position only); the rest of the matrix is never materialised.
 * cogaps_run_from_file does the same when params->subsetData is set. */
int cogaps_read_matrix_file_subset(const char *path, int byRows, const uint32_t *indices, uint32_t nIndices,
                                   uint32_t *nrow, uint32_t *ncol, float **data);
void cogaps_matrix_free(float *data);
/* getFileInfo_cpp (src/Cogaps.cpp:229-246): dimensions and the row / column names the file carries, '\n'-joined into the
 * caller's buffers (NULL / 0 to skip; *needed = bytes of a complete copy incl. the terminator).  Host only. */
int cogaps_file_info(const char *path, uint32_t *nrow, uint32_t *ncol, char *rowNames, size_t rowCap, size_t *rowNeeded,
                     char *colNames, size_t colCap, size_t *colNeeded);
void cogaps_result_free(cogaps_result *r);
const char *cogaps_last_error(void);
/* What kind of failure the calling thread's last failing call was -- so that a caller can react to device memory running out (fewer
 * sessions in flight) without reading message texts.  The reference has one failure path (GAPS_ERROR, utils/GapsAssert.h:19-25). */
enum { COGAPS_OK = 0, COGAPS_ERR_GENERIC = 1, COGAPS_ERR_OUT_OF_DEVICE_MEMORY = 2, COGAPS_ERR_OUT_OF_HOST_MEMORY = 3 };
int cogaps_last_error_code(void);

/* the HIP device ordinal that is current for the calling host thread (what cogaps_params.device = -1 resolves to) */
int cogaps_current_device(int *device);
/* free and total bytes of HBM on `device` (-1: the calling thread's current one); what a caller that keeps several sessions per GPU
 * sizes its batches by (cogaps_amd/distributed.py) */
int cogaps_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes);

/* the three trivial exports next to cogaps_cpp (src/Cogaps.cpp:217-246) */
const char *cogaps_build_report(void);
/* sha256 (first 16 hex digits) over the sources the library was built from, "unknown" for a build outside csrc/Makefile.  bench.py compares it
 * with the hash recorded beside a committed counter measurement (profiles/r*_pmc_traffic.json) and marks the measurement stale when they differ. */
const char *cogaps_source_hash(void);
int cogaps_checkpoints_enabled(void);
int cogaps_compiled_with_openmp(void);

/* ------------------------------------------------------------------------------------------------
 * Session interface: the same run, one step at a time.  Used by bench.py (inputs resident in HBM
 * before the timed region; `data_on_device` accepts a device pointer) and by the parity tests
 * (per-batch proposal traces).  cogaps_run is cogaps_session_create + phases + finish.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cogaps_session cogaps_session;

typedef struct cogaps_trace_rec {   /* one queued proposal, ProposalQueue.h:15-28 */
    uint64_t pos, rng_state;
    uint32_t atom1, atom2;          /* indices in the unsorted atom vector, 0xFFFFFFFF = none */
    uint32_t r1, c1, r2, c2;
    uint32_t type;                  /* 'B','D','M','E' */
    uint32_t batch;
} cogaps_trace_rec;

cogaps_session *cogaps_session_create(const float *data, uint32_t nrow, uint32_t ncol,
                                      const cogaps_params *params, const float *uncertainty,
                                      int data_on_device);
void cogaps_session_destroy(cogaps_session *s);
/* annealing temperature of both samplers (runOnePhase sets min(1, 2*iter/nIter) while equilibrating) */
int cogaps_session_set_annealing(cogaps_session *s, float temp);
/* nA, nP ~ Poisson(max(nAtoms,10)) from the runner's generator (GapsRunner.cpp:294-295) */
int cogaps_session_draw_steps(cogaps_session *s, uint32_t *nA, uint32_t *nP);
/* AsynchronousGibbsSampler::update for sampler `which` ('A' or 'P'); optional proposal trace */
int cogaps_session_update(cogaps_session *s, char which, uint32_t nSteps,
                          cogaps_trace_rec *trace, uint32_t traceCap, uint32_t *nTrace,
                          uint32_t *batchNproc, uint32_t *batchQlen, uint32_t batchCap, uint32_t *nBatches);
/* DenseNormalModel::sync for sampler `which` (copies the transposed AP of the other sampler) */
int cogaps_session_sync(cogaps_session *s, char which);
/* one iteration of runOnePhase: updateSampler(nA, nP) (+ statistics when sampling != 0) */
int cogaps_session_iterate(cogaps_session *s, uint32_t nA, uint32_t nP, int sampling);
/* `n` complete iterations of a phase starting at iteration `firstIter` (anneal, draw, update, stats,
 * history); phase 1 = equilibration, 2 = sampling.  Adds to *updates the proposals processed. */
int cogaps_session_run_iterations(cogaps_session *s, int phase, uint32_t firstIter, uint32_t n, uint64_t *updates);
int cogaps_session_natoms(cogaps_session *s, char which, uint32_t *n);
int cogaps_session_chisq(cogaps_session *s, char which, float *chisq);
/* copy-outs (host buffers): matrix row-major [M][K]; AP [M][N]; atoms in vector order */
int cogaps_session_get_matrix(cogaps_session *s, char which, float *out);
int cogaps_session_get_rows(cogaps_session *s, char which, float *out);   /* sparse model: the HybridMatrix row copy, row-major [rows][nPatterns]; dense model: = get_matrix */
int cogaps_session_get_ap(cogaps_session *s, char which, float *out);
int cogaps_session_get_atoms(cogaps_session *s, char which, uint64_t *pos, float *mass,
                             uint32_t *left, uint32_t *right);
int cogaps_session_dims(cogaps_session *s, char which, uint32_t *M, uint32_t *N, uint32_t *K);
int cogaps_session_avg_queue(cogaps_session *s, char which, float *avg);
int cogaps_session_finish(cogaps_session *s, cogaps_result *out);
/* counters for the roofline report: algorithmic bytes moved by the evaluation kernel so far, number
 * of evaluation launches, batches generated, and the accumulated HIP-event time of each kernel */
typedef struct cogaps_perf {
    uint64_t evalBytes;       /* sum over evaluated proposals of 16N/20N/32N + 12N per AP update */
    uint64_t evalLaunches, genLaunches, batches, proposalsQueued;
    double evalMs, genMs, syncMs;   /* HIP-event time (dispatch begin to end) of the launches that processed a batch since timing was switched on
                                       (a sample of them carries events; scaled to `timedBatches`) */
    double evalNoopMs, genNoopMs;   /* summed HIP-event time of sampled launches past the end of an update (empty queue) */
    uint64_t evalNoopTimed, genNoopTimed; /* ... and how many were sampled */
    uint64_t timedBatches;    /* batches processed since cogaps_session_set_timing(1): what evalMs / genMs are scaled to */
    uint64_t evalTimed, genTimed;   /* launches that carried events and processed a batch */
    uint64_t syncTimed, syncBytes;  /* sync launches timed since then (all of them; their summed time is syncMs) and their algorithmic bytes (8 M N each) */
} cogaps_perf;
int cogaps_session_set_timing(cogaps_session *s, int on);
int cogaps_session_perf(cogaps_session *s, cogaps_perf *out);                      /* both samplers */
int cogaps_session_perf_sampler(cogaps_session *s, char which, cogaps_perf *out);   /* the 'A' or the 'P' sampler alone */
/* 1 when the sampler's last update ran as chained launches (csrc/chain_kernel.h: ONE launch evaluates batch n and generates batch n + 1;
 * its time is reported as evalMs, genMs stays 0), 0 for a generator launch and an evaluation launch per batch.  The chained form serves
 * the one-chain fused evaluation (AsynchronousGibbsSampler.h:88-122, same results); environment COGAPS_NO_CHAIN=1 switches it off.  It is taken
 * only where the device shows a compute unit per workgroup of the launch (241; hipDeviceAttributeMultiprocessorCount, which honours
 * HSA_CU_MASK): the evaluation workgroups never wait for anything, so with fewer units the launch is still correct -- the workgroups run in
 * turns beside the generator's -- but slower than two launches; COGAPS_FORCE_CHAIN=1 (tests) takes it there anyway.  The hand-over inside the
 * launch assumes nothing about the order in which the dispatcher starts workgroups: only the generator workgroup waits, and only for
 * evaluation workgroups, which never wait, so every workgroup ends; the wait is bounded all the same (two seconds at least, for a workgroup
 * the GPU does not schedule at all) -- never a hang.  A wait that runs out applies nothing
 * of the decision it waited for and makes the launches already enqueued behind it no-ops; the host then completes the batch from the
 * decisions the evaluation workgroups have left by then (chain_recover_kernel), and the sampler goes on -- same chain, same bits -- with
 * two launches per batch for the rest of the session.  cogaps_session_chain_recoveries counts such events (0 in every run so far).  Only if
 * a decision is still missing then (the split evaluation's chained form, COGAPS_CHAIN_SPLIT: its deciding workgroups wait as well) does the
 * update end with an error (GAPS_ERR_SPIN) and the session refuse further steps. */
int cogaps_session_chained(cogaps_session *s, char which, int *chained);
int cogaps_session_chain_recoveries(cogaps_session *s, char which, uint32_t *n);
/* Attempts per round of the sampler's generator launches as of its last update (the library's instantiations: 128, 256, and 448 for the sparse
 * model's chained launch once batches are long; no result depends on it). */
int cogaps_session_generator_window(cogaps_session *s, char which, uint32_t *attempts);
/* Durations of the sampler's chained launches since cogaps_session_set_timing(1), EVERY launch -- replayed graphs included, where HIP
 * events cannot ride --, from the chip-wide 100 MHz clock read inside the launch (entry of its first workgroup to the end of its generator
 * workgroup, the last to finish: what rocprofv3 --kernel-trace reports as the dispatch's duration, minus the dispatcher's fill / drain).
 * meanUs; percentilesUs[5] = 10th, 50th, 75th, 90th, 99th (0.1 us bins); launches = how many were measured (0: the sampler does not chain). */
int cogaps_session_launch_clock(cogaps_session *s, char which, double *meanUs, double *percentilesUs, uint64_t *launches);
/* The same launches by their PERIOD: entry of a launch's first workgroup to the entry of the next launch's -- the launch with the dispatcher's
 * start-up and the end-of-kernel write-back around it (what rocprofv3 reports as the dispatch's duration, plus the idle gap to the next
 * dispatch); seams where the host read progress back (> 100 us) are left out.  The sum of the periods cannot exceed the wall time. */
int cogaps_session_launch_period(cogaps_session *s, char which, double *meanUs, double *percentilesUs, uint64_t *launches);

/* ------------------------------------------------------------------------------------------------
 * Batched multi-chain launches: the sessions of a batch -- the subsets of a GWCoGAPS / scCoGAPS job that share one GPU
 * (R/DistributedCogaps.R:60-68 hands them to BiocParallel workers), or replicas -- are stepped in lock-step by one stream: one
 * generator launch with a workgroup per chain, one evaluation launch over all chains' queues.  Every chain produces exactly the
 * bits it produces on its own.  The sessions must share the model (dense / sparse), whichMatrixFixed, the device and the
 * evaluation launch shape (equal cogaps_reduction_width of their vector lengths); create them first, then the batch; drive the
 * batch with cogaps_batch_run_iterations (phase 1, then phase 2), then cogaps_session_finish each session; destroy the batch
 * before the sessions.  While a session belongs to a batch only the batch may step it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cogaps_batch cogaps_batch;
cogaps_batch *cogaps_batch_create(cogaps_session **sessions, uint32_t n);
void cogaps_batch_destroy(cogaps_batch *b);
/* iterations [firstIter, firstIter + n) of `phase` for every chain; updates (may be NULL): [n chains], += proposals per chain */
int cogaps_batch_run_iterations(cogaps_batch *b, int phase, uint32_t firstIter, uint32_t n, uint64_t *updates);
int cogaps_batch_set_timing(cogaps_batch *b, int on);
/* mean HIP-event time (us) of the sampled generator / evaluation launches of sampler side 0 (A) or 1 (P) since set_timing(1) */
int cogaps_batch_perf(cogaps_batch *b, int side, double *genUs, double *evalUs, uint64_t *sampled, uint64_t *launches);

/* development aid: per-phase cycle counters of the generator kernel (all zero unless built with -DGEN_PROFILE) */
int cogaps_session_debug_prof(cogaps_session *s, char which, uint64_t *out16);
int cogaps_session_debug_replay(cogaps_session *s, char which, int kind, uint32_t n, uint32_t dbgFlags, double *usPerLaunch);
/* test hook: counts broken invariants of the atomic domain's redundant state (links, index vector, cached neighbour positions / masses) */
int cogaps_session_debug_check_domain(cogaps_session *s, char which, uint32_t *violations);

/* test hook for the math modes: y[i] = fn(x[i]) with fn 0 = logf, 1 = expf in math mode `mathMode`, evaluated by a kernel on
 * the current device (on_device != 0) or by the same source compiled for the host */
int cogaps_debug_math(int fn, int mathMode, const float *x, float *y, uint32_t n, int on_device);

/* lanes of the evaluation workgroup for data vectors of length N (the reduction-order contract) */
uint32_t cogaps_reduction_width(uint32_t N);
/* threads (= virtual lanes) of the sparse model's evaluation workgroup: one per 64-bit flag word of a data vector, 64..256 */
uint32_t cogaps_sparse_width(uint32_t N);

#ifdef __cplusplus
}
#endif
#endif
