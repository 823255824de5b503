/*
 * gaps_oracle.h -- CPU restatement of the CoGAPS asynchronous Gibbs sampler hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cogaps_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported CPU baseline.
 *
 * Every function in gaps_oracle.c cites the reference file:line (relative to
 * /root/reference/src) whose behaviour it restates.  Parity pin: the SURVEY.md section 8c
 * fingerprints (GIST.mtx K=7 seed=42 and modsimdata K=3 seed=42 atom histories,
 * totalUpdates, meanChiSq) -- see tests/test_oracle_pin.py.
 */
#ifndef GAPS_ORACLE_H
#define GAPS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* math_mode: which log/exp the fp32 accept tests use */
#define GO_MATH_LIBM     0  /* host libm logf/expf: what the reference binary does on this host */
#define GO_MATH_PORTABLE 1  /* fixed IEEE-double algorithm shared (as a spec) with the HIP kernels */
#define GO_MATH_GLIBC_FMA 2  /* glibc 2.35 logf/expf restated, the -mfma ifunc variant (what GO_MATH_LIBM resolves to on an FMA-capable x86-64) */
#define GO_MATH_GLIBC_SSE2 3 /* ... the generic variant (separate multiply and add) */

typedef struct go_params {
    uint32_t nPatterns;        /* GapsParameters.h:88  default 3 */
    uint32_t nIterations;      /* :89 default 1000 (per phase) */
    uint32_t seed;
    uint32_t outputFrequency;  /* :91 default 500 */
    uint32_t maxThreads;       /* OpenMP threads for the queue loop */
    float alphaA, alphaP;      /* :94-95 default 0.01 */
    float maxGibbsMassA, maxGibbsMassP; /* :96-97 default 100 */
    int32_t transposeData;
    int32_t subsetData;        /* dataIndicesSubset in use */
    int32_t subsetGenes;       /* subsetDim == 1 */
    const uint32_t *subsetIndices; /* 1-based, Matrix.cpp:55-62 */
    uint32_t nSubset;
    char whichMatrixFixed;     /* 'N','A','P' */
    const float *fixedPatterns;/* row-major [rows][nPatterns] */
    uint32_t fixedRows;
    int32_t math_mode;         /* GO_MATH_* */
    uint32_t redW_A, redW_P;   /* reduction lanes for A / P sampler; 0 or 1 = sequential (reference scalar order) */
    uint32_t redG;             /* lane granularity in elements (1 = reference PackedFloat pattern, 4 = float4) */
    int32_t useSparseOptimization; /* SparseNormalModel instead of DenseNormalModel (GapsRunner.cpp:65-91) */
    int32_t takePumpSamples;       /* GapsStatistics::updatePump per sampling iteration (GapsRunner.cpp:310-313) */
    uint32_t snapshotFrequency;    /* GapsRunner.cpp:316-322; 0 = none */
    int32_t snapshotPhase;         /* 0 = all phases (the default, GapsParameters.h:98), 1 = equilibration, 2 = sampling */
} go_params;

typedef struct go_result {
    uint32_t nGenes, nSamples, nPatterns;
    float *Amean, *Asd;        /* row-major [nGenes][nPatterns] */
    float *Pmean, *Psd;        /* row-major [nSamples][nPatterns] */
    uint32_t nHistory;
    float *chisqHistory;
    uint32_t *atomHistoryA, *atomHistoryP;
    uint64_t totalUpdates;
    float meanChiSq;
    float averageQueueLengthA, averageQueueLengthP;
    double samplerSeconds;     /* wall time of the two phases (first update to last) */
    float *pumpMatrix, *meanPatternAssignment;   /* row-major [nGenes][nPatterns], NULL unless takePumpSamples */
    uint32_t nEquilibrationSnapshots, nSamplingSnapshots;
    float *equilibrationSnapshotsA, *equilibrationSnapshotsP, *samplingSnapshotsA, *samplingSnapshotsP; /* [n][rows][nPatterns] */
} go_result;

/* one queued proposal as it leaves populate (ProposalQueue.h:15-28) */
typedef struct go_trace_rec {
    uint64_t pos;       /* move destination */
    uint64_t rng_state; /* PCG state when the proposal is queued (after populate-phase draws) */
    uint32_t atom1, atom2; /* indices into the unsorted atom vector (mAtoms); 0xFFFFFFFF = none */
    uint32_t r1, c1, r2, c2;
    uint32_t type;      /* 'B','D','M','E' */
    uint32_t batch;     /* batch ordinal inside this update() call */
} go_trace_rec;

typedef struct go_trace {
    go_trace_rec *rec; uint32_t cap, n;          /* queued proposals */
    uint32_t *batch_nproc; uint32_t *batch_qlen; /* per batch: nProcessed, queue size */
    uint32_t batch_cap, n_batches;
} go_trace;

typedef struct go_session go_session;

void go_default_params(go_params *p);

/* data: row-major [nrow][ncol] fp32; unc: same shape or NULL (default uncertainty) */
go_session *go_create(const float *data, uint32_t nrow, uint32_t ncol, const go_params *p,
                      const float *unc);
void go_destroy(go_session *s);

/* one full run = gaps::run (GapsRunner.cpp:382-499) */
int go_run(const float *data, uint32_t nrow, uint32_t ncol, const go_params *p,
           const float *unc, go_result *out);
void go_result_free(go_result *r);

/* step-wise access for parity tests */
void go_set_annealing(go_session *s, float temp);
uint32_t go_natoms(const go_session *s, char which);
/* draw nA,nP exactly like runOnePhase (GapsRunner.cpp:294-295) */
void go_draw_steps(go_session *s, uint32_t *nA, uint32_t *nP);
/* AsynchronousGibbsSampler::update for one sampler (no sync) */
void go_update(go_session *s, char which, uint32_t nSteps, go_trace *trace);
/* DenseNormalModel::sync: which = the sampler being refreshed */
void go_sync(go_session *s, char which);
/* one runOnePhase iteration (anneal temp must be set by caller); returns nA+nP */
uint64_t go_iterate(go_session *s, uint32_t nA, uint32_t nP);
void go_stats_update(go_session *s);
float go_chisq(const go_session *s, char which);
/* copy-outs: matrix is row-major [rows][K]; AP is [M][N] (one contiguous vector per factor row) */
void go_get_matrix(const go_session *s, char which, float *out);
void go_debug_set_matrices(go_session *s, const float *A, const float *P);
/* bench hook: a fresh session takes over a chain state (atoms in vector order + factor matrices, row-major); 0 on success */
int go_import_state(go_session *s, const uint64_t *posA, const float *massA, uint32_t nA, const float *A,
                    const uint64_t *posP, const float *massP, uint32_t nP, const float *P);
void go_debug_alpha(const go_session *s, char which, int mode, uint32_t r1, uint32_t c1, uint32_t r2, uint32_t c2, float ch, float *out2);
void go_get_rows(const go_session *s, char which, float *out); /* HybridMatrix row copy (sparse model); = go_get_matrix for the dense model */
void go_get_ap(const go_session *s, char which, float *out);
void go_get_atoms(const go_session *s, char which, uint64_t *pos, float *mass,
                  uint32_t *left, uint32_t *right); /* in mAtoms order; neighbours as indices */
void go_get_dims(const go_session *s, char which, uint32_t *M, uint32_t *N, uint32_t *K);
float go_lambda(const go_session *s, char which);
float go_max_gibbs_mass(const go_session *s, char which);
float go_avg_queue(const go_session *s, char which);
/* the three lookup tables (Random.cpp:269-295): 3001 + 5001 + 5001 floats */
void go_get_luts(const go_session *s, float *erf, float *erfinv, float *qgamma);
void go_finish(go_session *s, go_result *out);
/* iterations [first, first + n) of phase 1 (equilibration) / 2 (sampling) exactly as go_run's loop runs them (annealing, step counts,
 * statistics, snapshots, history; GapsRunner.cpp:272-327); stepsA / stepsP (may be NULL) receive the Poisson step counts per iteration */
void go_run_iterations(go_session *s, int phase, uint32_t first, uint32_t n, uint32_t *stepsA, uint32_t *stepsP);

/* free-standing pieces exposed for unit tests */
float go_portable_logf(float x);
float go_portable_expf(float x);
float go_glibc_logf(float x, int fused);
float go_glibc_expf(float x, int fused);
/* number of floats in [lo, hi] (bit patterns, same sign) on which go_glibc_{logf,expf}(., fused) differs from the host libm; fn 0 = logf, 1 = expf */
uint64_t go_glibc_mismatches(int fn, int fused, uint32_t lo_bits, uint32_t hi_bits, uint32_t step);
void go_build_luts(float *erf, float *erfinv, float *qgamma);
uint64_t go_seeder_stream(uint32_t seed, uint64_t *out, uint64_t n); /* first n seeder outputs */
uint32_t go_pcg_next(uint64_t *state);
float go_strtof(const char *s);

#ifdef __cplusplus
}
#endif
#endif
