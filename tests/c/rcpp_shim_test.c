/*
 * rcpp_shim_test.c -- the closest thing to the Rcpp binding this image allows (no R, no Rcpp headers): a plain C
 * program that does what cogaps_cpp / cogaps_from_file_cpp do (reference src/Cogaps.cpp:64-139, 148-186, 205-227)
 * against the C ABI of include/cogaps_hip.h.  `allParams` is a key = value list with exactly the keys
 * getGapsParameters reads -- the S4 slots of allParams$gaps (subsetDim, subsetIndices, takePumpSamples, seed,
 * nPatterns, nIterations, alphaA, alphaP, maxGibbsMassA, maxGibbsMassP, sparseOptimization, whichMatrixFixed,
 * fixedPatterns) and the list entries (transposeData, nThreads, workerID, messages, outputFrequency,
 * checkpointOutFile, checkpointInterval, nSnapshots, snapshotPhase, checkpointInFile, asynchronousUpdates) -- and
 * is turned into cogaps_params by the same rules, line for line.  The result is printed the way
 * cogapsRun names it (Cogaps.cpp:162-186) so that tests/test_gpu_parity.py can compare it with the ctypes path.
 *
 * Besides the two run entry points: entry=info prints what getFileInfo_cpp returns (Cogaps.cpp:243-254) through
 * cogaps_file_info (two calls: sizes, then names -- the '\n'-joined buffers split back into one name per line);
 * interruptAt=k installs the Rcpp::checkUserInterrupt() hook (GapsRunner.cpp:280) as cogaps_params.interrupt and
 * raises the "interrupt" when the hook is polled for the k-th time: the run must end there with a non-zero code
 * and the library's message, nothing leaked (cogaps_run releases its session on the error path).
 *
 * usage: rcpp_shim_test <matrix file> [key=value ...]          (no Python anywhere in this process)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cogaps_hip.h"

struct all_params {                 /* the R list + S4 object the reference's entry point receives */
    /* allParams$gaps slots */
    unsigned subsetDim; unsigned *subsetIndices; unsigned nSubsetIndices;
    int takePumpSamples; int seed; int nPatterns; int nIterations;
    float alphaA, alphaP, maxGibbsMassA, maxGibbsMassP;
    int sparseOptimization; char whichMatrixFixed; const float *fixedPatterns; unsigned fixedRows, fixedCols;
    /* allParams list entries */
    int transposeData, nThreads, workerID, messages, outputFrequency, checkpointInterval, nSnapshots, asynchronousUpdates;
    const char *checkpointOutFile, *checkpointInFile, *snapshotPhase;
};

static void defaults(struct all_params *a)
{   /* R/class-CogapsParams.R:99-123 and R/CoGAPS.R:90-95 */
    memset(a, 0, sizeof(*a));
    a->seed = 42; a->nPatterns = 7; a->nIterations = 50000; a->alphaA = a->alphaP = 0.01f; a->maxGibbsMassA = a->maxGibbsMassP = 100.f;
    a->whichMatrixFixed = 'N'; a->nThreads = 1; a->workerID = 1; a->messages = 1; a->outputFrequency = 1000; a->asynchronousUpdates = 1;
    a->checkpointOutFile = "gaps_checkpoint.out"; a->snapshotPhase = "sampling";
}

static int set(struct all_params *a, const char *k, const char *v)
{
#define I(name) if (!strcmp(k, #name)) { a->name = atoi(v); return 0; }
#define F(name) if (!strcmp(k, #name)) { a->name = (float)atof(v); return 0; }
    I(takePumpSamples) I(seed) I(nPatterns) I(nIterations) I(sparseOptimization) I(transposeData) I(nThreads) I(workerID) I(messages)
    I(outputFrequency) I(checkpointInterval) I(nSnapshots) I(asynchronousUpdates)
    F(alphaA) F(alphaP) F(maxGibbsMassA) F(maxGibbsMassP)
    if (!strcmp(k, "subsetDim")) { a->subsetDim = (unsigned)atoi(v); return 0; }
    if (!strcmp(k, "snapshotPhase")) { a->snapshotPhase = v; return 0; }
    if (!strcmp(k, "checkpointInFile")) { a->checkpointInFile = v; return 0; }
    if (!strcmp(k, "checkpointOutFile")) { a->checkpointOutFile = v; return 0; }
    if (!strcmp(k, "subsetIndices")) {          /* first:last, 1-based, as R's integer vector */
        unsigned lo = 0, hi = 0;
        if (sscanf(v, "%u:%u", &lo, &hi) != 2 || lo < 1 || hi < lo) return 1;
        a->nSubsetIndices = hi - lo + 1; a->subsetIndices = (unsigned *)malloc(sizeof(unsigned) * a->nSubsetIndices);
        for (unsigned i = 0; i < a->nSubsetIndices; ++i) a->subsetIndices[i] = lo + i;
        return 0;
    }
    return 1;
#undef I
#undef F
}

/* getGapsParameters (Cogaps.cpp:64-139), line for line against cogaps_params */
static int get_gaps_parameters(const struct all_params *a, cogaps_params *p)
{
    cogaps_default_params(p);
    if (a->subsetDim > 0) {                                              /* :69-81 */
        p->subsetData = 1; p->subsetGenes = (a->subsetDim == 1);
        p->dataIndicesSubset = a->subsetIndices; p->nSubset = a->nSubsetIndices;
    }
    p->transposeData = a->transposeData;
    p->runningDistributed = a->subsetDim > 0;                            /* :82 */
    p->maxThreads = (uint32_t)a->nThreads;                               /* :86-92 */
    p->workerID = (uint32_t)a->workerID;
    p->printMessages = a->messages && (a->workerID == 1);
    p->outputFrequency = (uint32_t)a->outputFrequency;
    p->checkpointInterval = (uint32_t)a->checkpointInterval;
    p->takePumpSamples = a->takePumpSamples;
    p->seed = (uint32_t)a->seed;                                         /* :95-103 */
    p->nPatterns = (uint32_t)a->nPatterns;
    p->nIterations = (uint32_t)a->nIterations;
    p->alphaA = a->alphaA; p->alphaP = a->alphaP;
    p->maxGibbsMassA = a->maxGibbsMassA; p->maxGibbsMassP = a->maxGibbsMassP;
    p->useSparseOptimization = a->sparseOptimization;
    p->asynchronousUpdates = a->asynchronousUpdates;
    if (a->nSnapshots > 0) p->snapshotFrequency = p->nIterations / (uint32_t)a->nSnapshots;      /* :106-110 */
    p->snapshotPhase = !strcmp(a->snapshotPhase, "equilibration") ? 1 : (!strcmp(a->snapshotPhase, "sampling") ? 2 : 0);   /* :113-121 */
    p->whichMatrixFixed = a->whichMatrixFixed;                           /* :124-130 */
    if (a->whichMatrixFixed != 'N') { p->fixedPatterns = a->fixedPatterns; p->fixedRows = a->fixedRows; p->fixedCols = (int32_t)a->fixedCols; }
    if (a->checkpointInFile) { fprintf(stderr, "checkpoints are disabled in this build (Cogaps.cpp:224-231)\n"); return 1; }   /* :133-137 */
    return 0;
}

/* Rcpp::checkUserInterrupt() stand-in: the k-th poll reports a pending interrupt */
struct interrupt_state { int polls, raiseAt; };
static int poll_interrupt(void *arg)
{
    struct interrupt_state *st = (struct interrupt_state *)arg;
    st->polls++;
    return st->raiseAt > 0 && st->polls >= st->raiseAt;
}

/* getFileInfo_cpp (Cogaps.cpp:243-254): dimensions + row / column names, one per line */
static int file_info(const char *path)
{
    uint32_t nr = 0, nc = 0; size_t rn = 0, cn = 0;
    if (cogaps_file_info(path, &nr, &nc, NULL, 0, &rn, NULL, 0, &cn)) { fprintf(stderr, "CoGAPS terminated: %s\n", cogaps_last_error()); return 1; }
    char *rows = (char *)calloc(rn ? rn : 1, 1), *cols = (char *)calloc(cn ? cn : 1, 1);
    if (cogaps_file_info(path, &nr, &nc, rows, rn, NULL, cols, cn, NULL)) { fprintf(stderr, "CoGAPS terminated: %s\n", cogaps_last_error()); free(rows); free(cols); return 1; }
    printf("dimensions %u %u\n", nr, nc);
    const char *label[2] = {"rowName", "colName"}; char *buf[2]; buf[0] = rows; buf[1] = cols;
    for (int w = 0; w < 2; ++w) {
        unsigned n = 0;
        for (char *tok = buf[w]; tok && *tok; ) {
            char *nl = strchr(tok, '\n'); if (nl) *nl = 0;
            printf("%s %u %s\n", label[w], n++, tok);
            tok = nl ? nl + 1 : NULL;
        }
        printf("%ss %u\n", label[w], n);
    }
    free(rows); free(cols);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <matrix file> [key=value ...]\n", argv[0]); return 2; }
    struct all_params a; defaults(&a);
    int fromFile = 0, info = 0;
    struct interrupt_state irq; irq.polls = 0; irq.raiseAt = 0;
    for (int i = 2; i < argc; ++i) {
        char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bad argument %s\n", argv[i]); return 2; }
        *eq = 0;
        if (!strcmp(argv[i], "entry")) { fromFile = !strcmp(eq + 1, "file"); info = !strcmp(eq + 1, "info"); continue; }    /* cogaps_cpp, cogaps_from_file_cpp or getFileInfo_cpp */
        if (!strcmp(argv[i], "interruptAt")) { irq.raiseAt = atoi(eq + 1); continue; }
        if (set(&a, argv[i], eq + 1)) { fprintf(stderr, "unknown key %s\n", argv[i]); return 2; }
    }
    if (info) return file_info(argv[1]);
    cogaps_params p;
    if (get_gaps_parameters(&a, &p)) return 1;
    if (irq.raiseAt > 0) { p.interrupt = poll_interrupt; p.interruptArg = &irq; }
    cogaps_result r; memset(&r, 0, sizeof(r));
    int rc;
    if (fromFile) rc = cogaps_run_from_file(argv[1], &p, NULL, &r);              /* cogaps_from_file_cpp, Cogaps.cpp:217-227 */
    else {                                                                        /* cogaps_cpp, Cogaps.cpp:205-215 */
        uint32_t nr = 0, nc = 0; float *d = NULL;
        if (cogaps_read_matrix_file(argv[1], &nr, &nc, &d)) { fprintf(stderr, "CoGAPS terminated: %s\n", cogaps_last_error()); return 1; }
        rc = cogaps_run(d, nr, nc, &p, NULL, &r);
        cogaps_matrix_free(d);
    }
    if (rc) {                                                                                    /* GAPS_ERROR -> Rcpp::stop */
        fprintf(stderr, "CoGAPS terminated: %s\n", cogaps_last_error());
        if (irq.raiseAt > 0) fprintf(stderr, "interrupt polls %d\n", irq.polls);
        free(a.subsetIndices);
        return 1;
    }
    /* the list cogapsRun returns (Cogaps.cpp:162-186) */
    printf("nGenes %u nSamples %u nPatterns %u\n", r.nGenes, r.nSamples, r.nPatterns);
    printf("seed %u\nmeanChiSq %.9g\ntotalUpdates %llu\n", r.seed, r.meanChiSq, (unsigned long long)r.totalUpdates);
    printf("averageQueueLengthA %.9g\naverageQueueLengthP %.9g\n", r.averageQueueLengthA, r.averageQueueLengthP);
    printf("chisq"); for (uint32_t i = 0; i < r.nHistory; ++i) printf(" %.9g", r.chisqHistory[i]); printf("\n");
    printf("atomsA"); for (uint32_t i = 0; i < r.nHistory; ++i) printf(" %u", r.atomHistoryA[i]); printf("\n");
    printf("atomsP"); for (uint32_t i = 0; i < r.nHistory; ++i) printf(" %u", r.atomHistoryP[i]); printf("\n");
    double sa = 0, sp = 0, sda = 0, sdp = 0;
    for (size_t i = 0; i < (size_t)r.nGenes * r.nPatterns; ++i) { sa += r.Amean[i]; sda += r.Asd[i]; }
    for (size_t i = 0; i < (size_t)r.nSamples * r.nPatterns; ++i) { sp += r.Pmean[i]; sdp += r.Psd[i]; }
    printf("sumAmean %.17g\nsumPmean %.17g\nsumAsd %.17g\nsumPsd %.17g\n", sa, sp, sda, sdp);
    printf("Amean00 %.9g Pmean00 %.9g\n", r.Amean[0], r.Pmean[0]);
    printf("snapshots %u %u pump %d\n", r.nEquilibrationSnapshots, r.nSamplingSnapshots, r.pumpMatrix != NULL);
    printf("buildReport %s\ncheckpointsEnabled %d\n", cogaps_build_report(), cogaps_checkpoints_enabled());
    cogaps_result_free(&r);
    free(a.subsetIndices);
    return 0;
}
