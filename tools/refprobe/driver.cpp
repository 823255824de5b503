// refprobe driver (tools/refprobe/README.md): fills the reference's GapsParameters from key=value arguments, calls its
// language-neutral entry gaps::run(path, params, uncertaintyPath, &randState) -- what cogaps_from_file_cpp reaches
// (src/Cogaps.cpp:148-160, 217-222; src/GapsRunner.cpp:119-123) -- and prints the GapsResult.  Own code: nothing here is
// taken from the reference besides calling its public interface.
//
// arguments (all key=value; data= is required):
//   data=PATH unc=PATH nPatterns=3 nIterations=1000 seed=0 threads=1 outFreq=500 sparse=0 transpose=0
//   subsetDim=0|1|2 subset=PATH (text file of 1-based indices)  fixed=N|A|P fixedFile=PATH (.csv/.tsv/.mtx, rows x nPatterns)
//   alphaA= alphaP= maxGibbsA= maxGibbsP= pump=0 snapshots=0 snapshotPhase=3 async=1 messages=0
// output grammar (one record per line, floats as %.9g):
//   dims G S K | atomsA .. | atomsP .. | chisq .. | totalUpdates N | meanChiSq x | qA x | qP x
//   row <Amean|Asd|Pmean|Psd> <r> v0 .. vK-1      (rows 0, middle, last)
//   hash <Amean|Asd|Pmean|Psd|pump|meanPattern> <fnv1a-64 of the float bit patterns, row-major> <count>
//   samplerSeconds x (the reference's own start-to-end interval, microsecond clock) | wallSeconds x (load + run)
#include "GapsRunner.h"
#include "GapsParameters.h"
#include "GapsResult.h"
#include "math/Random.h"
#include "data_structures/Matrix.h"
#include <boost/date_time/posix_time/posix_time.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include <stdint.h>

static uint64_t fnv_matrix(const Matrix &m, uint64_t *count)
{
    uint64_t h = 1469598103934665603ull, n = 0;
    for (unsigned i = 0; i < m.nRow(); ++i)
        for (unsigned j = 0; j < m.nCol(); ++j) {
            float f = m(i, j); uint32_t u; memcpy(&u, &f, 4);
            for (int b = 0; b < 4; ++b) { h ^= (u >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
            ++n;
        }
    *count = n;
    return h;
}
static void print_rows(const char *name, const Matrix &m)
{
    if (m.nRow() == 0) return;
    const unsigned rows[3] = {0u, m.nRow() / 2u, m.nRow() - 1u};
    for (int k = 0; k < 3; ++k) {
        printf("row %s %u", name, rows[k]);
        for (unsigned j = 0; j < m.nCol(); ++j) printf(" %.9g", m(rows[k], j));
        printf("\n");
    }
    uint64_t n = 0; const uint64_t h = fnv_matrix(m, &n);
    printf("hash %s %llu %llu\n", name, (unsigned long long)h, (unsigned long long)n);
}

int main(int argc, char **argv)
{
    std::map<std::string, std::string> a;
    for (int i = 1; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { fprintf(stderr, "bad argument %s\n", argv[i]); return 2; }
        a[std::string(argv[i], eq - argv[i])] = std::string(eq + 1);
    }
    auto S = [&](const char *k, const char *d) { return a.count(k) ? a[k] : std::string(d); };
    auto I = [&](const char *k, long d) { return a.count(k) ? atol(a[k].c_str()) : d; };
    auto F = [&](const char *k, double d) { return a.count(k) ? atof(a[k].c_str()) : d; };
    if (!a.count("data")) { fprintf(stderr, "data=PATH is required\n"); return 2; }
    const std::string data = a["data"], unc = S("unc", "");
    std::vector<unsigned> idx;
    const long subsetDim = I("subsetDim", 0);
    if (subsetDim > 0) {
        std::ifstream f(S("subset", "").c_str()); unsigned v;
        while (f >> v) idx.push_back(v);
        if (idx.empty()) { fprintf(stderr, "subset file empty\n"); return 2; }
    }
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    GapsParameters p(data, I("transpose", 0) != 0, subsetDim > 0, subsetDim == 1, idx);
    p.nPatterns = (unsigned)I("nPatterns", 3); p.nIterations = (unsigned)I("nIterations", 1000); p.seed = (uint32_t)I("seed", 0);
    p.maxThreads = (unsigned)I("threads", 1); p.outputFrequency = (unsigned)I("outFreq", 500);
    p.useSparseOptimization = I("sparse", 0) != 0; p.asynchronousUpdates = I("async", 1) != 0;
    p.printMessages = I("messages", 0) != 0; p.printThreadUsage = false;
    p.checkpointInterval = 0;                       // (createCheckpoint is compiled in whatever GAPS_DISABLE_CHECKPOINTS says, GapsRunner.cpp:229)
    p.alphaA = (float)F("alphaA", 0.01); p.alphaP = (float)F("alphaP", 0.01);
    p.maxGibbsMassA = (float)F("maxGibbsA", 100.0); p.maxGibbsMassP = (float)F("maxGibbsP", 100.0);
    p.takePumpSamples = I("pump", 0) != 0;
    const long snaps = I("snapshots", 0);
    if (snaps > 0) { p.snapshotFrequency = p.nIterations / (unsigned)snaps; p.snapshotPhase = (GapsAlgorithmPhase)I("snapshotPhase", 3); }
    const std::string fixed = S("fixed", "N");
    if (fixed != "N") {
        p.useFixedPatterns = true; p.whichMatrixFixed = fixed[0];
        p.fixedPatterns = Matrix(S("fixedFile", ""), false, false, std::vector<unsigned>());
    }
    GapsRandomState rs(p.seed);
    GapsResult r(gaps::run(data, p, unc, &rs));
    const double samplerSeconds = (double)refprobe_last_interval_us() * 1e-6;
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("dims %u %u %u\n", p.nGenes, p.nSamples, p.nPatterns);
    printf("atomsA"); for (size_t i = 0; i < r.atomHistoryA.size(); ++i) printf(" %u", r.atomHistoryA[i]); printf("\n");
    printf("atomsP"); for (size_t i = 0; i < r.atomHistoryP.size(); ++i) printf(" %u", r.atomHistoryP[i]); printf("\n");
    printf("chisq"); for (size_t i = 0; i < r.chisqHistory.size(); ++i) printf(" %.9g", r.chisqHistory[i]); printf("\n");
    printf("totalUpdates %llu\n", (unsigned long long)r.totalUpdates);
    printf("meanChiSq %.9g\nqA %.9g\nqP %.9g\n", r.meanChiSq, r.averageQueueLengthA, r.averageQueueLengthP);
    print_rows("Amean", r.Amean); print_rows("Asd", r.Asd); print_rows("Pmean", r.Pmean); print_rows("Psd", r.Psd);
    if (p.takePumpSamples) { print_rows("pump", r.pumpMatrix); print_rows("meanPattern", r.meanPatternAssignment); }
    for (size_t i = 0; i < r.equilibrationSnapshotsA.size(); ++i) { uint64_t n; printf("snapE %zu %llu %llu\n", i, (unsigned long long)fnv_matrix(r.equilibrationSnapshotsA[i], &n), (unsigned long long)fnv_matrix(r.equilibrationSnapshotsP[i], &n)); }
    for (size_t i = 0; i < r.samplingSnapshotsA.size(); ++i) { uint64_t n; printf("snapS %zu %llu %llu\n", i, (unsigned long long)fnv_matrix(r.samplingSnapshotsA[i], &n), (unsigned long long)fnv_matrix(r.samplingSnapshotsP[i], &n)); }
    printf("samplerSeconds %.6f\n", samplerSeconds);
    printf("wallSeconds %.6f\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    return 0;
}
