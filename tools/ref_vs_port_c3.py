"""The reference BUILD (tools/refprobe) against this repository's CPU port (oracle/, sequential sums + libm + OpenMP over the queue) on
BASELINE configs[2] -- synthetic dense 20000 x 2000, nPatterns = 50, seed 42 -- in the build container: same matrix, same parameters,
same thread counts, each timing the sampler alone (the reference's own start-to-end interval; the port's samplerSeconds).  Both must
print the same chain (atom histories, totalUpdates, meanChiSq, queue lengths): the comparison is of two programs computing the same thing.

    python tools/ref_vs_port_c3.py [--iterations 100] [--threads 8,1] [--one-thread-iterations 30] > profiles/r04_reference_vs_port_container.json
    python tools/ref_vs_port_c3.py --sparse --genes 5000 --samples 1250 --iterations 40 --threads 8 > profiles/r06_reference_vs_port_sparse_container.json
        (round 6: the SparseNormalModel path -- SparseNormalModel.cpp:152-292 under the same `omp parallel for` -- on SURVEY section 6's probe shape,
         95 % of the entries zeroed as bench.py --sparse zeroes them)

`bench.py` reads the committed record and prints `cpu_baseline.port_over_reference_build` from it, so that a reader can translate the
GPU / port ratio measured on the GPU box (where only the port can run) into a GPU / reference-build ratio -- with the host named.
"""
import argparse
import json
import os
import platform
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools", "refprobe")]
import bench            # noqa: E402
import pyoracle as po   # noqa: E402
import refprobe as rp   # noqa: E402


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--threads", default="8,1")
    ap.add_argument("--one-thread-iterations", type=int, default=30, help="iterations (per phase) of the one-thread leg: the whole run takes an hour on one core")
    ap.add_argument("--port-first", action="store_true", help="run the port before the reference build (the host is shared: a second pass in the other order shows how much of a difference is drift)")
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--sparse", action="store_true", help="the SparseNormalModel: the same product with 95 %% of the entries zeroed (bench.py --sparse), sparseOptimization on in both programs")
    a = ap.parse_args()
    binary = rp.build()
    data = bench.synthetic_dense(a.genes, a.samples)
    if a.sparse:
        data *= (np.random.Generator(np.random.MT19937(777)).random(data.shape) >= 0.95)
    legs = []
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c3.csv")
        t0 = time.time()
        rp.write_matrix(path, data)
        sys.stderr.write("matrix written as text in %.0f s\n" % (time.time() - t0))
        for thr in [int(x) for x in a.threads.split(",")]:
            it = a.iterations if thr > 1 else min(a.iterations, a.one_thread_iterations)
            out_freq = max(1, it // 10)
            def run_ref():
                r = rp.run(binary, path, nPatterns=50, nIterations=it, seed=42, outFreq=out_freq, threads=thr, sparse=1 if a.sparse else 0)
                sys.stderr.write("reference build, %d threads, %d + %d iterations: %.1f s\n" % (thr, it, it, r["samplerSeconds"]))
                return r

            def run_port():
                r = po.run(data, omp=thr > 1, nPatterns=50, nIterations=it, seed=42, outputFrequency=out_freq, maxThreads=thr, sparseOptimization=bool(a.sparse))
                sys.stderr.write("port, %d threads: %.1f s\n" % (thr, r["samplerSeconds"]))
                return r
            if a.port_first:
                o = run_port(); ref = run_ref()
            else:
                ref = run_ref(); o = run_port()
            same = (ref["atomsA"].tolist() == o["atomsA"].tolist() and ref["atomsP"].tolist() == o["atomsP"].tolist() and ref["totalUpdates"] == o["totalUpdates"]
                    and ref["meanChiSq"] == np.float32(o["meanChiSq"]) and ref["qA"] == np.float32(o["averageQueueLengthA"]) and ref["qP"] == np.float32(o["averageQueueLengthP"])
                    and all(ref["hashes"][n][0] == rp.fnv_matrix(o[n]) for n in ("Amean", "Asd", "Pmean", "Psd")))
            legs.append(dict(threads=thr, order=("port, reference build" if a.port_first else "reference build, port"), iterations_per_phase=it, totalUpdates=ref["totalUpdates"], same_chain_bit_for_bit=bool(same),
                             reference_build_seconds=round(ref["samplerSeconds"], 2), port_seconds=round(o["samplerSeconds"], 2),
                             reference_build_proposals_per_s=round(ref["totalUpdates"] / ref["samplerSeconds"], 1), port_proposals_per_s=round(o["totalUpdates"] / o["samplerSeconds"], 1),
                             port_over_reference_build=round(ref["samplerSeconds"] / o["samplerSeconds"], 4),
                             atomsA=ref["atomsA"].tolist(), meanChiSq=float(ref["meanChiSq"]), qA=float(ref["qA"]), qP=float(ref["qP"])))
    rec = dict(what="reference build (tools/refprobe: /root/reference/src + this repository's stand-in Boost headers, g++ -O2 -fopenmp, scalar SIMD path) vs the CPU port "
                    "(oracle/gaps_oracle.c, sequential sums, libm, OpenMP over the queue) on %s, sampler time only, same container" % ("the sparse model (SURVEY section 6's probe shape)" if a.sparse else "BASELINE configs[2]"),
               workload="synthetic %s %dx%d fp32, nPatterns=50, seed=42, asynchronous sampler" % ("sparse (95 % zeros, sparseOptimization)" if a.sparse else "dense", a.genes, a.samples),
               host=dict(cpu=cpu_model(), logical_cpus=os.cpu_count(), note="the build container, not the GPU box"),
               legs=legs)
    print(json.dumps(rec, indent=1))
    return 0 if all(l["same_chain_bit_for_bit"] for l in legs) else 1


if __name__ == "__main__":
    sys.exit(main())
