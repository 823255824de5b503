"""Generate the committed golden vectors under tests/golden/ with the CPU oracle (oracle/gaps_oracle.c).

Two kinds:
  *_seq.npz   sequential reductions + libm log/exp: the arithmetic of the reference's default scalar
              build.  The GIST / modsim K,seed=42,1000+1000 atom histories of this mode ARE the SURVEY.md
              section 8c fingerprints of the reference binary (asserted below before anything is written).
  *_lane.npz  the lane-strided reduction order of the HIP kernels (W lanes x float4, xor butterfly)
              + the portable log/exp: what the GPU path must reproduce bit for bit.
Run in the build container: python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
SURVEY_8C = {
    "gist": dict(atomsA=[2875, 3838, 3940, 3123, 3268, 3215, 3372, 3344, 3394, 3422, 3441, 3450, 3507, 3565, 3597, 3496, 3597, 3532, 3600, 3608],
                 atomsP=[36, 51, 58, 66, 70, 75, 78, 80, 81, 87, 85, 90, 88, 94, 100, 100, 101, 95, 96, 95], totalUpdates=6902140),
    "modsim": dict(atomsA=[29, 50, 59, 58, 57, 56, 56, 49, 57, 60, 64, 62, 60, 60, 58, 56, 58, 54, 55, 52],
                   atomsP=[31, 41, 47, 53, 63, 69, 57, 63, 62, 69, 62, 69, 70, 67, 72, 59, 57, 60, 58, 58], totalUpdates=225435),
}


def red_width(n):   # cogaps_reduction_width (cogaps_hip.cpp)
    need, w = (n + 3) // 4, 64
    while w < need and w < 16384:
        w <<= 1
    return w


def save(name, r, **meta):
    np.savez_compressed(os.path.join(G, name), Amean=r["Amean"], Pmean=r["Pmean"], Asd=r["Asd"], Psd=r["Psd"],
                        chisq=r["chisq"], atomsA=r["atomsA"], atomsP=r["atomsP"], totalUpdates=np.uint64(r["totalUpdates"]),
                        meanChiSq=np.float32(r["meanChiSq"]), avgQueueA=np.float32(r["averageQueueLengthA"]),
                        avgQueueP=np.float32(r["averageQueueLengthP"]), **meta)


def main():
    gist = po.read_mtx(os.path.join(G, "GIST.mtx"))
    modsim = np.loadtxt(os.path.join(G, "modsimdata.csv"), delimiter=",").astype(np.float32)
    for name, data, k in (("gist", gist, 7), ("modsim", modsim, 3)):
        r = po.run(data, nPatterns=k, nIterations=1000, seed=42, outputFrequency=100)
        fp = SURVEY_8C[name]
        assert r["atomsA"].tolist() == fp["atomsA"] and r["atomsP"].tolist() == fp["atomsP"] and r["totalUpdates"] == fp["totalUpdates"], name
        save("%s_k%d_s42_i1000_seq.npz" % (name, k), r)
        ng, ns = data.shape
        wA, wP = red_width(ns), red_width(ng)
        r = po.run(data, nPatterns=k, nIterations=300, seed=42, outputFrequency=30, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4)
        save("%s_k%d_s42_i300_lane.npz" % (name, k), r, redW_A=wA, redW_P=wP)
        print(name, "ok", r["atomsA"][-3:], r["totalUpdates"])
    # gene-wise shard (GWCoGAPS, subsetDim = 1): genes 1..600 of GIST, then the fixed-matrix second pass
    idx = np.arange(1, 601, dtype=np.uint32)
    wA, wP = red_width(gist.shape[1]), red_width(600)
    r1 = po.run(gist, nPatterns=5, nIterations=200, seed=7, outputFrequency=40, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4,
                subsetIndices=idx, subsetDim=1)
    save("gist_shard600_k5_s7_i200_lane.npz", r1, redW_A=wA, redW_P=wP)
    fixedP = r1["Pmean"] / np.maximum(r1["Pmean"].max(axis=0, keepdims=True), 1e-30)
    r2 = po.run(gist, nPatterns=5, nIterations=200, seed=7, outputFrequency=40, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4,
                subsetIndices=idx, subsetDim=1, whichMatrixFixed="P", fixedPatterns=fixedP.astype(np.float32))
    save("gist_shard600_k5_s7_i200_fixedP_lane.npz", r2, redW_A=wA, redW_P=wP, fixedP=fixedP.astype(np.float32))
    print("shard ok", r1["totalUpdates"], r2["totalUpdates"], float(np.abs(r2["Pmean"]).max()))


if __name__ == "__main__":
    main()
