"""Soak of the chained launch's hand-over beside a FOREIGN kernel (VERDICT round 5, next #5a): the headline chain is stepped for `--launches`
chained launches while a second process keeps the same GPU busy with long streaming kernels (a 4 GiB tensor scaled in place, back to back, on its
own HIP context).  A chained launch wants a compute unit per workgroup -- its workgroups carry 136 KB of LDS and a full register file -- so beside a
foreign kernel they are scheduled late and the generator workgroup's bounded wait (2^22 turns, >= 2 s) is what stands between a slow hand-over
and a lost one.  Reported: launches, recoveries (cogaps_session_chain_recoveries: hand-overs that ran out and were completed by
chain_recover_kernel; 0 expected), the launch-period tail (device clock of every launch) quiet and disturbed, and -- the chain is deterministic --
whether the disturbed run ended in the same state as the quiet one.

    python tools/dev_soak.py [--launches 1000000] [--genes 20000 --samples 2000 --patterns 50]      (through gpurun)
"""
import argparse, hashlib, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FOREIGN = r"""
import sys, time, torch
x = torch.ones(1 << 30, dtype=torch.float32, device="cuda")      # 4 GiB: one pass = 8 GiB of traffic, ~1.5 ms
n = 0
sys.stdout.write("ready\n"); sys.stdout.flush()
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(64):
        x.mul_(1.0000001)
    torch.cuda.synchronize(); n += 64
sys.stdout.write("kernels %d\n" % n); sys.stdout.flush()
"""


def digest(S):
    h = hashlib.sha256()
    for w in "AP":
        a = S.atoms(w)
        h.update(np.ascontiguousarray(a["pos"]).tobytes()); h.update(np.ascontiguousarray(a["mass"]).tobytes()); h.update(np.ascontiguousarray(S.matrix(w)).tobytes())
    return h.hexdigest()[:16]


def run(data, n_iter, iters, params, foreign_seconds=None):
    from cogaps_amd import _capi
    proc = None
    if foreign_seconds:
        proc = subprocess.Popen([sys.executable, "-c", FOREIGN, str(foreign_seconds)], stdout=subprocess.PIPE, text=True)
        assert proc.stdout.readline().strip() == "ready"
    S = _capi.Session(data, nIterations=n_iter, **params)
    S.set_timing(True)
    t0 = time.time(); updates = 0; done = 0
    while done < iters:
        m = min(iters - done, n_iter - done) if done < n_iter else iters - done
        updates += S.run_iterations(1 if done < n_iter else 2, done if done < n_iter else done - n_iter, m); done += m
    dt = time.time() - t0
    out = {"seconds": dt, "proposals": int(updates), "proposals_per_s": updates / dt, "state_digest": digest(S), "atoms": [S.natoms("A"), S.natoms("P")]}
    for w in "AP":
        out[w] = {"chained": bool(S.chained(w)), "recoveries": S.chain_recoveries(w), "launch_period_us": S.launch_period(w) if S.chained(w) else None, "inside_launch_us": S.launch_clock(w) if S.chained(w) else None}
    S.close()
    if proc:
        proc.wait(); out["foreign"] = proc.stdout.read().strip()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=1000000)
    ap.add_argument("--genes", type=int, default=20000); ap.add_argument("--samples", type=int, default=2000); ap.add_argument("--patterns", type=int, default=50)
    ap.add_argument("--sparse", action="store_true", help="the sparse model (95 %% zeros): chain_sparse_kernel -- two proposals per evaluation workgroup, the attempt lanes in the hand-over, the 448-attempt window")
    a = ap.parse_args()
    import bench
    data = bench.synthetic_dense(a.genes, a.samples)
    if a.sparse: data = (data * (np.random.Generator(np.random.MT19937(777)).random(data.shape) >= 0.95)).astype(np.float32)
    # ~2150 chained launches of the A sampler per iteration once the chain is populated (58 930 batches in 20 iterations, 73 % of them A's)
    iters = max(8, int(a.launches / (4900.0 if a.sparse else 2150.0)) + 1)
    n_iter = max(100, (iters + 1) // 2)
    params = dict(nPatterns=a.patterns, seed=42, outputFrequency=max(1, n_iter // 10))
    if a.sparse: params["sparseOptimization"] = True
    quiet = run(data, n_iter, iters, params)
    disturbed = run(data, n_iter, iters, params, foreign_seconds=max(60.0, 6.0 * quiet["seconds"]))
    rec = {"what": ("sparse model, %d x %d, " % (a.genes, a.samples) if a.sparse else "") + "headline chain, %d iterations (schedule of %d + %d), first alone, then beside a second process streaming over a 4 GiB tensor on the same GPU" % (iters, n_iter, n_iter),
           "quiet": quiet, "disturbed": disturbed,
           "same_final_state": quiet["state_digest"] == disturbed["state_digest"],
           "recoveries_total": sum(disturbed[w]["recoveries"] + quiet[w]["recoveries"] for w in "AP"),
           "chained_launches_disturbed_A": (disturbed["A"]["launch_period_us"] or {}).get("launches")}
    print(json.dumps(rec, indent=1))
    return 0 if rec["same_final_state"] else 1


if __name__ == "__main__":
    sys.exit(main())
