#!/usr/bin/env python
"""bench.py -- Gibbs atom proposals/sec of the asynchronous sampler hot path on MI355X.

A "step" is one Gibbs iteration (A.update, P.sync, P.update, A.sync [+ statistics when sampling]) =
one pass of runOnePhase's loop body (reference src/GapsRunner.cpp:272-327).  The workload at N=1 is
BASELINE.json configs[2]: synthetic dense 20000x2000 fp32, nPatterns=50, asynchronous sampler,
seed 42, default uncertainty, alpha 0.01, maxGibbsMass 100.  The chain runs W+K iterations: the
first ceil((W+K)/2) equilibrate (annealed), the rest sample; the first W are untimed warm-up and
exactly K are timed.  With --gpus N each rank runs one GWCoGAPS gene-wise shard of that size
(BASELINE.json configs[3], nSets = N): independent chains, no data-path collective; the single
exchange of the path, the all-gather of each shard's sample factor for findConsensusMatrix
(reference R/DistributedCogaps.R:71-78), is issued once after the timed iterations, inside the
timed region.

value = proposals (the reference's totalUpdates counter, GapsRunner.cpp:297,476) processed by all
ranks in the timed region / max-over-ranks wall time.  Inputs are resident in HBM before the timed
region starts (session creation uploads them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Gibbs atom proposals/sec, 20000x2000 dense D, nPatterns=50, at 1/2/4/8 GPUs"
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synthetic_dense(n_genes, n_samples, rank=10, seed=12345):
    """D = (A0 P0^T) * (0.9 + 0.2 u), true rank 10, A0 70% zeros else Gamma(2, 0.5), P0 50% zeros else
    Gamma(2, 0.5) (SURVEY.md section 8d recipe; numpy MT19937 stream)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    a0 = rng.gamma(2.0, 0.5, size=(n_genes, rank)) * (rng.random((n_genes, rank)) >= 0.7)
    p0 = rng.gamma(2.0, 0.5, size=(n_samples, rank)) * (rng.random((n_samples, rank)) >= 0.5)
    d = (a0 @ p0.T) * (0.9 + 0.2 * rng.random((n_genes, n_samples)))
    return np.ascontiguousarray(d, dtype=np.float32)


def cpu_baseline(data, params, budget_s, state, first_step, n_steps, threads_only=None):
    """The oracle (oracle/gaps_oracle.c: OpenMP over the queue exactly like the reference's `#pragma omp parallel for`,
    sequential fp32 reductions + libm = the reference's scalar build) timed on this host ON THE ITERATIONS THE GPU IS TIMED ON: the
    port's session takes over the chain state the GPU had at the start of its timed window (atoms and factor matrices, copied out
    before the timed region; go_import_state rebuilds the A*P caches) and runs the window's iterations -- as many as fit the time
    budget -- with its own generators: the same populated chain, the same batch lengths, like for like in phase.  `"kind": "port"`:
    only this repository reaches the GPU box, so the comparator is the port, not the reference binary; BASELINE.md section 4 records how the
    two compare where both can run (configs[2] whole, build container: the same speed within +-25 % at 8 threads, the port 1.4x faster on one;
    printed with the line as cpu_baseline.port_over_reference_build).  A batch holds only ~50-160 proposals, so
    threads beyond a handful only add fork/join cost: 8, 16, 24 and 32 threads share most of the budget (round 5: the best thread count is
    searched for, not assumed -- a 10x claim is against the best of them); the ONE-thread figure BASELINE.md section 3 quotes gets the
    rest -- whole iterations as far as its share reaches, at least one (it never sets `value`).  (Round 6: the nproc-thread leg
    is gone -- 256 threads on ~150 proposals per batch measured fork/join, 800 proposals/s, and cost six seconds of every run.)
    `value` is the best whole-iteration rate, every thread count's rate is listed in `by_threads`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    ncpu = os.cpu_count() or 1
    small = sorted({min(ncpu, c) for c in (8, 16, 24, 32)})
    plan = [(t, 0.88 * budget_s / len(small)) for t in small]
    if 1 not in small:
        plan.append((1, 0.12 * budget_s))
    if threads_only is not None:      # (N > 1: one leg per rank, all ranks at once)
        plan = [(int(threads_only), 0.4 * budget_s)]
    n_iter = params["nIterations"]
    best, by_threads = None, []
    for threads, share in plan:
        O = po.Session(data, omp=True, maxThreads=threads, math_mode=po.MATH_LIBM, redW_A=1, redW_P=1, redG=1, **params)
        O.import_state(state["atomsA"], state["A"], state["atomsP"], state["P"])
        props, it, t0 = 0, 0, time.time()
        # (every leg runs whole iterations: the nproc-thread leg of rounds 3-5 was cut into slices of an iteration -- each slice pays the
        # iteration's two A*P transposes, which is all a slice of one thread's work measures; the one-thread leg gets one or two whole
        # iterations out of its share)
        sliced = False
        while it < n_steps and time.time() - t0 < share:
            step = first_step + it
            O.set_annealing(min(1.0, 2.0 * step / n_iter) if step < n_iter else 1.0)
            nA, nP = O.draw_steps()
            if not sliced:
                O.iterate(nA, nP)
                props += nA + nP
                it += 1
                continue
            doneA = doneP = 0
            while doneA < nA and time.time() - t0 < share:
                a = min(2048, nA - doneA)
                p_ = min(nP - doneP, max(1, (a * nP) // max(nA, 1))) if doneA + a < nA else nP - doneP
                O.iterate(a, p_)
                doneA, doneP, props = doneA + a, doneP + p_, props + a + p_
            if doneA < nA:
                break
            it += 1
        dt = time.time() - t0
        atoms = (O.natoms("A"), O.natoms("P"))
        O.close()
        rate = props / max(dt, 1e-9)
        by_threads.append({"threads": threads, "value": rate, "iterations": it, "proposals": props, "seconds": dt, "atoms_at_end": atoms,
                           "granularity": "slices of an iteration (<= 2048 A steps + their share of P steps)" if sliced else "whole iterations"})
        if it and not sliced and (best is None or rate > best["value"]):
            best = {"value": rate, "unit": "proposals/s", "cores": threads, "kind": "port", "host_cpus": ncpu,
                    "sample": "schedule steps %d-%d of the SAME chain, started from the GPU chain's state at the start of its timed window (%d + %d atoms): "
                              "%d proposals in %.1f s; the GPU's timed window is steps %d-%d; best of OMP threads %s"
                              % (first_step + 1, first_step + it, len(state["atomsA"]["pos"]), len(state["atomsP"]["pos"]), props, dt, first_step + 1, first_step + n_steps, [t for t, _ in plan])}
    if best is None:      # no whole-iteration leg finished an iteration inside its share (a tiny --cpu-seconds, or --steps 0)
        best = {"value": None, "unit": "proposals/s", "cores": None, "kind": "port", "host_cpus": ncpu,
                "sample": "no whole iteration of the window fitted the time budget (%.0f s): see by_threads" % budget_s}
    best["by_threads"] = by_threads
    best["one_thread_value"] = next((b["value"] for b in by_threads if b["threads"] == 1), None)
    return best


def reference_translation(sparse=False):
    """How the port compares with the REFERENCE BUILD where both can run (the build container, tools/ref_vs_port_c3.py: configs[2] whole,
    sampler time only, same chain bit for bit).  Only this repository reaches the GPU box, so the bench's CPU figure is the port's; this
    record -- with its host named -- is what translates a GPU / port ratio into a GPU / reference-build ratio."""
    import glob
    # (the dense model: configs[2] whole; the sparse model -- round 6 -- SURVEY section 6's probe shape, 5000 x 1250 with 95 % zeros)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_vs_port_sparse_container.json" if sparse else "r*_reference_vs_port_container.json")))
    if not files:
        return None
    rec = json.load(open(files[-1]))
    by_thr = {}
    for l in rec["legs"]:
        by_thr.setdefault(l["threads"], []).append(l["port_over_reference_build"])
    best = max(rec["legs"], key=lambda l: l["reference_build_proposals_per_s"])
    return {"source": os.path.relpath(files[-1], ROOT), "host": rec["host"], "workload": rec["workload"],
            # every pass that was made (the host is shared: passes in both orders), and their geometric mean
            "port_over_reference_build": {str(t): {"passes": v, "geometric_mean": float(np.exp(np.mean(np.log(v))))} for t, v in sorted(by_thr.items())},
            "judge_round3_same_container_8_threads": rec.get("judge_round3"),
            "reference_build_best_proposals_per_s_on_that_host": best["reference_build_proposals_per_s"], "reference_build_best_threads": best["threads"],
            "note": "port_over_reference_build = reference build's sampler seconds / port's at that many OpenMP threads, in the build container; "
                    "> 1: the port is the faster of the two there.  The passes differ by more than the two programs do (shared host): read it as 'the same speed within +-25 %'"}


def chains_main(args):
    """--chains C (informational, not the headline configuration): C shards of the headline shape on this one GPU -- the per-GPU
    workload of a GWCoGAPS / scCoGAPS job with more subsets than GPUs.  Default: the batched multi-chain launches (cogaps_batch_*:
    the C chains stepped in lock-step by one stream, one generator and one evaluation launch per step for all of them);
    `--chains-mode threads`: one host thread, stream and session per chain (at most four run side by side)."""
    import threading
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    from cogaps_amd import _capi
    K, W, C = args.steps, args.warmup, args.chains
    n_iter = max(100, (W + K + 1) // 2)
    params = dict(nPatterns=args.patterns, nIterations=n_iter, seed=42, outputFrequency=max(1, n_iter // 10), sparseOptimization=args.sparse)

    def shard(c):
        d = synthetic_dense(args.genes, args.samples, seed=12345 + c)
        if args.sparse:
            d *= (np.random.Generator(np.random.MT19937(777 + c)).random(d.shape) >= 0.95)
        return d
    S = [_capi.Session(shard(c), device=0, **params) for c in range(C)]
    upd = [0] * C
    batched = args.chains_mode == "batched"
    # --chain-groups G: the chains as G batches driven by G host threads on their own streams, so that one group's one-workgroup-
    # per-chain generator launch runs under the other groups' evaluation launches
    G = max(1, min(args.chain_groups, C)) if batched else 1
    groups = [list(range(g, C, G)) for g in range(G)]
    BB = [_capi.Batch([S[c] for c in grp]) for grp in groups] if batched else []

    def span(first, n):      # (phase, first iteration, count) pieces of schedule steps [first, first + n)
        out, done = [], 0
        while done < n:
            it = first + done
            m = min(n - done, n_iter - it) if it < n_iter else n - done
            out.append((1 if it < n_iter else 2, it if it < n_iter else it - n_iter, m))
            done += m
        return out

    def steps(c, first, n):
        for ph, it, m in span(first, n):
            upd[c] += S[c].run_iterations(ph, it, m)

    def phase(first, n):
        t0 = time.perf_counter()
        if batched:
            def drive(g):
                for ph, it, m in span(first, n):
                    for c, u in zip(groups[g], BB[g].run_iterations(ph, it, m)):
                        upd[c] += u
            if G == 1:
                drive(0)
            else:
                th = [threading.Thread(target=drive, args=(g,)) for g in range(G)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
        else:
            th = [threading.Thread(target=steps, args=(c, first, n)) for c in range(C)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return time.perf_counter() - t0
    burn = max(0, 2 * n_iter - (W + K))      # as in main(): the timed steps are the last K of the schedule
    if burn:
        phase(0, burn)
    phase(burn, W)
    upd = [0] * C
    # every chain's state at the start of the timed window, for the CPU comparator's like-for-like legs (copied out before the timed region)
    cpu_states = None if args.no_cpu else [{"atomsA": s.atoms("A"), "A": s.rows("A"), "atomsP": s.atoms("P"), "P": s.rows("P")} for s in S]
    perf0 = [{w: s.perf(w) for w in "AP"} for s in S]
    for b_ in BB:
        b_.set_timing(True)
    torch.cuda.synchronize()
    dt = phase(burn + W, K)
    torch.cuda.synchronize()
    roof = None
    if batched:
        # path-level figure as in the headline line: algorithmic bytes of all chains / summed duration of the batched launches
        perf1 = [{w: s.perf(w) for w in "AP"} for s in S]
        kern, tot_ms, tot_bytes = [], 0.0, 0.0
        for w in "AP":
          for g, b_ in enumerate(BB):
            bp = b_.perf(w)
            mine = [(perf0[c], perf1[c]) for c in groups[g]]
            nbytes = sum(p1[w]["evalBytes"] - p0[w]["evalBytes"] for p0, p1 in mine)
            # steps of the batch in the timed window = the batches of its slowest chain
            steps_w = max(p1[w]["batches"] - p0[w]["batches"] for p0, p1 in mine)
            ev_ms, gen_ms = bp["eval_us"] * steps_w / 1e3, bp["gen_us"] * steps_w / 1e3
            # (no per-launch `achieved` / `frac` here -- round 6: the algorithmic bytes are more than these launches move (the uncertainty row is
            # recomputed, the other matrix's columns hit L2) and the sampled event time is shorter than the launch takes inside the replayed
            # graph, so the quotient exceeded the HBM peak; the path figure over the wall time below is the one that means something)
            kern.append({"kernel": "batched evaluation launch, sampler %s, group %d (%d chains)" % (w, g, len(groups[g])), "steps": int(steps_w), "sampled_launches": bp["sampled"], "avg_launch_us": bp["eval_us"],
                         "algorithmic_bytes_per_launch": nbytes / max(1, steps_w)})
            kern.append({"kernel": "batched generator launch, sampler %s, group %d" % (w, g), "steps": int(steps_w), "avg_launch_us": bp["gen_us"]})
            tot_ms += ev_ms + gen_ms
            tot_bytes += nbytes
        # The path-level figure is taken over the WALL time of the timed region: the sampled launches are the plain-launch remainder
        # of each chunk, whose HIP-event durations leave out what a launch replayed from the graph waits for at the kernel boundary
        # (the previous launch's write-back: 10-30 MB of A*P rows per batched evaluation) -- rocprofv3's in-graph durations are
        # 1.4x the sampled ones (profiles/r02_chains8_*), so bytes / summed samples would flatter the path; with several groups
        # the groups' launches overlap and only the wall clock adds up anyway.  The per-launch samples stay listed as what they are.
        ach = (tot_bytes / 1e9) / dt
        # HBM bytes from the PMC counters (two separate rocprofv3 --pmc passes over this mode with 8 chains in one batch,
        # tools/pmc_pass.sh, committed under profiles/): per step of the batch, weighted by this run's A and P step counts
        traffic, traffic_src = None, None
        if C == 8 and G == 1 and not args.sparse and (args.genes, args.samples, args.patterns) == (20000, 2000, 50):
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_chains8_pmc_traffic.json")))
            if files:
                per = json.load(open(files[-1]))["hbm_bytes_per_step"]
                st = {w: sum(k["steps"] for k in kern if k["kernel"].startswith("batched evaluation launch, sampler %s" % w)) for w in "AP"}
                if st["A"] + st["P"]:
                    traffic = (per["A (generator + fused evaluation)"] * st["A"] + per["P (generator + split evaluation)"] * st["P"]) / (st["A"] + st["P"])
                    traffic_src = "%s: HBM bytes per step of the 8-chain batch (FETCH_SIZE x2 + WRITE_SIZE), A and P steps weighted by this run's counts" % os.path.relpath(files[-1], ROOT)
        roof = {"bound": "hbm", "kernel": "path: batched generator + evaluation launches, algorithmic bytes over the wall time of the timed region", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_step": tot_bytes / max(1, sum(k["steps"] for k in kern if k["kernel"].startswith("batched evaluation"))),
                "sampled_kernel_time_over_wall": tot_ms / (1e3 * dt), "kernels": kern,
                "note": "`kernels` lists HIP-event samples of plain (not graph-replayed) launches: they exclude the boundary write-back a replayed launch waits for; no per-launch fraction is derived from them"}
    # CPU comparator (a job of nSets subsets on one host = nSets port runs): the C chains' port runs made SIDE BY SIDE,
    # one host thread and one OpenMP team of min(16, host threads / C) threads per chain, each from its chain's state at the start of the
    # timed window, over the window's iterations; the figure is the sum of their rates (as bench.py --gpus N measures it across ranks)
    cb = None
    if cpu_states is not None:
        thr = max(1, min(16, (os.cpu_count() or 1) // C))
        legs = [None] * C

        def leg(c):
            legs[c] = cpu_baseline(shard(c), params, args.cpu_seconds, cpu_states[c], burn + W, K, threads_only=thr)
        th = [threading.Thread(target=leg, args=(c,)) for c in range(C)]
        t_cpu = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        rows = [l["by_threads"][0] for l in legs]
        covered = min(r["iterations"] for r in rows) / float(K)
        total = sum(r["value"] for r in rows)
        cb = {"value": total if covered > 0 else None, "unit": "proposals/s", "cores": thr * C, "kind": "port", "host_cpus": os.cpu_count(),
              "sample": "SUM over the %d chains' port runs made side by side on this host (%d OpenMP threads each, every chain from its own state at the start of the timed window, "
                        "started together, %.0f s): schedule steps %d-%d" % (C, thr, time.time() - t_cpu, burn + W + 1, burn + W + K),
              "threads_per_chain": thr, "per_chain": [r["value"] for r in rows], "iterations_covered_by_the_slowest_chain": min(r["iterations"] for r in rows),
              "window_covered": covered}
        if cb["value"]:
            cb["gpu_over_cpu_same_window" if covered >= 1.0 else "gpu_over_cpu_partial_window"] = (sum(upd) / dt) / cb["value"]
    print(json.dumps({"metric": METRIC + " [informational: %d chains on one GPU, %s]" % (C, "batched multi-chain launches" if batched else "one thread and stream per chain"),
                      "value": sum(upd) / dt, "unit": "proposals/s",
                      "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "%d independent synthetic %s %dx%d shards on one GPU, nPatterns=%d" % (C, "sparse (95 %% zeros)" if args.sparse else "dense", args.genes, args.samples, args.patterns),
                                 "chains_mode": args.chains_mode, "chain_groups": G, "per_chain": [u / dt for u in upd],
                                 "algorithmic_GBps_over_wall": (tot_bytes / 1e9) / dt if batched else None},
                      "roofline": roof, "cpu_baseline": cb}))
    for b_ in BB:
        b_.close()
    for s in S:
        s.close()


def self_launch(n):
    """plain `python bench.py --gpus N`: start the N ranks the way the driver does and pass their exit code on"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this driver
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env, stdout=sys.stdout)      # the ranks' stdout = this process's real stdout (see __main__)
    if rc:
        raise SystemExit("bench.py --gpus %d: the %d-rank launch failed (exit code %d); no line was printed for fewer ranks" % (n, n, rc))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=190)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--patterns", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=64.0, help="time budget of the CPU baseline legs (the oracle port on this host: 8 / 16 / 24 / 32 OpenMP threads and one, from the GPU chain's state at the start of its timed window; a leg ends when it has covered the window).  N > 1 ranks, or --chains C: every shard's port run is made side by side -- one leg per shard, all at once, min(16, host threads / shards) threads each, 0.4 of the budget -- and the figure is their sum")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sparse", action="store_true",
                    help="not the headline: the SparseNormalModel on the same product with 95 %% of the entries zeroed (BASELINE configs[4] "
                         "uses --genes 50000 --samples 12500 per GPU)")
    ap.add_argument("--chains", type=int, default=1,
                    help="not the headline: that many independent chains (shards of the same shape) in flight per GPU, one host thread "
                         "and one stream each, as distributed.py runs a rank's shards; value = aggregate proposals/s")
    ap.add_argument("--chains-mode", choices=("batched", "threads"), default="batched")
    ap.add_argument("--chain-groups", type=int, default=1, help="with --chains: split the chains into this many batches, each driven by its own host thread and stream")
    args = ap.parse_args()
    # RCCL across processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails without it): set before torch / HIP initialise,
    # whoever launched the ranks (the driver's torch.distributed.run line never passes through self_launch)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.chains > 1:
        return chains_main(args)

    # `--gpus N` is a promise about the line that gets printed: N ranks, one per GPU.  Launched by torch.distributed.run (the
    # driver's form) the environment carries the world; launched as plain `python bench.py --gpus N` the script starts the N
    # ranks itself (the same launcher, 127.0.0.1 rendezvous on a free port) and hands over -- it never measures one GPU and
    # calls it N.
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or plain `python bench.py --gpus %d`, which starts its own ranks)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # test hook: COGAPS_BENCH_BACKEND=gloo lets the ranks of a multi-process run share the GPUs that exist (collectives on
    # host tensors), so the N > 1 code path can be exercised on a one-GPU box; the driver's runs use RCCL ("nccl")
    backend = os.environ.get("COGAPS_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and n_dev < world:
        raise SystemExit("--gpus %d needs %d GPUs on this node, %d visible: refusing to run (one rank per GPU over RCCL; nothing is measured on fewer)" % (args.gpus, world, n_dev))
    if backend != "nccl":
        local_rank %= n_dev
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also with one rank: the RCCL path end to end on one GPU)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, init_method="env://")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))

    from cogaps_amd import _capi

    K, W = args.steps, args.warmup
    # the chain is BASELINE's: nIterations = 100 equilibration + 100 sampling iterations; a step is one iteration, and the W + K
    # steps walk through that schedule from its start, whatever K is (the defaults cover it exactly; a longer request extends it)
    n_iter = max(100, (W + K + 1) // 2)
    # shard `rank` of the gene-wise partition: its own 20000-gene block (contiguous explicit sets)
    data = synthetic_dense(args.genes, args.samples, seed=12345 + rank)
    if args.sparse:     # the same product, 95 % of the entries zeroed i.i.d.
        data *= (np.random.Generator(np.random.MT19937(777 + rank)).random(data.shape) >= 0.95)
    params = dict(nPatterns=args.patterns, nIterations=n_iter, seed=42, outputFrequency=max(1, n_iter // 10), sparseOptimization=args.sparse)
    S = _capi.Session(data, device=local_rank, **params)
    shared_factor = "A" if args.sparse else "P"

    def run_steps(first, n):
        done, upd = 0, 0
        while done < n:
            it = first + done
            if it < n_iter:
                m = min(n - done, n_iter - it)
                upd += S.run_iterations(1, it, m)
            else:
                m = n - done
                upd += S.run_iterations(2, it - n_iter, m)
            done += m
        return upd

    # The timed steps are the LAST K iterations of the schedule, the W before them the warm-up: with the defaults that is the
    # whole BASELINE run but its first ten iterations.  A shorter request (--steps 5) first walks, untimed, through the part of
    # the schedule that precedes them, so that it measures the named configuration's populated chain and not the first
    # iterations of an empty one (a few dozen atoms, batches of three proposals).
    burn = max(0, 2 * n_iter - (W + K))
    run_steps(0, burn)
    run_steps(burn, W)
    # the chain state at the start of the timed window, for the CPU baseline's like-for-like sample (copied out before the timed region)
    cpu_state = None
    if (rank == 0 or world > 1) and not args.no_cpu:      # (N > 1: every rank times the port on ITS shard, all at once: below)
        cpu_state = {"atomsA": S.atoms("A"), "A": S.rows("A"), "atomsP": S.atoms("P"), "P": S.rows("P")}
    perf0 = {w: S.perf(w) for w in "AP"}
    S.set_timing(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    updates = run_steps(burn + W, K)
    if dist is not None:
        # the one exchange of the path: all-gather of the factor the subsets SHARE before findConsensusMatrix -- GWCoGAPS
        # (gene-wise subsets, the dense line) shares the sample factor P, scCoGAPS (cell-wise subsets, --sparse) the gene
        # factor A (reference R/DistributedCogaps.R:71-78)
        fac = torch.from_numpy(S.matrix(shared_factor)).to(comm_dev)
        gathered = [torch.empty_like(fac) for _ in range(world)]
        dist.all_gather(gathered, fac)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    S.set_timing(False)
    perf1 = {w: S.perf(w) for w in "AP"}

    tot_updates, max_dt, rank_seconds = float(updates), dt, [dt]
    if dist is not None:
        t = torch.tensor([float(updates), dt], dtype=torch.float64, device=comm_dev)
        u = t.clone()
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        m = t.clone()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        tot_updates, max_dt = float(u[0].item()), float(m[1].item())
        per_rank = [torch.zeros(1, dtype=torch.float64, device=comm_dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([dt], dtype=torch.float64, device=comm_dev))
        rank_seconds = sorted(float(x.item()) for x in per_rank)

    # N > 1: the CPU comparator of a GWCoGAPS / scCoGAPS job is nSets port runs, one per subset.  They are MEASURED side by
    # side, not extrapolated from one shard: every rank runs the port on its own shard's timed window at the same time (one leg,
    # min(16, host threads / ranks) OpenMP threads each, started behind a barrier), and the job's figure is the sum of the ranks' rates
    side_by_side = None
    if world > 1 and not args.no_cpu:
        thr = max(1, min(16, (os.cpu_count() or 1) // world))
        dist.barrier()
        mine = cpu_baseline(data, params, args.cpu_seconds, cpu_state, burn + W, K, threads_only=thr)
        leg = mine["by_threads"][0] if mine.get("by_threads") else {"value": 0.0, "iterations": 0}
        t = torch.tensor([float(leg["value"] or 0.0), float(leg["iterations"])], dtype=torch.float64, device=comm_dev)
        ssum = t.clone(); dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        smin = t.clone(); dist.all_reduce(smin, op=dist.ReduceOp.MIN)
        side_by_side = {"value": float(ssum[0].item()), "threads_per_shard": thr, "iterations_covered_by_the_slowest_rank": int(smin[1].item()),
                        "slowest_shard_value": float(smin[0].item()), "rank0": mine}
    if rank == 0:
        # HIP start/stop events ride on the dispatch packets of a sample of the launches (hipExtLaunchKernelGGL on the kernels'
        # own stream): begin-to-end time of the dispatch, the quantity rocprofv3 --kernel-trace reports.  The library scales the
        # sampled time to the batches processed since set_timing(1) (cogaps_perf.timedBatches); launches enqueued past the end
        # of an update (empty queue) are kept out of the averages; every sync (AP transpose) launch is timed.
        def window(w):
            d = {k: perf1[w][k] - perf0[w][k] for k in ("evalBytes", "batches", "evalLaunches", "genLaunches")}
            for k in ("evalMs", "genMs", "timedBatches", "evalTimed", "genTimed", "evalNoopMs", "evalNoopTimed", "genNoopMs", "genNoopTimed"):
                d[k] = perf1[w][k]           # counted from set_timing(1)
            return d
        kt = {w: window(w) for w in "AP"}
        tot = S.perf()
        batches = kt["A"]["batches"] + kt["P"]["batches"]
        fusedA = args.sparse or S.dims("A")[1] <= 4096
        fusedP = args.sparse or S.dims("P")[1] <= 4096
        chained = {w: S.chained(w) for w in "AP"}      # one launch per batch (csrc/chain_kernel.h): its time is the "evaluation" time, there is no generator launch
        CHAIN_NAME = "chain_kernel (one launch: workgroups 0..n-2 evaluate batch n -- fused, or in slices with a deciding workgroup per proposal --, the last workgroup generates batch n+1)"
        ev_name = lambda fused: "eval_sparse_kernel" if args.sparse else ("eval_kernel<EVAL_FUSED>" if fused else "eval_kernel<EVAL_DECIDE> (split evaluation, one launch; its A*P updates run beside the next generator launch)")

        def kernel_line(name, sampler, ms, launches, nbytes, sampled):
            us = 1e3 * ms / launches if launches else 0.0
            ach = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            return {"kernel": name, "sampler": sampler, "launches": int(launches), "sampled_launches": int(sampled), "avg_launch_us": us, "total_ms": ms,
                    "bytes_per_launch": nbytes / launches if launches else 0.0, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
        # chained launches: the launch clock (cogaps_session_launch_clock: the chip-wide clock read inside EVERY launch of the window, graph
        # replays included) replaces the HIP-event sample, which rides on plain launches only and overstated the replayed population by
        # ~10 % in round 4; the event figure stays in the line beside it
        clock = {w: (S.launch_clock(w) if chained[w] else None) for w in "AP"}
        period = {w: (S.launch_period(w) if chained[w] else None) for w in "AP"}
        events_us = {}
        for w in "AP":
            if clock[w] and clock[w]["launches"] and kt[w]["timedBatches"]:
                events_us[w] = 1e3 * kt[w]["evalMs"] / kt[w]["timedBatches"]
                # the kernel's time in the line: the launch-to-launch period where it was measured (the launch with the dispatcher's start-up and
                # the end-of-kernel write-back: comparable with rocprofv3's dispatch duration, and summing to no more than the wall time),
                # else the clock inside the launch
                use = period[w]["mean_us"] if (period[w] and period[w]["launches"] > clock[w]["launches"] // 2) else clock[w]["mean_us"]
                kt[w]["evalMs"] = use * kt[w]["timedBatches"] / 1e3
        kernels = [
            kernel_line(CHAIN_NAME if chained["A"] else ev_name(fusedA), "A", kt["A"]["evalMs"], kt["A"]["timedBatches"], kt["A"]["evalBytes"], kt["A"]["evalTimed"]),
            kernel_line(CHAIN_NAME if chained["P"] else ev_name(fusedP), "P", kt["P"]["evalMs"], kt["P"]["timedBatches"], kt["P"]["evalBytes"], kt["P"]["evalTimed"]),
            kernel_line("gen_kernel" if fusedA else "gen_apply_kernel (workgroup 0: generator; the others: the previous batch's A*P updates)", "A", kt["A"]["genMs"], 0 if chained["A"] else kt["A"]["timedBatches"], 0, kt["A"]["genTimed"]),
            kernel_line("gen_kernel" if fusedP else "gen_apply_kernel (workgroup 0: generator; the others: the previous batch's A*P updates)", "P", kt["P"]["genMs"], 0 if chained["P"] else kt["P"]["timedBatches"], 0, kt["P"]["genTimed"]),
            kernel_line("sparse_tables_kernel (not timed)" if args.sparse else "transpose_kernel (sync)", "A+P", tot["syncMs"], tot["syncTimed"], tot["syncBytes"], tot["syncTimed"]),
        ]
        for i_, w_ in ((2, "A"), (3, "P")):
            if chained[w_]: kernels[i_]["kernel"] = "(none: the generator is the last workgroup of chain_kernel)"
        for i_, w_ in ((0, "A"), (1, "P")):
            if w_ in events_us:
                kernels[i_]["timing"] = ("avg_launch_us: launch-to-launch period from the device clock of every launch of the timed window (entry of a launch's first workgroup to "
                                         "the next launch's: dispatcher start-up and end-of-kernel write-back included -- rocprofv3's dispatch duration plus the idle gap); "
                                         "inside_launch_us / launch_us_percentiles: entry of the first workgroup to the end of the generator workgroup")
                kernels[i_]["inside_launch_us"] = clock[w_]["mean_us"]
                kernels[i_]["launch_period_us"] = dict(period[w_]) if period[w_] else None
                kernels[i_]["launch_us_percentiles"] = {k: clock[w_][k] for k in ("p10_us", "p50_us", "p75_us", "p90_us", "p99_us")}
                kernels[i_]["launches_measured_by_device_clock"] = clock[w_]["launches"]
                kernels[i_]["avg_launch_us_hip_event_sample"] = events_us[w_]
        for i_, fused_ in ((0, fusedA), (1, fusedP)):
            if not fused_ and not args.sparse:
                # one-launch split evaluation: the A*P updates this kernel's proposals owe (12N bytes each, counted in bytes_per_launch by
                # SURVEY 8d's formula) are carried out by the update workgroups of the NEXT generator launch, whose time is in that line:
                # this line's achieved / frac are an upper bound for the evaluation launch alone; the path figure is unaffected
                kernels[i_]["note"] = "bytes_per_launch includes the A*P updates carried out inside the next gen_apply_kernel launch: achieved / frac overstate this launch alone"
        gen_ms = kt["A"]["genMs"] + kt["P"]["genMs"]
        ev_ms = kt["A"]["evalMs"] + kt["P"]["evalMs"]
        kernel_ms = gen_ms + ev_ms + tot["syncMs"]
        b_alg = kt["A"]["evalBytes"] + kt["P"]["evalBytes"] + tot["syncBytes"]
        # the kernels run back to back on one stream: their summed durations cannot exceed the wall time of the timed region
        consistent = kernel_ms <= 1.05 * (1e3 * dt) and all(k["sampled_launches"] > 0 for k in kernels[:4] if k["launches"])
        achieved = (b_alg / 1e9) / (kernel_ms / 1e3) if (consistent and kernel_ms > 0) else None
        dominant = max(kernels, key=lambda k: k["total_ms"])
        # HBM bytes from the PMC counters: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_pass.sh) over this
        # same workload, committed under profiles/ -- a bench run cannot collect them itself.  Quoted whenever the workload shape
        # is the one they were measured on (any --steps / --warmup: per-launch means of the populated chain).
        traffic, traffic_src, traffic_kernels, traffic_stale, traffic_measured_on = None, None, None, None, None
        lib_hash = _capi.load().cogaps_source_hash().decode()
        import glob
        files = []
        if not args.sparse and (args.genes, args.samples, args.patterns) == (20000, 2000, 50):
            files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) if "chains" not in os.path.basename(f) and "sparse" not in os.path.basename(f))
        elif args.sparse and (args.genes, args.samples, args.patterns) == (50000, 12500, 50):      # BASELINE configs[4]'s per-GPU shard
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sparse_c4shape_pmc_traffic.json")))
        if files:
            if True:
                rec_ = json.load(open(files[-1]))
                pk = rec_["kernels"]
                # the counters were collected on ONE build of the library: quoted only while the running library is that build
                traffic_measured_on = rec_.get("lib_source_hash")
                traffic_stale = traffic_measured_on != lib_hash
                traffic_kernels = {k: v["hbm_bytes_per_launch"] for k, v in pk.items()}
                tb = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in pk.values())
                nb = sum(v["launches"] for k, v in pk.items() if k.startswith("gen_kernel") or k.startswith("chain_") or k.startswith("void chain_"))      # one generator (or chained) launch per batch
                if nb and not traffic_stale:
                    traffic = tb / nb
                if nb:
                    traffic_src = ("%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + --pmc WRITE_SIZE, separate passes over this workload; sum over "
                                   "the generator and evaluation kernels of (bytes per launch x launches) / batches" % os.path.relpath(files[-1], ROOT))
        out = {
            "metric": METRIC, "value": tot_updates / max_dt, "unit": "proposals/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * max_dt / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("synthetic sparse %dx%d fp32 per GPU (95 %% zeros), sparseOptimization, nPatterns=%d, asynchronous sampler, seed 42 (%s%s)"
                                    if args.sparse else "synthetic dense %dx%d fp32 per GPU, nPatterns=%d, asynchronous sampler, seed 42 (%s%s)")
                                   % (args.genes, args.samples, args.patterns,
                                      (("one cell-wise shard of BASELINE configs[4]: 50000 genes x 12500 of the 100000 cells" if (args.genes, args.samples) == (50000, 12500)
                                        else "the SparseNormalModel on a configs[2]-sized product; BASELINE configs[4]'s shard is --genes 50000 --samples 12500") if args.sparse
                                       else ("BASELINE configs[2]" if (args.genes, args.samples, args.patterns) == (20000, 2000, 50) else "not a BASELINE shape")),
                                      ("; scCoGAPS nSets=%d cell-wise shards, one per GPU" % world if args.sparse else "; GWCoGAPS nSets=%d gene-wise shards, configs[3]" % world) if world > 1 else ""),
                       "rank_seconds": {"min": rank_seconds[0], "median": rank_seconds[len(rank_seconds) // 2], "max": rank_seconds[-1]},
                       "ranks": world, "collective_backend": (backend if dist is not None else None), "shared_factor_gathered": (shared_factor if dist is not None else None),
                       "nIterations": n_iter, "untimed_schedule_steps_before_warmup": burn, "proposals_timed": int(tot_updates), "batches_rank0": int(batches),
                       "avg_queue_A": S.avg_queue("A"), "avg_queue_P": S.avg_queue("P"),
                       "atoms_A": S.natoms("A"), "atoms_P": S.natoms("P"),
                       "timed_region_ms_rank0": 1e3 * dt, "gen_kernel_ms_rank0": gen_ms, "eval_kernel_ms_rank0": ev_ms, "sync_kernel_ms_rank0": tot["syncMs"],
                       "launches_per_batch": sum(kt[w]["evalLaunches"] + kt[w]["genLaunches"] for w in "AP") / max(1, batches),
                       # launches enqueued past the end of an update find nothing to do: what a launch costs before it does any work
                       "empty_launch_us": {w: {"gen": (1e3 * kt[w]["genNoopMs"] / kt[w]["genNoopTimed"]) if kt[w]["genNoopTimed"] else None,
                                               "eval": (1e3 * kt[w]["evalNoopMs"] / kt[w]["evalNoopTimed"]) if kt[w]["evalNoopTimed"] else None} for w in "AP"}},
            # roofline.achieved = B_alg / (sum of kernel time) over the whole path -- generator, evaluation and sync
            # kernels of the timed region -- with each kernel's own figure alongside.  One "launch" of the path = one batch.
            "roofline": {"bound": "hbm", "kernel": "path: generator + evaluation launches (one chained launch per batch where the fused evaluation serves) + sync, per batch (dominant by time: %s, sampler %s)" % (dominant["kernel"], dominant["sampler"]),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_bytes_per_launch_by_kernel": (traffic_kernels if not traffic_stale else None),
                         # the committed counter pass names the library build it measured; another build -> no figure, no ratio, re-run tools/pmc_pass.sh
                         "traffic_stale": traffic_stale, "traffic_measured_on_lib": traffic_measured_on, "lib_source_hash": lib_hash,
                         "bytes_per_launch": b_alg / max(1, batches), "avg_launch_us": 1e3 * kernel_ms / max(1, batches), "launches": int(batches),
                         "kernel_time_over_wall": kernel_ms / (1e3 * dt), "timing_consistent": bool(consistent),
                         # the same algorithmic bytes over the WALL time of the timed region (kernel boundaries and host gaps included): a lower bound of frac
                         "achieved_over_wall": (b_alg / 1e9) / dt, "frac_over_wall": (b_alg / 1e9) / dt / HBM_PEAK_GBS,
                         "kernels": kernels},
        }
        if not args.no_cpu:
            if world > 1:
                # measured above, all shards at once (round 4 multiplied rank 0's figure by nSets)
                cb = dict(side_by_side["rank0"])
                cb["value_rank0_shard"] = cb.get("value")
                cb["value"] = side_by_side["value"] if side_by_side["iterations_covered_by_the_slowest_rank"] > 0 else None
                cb["cores"] = side_by_side["threads_per_shard"] * world
                cb["side_by_side"] = {k: v for k, v in side_by_side.items() if k != "rank0"}
                cb["sample"] = ("SUM over the %d shards' port runs made side by side on this host (%d OpenMP threads each, every rank its own shard's timed window, "
                                "started together): rank 0's: " % (world, side_by_side["threads_per_shard"])) + str(cb.get("sample"))
            else:
                cb = cpu_baseline(data, params, args.cpu_seconds, cpu_state, burn + W, K)
            if cb["value"] is not None:
                # like for like only when the port ran the WHOLE timed window (the driver's --steps 20 does; the 190-step default lets
                # the port cover the window's first part, where the chain is still growing and the port is slower per proposal)
                covered = (side_by_side["iterations_covered_by_the_slowest_rank"] if world > 1 else max(b["iterations"] for b in cb["by_threads"] if b["threads"] == cb["cores"])) / float(K)
                cb["window_covered"] = covered
                cb["gpu_over_cpu_same_window" if covered >= 1.0 else "gpu_over_cpu_partial_window"] = out["value"] / cb["value"]
                # north_star's target is 10x the CPU path: where that line lies on this host and how far the GPU figure is from it
                cb["ten_x_line"] = 10.0 * cb["value"]
                cb["gpu_over_ten_x_line"] = out["value"] / (10.0 * cb["value"])
            # the port timed here against the REFERENCE BUILD where both can run (build container, committed record)
            cb["port_over_reference_build"] = reference_translation(args.sparse)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    S.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries the JSON line and nothing else: libraries that write to file descriptor 1 (RCCL prints its version block there
    # when the communicator is created) are sent to stderr, Python's own sys.stdout keeps the real descriptor
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(_real_stdout, "w", buffering=1)
    main()
    sys.stdout.flush()
