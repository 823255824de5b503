"""Host logic + kernel logic on the TEST-ONLY workgroup emulator (tests/emul), bit-compared with the
oracle batch by batch.  No GPU: the same kernel sources are compiled with g++ against a fiber-per-lane
emulator; the product library is not involved.  The generator's speculative window is exercised at three
widths (64 forces many multi-round batches and window cuts; 1024 is wider than any batch)."""
import numpy as np
import pytest

import parity_util as pu


@pytest.mark.parametrize("win", [64, 256])
def test_modsim_stepwise(emul_lib, modsim, win):
    a, p, props = pu.run_stepwise(emul_lib(win), modsim, 120 if win == 64 else 60, nPatterns=3, seed=42, total_iter=120)
    assert props > 2000


@pytest.mark.parametrize("win,n", [(64, 14), (256, 10), (1024, 6)])
def test_gist_stepwise(emul_lib, gist, win, n):
    pu.run_stepwise(emul_lib(win), gist, n, nPatterns=7, seed=42, total_iter=40)


@pytest.mark.parametrize("lds_rounds", [3, 2, 1])
def test_batches_of_several_rounds_use_the_lds_table_then_the_stamp_tables(emul_lib, lds_rounds):
    """a batch that outlives its window goes on in further rounds: the first GEN_LDS_ROUNDS keep the conflict sets in the LDS table (carried
    over; the marks of the attempts behind a cut and all gap / same-bin marks removed, committed births' atoms added), later ones use the stamp
    tables in HBM.  The product build admits three LDS rounds, after which further rounds are rare; the variants with two and one keep the
    hand-over to the stamp tables and the stamp-table rounds themselves under test (round counts from the test-only build's counters).
    2000 rows at a 64-attempt window (batches of ~56 attempts with a long tail) and a 5 x 2 domain full of hazards, against the oracle."""
    from cogaps_amd import _capi
    lib = emul_lib(64) if lds_rounds == 3 else emul_lib(64, extra="-DGEN_LDS_ROUNDS_MAX=%d" % lds_rounds, tag="_lds%d" % lds_rounds)
    data = pu.synthetic(2000, 10, seed=7)
    tiny = pu.synthetic(5, 6, rank=2, seed=3)
    pu.run_stepwise(lib, data, 24, nPatterns=3, seed=123, total_iter=40, check_every=4)
    pu.run_stepwise(lib, tiny, 100, nPatterns=2, seed=9, total_iter=100)
    n_lds = n_hbm = 0
    for d, kw, n in ((data, dict(nPatterns=3, seed=123, nIterations=40), 24), (tiny, dict(nPatterns=2, seed=9, nIterations=100), 100)):
        S = _capi.Session(d, lib=lib, **kw)
        S.run_iterations(1, 0, n)
        for w in "AP":
            prof = S.debug_prof(w)
            n_lds += prof[14]; n_hbm += prof[15]
        S.close()
    if lds_rounds == 1:
        assert n_lds == 0 and n_hbm > 100, (n_lds, n_hbm)
    elif lds_rounds == 2:
        assert n_lds > 100 and n_hbm > 10, (n_lds, n_hbm)
    else:
        assert n_lds > 100, (n_lds, n_hbm)


@pytest.mark.parametrize("variant", ["chained", "chained_fallback", "chained_redraw", "chained_few_appliers", "two_launches", "few_compute_units"])
def test_chained_launch_against_the_oracle(emul_lib, monkeypatch, variant):
    """csrc/chain_kernel.h: one launch evaluates batch n and generates batch n + 1 -- the generator workgroup's applier waves carry out the
    decisions they receive as tagged granules while its attempt waves classify AND DRAW the next window ahead of them; behind the join
    every lane whose reads the decisions or the flush touched draws again (gen_draw_valid, gen_round<.., AHEAD>).  1200 x 300: both
    samplers take the chained form at the 64-attempt window (workgroups of 128 / 512 threads: one to seven applier waves), the domains
    are small enough for a good share of the lanes to be invalidated.  Stepwise against the oracle (every proposal of every batch, the
    state after every update), then the test-only build's counters: every path was taken.
    Workgroups with several proposals evaluate them in pairs, one per half (eval_chain_pair).
    `chained_few_appliers`: a variant whose generator workgroup has five applier lanes -- the queue takes several passes over them, each
    pass fetching its records behind the previous one's wait (what a queue longer than the applier lanes takes on the hardware: the
    batch behind a generator launch of two rounds);
    `chained_fallback`: a build variant that declares every third window's classification unusable (the path a window takes when an
    attempt falls between the two birth / death thresholds -- too rare to meet otherwise): such a window is classified and drawn behind
    the decisions; `chained_redraw`: a variant that declares every fifth lane's draw invalid, whatever it read; `two_launches`:
    COGAPS_NO_CHAIN=1; `few_compute_units`: a device with fewer compute units than the chained launch has workgroups keeps two launches per batch."""
    from cogaps_amd import _capi
    if variant == "two_launches": monkeypatch.setenv("COGAPS_NO_CHAIN", "1")
    if variant == "few_compute_units": monkeypatch.setenv("COGAPS_TEST_COMPUTE_UNITS", "4")      # (a partitioned GPU: fewer compute units than the chained launch has workgroups)
    lib = (emul_lib(64, extra="-DGEN_SPEC_BAD_EVERY=3", tag="_specbad") if variant == "chained_fallback" else
           emul_lib(64, extra="-DGEN_AHEAD_BAD_EVERY=5", tag="_aheadbad") if variant == "chained_redraw" else
           emul_lib(64, extra="-DGEN_TEST_APPLIER_LANES=5", tag="_appl5") if variant == "chained_few_appliers" else emul_lib(64))
    data = pu.synthetic(1200, 300, seed=7)
    pu.run_stepwise(lib, data, 24, nPatterns=3, seed=123, total_iter=40, check_every=4)
    S = _capi.Session(data, lib=lib, nPatterns=3, seed=123, nIterations=40)
    S.run_iterations(1, 0, 24)
    tot = np.zeros(16, dtype=np.int64)
    for w in "AP":
        assert S.chained(w) == (variant in ("chained", "chained_fallback", "chained_redraw", "chained_few_appliers"))
        tot += np.array(S.debug_prof(w), dtype=np.int64)
    S.close()
    held, redrawn, usual, spec, chain_batches, pairs, kept_pick, seconds = (int(tot[i]) for i in (8, 9, 11, 12, 13, 7, 10, 6))
    redrawn += kept_pick      # (lanes that drew again: behind the flush, or -- a pick whose record or cells the decisions alone rewrote -- keeping their pick)
    if variant in ("two_launches", "few_compute_units"):
        assert chain_batches == 0 and spec == 0 and held == 0 and redrawn == 0 and pairs == 0 and seconds == 0
    else:
        assert chain_batches > 300 and spec > 200 and usual > 20 and held > 3000 and redrawn > 30, (chain_batches, spec, usual, held, redrawn)
        if variant == "chained_fallback": assert usual > spec // 3
        if variant == "chained_redraw": assert redrawn > held // 5 and kept_pick > 300
        assert kept_pick > 5, kept_pick
        assert pairs > 300, pairs        # (seven evaluation workgroups per launch in this build: most proposals are evaluated two at a time, eval_chain_pair)
        if variant == "chained_few_appliers": assert seconds > 500, seconds      # (decisions carried out in a pass beyond the first)
    print(variant, dict(chain_batches=chain_batches, ahead=spec, usual=usual, lanes_held=held, lanes_redrawn=redrawn, of_which_kept_their_pick=kept_pick, pairs=pairs, in_later_passes=seconds))


def test_a_lost_hand_over_is_completed_and_the_update_goes_on(emul_lib):
    """A generator lane that gives up waiting for a decision inside a chained launch (GAPS_ERR_SPIN: the bounded poll) drops the proposal,
    marks it, and the workgroup leaves without generating; the launches enqueued behind it are no-ops.  The host completes the batch from
    the decisions the evaluation workgroups have left (chain_recover_kernel: the marked proposals carried out, the erase cache and the
    scalars put back) and goes on with two launches per batch -- the same chain, bit for bit.  The emulator never waits (its workgroups run
    one after the other), so a build variant makes every third lane give up at one batch of each sampler; the run is compared with the
    oracle proposal by proposal, state by state, across the event."""
    from cogaps_amd import _capi
    lib = emul_lib(64, extra="-DGEN_TEST_SPIN_FAIL_EPOCH=57", tag="_spinfail")
    data = pu.synthetic(1200, 300, seed=7)
    pu.run_stepwise(lib, data, 24, nPatterns=3, seed=123, total_iter=40, check_every=4)
    S = _capi.Session(data, lib=lib, nPatterns=3, seed=123, nIterations=40)
    S.run_iterations(1, 0, 24)
    carried = 0
    for w in "AP":
        assert S.chain_recoveries(w) == 1, (w, S.chain_recoveries(w))
        assert not S.chained(w)            # two launches per batch from the event on
        carried += S.debug_prof(w)[5]
    S.close()
    assert carried >= 2, carried           # decisions the recovery carried out
    # ... and the sparse model's chained launch (chain_sparse_kernel: the decisions go to the HybridMatrix's two copies and its flags)
    sp = pu.synthetic_counts(600, 200, zeros=0.9, seed=5)
    pu.run_stepwise(lib, sp, 30, nPatterns=4, seed=11, total_iter=40, check_every=5, sparseOptimization=True)
    S = _capi.Session(sp, lib=lib, nPatterns=4, seed=11, nIterations=40, sparseOptimization=True)
    S.run_iterations(1, 0, 30)
    assert sum(S.chain_recoveries(w) for w in "AP") >= 1 and sum(S.debug_prof(w)[5] for w in "AP") >= 1
    S.close()


def test_sparse_chained_launch_two_proposals_per_evaluation_workgroup(emul_lib, monkeypatch):
    """chain_sparse_kernel (sparse_kernels.h): the launch's workgroups have 512 threads, the sparse evaluation the model's width (256 at
    most) -- lanes 256.. of an evaluation workgroup are a second group that takes the proposal one grid further on, with an LDS block of
    its own; the two groups pass through the workgroup's barriers side by side, each in its own order of phases (proposals of different
    types, one with a Gibbs mass and one without, vectors of one and of two words).  Stepwise against the oracle at model widths 64 and
    256, then the test-only build's counter of proposals the second groups evaluated; with COGAPS_NO_CHAIN the counter stays at zero."""
    from cogaps_amd import _capi
    lib = emul_lib(64)
    for data, kw, n in ((pu.synthetic_counts(600, 200, zeros=0.9, seed=5), dict(nPatterns=4, seed=11), 30),
                        (pu.synthetic_counts(1500, 9000, zeros=0.95, seed=6), dict(nPatterns=3, seed=12), 12)):
        pu.run_stepwise(lib, data, n, total_iter=40, check_every=5, sparseOptimization=True, **kw)
        S = _capi.Session(data, lib=lib, nIterations=40, sparseOptimization=True, **kw)
        S.run_iterations(1, 0, n)
        second = {w: S.debug_prof(w)[4] for w in "AP"}
        chained = {w: bool(S.chained(w)) for w in "AP"}
        S.close()
        assert all(chained.values()) and all(v > 20 for v in second.values()), (chained, second)
    # the hand-over in the sparse model: the attempt lanes, done with the window drawn ahead, take the queue slots behind the applier lanes' (the
    # variant with five applier lanes: slots 5 .. 68 go to the 64 attempt lanes, longer queues take further passes of 69)
    lib5 = emul_lib(64, extra="-DGEN_TEST_APPLIER_LANES=5", tag="_appl5")
    data = pu.synthetic_counts(600, 200, zeros=0.9, seed=5)
    pu.run_stepwise(lib5, data, 30, total_iter=40, check_every=5, sparseOptimization=True, nPatterns=4, seed=11)
    S = _capi.Session(data, lib=lib5, nIterations=40, sparseOptimization=True, nPatterns=4, seed=11)
    S.run_iterations(1, 0, 30)
    by_attempt_lanes = sum(S.debug_prof(w)[2] for w in "AP"); later = sum(S.debug_prof(w)[6] for w in "AP")
    S.close()
    assert by_attempt_lanes > 500, (by_attempt_lanes, later)
    # the third, wider window of the sparse model's chained launch (448 attempts in the build with GEN_WIN = 256: seven attempt waves and the
    # helper wave), taken from the first update on
    monkeypatch.setenv("COGAPS_TEST_WIDE_WINDOW", "1")
    lib256 = emul_lib(256)
    data = pu.synthetic_counts(900, 260, zeros=0.9, seed=15)
    pu.run_stepwise(lib256, data, 14, total_iter=30, check_every=2, sparseOptimization=True, nPatterns=4, seed=21)
    S = _capi.Session(data, lib=lib256, nIterations=30, sparseOptimization=True, nPatterns=4, seed=21)
    S.run_iterations(1, 0, 14)
    assert all(S.chained(w) for w in "AP") and [S.generator_window(w) for w in "AP"] == [448, 448], [S.generator_window(w) for w in "AP"]
    S.close()
    monkeypatch.delenv("COGAPS_TEST_WIDE_WINDOW")
    monkeypatch.setenv("COGAPS_NO_CHAIN", "1")
    data = pu.synthetic_counts(600, 200, zeros=0.9, seed=5)
    S = _capi.Session(data, lib=lib, nPatterns=4, seed=11, nIterations=40, sparseOptimization=True)
    S.run_iterations(1, 0, 10)
    assert sum(S.debug_prof(w)[4] for w in "AP") == 0
    S.close()


def test_a_batch_of_chains_in_one_chained_launch(emul_lib, monkeypatch):
    """cogaps_batch_* with the chained launch for ALL chains of the batch (chain_kernel.h, chain_kernel_multi; round 6): workgroups
    [c * wgPerChain, (c + 1) * wgPerChain) of one launch evaluate chain c's queue, the last of them generates its next batch.  Three
    chains of 1200 x 300 (both samplers' evaluations are the fused one and as large as the generator's workgroup: both sides take the
    form) stepped as one batch give, state for state, the three chains stepped one at a time; the counters say the chained form ran; with
    COGAPS_NO_CHAIN the batch keeps a generator launch and an evaluation launch per step and gives the same states."""
    from cogaps_amd import _capi
    lib = emul_lib(64)
    datas = [pu.synthetic(1200, 300, seed=7 + c) for c in range(3)]
    kw = dict(nPatterns=3, nIterations=40)

    def state(S):
        return [(S.atoms(w)["pos"].copy(), S.atoms(w)["mass"].copy(), S.matrix(w).copy(), S.ap(w).copy()) for w in "AP"]
    alone = []
    for c, d in enumerate(datas):
        S = _capi.Session(d, lib=lib, seed=100 + c, **kw); S.run_iterations(1, 0, 20); alone.append(state(S)); S.close()
    for no_chain in (False, True):
        if no_chain: monkeypatch.setenv("COGAPS_NO_CHAIN", "1")
        ss = [_capi.Session(d, lib=lib, seed=100 + c, **kw) for c, d in enumerate(datas)]
        B = _capi.Batch(ss)
        B.run_iterations(1, 0, 20)
        for c, S in enumerate(ss):
            for w in "AP": assert S.chained(w) == (not no_chain), (c, w, no_chain)
            for x, y in zip(state(S), alone[c]):
                for u, v in zip(x, y): assert np.array_equal(u, v), (c, no_chain)
            assert (sum(S.debug_prof(w)[13] for w in "AP") > 200) == (not no_chain)      # batches whose decisions arrived inside a chained launch
        B.close()
        for S in ss: S.close()


def test_tiny_domain_hazards(emul_lib):
    """5 rows x 2 patterns: every window is full of row conflicts, same-bin moves and neighbour hazards"""
    data = pu.synthetic(5, 6, rank=2, seed=3)
    pu.run_stepwise(emul_lib(64), data, 150, nPatterns=2, seed=9, total_iter=100)
    pu.run_stepwise(emul_lib(256), data, 60, nPatterns=2, seed=10, total_iter=100)


def test_multiwave_evaluation(emul_lib):
    """P-side data vectors of 2600 elements -> 1024 virtual lanes, one 1024-thread workgroup per proposal
    (cross-wave butterfly in LDS)"""
    data = pu.synthetic(2600, 12)
    pu.run_stepwise(emul_lib(256), data, 6, trace=False, nPatterns=3, seed=123, total_iter=10)


@pytest.mark.parametrize("genes,iters", [(6000, 8), (20000, 4), (70000, 3)])
def test_split_evaluation(emul_lib, genes, iters):
    """P-side data vectors longer than 4096 elements: W = 2048 / 8192 / 16384 virtual lanes, several workgroups
    per proposal (alpha kernel + apply kernel, per-slice partials folded in butterfly order); 70000 elements
    also gives every virtual lane more than one chunk"""
    data = pu.synthetic(genes, 8, seed=genes)
    pu.run_stepwise(emul_lib(256), data, iters, trace=False, nPatterns=3, seed=7, total_iter=10)


@pytest.mark.parametrize("genes,iters", [(6000, 6), (20000, 4)])
def test_split_evaluation_inside_the_chained_launch(emul_lib, monkeypatch, genes, iters):
    """COGAPS_CHAIN_SPLIT=1: the split evaluation as part of the chained launch (EVAL_CHAIN_SPLIT -- slices, a deciding workgroup per proposal
    that hands the decision to the launch's generator workgroup, the A*P updates from the decisions' granules; 3 and 10 slices per
    proposal) stepwise against the oracle.  Not the product's default (measured slower on the MI355X), kept under test."""
    from cogaps_amd import _capi
    monkeypatch.setenv("COGAPS_CHAIN_SPLIT", "1")
    lib = emul_lib(256)
    data = pu.synthetic(genes, 8, seed=genes)
    pu.run_stepwise(lib, data, iters, trace=True, nPatterns=3, seed=7, total_iter=10)
    S = _capi.Session(data, lib=lib, nPatterns=3, seed=7, nIterations=10)
    S.run_iterations(1, 0, 2)
    assert S.chained("P") == 1 and S.chained("A") == 0
    S.close()


def test_two_launch_split_evaluation_still_matches(emul_lib):
    """COGAPS_SPLIT_TWO_LAUNCHES=1 (the A/B switch of round 4) brings back the alpha + apply launches for one chain: the same bits as the
    one-launch form, i.e. as the oracle (the variable is read once per process: a child process runs the comparison)"""
    import os, subprocess, sys
    emul_lib(256)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "dev_parity_synth.py"), "256", "6000", "8", "3", "6"],
                         env=dict(os.environ, COGAPS_SPLIT_TWO_LAUNCHES="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("OK 6000 8 3 6"), out.stdout[-500:] + out.stderr[-500:]


@pytest.mark.parametrize("sparse,transpose,fixed,subset,k,with_unc", pu.option_cases()[1::3])
def test_option_combinations_stepwise(emul_lib, sparse, transpose, fixed, subset, k, with_unc):
    """a third of the 36 option combinations the GPU suite runs (parity_util.option_cases), on the emulator"""
    pu.run_option_case(emul_lib(win=256), sparse, transpose, fixed, subset, k, with_unc)


def test_transposed_and_uncertainty(emul_lib, modsim):
    unc = (np.maximum(modsim * 0.2, 0.05)).astype(np.float32)
    S, O = pu.make_pair(emul_lib(256), np.ascontiguousarray(modsim.T), unc=None, nPatterns=3, seed=4, nIterations=20, transposeData=True)
    S.close(), O.close()
    from cogaps_amd import _capi
    import pyoracle as po
    lib = emul_lib(256)
    S = _capi.Session(modsim, unc=unc, lib=lib, nPatterns=3, seed=4, nIterations=20)
    O = po.Session(modsim, unc=unc, math_mode=po.MATH_PORTABLE, redW_A=64, redW_P=64, redG=4, nPatterns=3, seed=4, nIterations=20)
    for it in range(20):
        nA, nP = S.draw_steps()
        assert (nA, nP) == O.draw_steps()
        S.iterate(nA, nP), O.iterate(nA, nP)
    pu.assert_state_equal(S, O, "uncertainty")


def test_gene_subset_and_fixed_matrix(emul_lib, gist):
    idx = np.arange(1, 301, dtype=np.uint32)
    lib = emul_lib(256)
    pu.run_stepwise(lib, gist, 6, trace=False, nPatterns=4, seed=7, total_iter=12, subsetIndices=idx, subsetDim=1)
    fixedP = np.abs(np.random.default_rng(1).normal(size=(9, 4))).astype(np.float32)
    pu.run_stepwise(lib, gist, 6, trace=False, nPatterns=4, seed=7, total_iter=12, subsetIndices=idx, subsetDim=1,
                    whichMatrixFixed="P", fixedPatterns=fixedP)


def test_full_run_matches_oracle(emul_lib, modsim, oracle):
    """cogaps_run end to end: histories, statistics, meanChiSq"""
    from cogaps_amd import _capi
    lib = emul_lib(256)
    r = _capi.run(modsim, lib=lib, nPatterns=3, nIterations=40, seed=42, outputFrequency=10)
    o = oracle.run(modsim, nPatterns=3, nIterations=40, seed=42, outputFrequency=10, math_mode=oracle.MATH_PORTABLE, redW_A=64, redW_P=64, redG=4)
    for f in ("atomsA", "atomsP", "chisq", "Amean", "Pmean", "Asd", "Psd"):
        assert np.array_equal(r[f], o[f]), f
    assert r["totalUpdates"] == o["totalUpdates"] and r["meanChiSq"] == o["meanChiSq"]
    assert r["averageQueueLengthA"] == o["averageQueueLengthA"]


def test_pump_statistics_and_snapshots(emul_lib, modsim, oracle):
    """takePumpSamples (GapsStatistics.h:65-126) and nSnapshots / snapshotPhase (GapsRunner.cpp:316-322,
    Cogaps.cpp:104-123) against the oracle, dense and sparse model"""
    from cogaps_amd import _capi
    lib = emul_lib(256)
    for sparse, data in ((False, modsim), (True, pu.synthetic_counts(80, 24, zeros=0.7, seed=4))):
        kw = dict(nPatterns=3, nIterations=40, seed=42, outputFrequency=10, takePumpSamples=True, sparseOptimization=sparse)
        for phase, code in (("all", 0), ("equilibration", 1), ("sampling", 2)):
            r = _capi.run(data, lib=lib, nSnapshots=4, snapshotPhase=phase, **kw)
            w_a, w_p = lib.cogaps_reduction_width(data.shape[1]), lib.cogaps_reduction_width(data.shape[0])
            o = oracle.run(data, math_mode=oracle.MATH_PORTABLE, redW_A=w_a, redW_P=w_p, redG=4, snapshotFrequency=10, snapshotPhase=code, **kw)
            for f in ("Amean", "Pmean", "pumpMatrix", "meanPatternAssignment", "equilibrationSnapshotsA", "equilibrationSnapshotsP",
                      "samplingSnapshotsA", "samplingSnapshotsP"):
                assert np.array_equal(r[f], o[f]), (sparse, phase, f)
            assert r["equilibrationSnapshotsA"].shape[0] == (4 if code != 2 else 0) and r["samplingSnapshotsP"].shape[0] == (4 if code != 1 else 0)
            assert np.allclose(r["pumpMatrix"].sum(axis=1), 1.0) and set(np.unique(r["meanPatternAssignment"])) <= {0.0, 1.0}


def test_atom_tables_grow(emul_lib, gist, monkeypatch):
    """grow_atoms: with a 64-atom initial capacity (COGAPS_INITIAL_ATOM_CAP, a test switch) the atom tables are
    reallocated many times on the way to ~1300 atoms; the chain stays bit-identical to the oracle"""
    monkeypatch.setenv("COGAPS_INITIAL_ATOM_CAP", "64")
    a, p, props = pu.run_stepwise(emul_lib(256), gist, 40, trace=False, nPatterns=7, seed=42, total_iter=40)
    assert a > 1000


def test_serial_flush_fallback(emul_lib, gist):
    """an erase cache longer than FLUSH_MAX is flushed by one lane exactly as the reference does it
    (ConcurrentAtomicDomain.cpp:71-79).  Erases are rare (this chain: 60 batches with one, 6 with two or three in 50
    iterations), so the variant is built with FLUSH_MAX = 1 and run long enough to take the path several times"""
    pu.run_stepwise(emul_lib(256, extra="-DFLUSH_MAX=1", tag="_flush1"), gist, 50, trace=False, nPatterns=7, seed=42, total_iter=40)


SEQ = dict(reductionMode="seq", mathMode="glibc-fma")


def test_verification_mode_stepwise(emul_lib, modsim, gist):
    """reductionMode SEQ + mathMode GLIBC_FMA (cogaps_hip.h): the kernels against the oracle in the reference's own arithmetic
    (sequential sums, glibc's logf / expf) -- traces, atoms, matrices, AP, chi2 bit for bit; dense, then the P side's 1363-element
    vectors (several 1024-element fold blocks)"""
    pu.run_stepwise(emul_lib(256), modsim, 60, nPatterns=3, seed=42, total_iter=60, check_every=10, **SEQ)
    pu.run_stepwise(emul_lib(256), gist, 5, nPatterns=5, seed=123, total_iter=20, **SEQ)


def test_verification_mode_sparse_stepwise(emul_lib):
    """... with the sparse model: gaps::dot back to front up to 25 elements (K = 4; the 24-element Z2 columns) and front to
    back above (K = 30), table terms first, one term per common non-zero in index order"""
    pu.run_stepwise(emul_lib(256), pu.synthetic_counts(120, 24, zeros=0.7, seed=4), 25, nPatterns=4, seed=77, total_iter=30, check_every=5, sparseOptimization=True, **SEQ)
    pu.run_stepwise(emul_lib(256), pu.synthetic_counts(200, 30, zeros=0.8, seed=5), 8, trace=False, nPatterns=30, seed=3, total_iter=20, check_every=4, sparseOptimization=True, **SEQ)


def test_verification_mode_full_run(emul_lib, modsim, oracle):
    """cogaps_run in verification mode = the oracle's reference-arithmetic run, statistics and meanChiSq included; on this
    container's host (glibc 2.35, FMA) that is also the libm mode, i.e. the reference binary's own numbers"""
    from cogaps_amd import _capi
    kw = dict(nPatterns=3, nIterations=100, seed=42, outputFrequency=10)
    r = _capi.run(modsim, lib=emul_lib(256), **SEQ, **kw)
    o = oracle.run(modsim, math_mode=oracle.MATH_GLIBC_FMA, **kw)
    for f in ("atomsA", "atomsP", "chisq", "Amean", "Pmean", "Asd", "Psd"):
        assert np.array_equal(r[f], o[f]), f
    assert r["totalUpdates"] == o["totalUpdates"] and r["meanChiSq"] == o["meanChiSq"] and r["averageQueueLengthP"] == o["averageQueueLengthP"]


def test_math_modes_on_the_host(emul_lib, oracle):
    """cogaps_debug_math (host side of the shared source): the glibc modes equal the committed libm vectors, the portable mode the
    oracle's portable functions"""
    import os
    from conftest import GOLDEN
    from cogaps_amd import _capi
    lib = emul_lib(256)
    g = np.load(os.path.join(GOLDEN, "glibc235_logf_expf.npz"))
    assert _capi.debug_math("log", g["x_log"], "glibc-fma", lib=lib).tobytes() == g["y_log"].tobytes()
    assert _capi.debug_math("exp", g["x_exp"], "glibc-fma", lib=lib).tobytes() == g["y_exp"].tobytes()
    L = oracle.lib()
    for mode, fused in (("glibc-fma", 1), ("glibc-sse2", 0)):
        y = _capi.debug_math("exp", g["x_exp"][:2000], mode, lib=lib)
        assert all(np.float32(L.go_glibc_expf(float(v), fused)).tobytes() == w.tobytes() for v, w in zip(g["x_exp"][:2000], y))
    y = _capi.debug_math("log", g["x_log"][:2000], "portable", lib=lib)
    assert all(np.float32(L.go_portable_logf(float(v))).tobytes() == w.tobytes() for v, w in zip(g["x_log"][:2000], y))


def test_mode_validation(emul_lib, modsim):
    from cogaps_amd import _capi
    with pytest.raises(RuntimeError, match="verification mode"):
        _capi.Session(modsim, lib=emul_lib(256), nPatterns=3, mathMode="glibc-fma")
    with pytest.raises(RuntimeError, match="outside 1"):
        _capi.Session(modsim, lib=emul_lib(256), nPatterns=3, subsetIndices=np.array([1, 2, 26], dtype=np.uint32), subsetDim=1)
    with pytest.raises(RuntimeError, match="outside 1"):
        _capi.Session(modsim, lib=emul_lib(256), nPatterns=3, subsetIndices=np.array([0, 2], dtype=np.uint32), subsetDim=2)
    with pytest.raises(ValueError, match="nPatterns"):
        _capi.Session(modsim, lib=emul_lib(256), nPatterns=3, whichMatrixFixed="P", fixedPatterns=np.ones((20, 4), np.float32))


def test_batched_chains_equal_single_sessions(emul_lib, gist):
    """cogaps_batch_* (batched multi-chain launches): chains stepped in lock-step through gen_kernel_multi / eval_kernel_multi give the
    bits of the same chains run one at a time -- dense with fused and split evaluation (shards of unequal size), the sparse
    model, a fixed matrix; sessions whose launch shapes differ are refused"""
    from cogaps_amd import _capi
    lib = emul_lib(256)

    def check(datas, kws, **common):
        for d, k, r in zip(datas, kws, _capi.run_batch(datas, lib=lib, kws=kws, **common)):
            o = _capi.run(d, lib=lib, **dict(common, **k))
            for f in ("Amean", "Pmean", "Asd", "Psd", "atomsA", "atomsP", "chisq", "totalUpdates", "meanChiSq", "averageQueueLengthA", "averageQueueLengthP"):
                assert np.array_equal(np.asarray(r[f]), np.asarray(o[f])), f
    check([gist[:200], gist[200:400], gist[400:610]], [dict(seed=5), dict(seed=6), dict(seed=7)], nPatterns=3, nIterations=15, outputFrequency=5)
    check([pu.synthetic(6000, 8, seed=1), pu.synthetic(6010, 8, seed=2)], [dict(seed=1), dict(seed=2)], nPatterns=3, nIterations=5, outputFrequency=5)
    check([pu.synthetic_counts(120, 40, zeros=0.8, seed=s) for s in (1, 2, 3)], [dict(seed=s) for s in (4, 5, 6)], nPatterns=4, nIterations=16, outputFrequency=4,
          sparseOptimization=True, takePumpSamples=True)
    fp = np.abs(np.random.default_rng(1).normal(size=(9, 3))).astype(np.float32)
    check([gist[:150], gist[150:300]], [dict(seed=1), dict(seed=2)], nPatterns=3, nIterations=10, outputFrequency=5, whichMatrixFixed="P", fixedPatterns=fp)
    with pytest.raises(RuntimeError, match="launch shape"):
        _capi.run_batch([pu.synthetic(6000, 8), pu.synthetic(300, 8)], lib=lib, nPatterns=3, nIterations=4)
    # dense subsets of 4095 and 4096 rows: one reduction width, one slice count -- and, for the dense kernels, nothing else to agree on (the
    # sparse model's workgroup width differs, 64 vs 128, and used to be compared for every model: distributed.py groups such shards into one
    # batch by the key _launch_shape computes, so the library must take them)
    from cogaps_amd.distributed import _launch_shape
    import cogaps_amd._capi as cc
    orig = cc.load
    cc.load = lambda: lib
    try:
        assert _launch_shape(4095, 6, False) == _launch_shape(4096, 6, False) and _launch_shape(4095, 6, True) != _launch_shape(4096, 6, True)
    finally:
        cc.load = orig
    check([pu.synthetic(4095, 6, seed=3), pu.synthetic(4096, 6, seed=4)], [dict(seed=8), dict(seed=9)], nPatterns=2, nIterations=3, outputFrequency=3)
    with pytest.raises(RuntimeError, match="launch shape"):
        _capi.run_batch([pu.synthetic_counts(4095, 6, seed=3), pu.synthetic_counts(4096, 6, seed=4)], lib=lib, nPatterns=2, nIterations=3, sparseOptimization=True)


def test_sparse_balanced_list_overflow(emul_lib):
    """a data vector whose round of flag words holds more common non-zeros than the lane-balancing list (SP_BAL_CAP = 2048): 8000
    genes, 70 % non-zero -- the owners' fallback loop; and a case that mixes both within one evaluation (two rounds of words)"""
    pu.run_stepwise(emul_lib(256), pu.synthetic_counts(8000, 10, zeros=0.3, seed=2), 4, trace=False, nPatterns=3, seed=5, total_iter=10, check_every=2, sparseOptimization=True)
    pu.run_stepwise(emul_lib(256), pu.synthetic_counts(20000, 6, zeros=0.85, seed=3), 3, trace=False, nPatterns=3, seed=6, total_iter=10, check_every=3, sparseOptimization=True)


def test_sparse_wide_vectors_merged_rounds(emul_lib):
    """data vectors of more than one round of flag words (more than 16384 elements: the kernel instantiation eval_sparse_kernel_wide):
    three rounds listed together (sp_partial_merged); four rounds whose common non-zeros overflow the merged list (falls back to the
    round-by-round form, which overflows its own list in turn); five rounds (beyond SP_MERGE_ROUNDS: round by round)"""
    lib = emul_lib(256)
    pu.run_stepwise(lib, pu.synthetic_counts(40000, 5, zeros=0.9, seed=9), 3, trace=False, nPatterns=3, seed=6, total_iter=10, check_every=3, sparseOptimization=True)
    pu.run_stepwise(lib, pu.synthetic_counts(60000, 4, zeros=0.5, seed=11), 2, trace=False, nPatterns=3, seed=7, total_iter=10, check_every=2, sparseOptimization=True)
    pu.run_stepwise(lib, pu.synthetic_counts(70000, 4, zeros=0.95, seed=12), 2, trace=False, nPatterns=3, seed=8, total_iter=10, check_every=2, sparseOptimization=True)


def test_update_of_zero_steps(emul_lib, gist, monkeypatch):
    """update(0) -- a Poisson draw of 0 has probability e^-10 per draw while a chain holds at most ten atoms -- is a no-op
    (AsynchronousGibbsSampler.h:94) for a single session (against the oracle) and inside a lock-stepped batch, which must not wait
    for the chain that has nothing to do: the batched chains equal the same chains run alone, zero-step iteration included"""
    from cogaps_amd import _capi
    lib = emul_lib(256)
    S, O = pu.make_pair(lib, gist[:200], nPatterns=3, seed=8, nIterations=20)
    for it in range(6):
        S.set_annealing(0.5), O.set_annealing(0.5)
        nA, nP = S.draw_steps()
        assert (nA, nP) == O.draw_steps()
        if it in (0, 3):
            nA = 0
        if it == 4:
            nP = 0
        S.iterate(nA, nP), O.iterate(nA, nP)
        pu.assert_state_equal(S, O, "it%d" % it)
    S.close(), O.close()
    monkeypatch.setenv("COGAPS_TEST_ZERO_STEPS", "2:3")
    kws = [dict(workerID=w, runningDistributed=True) for w in (1, 2, 3)]
    datas = [gist[:200], gist[200:400], gist[400:600]]
    common = dict(nPatterns=3, seed=8, nIterations=12, outputFrequency=6)
    batched = _capi.run_batch(datas, kws=kws, lib=lib, **common)
    for d, k, b in zip(datas, kws, batched):
        one = _capi.run(d, lib=lib, **dict(common, **k))
        assert one["totalUpdates"] == b["totalUpdates"] and np.array_equal(one["Amean"], b["Amean"]) and np.array_equal(one["Psd"], b["Psd"])
    monkeypatch.delenv("COGAPS_TEST_ZERO_STEPS")
    assert _capi.run(datas[1], lib=lib, **dict(common, **kws[1]))["totalUpdates"] != batched[1]["totalUpdates"]      # the hook did change worker 2's run
