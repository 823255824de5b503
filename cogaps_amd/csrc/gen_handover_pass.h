// gen_handover_pass.h -- NOT a header of its own: the passes of the chained launch's hand-over, included by gen_populate.h where a lane takes part
// (the applier lanes; in the sparse model the attempt lanes as well), with
//   HP_STRIDE   proposals per pass (all lanes that take part)          HP_LANE     this lane's number among them
//   HP_MINE     whether the lane takes proposals at all                HP_FETCH(b) whether pass b's record is still to be fetched
// One proposal per lane and pass: wait for its two granules (read past this workgroup's caches until both carry the batch's tag), note an
// erased atom in the erase cache, carry the decision out, note what changed.  Reads: it, e_prevQ, hot, tag, S, sh, gs, t, eraseList, eraseCap, isSparse.
            uint32_t unitAcc = 0;
            for (uint32_t base = 0; base < e_prevQ; base += HP_STRIDE) {
                const uint32_t q = base + (HP_LANE);
                bool have = (HP_MINE) && q < e_prevQ;
                if (HP_FETCH(base)) { chain_item_clear(it); if (have) chain_fetch(S, hot.queueRd, q, it, isSparse); }      // (a queue longer than the applier lanes: the batch after a generator launch of two rounds)
                // where the notes of this proposal go, whatever is decided (the hashes ahead of the wait)
                const GenNotePos nH1 = gen_note_pos<GEN_DIRTY_ATOMS>(it.h1), nHL = gen_note_pos<GEN_DIRTY_ATOMS>(it.hL), nHR = gen_note_pos<GEN_DIRTY_ATOMS>(it.hR),
                                 nH2 = gen_note_pos<GEN_DIRTY_ATOMS>(it.h2), nL2 = gen_note_pos<GEN_DIRTY_ATOMS>(it.l2), nIdx = gen_note_pos<GEN_DIRTY_ATOMS>(~it.idx),
                                 nC1 = gen_note_pos<GEN_DIRTY_CELLS>(it.cell1), nC2 = gen_note_pos<GEN_DIRTY_CELLS>(it.cell2);
                const unsigned long long *gr = hot.grans + (size_t)q * CHAIN_GRAN_STRIDE;
                unsigned long long g0 = 0ull, g1 = 0ull; uint32_t spins = 0;
                for (;;) {
                    if (have) { g0 = cg_load_l2_u64(&gr[0]); g1 = cg_load_l2_u64(&gr[1]); }
                    const bool ok = !have || ((uint32_t)(g0 >> 32) == tag && (uint32_t)(g1 >> 32) == tag);
                    if (cg_ballot(!ok) == 0ull) break;
                    // bounded (platform.h: two seconds at least), never a hang.  A bound that is hit applies NOTHING: a granule without this batch's
                    // tag is an older batch's decision -- the lane drops its proposal, the error word ends the update on the host (the session is
                    // then marked unusable: its domain lacks decisions) and the workgroup leaves behind the barrier below without generating
                    // (round 6: the dropped proposal is marked, the batch is completed by the host once the launch has ended -- chain_recover_kernel --
                    // and the update goes on with two launches per batch)
                    if (cg_poll_expired(++spins)) { if (!ok) { have = false; S.queueUnits[q] = CHAIN_DROPPED_MARK(tag); } if ((t & 63u) == 0u) { gs->error = GAPS_ERR_SPIN; sh.spinFail = 1u; } break; }
                    cg_poll_pause();
                }
#if defined(GEN_TEST_SPIN_FAIL_EPOCH)
                // test-only variant of the emulator build (whose workgroups run one after the other: nothing ever waits): at one batch every third
                // lane gives up as if its decision had not arrived -- the launch ends the way a lost hand-over ends it
                if (cg_ballot(have && tag == (uint32_t)GEN_TEST_SPIN_FAIL_EPOCH && (q % 3u) == 1u) != 0ull) {
                    if (have && (q % 3u) == 1u) { have = false; S.queueUnits[q] = CHAIN_DROPPED_MARK(tag); }
                    if ((t & 63u) == 0u) { gs->error = GAPS_ERR_SPIN; sh.spinFail = 1u; }
                }
#endif
                GEN_TS(33);
                if (base == 0u) GEN_RT_AT(3, WIN);
                const uint32_t code = have ? ((uint32_t)g0 & 0xFFu) : CHAIN_NONE;
                if (have) unitAcc += ((uint32_t)g0 >> 8) << (it.sparse ? 5u : 0u);      // (dense: units of 4N bytes; sparse: bytes / 32, GenScalars::evalBytes counts bytes there)
                // erase cache (ConcurrentAtomicDomain.cpp:62-69): one slot per erased atom, in any order -- the flush sorts by position.
                // (Before the stores: what the barrier below waits for is LDS traffic only.)
                const bool er = have && code == CHAIN_ERASE, ap = have && code == CHAIN_APPLY;
                // bitmap words whose bits or bins' heads this decision (or the flush, for an erased atom) changes: the births drawn ahead check them
                if (er) gen_mark_dirty(sh.dirty, it.cell1);
                if (ap && it.type == 'M') { gen_mark_dirty(sh.dirty, it.mb1); gen_mark_dirty(sh.dirty, it.mb2); }
                // atom records whose fields change: the atom's own (mass / position) and the neighbours that cache copies of them; an erased
                // atom's neighbours are relinked by the flush.  Matrix cells that are rewritten.
                if (er || ap) {
                    gen_note_set(sh.dAtom, nH1);
                    if (it.hL != CG_NONE) gen_note_set(sh.dAtom, nHL);
                    if ((er || it.type == 'M') && it.hR != CG_NONE) gen_note_set(sh.dAtom, nHR);
                    if (ap && it.type == 'E') { gen_note_set(sh.dAtom, nH2); if (it.l2 != CG_NONE) gen_note_set(sh.dAtom, nL2); }
                    if (er) {      // (the vector slot the flush refills from the tail; the records the flush rewrites: the erased atom's and its neighbours')
                        gen_note_set(sh.dAtom, nIdx);
                        gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.h1));
                        if (it.hL != CG_NONE) gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.hL));
                        if (it.hR != CG_NONE) gen_note_set(sh.dErase, gen_note_pos<GEN_DIRTY_ERASE>(it.hR));
                    }
                    if (ap || it.type == 'D') gen_note_set(sh.dCell, nC1);
                    if (ap && (it.type == 'M' || it.type == 'E')) gen_note_set(sh.dCell, nC2);
                }
                const unsigned long long em = cg_ballot(er);
                if (em) {
                    const uint32_t cntE = (uint32_t)cg_popc64(em);
                    uint32_t b0 = 0;
                    if ((t & 63u) == 0u) b0 = cg_atomic_add_u32(&sh.eraseN, cntE);
                    b0 = cg_wave_bcast_u32(b0, 0);
                    if (er) {
                        const uint32_t k = b0 + (uint32_t)cg_popc64(em & ((1ull << (t & 63u)) - 1ull));
                        const unsigned long long e = it.eraseEntry;
                        if (k < (uint32_t)FLUSH_MAX) sh.eraseTmp[k] = e;
                        if (k < eraseCap) eraseList[k] = e; else gs->error = GAPS_ERR_ERASE_CAP;
                    }
                }
                if (have) chain_apply(it, code, gm_u2f((uint32_t)g1));
#if defined(COGAPS_EMUL)
                if (have && base) cg_atomic_add_u64(&gs->prof[6], 1ull);      // test-only build: decisions carried out in a pass beyond the first (a queue longer than the applier lanes)
                if (have && HP_FETCH(0u)) cg_atomic_add_u64(&gs->prof[2], 1ull);      // ... by attempt lanes (sparse model)
#endif
            }
            const uint32_t waveUnits = cg_wave_sum_u32(unitAcc);
            if ((t & 63u) == 0u && waveUnits) cg_atomic_add_u32(&sh.unitSum, waveUnits);
