// gen_chain_apply.h -- chained launch: what the generator's applier lanes fetch, and how they carry a decision out.  (part of the generator: included from gen_populate.h, which documents the method)
#pragma once
// ---- chained launch (chain_kernel.h): the generator applies the previous batch's decisions itself -----------------------------------------
// One launch evaluates batch n (its other workgroups) and generates batch n + 1 (this workgroup).  The evaluation workgroups write
// nothing the generator reads except one pair of tagged granules per proposal -- {code, traffic units} and one float -- and the
// generator's lane q carries the decision out on the atomic domain and the factor matrix: the stores the evaluation's writer thread
// makes in the two-launch form (eval_kernel.h: atom_set_mass, eval_store_matrix, eval_domain_move, eval_cache_erase), the same values
// from the same operations.  Everything those stores need that does not depend on the decision -- the queue record, the atom's record
// (its links name the holders of the cached copies, as the evaluation looks them up when it runs), the old bin's head and the bitmap's
// upper words of a move -- is fetched while the evaluation workgroups still run.
// The lane's part of the hand-over is split in two.  chain_fetch (before the wait) turns the record into ADDRESSES and old values: every
// word a decision can rewrite, as a pointer held in vector registers (null: nothing to write there).  chain_apply (behind the wait, on the
// decide -> generate chain) only computes the new values and stores -- no field of the sampler's record is read there: the compiler
// re-loads such fields through the scalar cache wherever they are used, and two dozen of those loads, each waited for, in the four
// type branches a wave walks through one after the other cost the first version 3 k cycles per launch.
struct ChainItem {
    uint32_t type; float m1, m2, old1, old2; uint64_t pos; unsigned long long eraseEntry;
    float *mass1, *rm1, *mass2, *rm2;        // atoms[h1].mass and the copy its left neighbour caches; exchange: the same for the partner
    float *mat1, *mat2; uint32_t *col1, *col2;      // mMatrix(r1,c1), mMatrix(r2,c2), the columns' counts of positive entries
    // move (ConcurrentAtomicDomain.cpp:126-132 across bins, as eval_domain_move decides it: the atom's own record is current -- births queued
    // after the move may have changed the links, nothing moves next to a moving atom, ProposalQueue.cpp:167,218)
    uint64_t *pos1, *rposL, *lposR;          // atoms[h1].pos, atoms[left].rpos, atoms[right].lpos
    uint32_t mb1, mb2;                       // the move's old and new bin (sh.dirty marks)
    uint32_t *head1, *head2; uint32_t head1Val, h1;      // old bin's head word (null: the atom is not the head) and what it becomes; new bin's head word (null: stays)
    unsigned long long *b0clr, *b0set, *b1set, *b2set; uint32_t bit1, bit2, bit1w, bit2w;      // bitmap words (null: nothing to do) and bit numbers
    // what the decision touches, by name (the notes for the window drawn ahead): the neighbours, the partner and its left neighbour, the
    // two matrix cells (bins), the atom's slot in the unsorted vector
    uint32_t hL, hR, h2, l2, cell1, cell2, idx;
    // sparse model (HybridMatrix: row copy, column copy with its epsilon rule and flag word, flagged count -- sp_change_matrix / sp_safely_change_matrix)
    uint32_t sparse; float *rows1, *rows2; unsigned long long *fl1, *fl2; unsigned long long fbit1, fbit2; float colv1, colv2; uint32_t flg1, flg2;
};
CG_DEVICE void chain_item_clear(ChainItem &it)
{
    it.type = 0; it.m1 = 0.f; it.m2 = 0.f; it.old1 = 0.f; it.old2 = 0.f; it.pos = 0; it.eraseEntry = 0ull;
    it.mass1 = nullptr; it.rm1 = nullptr; it.mass2 = nullptr; it.rm2 = nullptr; it.mat1 = nullptr; it.mat2 = nullptr; it.col1 = nullptr; it.col2 = nullptr;
    it.mb1 = 0; it.mb2 = 0;
    it.pos1 = nullptr; it.rposL = nullptr; it.lposR = nullptr; it.head1 = nullptr; it.head2 = nullptr; it.head1Val = CG_NONE; it.h1 = 0;
    it.b0clr = nullptr; it.b0set = nullptr; it.b1set = nullptr; it.b2set = nullptr; it.bit1 = 0; it.bit2 = 0; it.bit1w = 0; it.bit2w = 0;
    it.hL = CG_NONE; it.hR = CG_NONE; it.h2 = CG_NONE; it.l2 = CG_NONE; it.cell1 = 0; it.cell2 = 0; it.idx = 0;
    it.sparse = 0; it.rows1 = nullptr; it.rows2 = nullptr; it.fl1 = nullptr; it.fl2 = nullptr; it.fbit1 = 0ull; it.fbit2 = 0ull; it.colv1 = 0.f; it.colv2 = 0.f; it.flg1 = 0; it.flg2 = 0;
}
// what the second trip brings: the atom's record, the partner's left link, a move's old bin head and upper bitmap words
struct ChainMid { AtomRec a; uint32_t l2, head1, b1, b2; unsigned long long x1, x2;
                  float colv1, colv2; unsigned long long fw1, fw2; };      // sparse model: the column copy's entries and their flag words (sp_cell_load)
CG_DEVICE ChainMid chain_fetch_mid(const SamplerDev &S, const PropRec &p, const bool sparse)
{
    // every lane issues every load (a lane without a proposal, or of another type, reads harmless words: handle 0, bin 0): loads inside
    // divergent branches made the compiler wait for the whole trip where the branches join, before the work meant to run under it
    ChainMid m;
    const uint32_t hE = p.type == 'E' ? p.h2 : p.h1;
    m.b1 = gen_bin_of(S, p.curPos); m.b2 = gen_bin_of(S, p.pos);
    const uint32_t w0 = m.b2 >> 6, w1 = w0 >> 6, w2 = w1 >> 6;
    m.a = S.atoms[p.h1];
    m.l2 = S.atoms[hE].left;
    m.head1 = S.binHead[m.b1];
    m.x1 = S.bits1[w1]; m.x2 = S.bits2[w2];
    m.colv1 = 0.f; m.colv2 = 0.f; m.fw1 = 0ull; m.fw2 = 0ull;
    if (sparse) {      // (wave-uniform) the HybridMatrix column copy and its flags: rows are proposal-exclusive for the whole batch, so what is read here is what the decision finds
        m.colv1 = S.mat[(size_t)p.c1 * S.Mpad + p.r1]; m.fw1 = S.mflags[(size_t)p.c1 * S.Mw + (p.r1 >> 6)];
        m.colv2 = S.mat[(size_t)p.c2 * S.Mpad + p.r2]; m.fw2 = S.mflags[(size_t)p.c2 * S.Mw + (p.r2 >> 6)];
    }
    return m;
}
CG_DEVICE void chain_fetch_build(const SamplerDev &S, const PropRec &p, const ChainMid &m, ChainItem &it, const bool sparse);
CG_DEVICE void chain_fetch(const SamplerDev &S, const PropRec *queueRd, uint32_t q, ChainItem &it, const bool sparse)
{
    const PropRec p = queueRd[q];
    const ChainMid m = chain_fetch_mid(S, p, sparse);
    chain_fetch_build(S, p, m, it, sparse);
}
CG_DEVICE void chain_fetch_build(const SamplerDev &S, const PropRec &p, const ChainMid &m, ChainItem &it, const bool sparse)
{
    chain_item_clear(it);
    it.type = p.type; it.m1 = p.m1; it.m2 = p.m2; it.old1 = p.old1; it.old2 = p.old2; it.pos = p.pos; it.h1 = p.h1;
    it.eraseEntry = ((unsigned long long)(p.r1 * S.K + p.c1) << 32) | (unsigned long long)p.h1;
    const AtomRec a = m.a;
    it.hL = a.left; it.hR = a.right; it.idx = a.idx; it.cell1 = p.r1 * S.K + p.c1; it.cell2 = p.r2 * S.K + p.c2; it.h2 = p.h2; it.l2 = m.l2;
    it.mass1 = &S.atoms[p.h1].mass; it.rm1 = a.left != CG_NONE ? &S.atoms[a.left].rmass : nullptr;
    it.mat1 = &S.mat[(size_t)p.c1 * S.Mpad + p.r1]; it.col1 = &S.colPos[p.c1];
    const bool two = p.type == 'M' || p.type == 'E';
    if (two) { it.mat2 = &S.mat[(size_t)p.c2 * S.Mpad + p.r2]; it.col2 = &S.colPos[p.c2]; }
    if (sparse) {
        it.sparse = 1u;
        it.rows1 = &S.rows[(size_t)p.r1 * S.Kpad + p.c1]; it.fl1 = &S.mflags[(size_t)p.c1 * S.Mw + (p.r1 >> 6)]; it.fbit1 = 1ull << (p.r1 & 63u);
        it.colv1 = m.colv1; it.flg1 = (uint32_t)((m.fw1 >> (p.r1 & 63u)) & 1ull);
        if (two) {
            it.rows2 = &S.rows[(size_t)p.r2 * S.Kpad + p.c2]; it.fl2 = &S.mflags[(size_t)p.c2 * S.Mw + (p.r2 >> 6)]; it.fbit2 = 1ull << (p.r2 & 63u);
            it.colv2 = m.colv2; it.flg2 = (uint32_t)((m.fw2 >> (p.r2 & 63u)) & 1ull);
        }
    }
    if (p.type == 'E') { const uint32_t l2 = m.l2; it.mass2 = &S.atoms[p.h2].mass; it.rm2 = l2 != CG_NONE ? &S.atoms[l2].rmass : nullptr; }
    if (p.type == 'M') {
        const uint32_t b1 = m.b1, b2 = m.b2;
        it.mb1 = b1; it.mb2 = b2;
        const uint32_t head1 = m.head1;
        const uint32_t w0 = b2 >> 6, w1 = w0 >> 6, w2 = w1 >> 6;
        const unsigned long long x1 = m.x1, x2 = m.x2;
        it.pos1 = &S.atoms[p.h1].pos; it.rposL = a.left != CG_NONE ? &S.atoms[a.left].rpos : nullptr; it.lposR = a.right != CG_NONE ? &S.atoms[a.right].lpos : nullptr;
        if (head1 == p.h1) {      // the old bin loses its lowest atom: the right neighbour takes over if it lies in the same bin, else the bin is empty
            it.head1 = &S.binHead[b1];
            if (a.right != CG_NONE && gen_bin_of(S, a.rpos) == b1) it.head1Val = a.right; else { it.head1Val = CG_NONE; it.b0clr = &S.bits0[b1 >> 6]; it.bit1 = b1 & 63u; }
        }
        if (a.left == CG_NONE || gen_bin_of(S, a.lpos) != b2) it.head2 = &S.binHead[b2];
        it.b0set = &S.bits0[w0]; it.bit2 = b2 & 63u;      // bm_set, the upper levels' words read ahead
        if (!((x1 >> (w0 & 63u)) & 1ull)) { it.b1set = &S.bits1[w1]; it.bit1w = w0 & 63u; }
        if (!((x2 >> (w1 & 63u)) & 1ull)) { it.b2set = &S.bits2[w2]; it.bit2w = w1 & 63u; }
    }
}
// mMatrix entry = newv with the per-column count of positive entries (eval_store_matrix)
CG_DEVICE void chain_store_matrix(float *cell, uint32_t *col, float oldv, float newv)
{
    *cell = newv;
    const bool was = oldv > 0.f, is = newv > 0.f;
    if (was != is) { if (is) cg_atomic_add_u32(col, 1u); else cg_atomic_sub_u32(col, 1u); }
}
// sparse model: row copy = rowNew; column copy = colNew, or 0 with the flag cleared when colNew < epsilon (sp_store_col)
CG_DEVICE void chain_store_hybrid(float *rowCell, float *colCell, unsigned long long *flagWord, unsigned long long bit, uint32_t *colCount, float rowNew, float colNew, bool wasFlagged)
{
    *rowCell = rowNew;
    const bool zero = colNew < GAPS_EPSILON;
    if (zero) {
        if (wasFlagged) { (void)cg_atomic_and_u64(flagWord, ~bit); (void)cg_atomic_sub_u32(colCount, 1u); }
        *colCell = 0.f;
    } else {
        if (!wasFlagged) { (void)cg_atomic_or_u64(flagWord, bit); (void)cg_atomic_add_u32(colCount, 1u); }
        *colCell = colNew;
    }
}
// Carries the decision out (the stores of eval_kernel.h's writer thread: atom_set_mass, eval_store_matrix, eval_domain_move); returns
// whether the atom goes to the erase cache.  AsynchronousGibbsSampler.h:127-144 birth, :148-180 death / rebirth, :184-196 move, :201-219 exchange.
CG_DEVICE bool chain_apply(const ChainItem &it, uint32_t code, float val)
{
    const uint32_t tB = it.type == 'B', tD = it.type == 'D', tM = it.type == 'M', tE = it.type == 'E';
    const bool app = code == CHAIN_APPLY, era = code == CHAIN_ERASE;
    // new masses: B: val, D: the rebirth mass val, E: m1 + val and m2 - val
    const float n1 = tE ? it.m1 + val : val, n2 = it.m2 - val;
    // new matrix entries (safelyChangeMatrix: gm_max(old + delta, 0); changeMatrix for a birth and a move's destination)
    float d1 = tB ? val : (tD ? (val - it.m1) : (tM ? -it.m1 : (n1 - it.m1)));
    d1 = (era && tD) ? -1.f * it.m1 : d1;
    const float s1 = it.old1 + d1;
    const float nv1 = tB ? s1 : gm_max(s1, 0.f);
    const float s2 = it.old2 + (tM ? it.m1 : (n2 - it.m2));
    const float nv2 = tM ? s2 : gm_max(s2, 0.f);
    const bool doMat1 = app || (era && tD != 0u), doMat2 = app && (tM | tE) != 0u;
    const bool doMass1 = app && tM == 0u, doMass2 = app && tE != 0u;
    if (doMass1) { *it.mass1 = n1; if (it.rm1) *it.rm1 = n1; }
    if (doMass2) { *it.mass2 = n2; if (it.rm2) *it.rm2 = n2; }
    if (it.sparse) {
        // the HybridMatrix entries (sparse_kernels.h: sp_change_matrix for a birth and a move's destination, sp_safely_change_matrix elsewhere):
        // the row copy takes the new value; the column copy the new value -- for changeMatrix its OWN old value plus the change -- or zero
        // below epsilon, with its flag and the column's flagged count (HybridVector.cpp:55-86)
        if (doMat1) { const float colNew = tB ? it.colv1 + d1 : nv1; chain_store_hybrid(it.rows1, it.mat1, it.fl1, it.fbit1, it.col1, nv1, colNew, it.flg1 != 0u); }
        if (doMat2) { const float colNew = tM ? it.colv2 + it.m1 : nv2; chain_store_hybrid(it.rows2, it.mat2, it.fl2, it.fbit2, it.col2, nv2, colNew, it.flg2 != 0u); }
    } else {
        if (doMat1) chain_store_matrix(it.mat1, it.col1, it.old1, nv1);
        if (doMat2) chain_store_matrix(it.mat2, it.col2, it.old2, nv2);
    }
    if (app && tM != 0u) {
        *it.pos1 = it.pos; if (it.rposL) *it.rposL = it.pos; if (it.lposR) *it.lposR = it.pos;
        if (it.head1) *it.head1 = it.head1Val;
        if (it.b0clr) cg_atomic_and_u64(it.b0clr, ~(1ull << it.bit1));
        if (it.head2) *it.head2 = it.h1;
        cg_atomic_or_u64(it.b0set, 1ull << it.bit2);
        if (it.b1set) cg_atomic_or_u64(it.b1set, 1ull << it.bit1w);
        if (it.b2set) cg_atomic_or_u64(it.b2set, 1ull << it.bit2w);
    }
    return era;
}
