// lane_wave.h -- TEST MODEL ONLY: a lane-serial wave policy for PairMachine (lz-ani_amd/csrc/lzani_core.h).
// One "lane" computes every wave primitive alone with word-parallel bit tricks; used by tests/model/ as a
// second, independent host model of the pair path.  Not part of the product library.
#pragma once
#include "../../lz-ani_amd/csrc/lzani_core.h"

namespace lzani {

// ---- lane-serial policy ------------------------------------------------------------------------
// The same machine driven by ONE lane per pair: every "wave" primitive is computed by the lane alone
// with word-parallel bit tricks (2-bit XOR + even-bit compress for the mismatch masks, shift-and for
// the match runs, a walk over the mismatch bits for the window break), and the scan advances one
// step at a time.  A thread-per-pair GPU kernel built on it was measured 8x slower than the wave kernel
// (rejected experiments) and removed from the product in round 2.
// Anchor-style index lookup of the close seeds of one tracking step (replaces the ht_short bucket walk,
// parser.cpp:548-580): positions p in [r_end, ref_pred + mrd) holding the step's msl-mer, ascending.
LZ_HD void seed_lookup(const Params& P, const TextView& R, const TextView& Q, const IndexView& S,
                       u64 qk, int qp, int r_end, int lit, int& sp, int& sl)
{
    sp = 0; sl = 0;
    const u64 h = mix_key(qk, S.kb);
    const int tb = S.kb - S.dirbits;
    const u32 b = (u32)(h >> tb), tag = (u32)(h & lowmask(tb)) & S.tagmask;
    u32 s = S.dirz[b], e = S.dirz[b + 1];
    if (e - s > (u32)R.len || e < s) { LZ_GUARD_TRIP(5); return; }
    const u32 pm = (u32)lowmask(S.posbits);
    const int ref_pred = r_end + lit;
    const int hi = imin(ref_pred + P.mrd, R.len - P.msl + 1);
    for (u32 j = s; j < e; ++j) {
        const u32 en = S.ent[j];
        if ((en >> S.posbits) != tag) continue;
        const int p = (int)(en & pm);
        if (p < r_end || p >= hi) continue;
        seed_consider(p, equal_len(R, p, Q, qp, P.msl), ref_pred, sp, sl);
    }
}

struct LaneWave {
    const Params& P;
    TextView R, Q;
    IndexView A, S;            // anchor (mal) and seed (msl) indexes of the reference; S tags must be exact
    const u32* qkL;            // optional per-position k-mer words of the query (nullptr: extract from the text)
    const u32* qkS;

    LZ_HD void stamp(int) const {}
    LZ_HD bool ext_record(u32&) const { return false; }

    LZ_HD u64 mism_fwd(int q0, int r0, int n) const
    {
        if (n <= 0) return 0;
        if (q0 < 0 || r0 < 0) {                                   // only near the text start: symbol by symbol
            u64 m = 0;
            for (int j = 0; j < n; ++j) if (!sym_match(R, r0 + j, Q, q0 + j)) m |= 1ULL << j;
            return m;
        }
        u64 mm = 0;
        for (int k = 0; k < 64 && k < n; k += 32) {
            u64 x = win2(R.t2, r0 + k) ^ win2(Q.t2, q0 + k);
            mm |= compress_even(x | (x >> 1)) << k;
        }
        u64 valid;
        if (R.nfree && Q.nfree) valid = lzani::valid_bits(R, r0) & lzani::valid_bits(Q, q0);
        else valid = ~(winN(R.nm, r0) | winN(Q.nm, q0)) & bits_below(R.len - r0) & bits_below(Q.len - q0);
        return (mm | ~valid) & lowmask(n);
    }
    LZ_HD u64 mism_bwd(int q0, int r0, int n) const                // bit j: Q[q0-1-j] vs R[r0-1-j]
    {
        if (n <= 0) return 0;
        u64 f = mism_fwd(q0 - n, r0 - n, n);                      // bit t: Q[q0-n+t]; reverse the n bits
        return brev64(f) >> (64 - n);
    }
    LZ_HD void mism2(int qa, int ra, int da, int na, int qb, int rb, int db, int nb, u64& A, u64& B) const
    {
        A = da > 0 ? mism_fwd(qa, ra, na) : mism_bwd(qa + 1, ra + 1, na);
        B = db > 0 ? mism_fwd(qb, rb, nb) : mism_bwd(qb + 1, rb + 1, nb);
    }
    LZ_HD ExtMasks ext_scan(u64 prevB, u64 B, int n) const
    {
        ExtMasks m{0, 0};
        const int a = P.ar < 1 ? 1 : P.ar;
        const u64 Z = ~B, Zp = ~prevB;
        u64 acc = Z;
        for (int k = 1; k < a; ++k) acc &= (Z << k) | (Zp >> (64 - k));
        m.qual = acc & lowmask(n);
        u64 mmbits = B & lowmask(n);                               // a break can only sit on a mismatch (am >= 0)
        while (mmbits) {
            const int j = ctz64(mmbits);
            mmbits &= mmbits - 1;
            const u64 W = (B << (63 - j)) | ((prevB >> 1) >> j);
            if (popc64(W >> (64 - P.aw)) > P.am) { m.brk = 1ULL << j; break; }
        }
        return m;
    }
    LZ_HD int best_split(u64 Lm, u64 Rm, int to_scan) const
    {
        int best = -1, bs = 0;
        for (int s = 0; s <= to_scan; ++s) {
            int v = popc64(Lm & lowmask(s)) + (s >= 64 ? 0 : popc64(Rm >> s));
            if (v >= best) { best = v; bs = s; }
        }
        return bs;
    }
    LZ_HD bool find_event(int i, int n, bool trk, int r_end, int lit, int& lane, int& bpos, int& blen) const
    {
        lane = n;                                          // the steps looked at when none hits: all of them
        for (int l = 0; l < n; ++l) {
            const int qp = i + l;
            int ap = 0, al = 0;
            if (qkL) { const u32 h = qkL[qp]; if (h != 0xFFFFFFFFu) anchor_lookup(P, R, Q, A, h, qp, ap, al); }
            else best_anchor(P, R, Q, A, qp, ap, al);
            int bp = ap, bl = al;
            if (trk && lit + l <= P.mqd) {
                int sp = 0, sl = 0;
                u64 qk = 0;
                bool ok;
                if (qkS) { const u32 v = qkS[qp]; ok = v != 0xFFFFFFFFu; qk = v; }
                else ok = kmer_at(Q, qp, P.msl, qk);
                if (ok) seed_lookup(P, R, Q, S, qk, qp, r_end, lit + l, sp, sl);
                arbitrate(P, R.len, lit + l, ap, al, sp, sl);
                bp = sp; bl = sl;
            }
            if (bl >= P.msl) { lane = l; bpos = bp; blen = bl; return true; }
        }
        return false;
    }
    LZ_HD void emit_region(const RegionCoords&) const {}
};

}  // namespace lzani
