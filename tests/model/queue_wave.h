// queue_wave.h -- TEST MODEL ONLY: the anchor-queue formulation of the device's find_event (DevWave with tag
// words, lz-ani_amd/csrc/lzani_kernels_pairs.h), as plain sequential code over the same data structures.
//
// The formulation: every query position whose mal-mer occurs in the reference (a tag-word hit: the tag is
// exact, so it is a true k-mer hit, hence a valid anchor of length >= mal) is a CANDIDATE.  Candidates are
// detected ahead of the scan in chunks of 64 positions (state-free), collected in order, and resolved in one
// lane-parallel pass: the single bucket entry carrying the tag gives the reference position, a 32-symbol
// word compare the (capped) match length; candidates whose bucket overflows or holds the tag more than once
// are flagged and evaluated by the whole wave when they are consumed.  The scan then
//   * in tracking mode runs one round over the <= mqd+1 tracking steps (close seeds as before, anchors from
//     the queue), and
//   * in lost mode JUMPS to the next queued candidate instead of walking rounds of 64 steps.
// The model checks that this yields exactly the reference's events (tests/test_model.py).
#pragma once
#include <vector>

#include "../../lz-ani_amd/csrc/lzani_core.h"

namespace lzani {

enum { AQ_CAP = 64, AQ_MAXCHUNKS = 16, AQ_LANE_CAP = 32 };
enum : u32 { AQ_COMPLEX = 0x80000000u, AQ_LONG = 0x40000000u };

struct QueueWaveTables {          // what the device's bucket table / tag words say, rebuilt from directory + entries
    std::vector<u32> bk;          // 4 per bucket, BK_EMPTY padded, [3] = BK_OVERFLOW when the bucket holds more
};

template <class Base>
struct QueueWave : Base {
    const QueueWaveTables& tab;
    const u32* qkL;               // mixed mal-mer hash per query position (KM 0xFFFFFFFF = none)
    int iend;
    mutable int scan_pos = 0, head = 0, cnt = 0;
    mutable int a_pos[AQ_CAP];
    mutable u32 a_ref[AQ_CAP];    // reference position | flags
    mutable int a_len[AQ_CAP];
    mutable u32 a_ext[AQ_CAP];     // null-extension record of a simple, short candidate, made at resolve time
    mutable int last_src = -1;     // queue entry the last event came from, if its record applies

    QueueWave(const Base& b, const QueueWaveTables& t, const u32* qk, int iend_) : Base(b), tab(t), qkL(qk), iend(iend_) {}

    // detect: is the step at p a candidate, and which bucket slot carries its tag (or complex)
    bool detect(int p, u32& slot_or_complex) const
    {
        const u32 hq = qkL[p];
        if (hq == 0xFFFFFFFFu) return false;
        const IndexView& I = this->I;
        const int tb = I.kb - I.dirbits;
        const u32 b = hq >> tb, tag = hq & I.tagmask;
        const u32* e = &tab.bk[4 * (size_t)b];
        if (e[3] == BK_OVERFLOW) { slot_or_complex = AQ_COMPLEX; return true; }
        int m = 0, k0 = -1;
        for (int k = 0; k < 4; ++k) if (e[k] != BK_EMPTY && (e[k] >> I.posbits) == tag) { if (!m) k0 = k; ++m; }
        if (!m) return false;
        slot_or_complex = m > 1 ? (u32)AQ_COMPLEX : 4 * b + (u32)k0;
        return true;
    }
    // lane-serial capped match length (the device compares one 32-symbol window)
    void resolve(int k) const
    {
        if (a_ref[k] & AQ_COMPLEX) { a_len[k] = 0; return; }
        const IndexView& I = this->I;
        const int pos = (int)(tab.bk[a_ref[k]] & (u32)lowmask(I.posbits)), qp = a_pos[k];
        const TextView &R = this->R, &Q = this->Q;
        int bound, n = 0;
        if (R.nfree && Q.nfree) bound = imin(run_end(R, pos) - pos, run_end(Q, qp) - qp);
        else bound = imin(R.len - pos, Q.len - qp);
        while (n < AQ_LANE_CAP && n < bound && sym_match(R, pos + n, Q, qp + n)) ++n;
        const bool lng = n == AQ_LANE_CAP && bound > AQ_LANE_CAP;
        a_ref[k] = (u32)pos | (lng ? (u32)AQ_LONG : 0u);
        a_len[k] = n;
        a_ext[k] = lng ? ext_rec_none(this->P.aw) : null_ext_record(this->P, R, Q, qp, pos, n);
    }
    bool ext_record(u32& x) const
    {
        if (last_src < 0) return false;
        x = a_ext[last_src];
        return true;
    }
    // the record of entry k describes the event (bpos, blen) at its step iff the entry is simple and short, long
    // enough to be an anchor, and the event is its anchor
    void note_src(int k, int ap, int al, int bpos, int blen) const
    {
        last_src = (!(a_ref[k] & (AQ_COMPLEX | AQ_LONG)) && al >= this->P.mal && bpos == ap && blen == al) ? k : -1;
    }
    void refill(int from) const
    {
        scan_pos = imax(scan_pos, from);
        head = 0; cnt = 0;
        int ncand = 0, first_unresolved = -1;
        for (int ch = 0; ch < AQ_MAXCHUNKS && scan_pos < iend && ncand < AQ_CAP; ++ch) {
            const int n = imin(64, iend - scan_pos);
            for (int l = 0; l < n; ++l) {
                u32 s;
                if (!detect(scan_pos + l, s)) continue;
                if (ncand < AQ_CAP) { a_pos[ncand] = scan_pos + l; a_ref[ncand] = s; }
                else if (first_unresolved < 0) first_unresolved = scan_pos + l;
                ++ncand;
            }
            scan_pos += n;
        }
        if (first_unresolved >= 0) scan_pos = first_unresolved;     // the surplus candidates are detected again later
        cnt = imin(ncand, AQ_CAP);
        for (int k = 0; k < cnt; ++k) resolve(k);
    }
    // the anchor of the queued step k, exactly (flags resolved by "the wave")
    void anchor_of(int k, int& ap, int& al) const
    {
        const int qp = a_pos[k];
        if (a_ref[k] & AQ_COMPLEX) { best_anchor(this->P, this->R, this->Q, this->I, qp, ap, al); return; }
        ap = (int)(a_ref[k] & 0x3FFFFFFFu);
        al = a_len[k];
        if (a_ref[k] & AQ_LONG) al = equal_len(this->R, ap, this->Q, qp, AQ_LANE_CAP);
        if (al < this->P.mal) { ap = 0; al = 0; }       // the k-mer words are those of the reference text: see the device code
    }
    void drop_before(int pos) const { while (head < cnt && a_pos[head] < pos) ++head; }

    bool find_event(int i, int n, bool trk, int r_end, int lit, int& adv, int& bpos, int& blen) const
    {
        const Params& P = this->P;
        int off = 0;
        last_src = -1;
        if (trk && lit <= P.mqd) {
            const int nt = imin(n, P.mqd - lit + 1);
            drop_before(i);
            if (scan_pos < i + nt) { scan_pos = i; refill(i); }          // the queue must cover the tracking steps
            for (int l = 0; l < nt; ++l) {
                const int qp = i + l;
                int ap = 0, al = 0, src = -1;
                drop_before(qp);
                if (head < cnt && a_pos[head] == qp) { anchor_of(head, ap, al); src = head; }
                int sp = 0, sl = 0;
                seed_search_window(P, this->R, this->Q, qp, r_end, lit + l, sp, sl);
                arbitrate(P, this->R.len, lit + l, ap, al, sp, sl);
                if (sl >= P.msl) { adv = l; bpos = sp; blen = sl; if (src >= 0) note_src(src, ap, al, sp, sl); return true; }
            }
            off = nt;
        }
        int pos = i + off;
        for (;;) {
            if (pos >= i + n) { adv = n; return false; }
            drop_before(pos);
            if (head >= cnt) {
                if (scan_pos >= iend) { adv = n; return false; }
                refill(pos);
                continue;
            }
            const int qp = a_pos[head];
            int ap, al;
            anchor_of(head, ap, al);
            ++head;
            if (al >= P.msl) { adv = qp - i; bpos = ap; blen = al; note_src(head - 1, ap, al, ap, al); return true; }
            pos = qp + 1;
        }
    }
};

}  // namespace lzani
