// lzani_kernels_split.h -- one directed pair by SEVERAL waves (round 4).  Included by lzani_hip.hip only (after
// lzani_kernels_pairs.h).
//
// Why: the reference scans a pair with one thread (CParser::parse, /root/reference/src/parser.cpp:482-716), the pair kernel
// with one wave -- and a batch of few, long pairs (32 bacterial genomes = 992 pairs on 8,192 wave slots) then lasts as long
// as its slowest pair, a related one: ~25 k events x ~6 us of dependent memory round trips.  The scan is sequential, but
// all of its state that later EVENTS depend on is where it stands behind an event (lzani_core.h: SplitStart): a query is
// cut every `seglen` positions, a wave per cut finds the cut's checkpoint (a fresh scan through its first event), a wave
// per segment scans from its checkpoint until its state equals a later checkpoint, and a stitch folds the segments'
// region counts along the chain of hand-overs.  Every hand-over is an equality of states, so the result is the whole
// scan's, bit for bit; a segment whose work turns out void (its look-back bound cut a backward extension short, or the
// region it started in ended the other way than it assumed) is dropped from the chain -- its cut disabled -- and the
// segment before it goes on from where it stopped, round by round, until the stitch gets through (at worst segment 0 scans
// the whole pair).
//
//   k_split<.., 0>  checkpoints: one wave per (pair, cut >= 1)
//   k_split<.., 1>  segments:    one wave per work item (pair, segment)
//   k_split_stitch  one thread per unfinished pair: the result, or the next round's work item
#pragma once

namespace lzani {

struct SplitArgs {
    PairArgs pa;                 // the batch, as the pair kernel gets it (candidate bitmaps: CAND 2)
    u32 rows;                    // rows of the batch
    u32 n_pairs;                 // its pairs: batch-relative pair p = absolute offset - pa.cb_e0
    u32 S;                       // cuts per pair (cut 0 = the start of the query)
    int seglen;                  // a cut every seglen query positions
    SplitStart* cuts;            // [n_pairs * S]
    SplitOut* outs;              // [n_pairs * S]
    const u32* work;             // this launch's items: p * S + segment (checkpoints: every (p, cut >= 1), no list)
    u32 n_work;
    u32* work_next;              // the stitch's: segments to run again
    u32* counters;               // [0] tickets of the running launch, [1] items in work_next, [2] pairs finished, [4..11] void segments by cause
    unsigned char* done;         // per pair: stitched and stored
    int reg;
    int last_round;              // the stitch: a pair that still does not get through has all its cuts disabled and segment 0 listed
    const unsigned char* heavy;  // per pair: cut it (the others are scanned whole, by segment 0)
};

// the pair of batch-relative index p: its row (slot), reference, query and absolute result offset
__device__ __forceinline__ void split_pair_of(const SplitArgs& a, u32 p, u32& slot, u32& r, u32& q, u64& e)
{
    e = a.pa.cb_e0 + p;
    u32 lo = 0, hi = a.rows;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (a.pa.row_off[mid] <= e) lo = mid; else hi = mid; }
    slot = lo;
    r = a.pa.ref_ids[slot];
    const u32 j = (u32)(e - a.pa.row_off[slot]);
    q = a.pa.query_ids ? a.pa.query_ids[e] : j + (j >= r ? 1u : 0u);
}

// MODE 0: the checkpoint of cut `seg` of pair p;  MODE 1: segment `seg` of pair p
template <bool NFREE, int DEFP, int MODE>
__device__ __forceinline__ void split_body(const SplitArgs& a, u32 p, u32 seg, bool resume, int lane, u32* lds)
{
    const Params Pk = DEFP ? folded_params(DEFP) : a.pa.P;
    u32 slot, r, q;
    u64 e;
    split_pair_of(a, p, slot, r, q, e);
    const int Lr = a.pa.G.L[r], Lq = a.pa.G.L[q];
    const u64 ro = a.pa.G.nmoff[r], qo = a.pa.G.nmoff[q];
    const int T = ref_text_len(Lr, Pk.mrd), D = Lq + Pk.mrd;
    IndexView iv;
    iv.dirz = a.pa.dirz + slot * a.pa.dir_stride;
    iv.ent = a.pa.ent + slot * a.pa.ent_stride;
    iv.kb = a.pa.geo.kb; iv.dirbits = a.pa.geo.dirbits; iv.posbits = a.pa.geo.posbits; iv.tagmask = a.pa.geo.tagmask;
    iv.bk = a.pa.bk + slot * a.pa.bk_stride;
    iv.tw = a.pa.tw ? a.pa.tw + slot * a.pa.tw_stride : nullptr;
    const bool nfree = NFREE ? true : !(a.pa.G.hasN[r] | a.pa.G.hasN[q]);
    constexpr int CHAIN = MODE == 1 ? chain_of(DEFP, NFREE) : 0;
    typedef DevWave<true, true, true, CHAIN, false, true> Wave;
    Wave w{Pk, ref_view(a.pa.G.t2 + 2 * ro, a.pa.G.nm + ro, Lr, Pk.mrd, nfree), qry_view(a.pa.G.t2 + 2 * qo, a.pa.G.nm + qo, Lq, Pk.mrd, nfree), iv, lane, lds,
           a.pa.G.kmS + 64 * ro, a.pa.G.kmL + 64 * qo, a.pa.G.kmS + 64 * qo, nullptr, nullptr, 0, e};
    w.iend = D - Pk.msl;
    w.cand_bits = a.pa.cbits + (e - a.pa.cb_e0) * a.pa.cbits_stride;
    PairMachine<Wave, false> m(w, Pk, T, D);
    SplitStart* const cuts = a.cuts + (u64)p * a.S;
    if (MODE == 0) {
        const int p0 = (int)seg * a.seglen;
        SplitStart cp;
        cp.i = -1; cp.r_end = cp.prev_rs = cp.pre_lit = cp.cl = cp.clit = 0;
        if (p0 < w.iend && a.heavy[p]) { w.scan_pos = p0; m.run_checkpoint(p0, &cp); }
        cuts[seg] = cp;                       // (wave-uniform: every lane stores the same values)
    } else {
        // (the segment's record is written where it belongs as the scan goes: a handful of stores at rare events, no registers held)
        SplitOut* const so = a.outs + (u64)p * a.S + seg;
        const SplitStart st = cuts[seg];
        if (resume) {                                    // on from where it stopped, past the (now disabled) cut it stopped at
            w.scan_pos = Wave::uniform(so->stop_i);
            m.resume_segment((int)seg, cuts, (int)a.S, so);
        } else
        if (seg > 0 && Wave::uniform(st.i) < 0) {        // (its cut has no checkpoint: nothing hands over to it)
            so->tm = so->tl = so->tc = 0; so->first = 2; so->first_cl = so->first_clit = so->first_re = 0; so->stop = -1;
            so->open_cl = so->open_clit = 0; so->open_rs = -1; so->assumed = 0; so->first_floor = 0; so->synced = 0; so->floor = 0;
        } else {
            if (seg > 0) w.scan_pos = Wave::uniform(st.i);
            m.run_segment((int)seg, seg > 0 ? &st : nullptr, cuts, (int)a.S, so);
        }
    }
}

template <bool NFREE, int DEFP, int MODE>
__global__ void __launch_bounds__(256, LZANI_WAVES_PER_SIMD) k_split(SplitArgs a)
{
    const int lane = threadIdx.x & 63;
    __shared__ u32 s_seed[4][SEED_LDS_WORDS];
    u32* const lds = s_seed[threadIdx.x >> 6];
    for (int k = lane; k < SEED_LDS_WORDS; k += 64) lds[k] = 0;
    for (;;) {
        u32 t = 0;
        if (lane == 0) t = atomicAdd(&a.counters[0], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= a.n_work) break;
        u32 p, seg;
        bool resume = false;
        if (MODE == 0) { p = t / (a.S - 1); seg = 1 + t % (a.S - 1); }      // (n_work = n_pairs * (S - 1))
        else { const u32 it = a.work[t]; resume = (it >> 31) != 0; p = (it & 0x7FFFFFFFu) / a.S; seg = (it & 0x7FFFFFFFu) % a.S; }
        split_body<NFREE, DEFP, MODE>(a, p, seg, resume, lane, lds);
    }
}

// One thread per pair that is not finished: the stitch.  Through -> the result is stored; a void segment -> its cut is
// disabled and the segment that handed over to it is listed for the next round.
__global__ void __launch_bounds__(256) k_split_stitch(SplitArgs a)
{
    for (u32 p = blockIdx.x * blockDim.x + threadIdx.x; p < a.n_pairs; p += gridDim.x * blockDim.x) {
        if (a.done[p]) continue;
        SplitStart* const cuts = a.cuts + (u64)p * a.S;
        int res[3], at = -1, from = -1, why = 5;
        if (split_stitch(cuts, a.outs + (u64)p * a.S, (int)a.S, a.reg, res, &at, &from, &why)) {
            int* o = a.pa.out + 3 * (a.pa.cb_e0 + p);
            o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
            a.done[p] = 1;
            atomicAdd(&a.counters[2], 1u);
            continue;
        }
        atomicAdd(&a.counters[4 + (why & 7)], 1u);          // (void segments by cause: the host's trace)
        u32 item;
        if (at < 0 || from < 0 || a.last_round) {          // nothing to retry (or out of rounds): segment 0 scans the pair whole
            for (u32 k = 1; k < a.S; ++k) cuts[k].i = -1;
            item = p * a.S;
        } else { cuts[at].i = -1; item = (p * a.S + (u32)from) | 0x80000000u; }     // (bit 31: resume, see split_body)
        a.work_next[atomicAdd(&a.counters[1], 1u)] = item;
    }
}

}  // namespace lzani
