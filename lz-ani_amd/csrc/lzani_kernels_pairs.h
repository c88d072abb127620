// lzani_kernels_pairs.h -- the pair kernels.  Included by lzani_hip.hip only (after lzani_kernels_index.h).
//   DevWave       the wave policy of PairMachine (lzani_core.h) on the device: one wavefront per pair
//   k_pairs       one wavefront per directed genome pair, persistent waves pulling pairs from
//                 per-XCD work queues (replaces prepare_data + parse + calc_stats,
//                 parser.cpp:37-50, 482-716, 734-783, and the worker loop of do_matching,
//                 lz_matcher.cpp:192-269); instantiations FAST/NFREE/DEFP/ALN/BK, see the kernel
#pragma once

namespace lzani {

// ------------------------------------------------------------------------------------------
// k_pairs: the pair kernel.
// ------------------------------------------------------------------------------------------
enum { SEED_BM_BITS = 14, SEED_BM_WORDS = 1 << (SEED_BM_BITS - 5), NQUEUES = 8 };
// anchor queue of a wave (tag-word instantiation): up to AQ_CAP resolved candidates, one per lane; candidates are
// detected ahead of the scan in chunks of 64 query positions, at most AQ_MAXCHUNKS per refill, and compacted
// through 2 x AQ_LDS_CAND words of LDS (positions, bucket slots) -- the first words of the wave's seed bitmap, which no
// round is using while a refill runs and which the refill leaves all zero again (2 KB of LDS per wave in all)
enum { AQ_CAP = 64, AQ_MAXCHUNKS = 32, AQ_LANE_CAP = 32, AQ_LDS_CAND = 128,
       SEED_PAD = 96,                 // words behind a wave's bitmap: 64 that take what the null chain's lanes without a k-mer write
                                      // (one per lane: LDS atomics on one address serialise), then one they read (always zero)
       SEED_LDS_WORDS = SEED_BM_WORDS + SEED_PAD };
static_assert(2 * AQ_LDS_CAND <= SEED_BM_WORDS, "the candidate buffers of refill live in the seed bitmap");
enum : u32 { AQ_COMPLEX = 0x80000000u, AQ_LONG = 0x40000000u, AQ_POS = 0x3FFFFFFFu };
enum { AQ_NONE = 0x7FFFFFFF };

// FAST: per-position k-mer words exist;  BK: the bucket table and its tag words exist;  JOIN: the candidates of a pair
// come from a join with the query's sorted k-mer list (long genomes) instead of a probe per query position
// the wave's vote as the compare result itself (HIP's __ballot goes through an int: v_cndmask 0/1 + v_cmp_ne per vote)
__device__ __forceinline__ u64 wballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
// a byte in all four bytes of a word: one byte permute (a 32-bit multiply by 0x01010101 runs at quarter rate)
__device__ __forceinline__ u32 rep4(u32 b) { return __builtin_amdgcn_perm(b, b, 0u); }

// The parameter sets a pair kernel folds into its code (template parameter DEFP of pair_body / k_pairs):
//   0 = none (the eight ints are read from the kernel's arguments), 1 = the reference's defaults (params.h:34-48),
//   2 = --mal 15 --msl 9 --reg 60 (BASELINE configs[3]), 9 = the set a run-time compile was made for (lzani_rtc.h: the
//   eight ints arrive as the macros LZANI_P_*; the reference reads them at run time and has one speed for all of them,
//   lz-ani.cpp:205-260 -- here every other tuple gets its own code object, built when a context first needs it).
#if defined(LZANI_RTC)
#define LZ_RTC_PARAMS Params{LZANI_P_MAL, LZANI_P_MSL, LZANI_P_MRD, LZANI_P_MQD, LZANI_P_REG, LZANI_P_AW, LZANI_P_AM, LZANI_P_AR}
#else
#define LZ_RTC_PARAMS Params{11, 7, 40, 40, 35, 15, 7, 3}
#endif
LZ_HD constexpr Params folded_params(int defp)
{
    return defp == 2 ? Params{15, 9, 40, 40, 60, 15, 7, 3} : defp == 9 ? LZ_RTC_PARAMS : Params{11, 7, 40, 40, 35, 15, 7, 3};
}
// What the hand-written null chain (DevWave::null_chain) takes for granted about the parameters: one lane per tracking
// step (mqd + 1 <= 64), a seed window of 64 .. 128 positions (two loads of window k-mers, the spare lanes of the second
// wrapping around to position 0), extensions that fit the null-extension record (aw <= 15, lzani_core.h), the window of
// an extension step out of a 96-bit funnel (aw >= 2, ar <= aw), msl-mers that index the 16 Kbit seed bitmap directly
// (msl <= 7) or through the hash bits k_kmers packs above them (msl 8, 9).
LZ_HD constexpr bool chain_params_ok(const Params& p)
{
    return p.mal >= 1 && p.mal <= 15 && p.msl >= 1 && p.msl <= 9 && p.mqd >= 0 && p.mqd <= 63 && p.mrd >= 1 &&
           p.mqd + p.mrd >= 64 && p.mqd + p.mrd <= 128 && p.aw >= 2 && p.aw <= 15 && p.ar <= p.aw && p.am >= 0 && p.reg >= 0 && p.reg < (1 << 20);
}
// CHAIN (template parameter of DevWave): 0 = no hand-written loop; 1 = the defaults, 3 = the defaults in a kernel for
// genomes without N; 2 = the long-genome set; 9 / 10 = the run-time set (10: genomes without N)
LZ_HD constexpr int chain_of(int defp, bool nfree)
{
    return defp == 1 ? (nfree ? 3 : 1) : defp == 2 ? 2 : defp == 9 ? (chain_params_ok(folded_params(9)) ? (nfree ? 10 : 9) : 0) : 0;
}
template <int CHAIN> struct ChainP {
    static constexpr Params P = folded_params(CHAIN == 2 ? 2 : CHAIN >= 9 ? 9 : 1);
    static constexpr bool NF = CHAIN == 3 || CHAIN == 10;
    enum { MAL = P.mal, MSL = P.msl, MRD = P.mrd, MQD = P.mqd, REG = P.reg, AW = P.aw, AM = P.am > 16 ? 16 : P.am, AR = P.ar < 1 ? 1 : P.ar,
           NT = MQD + 1, WIN = MQD + MRD };
};

#ifndef LZANI_STRETCH_DENSE
#define LZANI_STRETCH_DENSE 0
#endif
// SPLITW: the wave scans one SEGMENT of a pair (lzani_core.h: SplitStart; lzani_kernels_split.h): its hand-written loops
// commit several events a call, none of them across the next checkpoint (split_limit)
template <bool FAST, bool BK = false, bool JOIN = false, int CHAIN = 0, bool LFLT = false, bool SPLITW = false>
struct DevWave {
    static constexpr bool NULL_CHAIN = CHAIN != 0;      // (see ChainP)
    const Params& P;
    TextView R, Q;
    IndexView I;
    int lane;
    u32* bitmap;     // per-wave LDS: SEED_BM_WORDS words, all zero between rounds
    const u32* rkS;  // FAST: msl-mers of the reference text, one per position
    const u32* qkL;  // FAST: hashed mal-mers of the query text
    const u32* qkS;  // FAST: msl-mers of the query text
    // alignment instantiation: region sink
    lzani_region* reg_out;
    unsigned long long* reg_count;
    unsigned long long reg_cap, pair_e;
    // Anchor queue (BK instantiation, see find_event): lane k holds the k-th queued candidate of the pair --
    // its query position, its reference position (| AQ_LONG / AQ_COMPLEX) and its match length (capped at
    // AQ_LANE_CAP when AQ_LONG is set); q_head .. q_cnt are live, the query is scanned up to scan_pos.
    int iend = 0;    // steps exist for query positions < iend
    int scan_pos = 0, q_head = 0, q_cnt = 0;
    int last_src = -1;   // queue entry the event just returned came from, if its null-extension record applies
    int stop_i = 0x7FFFFFFF;     // SPLITW: the query position of the nearest checkpoint ahead
    __device__ __forceinline__ void split_limit(int at) { stop_i = at; }
    bool split_taint = false;    // SPLITW: the machine's look-back is a lower bound of the true scan's (lzani_core.h: SplitOut)
    __device__ __forceinline__ void split_taint_set(bool t) { split_taint = t; }
    static __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
    enum { SPLIT_MARGIN = 256 };  // a chain call's last commit starts below the limit and moves less than this (gap <= mqd, match < 72 + a first extension chunk)
    u32 a_ext = EXT_REC_NONE;    // lane k: null-extension record of candidate k (lzani_core.h: null_ext_record)
    // Join form of candidate detection (long genomes): the wave's candidate bitmap over the query positions of the
    // pair, filled by join() before the scan; nullptr = candidates are probed position by position (refill)
    unsigned long long* cand_bits = nullptr;
    int a_pos = AQ_NONE, a_len = 0;
    u32 a_ref = 0;
#ifdef LZANI_CHAIN_STATS
    unsigned st[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // diagnostic build: chain calls, commits, exits by kind, events by the general path, refills
    // wave cycles between two null_chain calls by what the first one handed back (0 nothing, 1 round done, 2 event found),
    // slot 3 = inside null_chain itself; stc_open = the slot the running interval belongs to
    unsigned sw[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // exit_seed by reason (see null_chain)
    unsigned sr[4] = {0, 0, 0, 0};                  // events found but not null, by reason: close, region kept (or none open), no forward record, backward
    unsigned long long stc[4] = {0, 0, 0, 0}, stc_t0 = 0;
    int stc_open = -2;
    __device__ __forceinline__ void cycles_mark(int code)
    {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if (stc_open == 0) stc[0] += t - stc_t0; else if (stc_open == 1) stc[1] += t - stc_t0; else if (stc_open == 2) stc[2] += t - stc_t0; else if (stc_open == 3) stc[3] += t - stc_t0;
        stc_open = code < 0 ? 3 : code;
        stc_t0 = t;
    }
#endif
#ifdef LZANI_PATH_STATS                             // diagnostic build: which path found the events of a pair (LZ_PS slots, see lzani_hip.hip)
    unsigned ps[24] = {0};
    unsigned pw[12] = {0};                          // stretch chain: exits by reason
#define LZ_PS(k) (ps[k] += 1)
#define LZ_PSN(k, n) (ps[k] += (unsigned)(n))
#else
#define LZ_PS(k) ((void)0)
#define LZ_PSN(k, n) ((void)0)
#endif
#ifdef LZANI_PHASE_TIME                             // slim diagnostic build: wave cycles inside the null chain / inside refill / per pair
    unsigned long long pt_chain = 0, pt_refill = 0;
    static __device__ __forceinline__ unsigned long long pt_now()
    {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        return t;
    }
#endif
    // null_chain found the queue short of what it needs -- empty, or ending inside the tracking steps of i -- with more
    // of the pair's bitmap to read: the next find_event call only refills (from i, if queued candidates are left)
    bool refill_only = false;
    int restart_at = -1;         // the last i a refill restarted the detection at (no second restart at the same place)
    // the stretch chain has done the tracking round of the next find_event call and found no seed in it
    bool round_no_seed = false;
    // the tracking round null_chain has done for the next find_event call (CHAIN)
    bool pre_round = false;
    u64 pre_seed = 0;
    u32 pre_rk0 = KM_INVALID, pre_rk1 = KM_INVALID, pre_qk = KM_INVALID;
    // LFLT: the block's copy, in LDS, of the presence filter of the reference (k_idx_filter): bit (h & fmask)
    const u32* flt = nullptr;
    u32 fmask = 31;
    __device__ __forceinline__ void emit_region(const RegionCoords& c) const
    {
        // One slot per wave, reserved by ONE lane and without a lane-dependent branch (see the note at the ticket fetch:
        // a second `if (lane == 0)` inside the persistent loop let the compiler split lane 0 from the rest): the exec
        // mask is narrowed to lane 0 around the returning atomic by hand, 63 no-op atomics on one address less per region.
        unsigned long long old, saved;
        const unsigned long long one = 1ULL;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "global_atomic_add_x2 %[old], %[addr], %[one], off sc0\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [old] "=&v"(old), [sv] "=&s"(saved)
                     : [addr] "v"(reg_count), [one] "v"(one)
                     : "memory");
        const u32 lo = __builtin_amdgcn_readfirstlane((u32)old), hi = __builtin_amdgcn_readfirstlane((u32)(old >> 32));
        const unsigned long long slot = ((unsigned long long)hi << 32) | lo;
        if (slot < reg_cap) {
            lzani_region* o = reg_out + slot;
            o->pair = pair_e;
            o->ref_start = c.ref_start; o->ref_end = c.ref_end; o->seq_start = c.seq_start; o->seq_end = c.seq_end;
            o->num_matches = c.nm; o->num_mismatches = c.nmm;
        }
    }
#ifdef LZANI_STAMPS
    // diagnostic build only: cycles per section, summed per wave, added to g_stamp_acc at pair end
    mutable unsigned long long t0;
    mutable unsigned long long acc[8];
    mutable int cur;
    __device__ __forceinline__ void stamp(int k) const
    {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        acc[cur] += t - t0; cur = k; t0 = t;
    }
#else
    __device__ __forceinline__ void stamp(int) const {}
#endif

    // One lane's mismatch test.  N-free texts: both 32-bit text words are requested before either is
    // used (the empty asm pins the two loads ahead of the first use - under the 64-VGPR budget the
    // scheduler otherwise waits for the first load before it issues the second), no branch.
    struct SymReq { u32 wr, wq; int sr, sq; bool ok; };
    __device__ __forceinline__ SymReq sym_request(int rp, int qp) const
    {
        const bool vr = pos_valid(R, rp), vq = pos_valid(Q, qp);
        const int a = vr ? rp : 0, b = vq ? qp : 0;
        SymReq x;
        x.wr = reinterpret_cast<const u32a*>(R.t2)[(u32)a >> 4];
        x.wq = reinterpret_cast<const u32a*>(Q.t2)[(u32)b >> 4];
        x.sr = (a & 15) * 2; x.sq = (b & 15) * 2; x.ok = vr & vq;
        return x;
    }
    static __device__ __forceinline__ bool sym_differs(const SymReq& x)
    {
        return !(x.ok & ((((x.wr >> x.sr) ^ (x.wq >> x.sq)) & 3u) == 0));
    }
    __device__ __forceinline__ bool lane_mismatch(int rp, int qp) const
    {
        if (R.nfree && Q.nfree) {
            SymReq x = sym_request(rp, qp);
            asm volatile("" : "+v"(x.wr), "+v"(x.wq));
            return sym_differs(x);
        }
        return !sym_match(R, rp, Q, qp);
    }
    __device__ __forceinline__ u64 mism_fwd(int q0, int r0, int n) const
    {
        return wballot((lane < n) & lane_mismatch(r0 + lane, q0 + lane));
    }
    __device__ __forceinline__ u64 mism_bwd(int q0, int r0, int n) const
    {
        return wballot((lane < n) & lane_mismatch(r0 - 1 - lane, q0 - 1 - lane));
    }
    // two masks, four independent loads in flight, one wait
    __device__ __forceinline__ void mism2(int qa, int ra, int da, int na, int qb, int rb, int db, int nb, u64& A, u64& B) const
    {
        const int rpa = ra + da * lane, qpa = qa + da * lane, rpb = rb + db * lane, qpb = qb + db * lane;
        bool ma, mb;
        if (R.nfree && Q.nfree) {
            SymReq x = sym_request(rpa, qpa), y = sym_request(rpb, qpb);
            asm volatile("" : "+v"(x.wr), "+v"(x.wq), "+v"(y.wr), "+v"(y.wq));
            ma = sym_differs(x); mb = sym_differs(y);
        } else {
            ma = !sym_match(R, rpa, Q, qpa);
            mb = !sym_match(R, rpb, Q, qpb);
        }
        A = wballot((lane < na) & ma);
        B = wballot((lane < nb) & mb);
    }
    // three forward masks, six independent loads in flight, one wait (a close match: the two diagonals of its gap fill and
    // the first chunk of the forward extension behind it)
    static constexpr bool HAS_MISM3 = true;
    __device__ __forceinline__ void mism3(int qa, int ra, int na, int qb, int rb, int nb, int qc, int rc, int nc, u64& A, u64& B, u64& C) const
    {
        bool ma, mb, mc;
        if (R.nfree && Q.nfree) {
            SymReq x = sym_request(ra + lane, qa + lane), y = sym_request(rb + lane, qb + lane), z = sym_request(rc + lane, qc + lane);
            asm volatile("" : "+v"(x.wr), "+v"(x.wq), "+v"(y.wr), "+v"(y.wq), "+v"(z.wr), "+v"(z.wq));
            ma = sym_differs(x); mb = sym_differs(y); mc = sym_differs(z);
        } else {
            ma = !sym_match(R, ra + lane, Q, qa + lane);
            mb = !sym_match(R, rb + lane, Q, qb + lane);
            mc = !sym_match(R, rc + lane, Q, qc + lane);
        }
        A = wballot((lane < na) & ma);
        B = wballot((lane < nb) & mb);
        C = wballot((lane < nc) & mc);
    }
    // Close-seed search of the tracking steps of a round (replaces the ht_short bucket walk, parser.cpp:548-580).
    // rk0/rk1 = msl-mers of the window positions r_end+lane / r_end+64+lane, qk = msl-mer of this lane's step
    // (KM_INVALID where there is none).
    //  1. seed_prefilter: the window k-mers set bits in a per-wave 16 Kbit LDS bitmap (exact for msl <= 7, a
    //     Bloom filter above), each tracking lane tests its own k-mer; the bits are cleared again (the bitmap
    //     is always zero between rounds).  In four rounds out of five of an unrelated pair no lane hits.
    //  2. seed_candidates: when the verify loop reaches a hit lane, that lane's k-mer is broadcast and compared
    //     with the window k-mers, which are still in registers: two ballots give the lane's candidate positions
    //     (ascending = the order of the reference's bucket) - no join structure at all.
    __device__ __forceinline__ u32 bm_hash(u32 k) const
    {
        // (msl 8, 9: k_kmers packs 14 hash bits above the msl-mer)
        return P.msl <= 7 ? k : P.msl <= 9 ? k >> (2 * P.msl) : (k * 0x9E3779B1u) >> (32 - SEED_BM_BITS);
    }
    __device__ __forceinline__ bool seed_prefilter(u32 rk0, u32 rk1, u32 qk) const
    {
        // no lane-dependent branch (each costs two to three scalar instructions of exec-mask bookkeeping and the
        // kernel is bound by the scalar pipe): a lane without a k-mer ORs nothing into / reads (masked) / clears word
        // `lane` of the bitmap itself -- the clear comes after every lane's read, and every word a round sets is
        // cleared by the round anyway
        const u32 scratch = (u32)lane;
        const u32 b0 = bm_hash(rk0), b1 = bm_hash(rk1), bq = bm_hash(qk);
        const bool v0 = rk0 != KM_INVALID, v1 = rk1 != KM_INVALID, vq = qk != KM_INVALID;
        const u32 w0 = v0 ? b0 >> 5 : scratch, w1 = v1 ? b1 >> 5 : scratch, wq = vq ? bq >> 5 : scratch;
        atomicOr(&bitmap[w0], v0 ? 1u << (b0 & 31) : 0u);
        atomicOr(&bitmap[w1], v1 ? 1u << (b1 & 31) : 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool hit = vq & ((bitmap[wq] >> (bq & 31)) & 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bitmap[w0] = 0;
        bitmap[w1] = 0;
        return hit;
    }
    // window positions idx < lim (the step's own window) whose msl-mer equals the step's (qkl, wave-uniform)
    __device__ __forceinline__ void seed_candidates(u32 qkl, int lim, u32 rk0, u32 rk1, u64& d0, u64& d1) const
    {
        d0 = wballot(rk0 == qkl) & lowmask(lim);
        d1 = wballot(rk1 == qkl) & lowmask(lim - 64);
    }

    __device__ __forceinline__ u64 bcast64(u64 v, int l) const      // readlane returns a signed int: widen as u32
    {
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
        return ((u64)hi << 32) | lo;
    }
    // equal_len by the whole wave: 64 symbols per step (ballot + ctz), same value as lzani::equal_len
    __device__ __forceinline__ int wave_equal_len(int rp, int qp, int start) const
    {
        const int bound = imin(R.len - rp, Q.len - qp);
        int n = start;
        while (n < bound) {
            u64 B = mism_fwd(qp + n, rp + n, imin(64, bound - n));
            if (B) { n += ctz64(B); break; }
            n += 64;
        }
        n = imin(n, bound);
        return n > start ? n : start;
    }

    // Generic round: every lane evaluates its step completely (portable code of lzani_core.h).
    __device__ __forceinline__ bool find_event_generic(int i, int n, bool trk, int r_end, int lit,
                                                       int& ev_lane, int& bpos, int& blen) const
    {
        int bp = 0, bl = 0;
        if (lane < n)
            eval_step(P, R, Q, I, i + lane, trk && (lit + lane <= P.mqd), r_end, lit + lane, bp, bl);
        u64 hit = wballot(lane < n && bl >= P.msl);
        if (!hit) { ev_lane = n; return false; }
        ev_lane = ctz64(hit);
        bpos = __builtin_amdgcn_readlane(bp, ev_lane);     // ev_lane is wave-uniform (from the ballot)
        blen = __builtin_amdgcn_readlane(bl, ev_lane);
        return true;
    }

    // best_anchor over a whole bucket, by the wave: the longest match >= mal among the entries carrying the
    // tag, the smallest position among equals - the reference's "first wins" over ascending positions
    // (parser.cpp:514-531), stated without relying on the order (buckets beyond IDX_SORT_MAX stay unsorted).
    __device__ __forceinline__ void walk_bucket(u32 b, u32 tag, int qp, int& ap, int& al) const
    {
        const u32 pm = (u32)lowmask(I.posbits);
        u32 s = I.dirz[b], e = I.dirz[b + 1];
        if (e - s > (u32)R.len || e < s) { LZ_GUARD_TRIP(2); e = s; }
        for (u32 j = s; j < e; ++j) {
            const u32 x = (u32)__builtin_amdgcn_readfirstlane((int)I.ent[j]);
            if ((x >> I.posbits) != tag) continue;
            const int p = (int)(x & pm);
            const int m = wave_equal_len(p, qp, 0);
            if (m >= P.mal && (m > al || (m == al && p < ap))) { al = m; ap = p; }
        }
    }

    // best_anchor of one step by the whole wave, from the bucket table (hql = the step's mixed mal-mer hash,
    // wave-uniform): the bucket by a scalar load, its entries carrying the tag verified with wave_equal_len.
    __device__ __forceinline__ void anchor_by_wave(u32 hql, int qp, int& ap, int& al) const
    {
        const int tb = I.kb - I.dirbits;
        const u32 pm = (u32)lowmask(I.posbits), tag = hql & I.tagmask;
        // the bucket's address is wave-uniform: a scalar load (the table is read-only during the launch)
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        u32x4 bq;
        const u64 baddr = (u64)(I.bk + 4 * (u64)(hql >> tb));
        // (the address through readfirstlane: an "s" operand must be in SGPRs whatever the allocator did)
        const u64 bsgpr = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(baddr >> 32)) << 32) |
                          (u32)__builtin_amdgcn_readfirstlane((int)(u32)baddr);
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bq) : "s"(bsgpr) : "memory");
        const u32 en[4] = {bq.x, bq.y, bq.z, bq.w};
        if (__builtin_expect(en[3] == BK_OVERFLOW, 0)) {     // the whole bucket (big buckets are not sorted)
            walk_bucket(hql >> tb, tag, qp, ap, al);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((en[k] >> I.posbits) != tag) continue;
                const int p = (int)(en[k] & pm);
                const int m = wave_equal_len(p, qp, 0);
                if (m >= P.mal && m > al) { al = m; ap = p; }
            }
        }
    }

    // Fast round (k-mer words available, seed window <= 128): the lanes only DETECT candidates --
    // a bucket entry whose tag equals the step's mal-mer, a window position whose msl-mer equals the
    // step's -- and the wave then verifies the candidates of the first candidate lane together
    // (wave_equal_len), exactly as eval_step would for that step; if that step turns out not to hit
    // (quirk Q1, or mal < msl) the next candidate lane is taken.
    __device__ __forceinline__ bool find_event_round(int i, int n, bool trk, int r_end, int lit,
                                                     int& ev_lane, int& bpos, int& blen) const
    {
        n = imin(n, 64);
        ev_lane = n;                                                 // steps looked at when none hits
        const int nt = trk ? imin(n, P.mqd - lit + 1) : 0;          // lanes [0, nt) are tracking steps
        const int W = nt > 0 ? imin(lit + nt - 1 + P.mrd, R.len - P.msl + 1 - r_end) : 0;
        const int tb = I.kb - I.dirbits;
        if (__builtin_expect(!FAST || W > 128 || (!BK && I.tagmask != (u32)lowmask(tb)), 0)) // the stored tag must identify the k-mer
            return find_event_generic(i, n, trk, r_end, lit, ev_lane, bpos, blen);

        // every independent load of the round first, unconditionally (the k-mer arrays are padded by two
        // 64-entry blocks, so the addresses are always in bounds) and masked afterwards: no branches
        u32 hq = qkL[(u32)(i + lane)], rk0 = KM_INVALID, rk1 = KM_INVALID, qk = KM_INVALID;
        hq = lane < n ? hq : KM_INVALID;
        if (W > 0) {                                                 // wave-uniform
            const int w0 = imin(lane, W - 1), w1 = imin(lane + 64, W - 1);
            qk = qkS[(u32)(i + lane)];
            rk0 = rkS[(u32)(r_end + w0)];
            rk1 = rkS[(u32)(r_end + w1)];
            qk = lane < nt ? qk : KM_INVALID;
            rk0 = lane < W ? rk0 : KM_INVALID;
            rk1 = lane + 64 < W ? rk1 : KM_INVALID;
        }
        // anchor candidates of this lane's step: bucket entries carrying the step's (exact) tag.
        //   BK: the bucket's tag word (4 B) says whether one of its first four entries carries the tag, or
        //   that the bucket overflows (~0.03 % of buckets); the 16-byte bucket itself is read by the verify
        //   step of a candidate only;
        //   bucket table without tag words: one 16-byte load brings the bucket's first four entries, which stay
        //   in registers for the verify step, an overflowing bucket sends the lane through the directory;
        //   no bucket table (large genomes): directory + entries.
        u32 ac = 0;
        uint4 bkv = {BK_EMPTY, BK_EMPTY, BK_EMPTY, BK_EMPTY};
        bool viadir = !BK && I.bk == nullptr;
        if constexpr (BK) {
            const bool valid = hq != KM_INVALID;
            const u32 w = I.tw[valid ? hq >> tb : 0u];
            const u32 x = w ^ rep4(0x80u | (hq & I.tagmask));       // a zero byte = a slot with this tag
            ac = (u32)(valid & ((((x - 0x01010101u) & ~x & 0x80808080u) != 0) | (w == TW_OVERFLOW)));
        } else if (hq != KM_INVALID) {
            const u32 b = hq >> tb, tag = hq & I.tagmask;
            if (I.bk) {
                bkv = reinterpret_cast<const uint4*>(I.bk)[b];
                viadir = bkv.w == BK_OVERFLOW;
                if (!viadir)                              // BK_EMPTY never carries a real tag (tag + position bits <= 30)
                    ac = (u32)((bkv.x >> I.posbits) == tag) + (u32)((bkv.y >> I.posbits) == tag) +
                         (u32)((bkv.z >> I.posbits) == tag) + (u32)((bkv.w >> I.posbits) == tag);
            }
            if (viadir) {
                u32 s = I.dirz[b], e = I.dirz[b + 1];
                if (e - s > (u32)R.len || e < s) { LZ_GUARD_TRIP(2); e = s; }
                for (u32 j = s; j < e; ++j) ac += (I.ent[j] >> I.posbits) == tag;
            }
        }
        stamp(2);
        bool shit = false;
        if (W > 0) shit = seed_prefilter(rk0, rk1, qk);
        u64 todo = wballot(ac != 0 || shit);
        stamp(7);
        const u32 pm = (u32)lowmask(I.posbits);
        while (todo) {
            const int l = ctz64(todo);
            todo &= todo - 1;
            const int qp = i + l;
            int ap = 0, al = 0;
            const u32 cnt = __builtin_amdgcn_readlane(ac, l);
            if (BK && cnt) {                                         // candidate step: now its bucket is read, by the wave
                anchor_by_wave((u32)__builtin_amdgcn_readlane((int)hq, l), qp, ap, al);
            } else if (cnt) {
                const u32 hql = (u32)__builtin_amdgcn_readlane((int)hq, l), tag = hql & I.tagmask;
                if (__builtin_amdgcn_readlane((int)viadir, l)) {
                    walk_bucket(hql >> tb, tag, qp, ap, al);
                } else {                                             // the step's bucket is still in registers
                    const u32 en[4] = {(u32)__builtin_amdgcn_readlane((int)bkv.x, l), (u32)__builtin_amdgcn_readlane((int)bkv.y, l),
                                       (u32)__builtin_amdgcn_readlane((int)bkv.z, l), (u32)__builtin_amdgcn_readlane((int)bkv.w, l)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((en[k] >> I.posbits) != tag) continue;
                        const int p = (int)(en[k] & pm);
                        const int m = wave_equal_len(p, qp, 0);
                        if (m >= P.mal && m > al) { al = m; ap = p; }
                    }
                }
            }
            int bp = ap, bl = al;
            if (l < nt) {
                int sp = 0, sl = 0;
                const int ref_pred = r_end + lit + l;
                u64 d0 = 0, d1 = 0;
                const u32 qkl = (u32)__builtin_amdgcn_readlane((int)qk, l);
                if (qkl != KM_INVALID) seed_candidates(qkl, lit + l + P.mrd, rk0, rk1, d0, d1);
                while (d0 | d1) {
                    int idx;
                    if (d0) { idx = ctz64(d0); d0 &= d0 - 1; }
                    else { idx = 64 + ctz64(d1); d1 &= d1 - 1; }
                    seed_consider(r_end + idx, wave_equal_len(r_end + idx, qp, P.msl), ref_pred, sp, sl);
                }
                arbitrate(P, R.len, lit + l, ap, al, sp, sl);
                bp = sp; bl = sl;
            }
            if (__builtin_expect(bl >= P.msl, 1)) { ev_lane = l; bpos = bp; blen = bl; return true; }
        }
        return false;
    }
    // ---- anchor queue (tag-word instantiation) ------------------------------------------------------------
    // Every query position whose mal-mer occurs in the reference (a tag-word hit; the tag is exact, so the
    // k-mers are equal and the step has an anchor of length >= mal) is a CANDIDATE.  Candidates are found ahead
    // of the scan, state-free, 64 positions per instruction, compacted in order through LDS and resolved in ONE
    // lane-parallel pass (lane k resolves candidate k: the bucket entry carrying the tag gives the reference
    // position, a 32-symbol word compare the match length).  The sequential scan then jumps from candidate to
    // candidate in lost mode and runs a round only over the <= mqd+1 tracking steps after an event.  Replaces the
    // rounds of 64 speculative steps with one wave-wide verification per candidate (find_event_round), which stay
    // for the other index forms.  Host model: tests/model/queue_wave.h.
    // inclusive prefix sum over the 64 lanes: row shifts inside the rows of 16, then the row broadcasts (gfx9 DPP)
    static __device__ __forceinline__ int wave_incl_scan(int v)
    {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
        return v;
    }
    __device__ __forceinline__ void lds_order() const
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void refill(int from)
    {
#ifdef LZANI_CHAIN_STATS
        st[7] += 1;
#endif
#ifdef LZANI_STAMPS
        const int stamp_sv = cur;
        stamp(2);
#endif
#ifdef LZANI_PHASE_TIME
        const unsigned long long pt_r0 = pt_now();
#endif
        LZ_PS(15);
        u32* const cq = bitmap;                                 // (all zero again when refill returns)
        const int tb = I.kb - I.dirbits;
        scan_pos = imax(scan_pos, from);
        int ncand = 0;
        // Four chunks of 64 positions per turn: their k-mer words are requested together, then their tag words (two
        // memory round trips per 256 positions instead of eight), then the candidates are compacted chunk by chunk.
        // A turn may look beyond iend or find more than the queue takes: the surplus is masked / detected again.
        if (JOIN && scan_pos < iend) {
            // Bitmap form: ONE 64-bit word of the pair's candidate bitmap per lane = 4,096 query positions a refill, all
            // of it lane-parallel (the wave-uniform chunk loop this replaces cost ~20 scalar instructions per 64
            // positions, and the kernel is bound by the scalar unit): popcount, a prefix sum over the lanes (DPP), then
            // every lane peels its own candidates into the compaction buffer, lowest bit first -- as many turns as the
            // fullest word has candidates (~5 for chance anchors).  Ranks 0 .. AQ_CAP go out (the last one only to tell
            // the next refill where to start).
            // Two such blocks are requested together; the second one is peeled only if the first leaves room in the queue
            // (sparse candidates: long k-mers -- a queue then covers twice the positions, and a refill is a chain of
            // dependent memory round trips the wave sits out).
            const u32 w0 = (u32)scan_pos >> 6;
            const int p_lo = (int)((w0 + (u32)lane) << 6), p_hi = p_lo + 4096;  // first position of this lane's words
            unsigned long long x = cand_bits[p_lo < iend ? w0 + (u32)lane : w0];
            unsigned long long x2 = cand_bits[p_hi < iend ? w0 + 64u + (u32)lane : w0];
            x &= ~lowmask(scan_pos - p_lo) & lowmask(iend - p_lo);          // positions in [scan_pos, iend)
            x2 = p_hi < iend ? x2 & lowmask(iend - p_hi) : 0ULL;
            int blocks = 1, base = 0;
            for (;;) {
                const int cnt = popc64(x);
                const int incl = wave_incl_scan(cnt);
                int at = base + incl - cnt;
                ncand = base + __builtin_amdgcn_readlane(incl, 63);
                int p0 = blocks == 1 ? p_lo : p_hi;
                while (wballot((x != 0) & (at <= AQ_CAP)) != 0) {
                    const bool has = (x != 0) & (at <= AQ_CAP);
                    const u32 lo = (u32)x, hi = (u32)(x >> 32);
                    const int b = lo ? (int)__builtin_ctz(lo) : 32 + (int)__builtin_ctz(hi | 0x80000000u);
                    cq[has ? at : AQ_LDS_CAND - 1] = (u32)(p0 + b);
                    x &= x - 1;
                    at += 1;
                }
                if (blocks == 2 || ncand >= AQ_CAP || (int)((w0 + 64u) << 6) >= iend) break;      // wave-uniform
                blocks = 2; base = ncand; x = x2;
            }
            scan_pos = imin((int)((w0 + 64u * (u32)blocks) << 6), iend);    // (more than AQ_CAP candidates: corrected below)
        }
        for (int turn = 0; !JOIN && turn < AQ_MAXCHUNKS / 4 && scan_pos < iend && ncand < AQ_CAP; ++turn) {
            u32 hq[4], w[4];
            bool valid[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) hq[c] = qkL[(u32)imin(scan_pos + 64 * c + lane, iend + 63)];     // (the arrays are padded by 128 entries)
            asm volatile("" : "+v"(hq[0]), "+v"(hq[1]), "+v"(hq[2]), "+v"(hq[3]));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                valid[c] = (scan_pos + 64 * c + lane < iend) & (hq[c] != KM_INVALID);
                // presence filter (LDS): no such k-mer in the reference = no probe
                if (LFLT) valid[c] &= (bool)((flt[(hq[c] & fmask) >> 5] >> (hq[c] & 31u)) & 1u);
                w[c] = I.tw[valid[c] ? hq[c] >> tb : 0u];
            }
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
#ifdef LZANI_PROBE2X                      // diagnostic build: every tag-word probe twice (a second, unrelated line)
            {
                u32 d[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = I.tw[valid[c] ? ((hq[c] * 0x9E3779B1u) >> (32 - I.dirbits)) : 0u];
                asm volatile("" :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
            }
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (scan_pos >= iend || ncand >= AQ_CAP) break;               // wave-uniform
                const int p = scan_pos + lane;
                const u32 x = w[c] ^ rep4(0x80u | (hq[c] & I.tagmask));   // a zero byte = a slot with this tag
                const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;                     // its lowest flag is exact
                const bool ovf = w[c] == TW_OVERFLOW;
                const bool cnd = valid[c] & ((z != 0) | ovf);
                const u64 bal = wballot(cnd);
                {
                    // no lane-dependent branch (exec-mask bookkeeping is scalar work): a lane without a candidate writes
                    // the spare last entry (ncand < AQ_CAP at this point, so at <= AQ_CAP - 1 + 63 < AQ_LDS_CAND - 1)
                    const int at = cnd ? ncand + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u)) : AQ_LDS_CAND - 1;
                    const bool cplx = ovf | ((z & (z - 1)) != 0);                       // bucket overflow, or the tag in two slots
                    cq[at] = (u32)p;
                    cq[AQ_LDS_CAND + at] = cplx ? (u32)AQ_COMPLEX : 4u * (hq[c] >> tb) + ((u32)__builtin_ctz(z | 0x80000000u) >> 3);
                    ncand += popc64(bal);
                }
                scan_pos = imin(scan_pos + 64, iend);
            }
        }
        lds_order();
        q_head = 0;
        q_cnt = imin(ncand, AQ_CAP);
        if (__builtin_expect(ncand > AQ_CAP, 0))       // the surplus candidates are detected again by the next refill
            scan_pos = (int)__builtin_amdgcn_readfirstlane(cq[AQ_CAP]);
        // resolve: lane k owns candidate k
        const bool live = lane < q_cnt;
        const int qp = live ? (int)cq[lane] : 0;
        u32 slot = live ? cq[AQ_LDS_CAND + lane] : (u32)AQ_COMPLEX;
        lds_order();
#pragma unroll
        for (int k = 0; k < 2 * AQ_LDS_CAND; k += 64) cq[k + lane] = 0;     // the seed bitmap is all zero between rounds
        lds_order();
        bool simple, dead = false;
        int pos;
        if (JOIN) {
            // bitmap form: the bitmap says where, not which bucket slot -- the candidate's bucket (its first four entries,
            // 16 bytes) is read whole and the entry carrying the tag picked from it: one dependent load instead of tag word
            // + entry, and no tag-word table needed (tags of any width: mid-size genomes with long k-mers)
            const u32 hq = qkL[(u32)qp];
            const bool ok = live & (hq != KM_INVALID);
            const uint4 bkv = reinterpret_cast<const uint4*>(I.bk)[ok ? hq >> tb : 0u];
            const u32 tag = hq & I.tagmask;                  // (BK_EMPTY / BK_OVERFLOW never carry a real tag: tag + position bits <= 30)
            const bool m0 = (bkv.x >> I.posbits) == tag, m1 = (bkv.y >> I.posbits) == tag, m2 = (bkv.z >> I.posbits) == tag,
                       m3 = (bkv.w >> I.posbits) == tag;
            const int cnt = (int)m0 + (int)m1 + (int)m2 + (int)m3;
            const bool ovf = bkv.w == BK_OVERFLOW;
            simple = ok & !ovf & (cnt == 1);                 // the tag in two slots, an overflowing bucket: by the wave, when the scan gets there
            dead = !ok | (!ovf & (cnt == 0));                // (no such k-mer in the reference after all: never an event)
            const u32 en = m0 ? bkv.x : m1 ? bkv.y : m2 ? bkv.z : bkv.w;
            pos = simple ? (int)(en & (u32)lowmask(I.posbits)) : 0;
        } else {
            simple = !(slot & AQ_COMPLEX);
            // (a lane without a simple candidate must not wander: slot 0 may hold BK_EMPTY, whose position bits point
            // far beyond the text)
            pos = simple ? (int)(I.bk[simple ? slot : 0u] & (u32)lowmask(I.posbits)) : 0;
        }
        // 32 symbols of both texts from pos / qp on, as two 32-bit words each (funnel of three dwords by v_alignbit, no branch,
        // no 64-bit shift)
        const u32* const pr = reinterpret_cast<const u32*>(R.t2) + ((u32)pos >> 4);
        const u32* const pq = reinterpret_cast<const u32*>(Q.t2) + ((u32)qp >> 4);
        const u32 sr = ((u32)pos & 15u) * 2u, sq = ((u32)qp & 15u) * 2u;
        const u32 r0 = pr[0], r1 = pr[1], r2 = pr[2], q0 = pq[0], q1 = pq[1], q2 = pq[2];
        const u32 dlo = __builtin_amdgcn_alignbit(r1, r0, sr) ^ __builtin_amdgcn_alignbit(q1, q0, sq);
        const u32 dhi = __builtin_amdgcn_alignbit(r2, r1, sr) ^ __builtin_amdgcn_alignbit(q2, q1, sq);
        const u32 mlo = (dlo | (dlo >> 1)) & 0x55555555u, mhi = (dhi | (dhi >> 1)) & 0x55555555u;
        int same = mlo ? ((int)__builtin_ctz(mlo) >> 1) : mhi ? 16 + ((int)__builtin_ctz(mhi) >> 1) : (int)AQ_LANE_CAP;
        int bound;
        if (R.nfree && Q.nfree) bound = imin(run_end(R, pos) - pos, run_end(Q, qp) - qp);
        else {
            bound = imin(R.len - pos, Q.len - qp);
            const u32 nn = (u32)(winN(R.nm, pos) | winN(Q.nm, qp));
            same = imin(same, nn ? (int)__builtin_ctz(nn) : AQ_LANE_CAP);
        }
        const bool lng = (same == AQ_LANE_CAP) & (bound > AQ_LANE_CAP);
        a_pos = live ? qp : (int)AQ_NONE;
        a_ref = simple ? ((u32)pos | (lng ? (u32)AQ_LONG : 0u)) : (u32)AQ_COMPLEX;
        // A PLAIN candidate -- resolved by the lane, an anchor (length >= mal), long enough to be an event (>= msl)
        // and not at reference position 0 (quirk Q1) -- is an event wherever the scan meets it unopposed: it
        // carries its length as it is, every other candidate as -1 - length (the jump path tests one sign).
        const int al0 = simple ? imin(same, bound) : 0;
        const bool plain = simple & !lng & (al0 >= P.mal) & (al0 >= P.msl) & (pos != 0);
        a_len = plain ? al0 : -1 - al0;
        // what a distant event at this candidate needs to see that neither extension moves, worked out now, by the
        // lane, for the whole batch at once
        a_ext = (simple & !lng) ? null_ext_record(P, R, Q, qp, pos, al0) : ext_rec_none(P.aw);
        // A candidate the lane has resolved to a match shorter than mal is no anchor and never an event -- the tag of
        // another k-mer in the bucket, about as many per pair as there are true anchors: it leaves the queue here, all
        // of a batch in one stable compaction (lane permute, no LDS memory), instead of costing the sequential scan a
        // step each.  (The lanes behind the last kept candidate receive leftovers; their position says AQ_NONE.)
        const bool kept = live & !dead & !(simple & !lng & (al0 < P.mal));
        const u64 keep = wballot(kept);
        if (keep != lowmask(q_cnt)) {
            const int to = 4 * (kept ? (int)__builtin_amdgcn_mbcnt_hi((u32)(keep >> 32), __builtin_amdgcn_mbcnt_lo((u32)keep, 0u)) : 63);
            q_cnt = popc64(keep);
            const int mp = __builtin_amdgcn_ds_permute(to, a_pos), ml = __builtin_amdgcn_ds_permute(to, a_len);
            const int mr = __builtin_amdgcn_ds_permute(to, (int)a_ref), mx = __builtin_amdgcn_ds_permute(to, (int)a_ext);
            a_pos = lane < q_cnt ? mp : (int)AQ_NONE;
            a_len = ml; a_ref = (u32)mr; a_ext = (u32)mx;
        }
        if constexpr (CHAIN) chain_classes();
#ifdef LZANI_PHASE_TIME
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        pt_refill += pt_now() - pt_r0;
#endif
#ifdef LZANI_STAMPS
        stamp(stamp_sv);
#endif
    }
    // the length word of queue entry k (wave-uniform k): with the null chain its low byte, the rest is chain_classes'
    __device__ __forceinline__ int len_at(int k) const
    {
        const int v = __builtin_amdgcn_readlane(a_len, k);
        return CHAIN ? (int)(signed char)(v & 0xFF) : v;
    }
    // What the null chain (below) may take for granted about candidate j, worked out for the whole batch at once by the
    // lanes -- the scalar unit is what bounds this kernel, and this is ~25 of its instructions per event: IF j has been
    // committed as a null event by its record (backward extension from the record, i.e. the machine could look back
    // at least aw symbols), THEN bits 8..14 = the next candidate the scan meets (the first one at or behind the end of
    // j's match and forward extension) and bit 15 (GO) = that candidate is the next null event, for certain: j's
    // tracking round is complete (query and reference ends far enough), the region j opened is short (dropped), the
    // successor is plain, distant from j, its record holds both extensions and it sits at least aw symbols into both
    // texts (that last property of a candidate alone is bit 16: the general turn's short cut).  (Then its reach is >= aw as well: what it may look back at grows from event to event while regions are
    // dropped.)  Only "no seed candidate in j's tracking round" is left to the loop.  The parameters are the defaults.
    __device__ __forceinline__ void chain_classes()
    {
        typedef ChainP<CHAIN> CP;
        enum { MQD = CP::MQD, MRD = CP::MRD, MSL = CP::MSL, REG = CP::REG, AW = CP::AW, NT = CP::NT, WIN = CP::WIN };
        // (N-free pair: the rounds of the fast turns -- the ones a GO bit lets through -- only ever see real msl-mers: all 64
        // query lanes, all 80 window positions inside one strand; their address arithmetic then needs no clamp, LZ_NC_WORD7N)
        constexpr bool nf = CP::NF;                 // (a kernel for genomes without N)
        int ilim = imin(imin(scan_pos, iend) - NT, nf ? Q.L - MSL + 1 - 64 : iend);
        const int rlim = R.len - MSL + 1 - WIN;
        if constexpr (SPLITW) ilim = imin(ilim, stop_i - (int)SPLIT_MARGIN);
        const int len = a_len;
        const bool plain = len > 0;
        const u32 rec = a_ext;
        const int pos = (int)(a_ref & AQ_POS);
        const int t2 = len + (int)((rec >> 24) & 31u);
        const int end = a_pos + t2, rend = pos + t2;
        const bool both = (rec & (EXT_REC_FWDK | EXT_REC_BRKB)) == (u32)(EXT_REC_FWDK | EXT_REC_BRKB);
        const bool cap = plain & both & (imin(a_pos, pos) >= AW);                      // as a successor
        const bool rwin = !nf | (rend + WIN <= R.L - MSL + 1) | ((rend >= R.rc0) & (rend + WIN <= R.rc0 + R.L - MSL + 1));
        const bool pred = plain & both & (end <= ilim) & (rend <= rlim) & rwin & (t2 + (int)(rec & 15u) < REG);   // as a predecessor
        const int mine = (int)((u32)pos | (cap ? 0x80000000u : 0u));
        int succ = 64, s_pos = 0, s_u = 0;
#pragma unroll
        for (int d = 3; d >= 1; --d) {               // (positions ascend: the smallest d that qualifies is written last)
            const int l = lane + d;
            const int p = __builtin_amdgcn_ds_bpermute(4 * l, a_pos), u = __builtin_amdgcn_ds_bpermute(4 * l, mine);
            const bool ok = (l < 64) & (p >= end);
            succ = ok ? l : succ; s_pos = ok ? p : s_pos; s_u = ok ? u : s_u;
        }
        const int gap = s_pos - end;
        const bool distant = (gap > MQD) | (iabs((s_u & 0x7FFFFFFF) - (rend + gap)) > MRD);
        bool go = pred & (succ < q_cnt) & (s_u < 0) & distant;
        if constexpr (SPLITW) go &= s_pos <= ilim;              // (a fast turn commits the successor: it, too, below the limit)
        a_len = (len & 0xFF) | (succ << 8) | (go ? 0x8000 : 0) | (cap ? 0x10000 : 0) | ((t2 & 0xFF) << 17);      // (t2 <= 127 + 31)
    }
    __device__ __forceinline__ bool ext_record(u32& x) const
    {
        if (last_src < 0) return false;
        x = (u32)__builtin_amdgcn_readlane((int)a_ext, last_src);
        return true;
    }
    // the record of entry k describes the event (bpos, blen) iff the entry is simple and short, an anchor, and the
    // event is that anchor
    __device__ __forceinline__ void note_src(int k, int ap, int al, int bpos, int blen)
    {
        const u32 ref = (u32)__builtin_amdgcn_readlane((int)a_ref, k);
        last_src = (!(ref & (AQ_COMPLEX | AQ_LONG)) && al >= P.mal && bpos == ap && blen == al) ? k : -1;
    }
    // the queue without the candidates before query position pos
    __device__ __forceinline__ void drop_before(int pos)
    {
        const u64 m = wballot((lane >= q_head) & (a_pos >= pos));          // lanes beyond q_cnt hold AQ_NONE
        q_head = m ? ctz64(m) : 64;
    }
    // best_anchor of the queued step k (wave-uniform k), exactly
    __device__ __forceinline__ void anchor_of(int k, int qp, int& ap, int& al) const
    {
        const u32 ref = (u32)__builtin_amdgcn_readlane((int)a_ref, k);
        ap = 0; al = 0;
        if (__builtin_expect((ref & AQ_COMPLEX) != 0, 0)) {
            anchor_by_wave((u32)__builtin_amdgcn_readfirstlane((int)qkL[(u32)qp]), qp, ap, al);
            return;
        }
        ap = (int)(ref & AQ_POS);
        al = len_at(k);
        al = al < 0 ? -1 - al : al;
        if (__builtin_expect((ref & AQ_LONG) != 0, 0)) al = wave_equal_len(ap, qp, AQ_LANE_CAP);
        // a candidate is a k-mer hit in the genome's REFERENCE text; as a query the text ends at D, and with
        // mrd < mal - msl a step near the end holds a k-mer that runs past it: shorter than mal = no anchor
        if (__builtin_expect(al < P.mal, 0)) { ap = 0; al = 0; }
    }

    // which of the steps [i, i + nt), nt <= 64, are candidates (have an anchor): the detect of refill for one chunk, nothing
    // resolved.  For tracking rounds the queue does not cover -- the scan has jumped over it, i.e. an extension moved: a
    // related stretch, where every position is a candidate and resolving 64 of them per event would be wasted.
    // (probe form: the steps [from, nt) only -- a lane outside makes no request, and every probe is a 128-byte line)
    __device__ __forceinline__ u64 detect_steps(int i, int nt, int from = 0) const
    {
        if (JOIN) {
            const int sh = i & 63;
            const unsigned long long wd = cand_bits[((u32)i >> 6) + (u32)imin(lane, 1)];
            return ((bcast64(wd, 0) >> sh) | ((bcast64(wd, 1) << 1) << (63 - sh))) & lowmask(nt) & ~lowmask(from);
        }
        const int tb = I.kb - I.dirbits;
        const u32 hq = qkL[(u32)(i + lane)];
        const bool valid = (lane >= from) & (lane < nt) & (hq != KM_INVALID);
        const u32 w = I.tw[valid ? hq >> tb : 0u];
        const u32 x = w ^ rep4(0x80u | (hq & I.tagmask));
        const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;
        return wballot(valid & ((z != 0) | (w == TW_OVERFLOW)));
    }

    // the close-seed probe of the tracking steps [i, i + nt): which steps have a candidate in their window
    __device__ __forceinline__ u64 track_round(int i, int nt, int r_end, int lit, u32& rk0, u32& rk1, u32& qk) const
    {
        rk0 = rk1 = qk = KM_INVALID;
        const int W = imin(lit + nt - 1 + P.mrd, R.len - P.msl + 1 - r_end);
        if (W <= 0) return 0;
        const int w0 = imin(lane, W - 1), w1 = imin(lane + 64, W - 1);
#ifdef LZANI_PRIO
        __builtin_amdgcn_s_setprio(LZANI_PRIO);
#endif
        qk = qkS[(u32)(i + lane)];
        rk0 = rkS[(u32)(r_end + w0)];
        rk1 = rkS[(u32)(r_end + w1)];
#ifdef LZANI_PRIO
        asm volatile("" : "+v"(qk), "+v"(rk0), "+v"(rk1));
        __builtin_amdgcn_s_setprio(0);
#endif
        qk = lane < nt ? qk : KM_INVALID;
        rk0 = lane < W ? rk0 : KM_INVALID;
        rk1 = lane + 64 < W ? rk1 : KM_INVALID;
        return wballot(seed_prefilter(rk0, rk1, qk));
    }


    // a wave-uniform pointer the compiler keeps in vector registers, as a scalar pair (the base of a global_load)
    static __device__ __forceinline__ const u32* uniform_ptr(const u32* p)
    {
        const u64 a = (u64)p;
        const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(a >> 32));
        return (const u32*)(((u64)hi << 32) | lo);
    }
    // The null chain (CHAIN = default LZ parameters, no alignment output): the cycle an unrelated pair spends four
    // fifths of its events in -- tracking round over the 41 steps behind a match, no seed candidate, the next queued
    // candidate is plain and distant, the region behind it is short and dropped, its record proves both extensions
    // empty -- as ONE hand-scheduled loop in GCN, scalar state in fixed registers: ~45 scalar instructions per event
    // against the ~140 the compiler spends on the same steps spread over find_event and PairMachine::run (phi copies
    // and exit flags at every merge point).  Semantics: exactly find_event's common call followed
    // by run()'s null event, repeated; it stops, with nothing half-done, at the first thing that is not that cycle:
    //   returns 0   nothing in hand (queue empty or not covering the tracking steps, query or reference end near)
    //   returns 1   the tracking round of step i is done: pre_seed / pre_rk0 / pre_rk1 / pre_qk, for find_event
    //   returns 2   the next event is found but is not a null event: adv, bpos, blen (consumed from the queue)
    //   returns 4   the tracking round of step i found a seed candidate and the event is simple (see Lnc_seedev): adv,
    //               bpos, blen are the close match's, nothing consumed from the queue
    // i, r_end, prev_rs, prev_re, pre_lit are the machine's; last_cl != 0 = events were committed, the open region is the
    // last one's match and forward extension (cl = last_cl, clit = last_clit, nl = 0).  A null event over a region that is
    // KEPT (its query span reached reg: seed events grew it) or over none is committed as well: the kept region --
    // (open_cl, open_clit) at the call, the last committed event's afterwards -- is closed the way calc_stats closes it
    // at a match_distant factor, into add_tm / add_tl / add_tc (the machine adds them to its totals), and the candidate
    // looks back over the literals since the last match only.  Wait states of gfx950 are placed by hand
    // (VALU-written mask -> VALU use: 2; lane select / VMEM base written by a VALU: the prologue is long enough).
    // The chain has nothing to work with where the scan stands beyond everything detected so far (an extension has moved
    // over the queue: a related stretch, find_event's light rounds): it would leave at once, nothing touched.
    __device__ __forceinline__ bool chain_covers(int i) const { return scan_pos >= i; }
    __device__ __forceinline__ int null_chain(int& i, int& r_end, int& prev_rs, int& prev_re, int& pre_lit, int& last_cl,
                                              int& last_clit, int& adv, int& bpos, int& blen, int open_cl, int open_clit,
                                              int& add_tm, int& add_tl, int& add_tc)
    {
        static_assert(!CHAIN || (FAST && BK), "the null chain reads the anchor queue");
        typedef ChainP<CHAIN> CP;                              // params.h:34-48, or the set this kernel was compiled for
        enum { MQD = CP::MQD, MRD = CP::MRD, MSL = CP::MSL, REG = CP::REG, AW = CP::AW, AM = CP::AM, AR = CP::AR, NT = CP::NT, WIN = CP::WIN };
        static_assert(!CHAIN || chain_params_ok(CP::P), "the null chain is not written for these parameters");
        const int ilim_q = imin(scan_pos, iend) - NT;           // the queue and the query cover the tracking steps of i <= ilim
        const int ilim = SPLITW ? imin(ilim_q, stop_i - (int)SPLIT_MARGIN) : ilim_q;
        const int rlim = R.len - MSL + 1 - WIN;                 // the seed window of r_end <= rlim is complete
        const u32 ldsb = (u32)(size_t)bitmap;                   // LDS byte offset (low half of the flat address)
        const u32 zero = 0, one = 1;
        const u32* const qks = uniform_ptr(qkS);
        const u32* const rks = uniform_ptr(rkS);
        int code, ap, rec, t0, t1, t2, kb, kc, qh = q_head;
        u64 m, seed;
#if defined(LZANI_STAMPS) || defined(LZANI_PATH_STATS)    // (diagnostic builds: the bookkeeping makes the compiler lose sight of the uniformity)
        constexpr bool unify = true;
#else
        constexpr bool unify = SPLITW;                    // (a segment's state comes out of memory)
#endif
        int qc_u = q_cnt, ilim_u = ilim, rlim_u = rlim;
        [[maybe_unused]] const int tnt_u = SPLITW ? __builtin_amdgcn_readfirstlane((int)split_taint) : 0;
        if constexpr (unify) {
            i = __builtin_amdgcn_readfirstlane(i); r_end = __builtin_amdgcn_readfirstlane(r_end); qh = __builtin_amdgcn_readfirstlane(qh);
            prev_rs = __builtin_amdgcn_readfirstlane(prev_rs); prev_re = __builtin_amdgcn_readfirstlane(prev_re);
            pre_lit = __builtin_amdgcn_readfirstlane(pre_lit);
            qc_u = __builtin_amdgcn_readfirstlane(q_cnt); ilim_u = __builtin_amdgcn_readfirstlane(ilim); rlim_u = __builtin_amdgcn_readfirstlane(rlim);
        }
        u32 rk0, rk1, qk, a0, a1, aq, t, bq, w1, dumv, qkb, rk0b, rk1b;
#ifdef LZANI_CHAIN_STATS
        int ncnt = 0;
#define LZ_NC_COUNT "s_add_i32 %[ncnt], %[ncnt], 1\n\t"
        int why = 0;                                // which test of the loop's seed event handed the round back (exit_seed)
#define LZ_NC_COUNT_OPERAND [ncnt] "+s"(ncnt), [why] "+s"(why),
#define LZ_NC_WHY(n) "s_mov_b32 %[why], " #n "\n\t"
#else
#define LZ_NC_COUNT
#define LZ_NC_COUNT_OPERAND
#define LZ_NC_WHY(n)
#endif
#if defined(LZANI_EXP) && LZANI_EXP == 4           // diagnostic build (wrong results): seed candidates ignored inside the chain
#define LZ_NC_NOSEED "s_cmp_lg_u32 0, 0\n\t"
#else
#define LZ_NC_NOSEED
#endif
        // the three loads of a tracking round (msl-mers of the 41 steps and of the 80 window positions).  The lanes beyond
        // (steps >= NT, window positions >= WIN) are not masked: their steps are cut from the result (LZ_NC_SEEDS), their
        // window positions are positions of the first load again; what leaves the loop with the round in hand is masked then (LZ_NC_FIX)
#define LZ_NC_LOADS_X(I, R, QK, K0, K1) \
            "v_add_lshl_u32 %[a0], %[lane], %[" I "], 2\n\t" \
            "v_add_lshl_u32 %[a1], %[lane], %[" R "], 2\n\t" \
            "v_add_lshl_u32 %[aq], %[w1], %[" R "], 2\n\t" \
            "global_load_dword %[" QK "], %[a0], %[qks]\n\t" \
            "global_load_dword %[" K0 "], %[a1], %[rks]\n\t" \
            "global_load_dword %[" K1 "], %[aq], %[rks]\n\t"
#define LZ_NC_LOADS_F LZ_NC_LOADS_X("i", "rend", "qk", "rk0", "rk1")
        // the round itself (track_round + seed_prefilter): window k-mers into the LDS bitmap, every step tests its own, the
        // bits are cleared again; leaves the steps with a seed candidate in seed.  A lane without a k-mer (KM_INVALID >> 5 is
        // beyond every word of the bitmap) works on its own word behind the bitmap (SEED_PAD) by a minimum: no lane select,
        // no mask (a step without a k-mer may see its own window position's bit there: LZ_NC_FIX)
        // (word and bit of a k-mer word in the bitmap: msl 7 = the msl-mer itself, 14 bits; msl 9 = the 14 hash bits k_kmers packs
        // above the msl-mer, where KM_INVALID falls on a bit no msl-mer's hash takes)
#define LZ_NC_WORD7(A, T, K) \
            "v_lshrrev_b32_e32 %[" A "], 5, %[" K "]\n\t" \
            "v_lshlrev_b32_e32 %[" T "], %[" K "], %[one]\n\t" \
            "v_min_u32_e32 %[" A "], %[dumv], %[" A "]\n\t" \
            "v_lshl_add_u32 %[" A "], %[" A "], 2, %[ldsb]\n\t"
#define LZ_NC_WORD7N(A, T, K) \
            "v_lshrrev_b32_e32 %[" A "], 5, %[" K "]\n\t" \
            "v_lshlrev_b32_e32 %[" T "], %[" K "], %[one]\n\t" \
            "v_lshl_add_u32 %[" A "], %[" A "], 2, %[ldsb]\n\t"
#define LZ_NC_WORD9(A, T, K) \
            "v_lshrrev_b32_e32 %[" A "], 23, %[" K "]\n\t" \
            "v_lshrrev_b32_e32 %[" T "], 18, %[" K "]\n\t" \
            "v_lshlrev_b32_e32 %[" T "], %[" T "], %[one]\n\t" \
            "v_lshl_add_u32 %[" A "], %[" A "], 2, %[ldsb]\n\t"
#define LZ_NC_WORD8(A, T, K)                /* (msl 8: the hash sits at bit 16, and KM_INVALID must stay inside the bitmap) */ \
            "v_bfe_u32 %[" A "], %[" K "], %[KS5], 9\n\t" \
            "v_lshrrev_b32_e32 %[" T "], %[KS], %[" K "]\n\t" \
            "v_lshlrev_b32_e32 %[" T "], %[" T "], %[one]\n\t" \
            "v_lshl_add_u32 %[" A "], %[" A "], 2, %[ldsb]\n\t"
#define LZ_NC_ROUND_X(WORD, WAIT, QK, K0, K1) \
            WAIT "\n\t" \
            WORD("a0", "t", K0) \
            "ds_or_b32 %[a0], %[t]\n\t" \
            WORD("a1", "bq", K1) \
            "ds_or_b32 %[a1], %[bq]\n\t" \
            WORD("aq", "t", QK) \
            "ds_read_b32 %[aq], %[aq]\n\t" \
            "ds_write_b32 %[a0], %[zero]\n\t" \
            "ds_write_b32 %[a1], %[zero]\n\t" \
            "s_waitcnt lgkmcnt(2)\n\t" \
            "v_and_b32_e32 %[aq], %[aq], %[t]\n\t" \
            "v_cmp_ne_u32_e64 %[seed], 0, %[aq]\n\t"
#define LZ_NC_ROUND_F(WORD) LZ_NC_ROUND_X(WORD, "s_waitcnt vmcnt(0)", "qk", "rk0", "rk1")
        // one fast turn (see Lnc_fast)
#define LZ_NC_FTURN(WORDF, P, N, IC, IN, RN, CQ, C0, C1, NQ, N0, N1, SFX) \
            "Lnc_fturn" SFX "_%=:\n\t" \
            "s_bitcmp0_b32 %[" P "], 15\n\t" \
            "s_cbranch_scc1 Lnc_fnogo" SFX "_%=\n\t" \
            "s_bfe_u32 %[qh], %[" P "], 0x70008\n\t"        /* the successor (the entries between are passed) */ \
            "v_readlane_b32 %[" N "], %[alen], %[qh]\n\t"   /* (its word: bits 17..24 = the match and its forward extension) */ \
            "v_readlane_b32 %[ap], %[apos], %[qh]\n\t" \
            "v_readlane_b32 %[bpos], %[aref], %[qh]\n\t" \
            "s_bfe_u32 %[t0], %[" N "], 0x80011\n\t" \
            "s_sub_i32 %[gap], %[ap], %[" IC "]\n\t" \
            "s_add_i32 %[" IN "], %[ap], %[t0]\n\t" \
            "s_add_i32 %[" RN "], %[bpos], %[t0]\n\t" \
            LZ_NC_LOADS_X(IN, RN, NQ, N0, N1) \
            LZ_NC_ROUND_X(WORDF, "s_waitcnt vmcnt(3)", CQ, C0, C1) \
            LZ_NC_SEEDS \
            "s_cbranch_scc1 Lnc_frec" SFX "_%=\n\t"         /* a seed candidate: the state first, then the seed event */ \
            "s_mov_b32 %[t2], %[qh]\n\t" \
            LZ_NC_COUNT
        // the lane masks a round went without, for what leaves the loop with the round in hand (find_event, the seed event)
#define LZ_NC_FIX \
            "s_bfm_b64 %[m2], %[NT], 0\n\t" \
            "s_and_b64 %[seed], %[seed], %[m2]\n\t" \
            "v_cmp_gt_u32_e64 %[m2], %[NR1], %[lane]\n\t" \
            "v_cmp_gt_u32_e32 vcc, %[NT], %[lane]\n\t" \
            "s_nop 1\n\t" \
            "v_cndmask_b32_e64 %[rk1], -1, %[rk1], %[m2]\n\t" \
            "v_cndmask_b32_e32 %[qk], -1, %[qk], vcc\n\t" \
            "v_cmp_ne_u32_e32 vcc, -1, %[qk]\n\t"          /* (a step without a k-mer has read its own word behind the bitmap) */ \
            "s_and_b64 %[seed], %[seed], vcc\n\t" \
            "s_and_b64 %[m], %[m], vcc\n"
        // a seed candidate matters only up to the step of the queued candidate itself (gap steps ahead): m = the steps
        // with a seed candidate among them (scc = any); seed keeps them all for find_event
#define LZ_NC_SEEDS \
            "s_add_i32 %[t0], %[gap], 1\n\t" \
            "s_min_u32 %[t0], %[t0], %[NT]\n\t" \
            "s_bfm_b64 %[m], %[t0], 0\n\t" \
            "s_and_b64 %[m], %[m], %[seed]\n\t" LZ_NC_NOSEED
        // the null event: the machine's state after it (see PairMachine::run)
#define LZ_NC_COMMIT \
            "s_sub_i32 %[plit], %[t1], %[kb]\n\t"           /* what is left of the literals before the backward extension */ \
            "s_sub_i32 %[prs], %[ap], %[kb]\n\t"            /* the region starts with it */ \
            "s_bfe_u32 %[t0], %[rec], 0x50018\n\t"          /* e: the forward extension's length (bits 24..28) */ \
            "s_bfe_u32 %[lastlit], %[rec], 0x40014\n\t"     /* its mismatches (bits 20..23) */ \
            "s_add_i32 %[t2], %[blen], %[t0]\n\t" \
            "s_sub_i32 %[lastb], %[t2], %[lastlit]\n\t"     /* the region's matches: the anchor, the extensions' */ \
            "s_add_i32 %[lastb], %[lastb], %[kc]\n\t" \
            "s_add_i32 %[lastlit], %[lastlit], %[kb]\n\t"   /* and its literals: the extensions' mismatches */ \
            "s_sub_i32 %[lastlit], %[lastlit], %[kc]\n\t" \
            "s_add_i32 %[i], %[ap], %[t2]\n\t" \
            "s_add_i32 %[rend], %[bpos], %[t2]\n\t" \
            "s_mov_b32 %[pre], %[i]\n\t" \
            "s_add_i32 %[qh], %[qh], 1\n\t" \
            LZ_NC_COUNT
        int gap, cls, fok;
        u64 m2;
        // the packed texts as 32-bit words (16 symbols each) and the end of the scan -- far below zero for a pair with an N in
        // it, which the loop's own seed event does not take (its lanes carry no bounds)
        const u32* const rt2 = uniform_ptr(reinterpret_cast<const u32*>(R.t2));
        const u32* const qt2 = uniform_ptr(reinterpret_cast<const u32*>(Q.t2));
        const int qend = __builtin_amdgcn_readfirstlane((R.nfree && Q.nfree) ? iend : -(1 << 30));
        // (the machine's accumulators may live in vector registers -- they come out of popcounts: as scalars for the loop)
        const int ocl_u = __builtin_amdgcn_readfirstlane(open_cl), oclit_u = __builtin_amdgcn_readfirstlane(open_clit);
#ifdef LZANI_CHAIN_PRIO                             // (experiment: the loop's waves ahead of the others in the issue arbitration)
#define LZ_NC_PRIO_ON "s_setprio 2\n\t"
#define LZ_NC_PRIO_OFF "s_setprio 0\n\t"
#else
#define LZ_NC_PRIO_ON
#define LZ_NC_PRIO_OFF
#endif
        /* (a segment of a split pair: the queued candidate itself must lie below the limit -- a lost-mode jump to it may be long) */
#define LZ_NC_SPLITCHK \
            "v_readlane_b32 %[t0], %[apos], %[qh]\n\t" \
            "s_cmp_gt_i32 %[t0], %[ilim]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"
        /* (... and while its look-back is a lower bound of the true scan's -- split_taint -- only the record's own answers stand: they
           hold for every look-back of aw symbols and more; what depends on the very reach is left to the machine) */
#define LZ_NC_SPLITGEN \
            "s_cmp_lg_u32 %[tnt], 0\n\t" \
            "s_cbranch_scc1 Lnc_chk_%=\n\t"
#define LZ_NC_SPLITIN , [tnt] "s"(tnt_u)
#define LZ_NC_ASM(WORD, WORDF) LZ_NC_ASM_X(WORD, WORDF, "", "", )
#define LZ_NC_ASM_X(WORD, WORDF, SPLCHK, SPLGEN, SPLIN) \
        asm volatile( \
            LZ_NC_PRIO_ON \
            "s_mov_b32 %[code], 0\n\t" \
            "s_mov_b32 %[lastb], 0\n\t" \
            "s_mov_b32 %[lastlit], 0\n\t" \
            "s_mov_b32 %[atm], 0\n\t" \
            "s_mov_b32 %[atl], 0\n\t" \
            "s_mov_b32 %[atc], 0\n\t" \
            "s_mov_b64 %[m2], -1\n\t"                      /* (0 = the seed candidates of the turn that follows are known to be false, Lnc_snone) */ \
            /* the window positions of a round's third load: 64 + lane up to the window's last, then the positions from 0 on */ \
            /* again (their bits are set already, and no two lanes of one LDS atomic meet on a word more often than k-mers do) */ \
            "v_add_u32_e32 %[w1], 64, %[lane]\n\t" \
            "v_subrev_u32_e32 %[dumv], %[WIN], %[w1]\n\t" \
            "v_min_u32_e32 %[w1], %[w1], %[dumv]\n\t" \
            "v_add_u32_e32 %[dumv], %[wdum], %[lane]\n\t"  /* a lane's own word behind the bitmap */ \
            "s_nop 3\n" \
            "Lnc_top_%=:\n\t" \
            /* the queue head: candidates the last match has passed go */ \
            "s_cmp_ge_i32 %[qh], %[qc]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "v_readlane_b32 %[t0], %[apos], %[qh]\n\t" \
            "s_cmp_ge_i32 %[t0], %[i]\n\t" \
            "s_cbranch_scc1 Lnc_nodrop_%=\n\t" \
            "v_cmp_le_i32_e32 vcc, %[qh], %[lane]\n\t" \
            "v_cmp_le_i32_e64 %[m], %[i], %[apos]\n\t" \
            "s_and_b64 %[m], %[m], vcc\n\t" \
            "s_ff1_i32_b64 %[qh], %[m]\n\t" \
            "s_min_u32 %[qh], %[qh], 64\n\t" \
            "s_cmp_ge_i32 %[qh], %[qc]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n" \
            "Lnc_nodrop_%=:\n\t" \
            "s_cmp_gt_i32 %[i], %[ilim]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_cmp_gt_i32 %[rend], %[rlim]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            SPLCHK \
            LZ_NC_LOADS_F \
            /* While the loads fly: the next queued candidate and, should the round find no seed candidate, whether it is */ \
            /* a null event (t2 = 1): plain, distant, the open region short (dropped), both extensions empty by the record */ \
            "v_readlane_b32 %[blen], %[alen], %[qh]\n\t" \
            "v_readlane_b32 %[ap], %[apos], %[qh]\n\t" \
            "v_readlane_b32 %[bpos], %[aref], %[qh]\n\t" \
            "v_readlane_b32 %[rec], %[aext], %[qh]\n\t" \
            "s_mov_b32 %[t2], 0\n\t" \
            "s_mov_b32 %[cls], 0\n\t"                        /* (1 = the open region is kept, or there is none; bit 15 = a fast turn, Lnc_snone) */ \
            "s_bfe_u32 %[code], %[blen], 0x10010\n\t"       /* chain_classes' bit 16: plain, both extensions in the record, aw symbols into both texts */ \
            "s_sext_i32_i8 %[blen], %[blen]\n\t"            /* (the rest of the word is chain_classes') */ \
            "s_sub_i32 %[gap], %[ap], %[i]\n\t" \
            "s_cmp_lt_i32 %[blen], 1\n\t" \
            "s_cbranch_scc1 Lnc_chk_%=\n\t" \
            "s_cmp_le_i32 %[gap], %[MQD]\n\t" \
            "s_cbranch_scc1 Lnc_close_%=\n" \
            "Lnc_distant_%=:\n\t" \
            "s_cmp_lt_i32 %[prs], 0\n\t" \
            "s_cbranch_scc1 Lnc_kept_%=\n\t" \
            "s_sub_i32 %[t1], %[pre], %[prs]\n\t" \
            "s_cmp_ge_i32 %[t1], %[REG]\n\t" \
            "s_cbranch_scc1 Lnc_kept_%=\n\t" \
            "s_sub_i32 %[t1], %[ap], %[prs]\n\t" \
            "s_add_i32 %[t1], %[t1], %[plit]\n"             /* avail: the dropped region and the literals before it */ \
            "Lnc_avail_%=:\n\t" \
            /* the short cut: the candidate's own properties are in bit 16; with at least aw symbols to look back at, its */ \
            /* reach is >= aw and the backward extension is the record's */ \
            "s_cmp_eq_u32 %[code], 0\n\t" \
            "s_cbranch_scc1 Lnc_gen_%=\n\t" \
            "s_cmp_lt_i32 %[t1], %[AW]\n\t" \
            "s_cbranch_scc1 Lnc_gen_%=\n\t" \
            "s_and_b32 %[kb], %[rec], 15\n\t" \
            "s_bfe_u32 %[kc], %[rec], 0x40004\n\t" \
            "s_mov_b32 %[fok], 1\n\t" \
            "s_branch Lnc_ok_%=\n" \
            "Lnc_gen_%=:\n\t" \
            SPLGEN \
            "s_bitcmp0_b32 %[rec], 29\n\t"                  /* the forward extension must be in the record (empty or not) */ \
            "s_cbranch_scc1 Lnc_chk_%=\n\t" \
            "s_min_i32 %[t0], %[t1], %[ap]\n\t" \
            "s_min_i32 %[t0], %[t0], %[bpos]\n\t"           /* reach */ \
            "s_mov_b32 %[kb], 0\n\t"                          /* the backward extension: empty, unless the record holds it */ \
            "s_mov_b32 %[kc], 0\n\t" \
            "s_mov_b32 %[fok], 0\n\t" \
            "s_cmp_lt_i32 %[t0], 1\n\t" \
            "s_cbranch_scc1 Lnc_ok_%=\n\t" \
            "s_bitcmp1_b32 %[rec], 30\n\t" \
            "s_cbranch_scc1 Lnc_brkb_%=\n\t" \
            "s_min_i32 %[code], %[t0], %[AW]\n\t"           /* no break inside the first aw symbols: the qual bits decide, */ \
            "s_bfm_b32 %[code], %[code], 0\n\t"             /* if the machine may not look further back than they reach */ \
            "s_and_b32 %[code], %[code], %[rec]\n\t" \
            "s_cbranch_scc1 Lnc_chk_%=\n\t" \
            "s_cmp_le_i32 %[t0], %[AW]\n\t" \
            "s_cbranch_scc1 Lnc_ok_%=\n\t" \
            "s_branch Lnc_chk_%=\n" \
            "Lnc_brkb_%=:\n\t"                               /* the scan breaks inside them: its result is in the record */ \
            "s_cmp_lt_i32 %[t0], %[AW]\n\t"                 /* (given the full first window) */ \
            "s_cbranch_scc1 Lnc_chk_%=\n\t" \
            "s_and_b32 %[kb], %[rec], 15\n\t" \
            "s_bfe_u32 %[kc], %[rec], 0x40004\n\t" \
            "s_mov_b32 %[fok], 1\n"                          /* committed by its record: what chain_classes assumes */ \
            "Lnc_ok_%=:\n\t" \
            "s_mov_b32 %[t2], 1\n" \
            "Lnc_chk_%=:\n\t" \
            LZ_NC_ROUND_F(WORD) \
            "s_mov_b32 %[code], 1\n\t" \
            LZ_NC_SEEDS \
            "s_cbranch_scc1 Lnc_sseed_%=\n"                 /* a seed candidate: the round is done; the event itself, if it is simple */ \
            "Lnc_noseed_%=:\n\t" \
            "s_cmp_lt_i32 %[blen], 1\n\t" \
            "s_cbranch_scc1 Lnc_npl_%=\n\t"                 /* the candidate is not plain: likewise */ \
            "s_mov_b32 %[code], 2\n\t" \
            "s_cmp_eq_u32 %[t2], 0\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"                 /* the event is found but is not a null event */ \
            "s_cmp_eq_u32 %[cls], 0\n\t" \
            "s_cbranch_scc1 Lnc_commit_%=\n\t" \
            /* the open region is kept (not dropped): calc_stats' step at the match_distant factor that follows -- it counts */ \
            /* if its matches and literals reach reg (parser.cpp:743-751, 775).  The open region is the last committed */ \
            /* event's if this call has committed any, else the machine's */ \
            "s_cmp_lg_u32 %[lastb], 0\n\t" \
            "s_cselect_b32 %[t0], %[lastb], %[ocl]\n\t" \
            "s_cselect_b32 %[t2], %[lastlit], %[oclit]\n\t" \
            "s_cmp_eq_u32 %[t0], 0\n\t" \
            "s_cbranch_scc1 Lnc_commit_%=\n\t" \
            "s_add_i32 %[t2], %[t2], %[t0]\n\t" \
            "s_cmp_lt_i32 %[t2], %[REG]\n\t" \
            "s_cbranch_scc1 Lnc_commit_%=\n\t" \
            "s_sub_i32 %[t2], %[t2], %[t0]\n\t" \
            "s_add_i32 %[atm], %[atm], %[t0]\n\t" \
            "s_add_i32 %[atl], %[atl], %[t2]\n\t" \
            "s_add_i32 %[atc], %[atc], 1\n" \
            "Lnc_commit_%=:\n\t" \
            LZ_NC_COMMIT \
            "s_mov_b32 %[code], 0\n\t" \
            "s_cmp_eq_u32 %[fok], 0\n\t" \
            "s_cbranch_scc1 Lnc_top_%=\n" \
            /* The fast turn: the candidate just committed (queue entry qh - 1) was a null event by its record, so */ \
            /* chain_classes' word says which entry the scan meets next and whether that is the next null event for */ \
            /* certain (GO) -- then only this entry's tracking round is left to do */ \
            "Lnc_fast_%=:\n\t" \
            /* (inside a run of fast turns the machine's state is kept as far as the turns need it: i, rend, the queue head; */ \
            /* what the general turn and the seed event need besides -- prs, plit, pre, the open region -- is a function of */ \
            /* the last committed entry and of fok = prs - plit, the same for every null event over a dropped region, and is */ \
            /* rebuilt when the run ends, Lnc_frec) */ \
            "s_sub_i32 %[fok], %[prs], %[plit]\n\t" \
            "s_sub_i32 %[t2], %[qh], 1\n\t"                /* the last committed entry (t2 stays untouched to the end of the turn) */ \
            "v_readlane_b32 %[cls], %[alen], %[t2]\n"      /* its word */ \
            "Lnc_fstart_%=:\n\t" \
            LZ_NC_LOADS_F \
            /* Two turns a pass, in two sets of registers that take turns: a turn requests the k-mers of the NEXT turn's round */ \
            /* -- where that round stands follows from the successor's queue entry alone, not from this round's outcome -- */ \
            /* before it waits for its own, so the loads of one turn fly during the LDS phase of the turn before.  A turn that */ \
            /* ends the run (no GO, a seed candidate) leaves the request behind; what leaves the second turn puts the */ \
            /* registers back: cls the committed entry's word, blen the successor's, i / rend the round's, qk / rk0 / rk1 its k-mers. */ \
            LZ_NC_FTURN(WORDF, "cls", "blen", "i", "t1", "kb", "qk", "rk0", "rk1", "qkb", "rk0b", "rk1b", "") \
            LZ_NC_FTURN(WORDF, "blen", "cls", "t1", "i", "rend", "qkb", "rk0b", "rk1b", "qk", "rk0", "rk1", "2") \
            "s_branch Lnc_fturn_%=\n" \
            "Lnc_fnogo2_%=:\n\t" \
            "s_mov_b32 %[i], %[t1]\n\t" \
            "s_mov_b32 %[rend], %[kb]\n\t" \
            "s_mov_b32 %[cls], %[blen]\n\t" \
            "s_branch Lnc_fnogo_%=\n" \
            "Lnc_frec2_%=:\n\t" \
            "s_mov_b32 %[i], %[t1]\n\t" \
            "s_mov_b32 %[rend], %[kb]\n\t" \
            "s_mov_b32 %[t0], %[cls]\n\t" \
            "s_mov_b32 %[cls], %[blen]\n\t" \
            "s_mov_b32 %[blen], %[t0]\n\t" \
            "s_waitcnt vmcnt(0)\n\t"                        /* (the request left behind is for qk / rk0 / rk1) */ \
            "v_mov_b32_e32 %[qk], %[qkb]\n\t" \
            "v_mov_b32_e32 %[rk0], %[rk0b]\n\t" \
            "v_mov_b32_e32 %[rk1], %[rk1b]\n\t" \
            "s_branch Lnc_frec_%=\n" \
            /* back into the run behind a round whose seed candidates were all false (Lnc_snone): the successor's null event */ \
            "Lnc_fcont_%=:\n\t" \
            "s_bfe_u32 %[t0], %[blen], 0x80011\n\t" \
            "s_add_i32 %[i], %[ap], %[t0]\n\t" \
            "s_add_i32 %[rend], %[bpos], %[t0]\n\t" \
            "s_mov_b32 %[t2], %[qh]\n\t" \
            "s_mov_b32 %[cls], %[blen]\n\t" \
            LZ_NC_COUNT \
            "s_branch Lnc_fstart_%=\n" \
            "Lnc_fnogo_%=:\n\t"                            /* the run ends behind entry t2 */ \
            "s_add_i32 %[qh], %[t2], 1\n" \
            /* the machine's state after the null event of entry t2 (see LZ_NC_COMMIT): avail = apos - fok */ \
            "Lnc_frec_%=:\n\t" \
            "v_readlane_b32 %[t0], %[apos], %[t2]\n\t" \
            "v_readlane_b32 %[t1], %[alen], %[t2]\n\t" \
            "v_readlane_b32 %[kc], %[aext], %[t2]\n\t" \
            "s_sext_i32_i8 %[t1], %[t1]\n\t" \
            "s_and_b32 %[kb], %[kc], 15\n\t" \
            "s_sub_i32 %[prs], %[t0], %[kb]\n\t" \
            "s_sub_i32 %[plit], %[t0], %[fok]\n\t" \
            "s_sub_i32 %[plit], %[plit], %[kb]\n\t" \
            "s_bfe_u32 %[lastlit], %[kc], 0x40014\n\t" \
            "s_bfe_u32 %[t0], %[kc], 0x50018\n\t" \
            "s_add_i32 %[t1], %[t1], %[t0]\n\t" \
            "s_sub_i32 %[lastb], %[t1], %[lastlit]\n\t" \
            "s_bfe_u32 %[t0], %[kc], 0x40004\n\t" \
            "s_add_i32 %[lastb], %[lastb], %[t0]\n\t" \
            "s_add_i32 %[lastlit], %[lastlit], %[kb]\n\t" \
            "s_sub_i32 %[lastlit], %[lastlit], %[t0]\n\t" \
            "s_mov_b32 %[pre], %[i]\n\t" \
            "s_bitcmp0_b32 %[cls], 15\n\t" \
            "s_cbranch_scc1 Lnc_top_%=\n"                   /* the run ended at an entry that is not GO: the general turn */ \
            "Lnc_fseed_%=:\n\t"                             /* a seed candidate: as above (the queue head is the successor) */ \
            "s_mov_b32 %[code], 1\n\t" \
            LZ_NC_FIX \
            "s_cmp_eq_u64 %[m], 0\n\t"                      /* (only steps without a k-mer) */ \
            "s_cbranch_scc1 Lnc_snone_%=\n" \
            /* The seed event (code = 1 here; every way out before the last line leaves it so, and seed / rk0 / rk1 / qk */ \
            /* untouched: find_event then does the same from the round).  The simple case is found here: the first step with */ \
            /* a seed candidate lies before the queued candidate's, ONE window position carries its msl-mer, the 64 symbols */ \
            /* behind the msl-mer are real symbols of one strand on both sides (no lane needs a bound) and hold a mismatch: */ \
            /* that is the event -- a seed is at least msl long and no anchor stands at its step to arbitrate with */ \
            /* (parser.cpp:548-580, 604-606) -- adv = the step, bpos, blen = msl + the matching symbols behind. */ \
            "Lnc_seedev_%=:\n\t" \
            "s_ff1_i32_b64 %[t0], %[m]\n"                   /* l */ \
            "Lnc_sdl_%=:\n\t" \
            "v_readlane_b32 %[t1], %[qk], %[t0]\n\t"        /* the step's msl-mer */ \
            "s_add_i32 %[t2], %[t0], %[MRD]\n\t"            /* its window: the positions idx < l + mrd */ \
            "s_sub_i32 %[kc], %[t2], 64\n\t" \
            "s_max_i32 %[kc], %[kc], 0\n\t" \
            "s_min_u32 %[kb], %[t2], 64\n\t" \
            "s_sub_i32 %[kb], 64, %[kb]\n\t"                /* (the positions of the first load beyond it leave by a shift) */ \
            "v_cmp_eq_u32_e32 vcc, %[t1], %[rk1]\n\t" \
            "s_bfm_b64 %[m2], %[kc], 0\n\t" \
            "s_and_b64 %[m2], %[m2], vcc\n\t" \
            "v_cmp_eq_u32_e32 vcc, %[t1], %[rk0]\n\t" \
            "s_bcnt1_i32_b64 %[kc], %[m2]\n\t" \
            "s_lshl_b64 vcc, vcc, %[kb]\n\t" \
            "s_bcnt1_i32_b64 %[t1], vcc\n\t" \
            "s_add_i32 %[t2], %[t1], %[kc]\n\t" \
            LZ_NC_WHY(2) \
            "s_cmp_eq_u32 %[t2], 0\n\t" \
            "s_cbranch_scc1 Lnc_sfalse_%=\n\t"              /* no window position: the prefilter's false candidate (the bitmap holds all 80 positions) */ \
            "s_cmp_lg_u32 %[t2], 1\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"                 /* several window positions: the longest, then the nearest (find_event) */ \
            "s_ff1_i32_b64 %[t2], vcc\n\t" \
            "s_sub_i32 %[t2], %[t2], %[kb]\n\t" \
            "s_ff1_i32_b64 %[kc], %[m2]\n\t" \
            "s_add_i32 %[kc], %[kc], 64\n\t" \
            "s_cmp_lg_u32 %[t1], 0\n\t" \
            "s_cselect_b32 %[t2], %[t2], %[kc]\n\t"         /* idx */ \
            "s_add_i32 %[rec], %[rend], %[t2]\n\t"          /* the seed in the reference ... */ \
            "s_add_i32 %[cls], %[i], %[t0]\n\t"             /* ... and in the query */ \
            /* The queued candidate's own step: its anchor arbitrates with the seed (parser.cpp:604-623) -- unless it IS the */ \
            /* seed: a plain candidate (resolved by its lane, >= mal, not at position 0) at the seed's own position is the */ \
            /* homologous mal-mer of a related stretch, same match, nothing to decide; it leaves the queue when the scan has */ \
            /* passed it (Lnc_top) */ \
            LZ_NC_WHY(1) \
            "s_cmp_lg_u32 %[t0], %[gap]\n\t" \
            "s_cbranch_scc1 Lnc_snoq_%=\n\t" \
            "s_cmp_lt_i32 %[blen], 1\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_cmp_lg_u32 %[bpos], %[rec]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n" \
            "Lnc_snoq_%=:\n\t" \
            /* bounds: query [cls + 7, cls + 71) inside [0, Lq), Lq = qend - 33 (qend = the scan's end D - msl = Lq + mrd - msl; a pair */ \
            /* with an N in it has qend far below zero); reference the same inside [0, L) or [rc0, rc0 + L), L = (rlim - C41M) / 2 */ \
            /* (rlim = 2 L + 3 mrd - msl + 1 - 80), rc0 = L + 2 mrd */ \
            "s_sub_i32 %[t1], %[qend], %[CQ64]\n\t" \
            LZ_NC_WHY(3) \
            "s_cmp_gt_i32 %[cls], %[t1]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_sub_i32 %[t1], %[rlim], %[C41M]\n\t" \
            "s_lshr_b32 %[t1], %[t1], 1\n\t" \
            "s_add_i32 %[t2], %[rec], %[MSL64]\n\t" \
            "s_cmp_le_i32 %[t2], %[t1]\n\t" \
            "s_cbranch_scc1 Lnc_sok_%=\n\t" \
            "s_add_i32 %[kb], %[t1], %[C2MRD]\n\t"                /* rc0 */ \
            "s_cmp_lt_i32 %[rec], %[kb]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"                 /* runs from the forward strand into the pad */ \
            "s_add_i32 %[kb], %[kb], %[t1]\n\t" \
            "s_cmp_gt_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n" \
            "Lnc_sok_%=:\n\t" \
            "s_add_i32 %[t1], %[rec], %[MSL]\n\t" \
            "s_add_i32 %[t2], %[cls], %[MSL]\n\t" \
            "v_add_u32_e32 %[a0], %[t1], %[lane]\n\t" \
            "v_add_u32_e32 %[a1], %[t2], %[lane]\n\t" \
            "v_lshrrev_b32_e32 %[aq], 4, %[a0]\n\t" \
            "v_lshrrev_b32_e32 %[t], 4, %[a1]\n\t" \
            "v_lshlrev_b32_e32 %[aq], 2, %[aq]\n\t" \
            "v_lshlrev_b32_e32 %[t], 2, %[t]\n\t" \
            "global_load_dword %[aq], %[aq], %[rt2]\n\t" \
            "global_load_dword %[t], %[t], %[qt2]\n\t" \
            "v_and_b32_e32 %[a0], 15, %[a0]\n\t" \
            "v_and_b32_e32 %[a1], 15, %[a1]\n\t" \
            "v_lshlrev_b32_e32 %[a0], 1, %[a0]\n\t" \
            "v_lshlrev_b32_e32 %[a1], 1, %[a1]\n\t" \
            "s_waitcnt vmcnt(0)\n\t" \
            "v_lshrrev_b32_e32 %[aq], %[a0], %[aq]\n\t" \
            "v_lshrrev_b32_e32 %[t], %[a1], %[t]\n\t" \
            "v_xor_b32_e32 %[aq], %[aq], %[t]\n\t" \
            "v_and_b32_e32 %[aq], 3, %[aq]\n\t" \
            "v_cmp_ne_u32_e64 %[m], 0, %[aq]\n\t" \
            "s_nop 0\n\t" \
            LZ_NC_WHY(4) \
            "s_cmp_eq_u64 %[m], 0\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"                 /* 64 more symbols match: the general path measures on */ \
            "s_ff1_i32_b64 %[blen], %[m]\n\t" \
            "s_add_i32 %[blen], %[blen], %[MSL]\n\t" \
            "s_mov_b32 %[ap], %[t0]\n\t" \
            "s_mov_b32 %[bpos], %[rec]\n\t" \
            LZ_NC_WHY(5) \
            "s_mov_b32 %[code], 4\n\t"                      /* from here on the event is known: any way out hands it to the machine */ \
            /* The close match itself (PairMachine::run: gap_fill, match_run, the first chunk of extend_forward), when every */ \
            /* symbol it looks at is a real symbol of one strand and the forward extension breaks inside its first chunk. */ \
            /* What it does to the open region (cl, clit; nl = 0 before and after): the gap's l symbols give `score` matches */ \
            /* -- the best split's count, compare_ranges_both_ways (parser.cpp:251-374); which split wins a tie only matters to */ \
            /* the alignment output -- and l - score literals, the match blen matches, the extension e - mm and mm. */ \
            /*   t0 = l, rec = the seed in the reference, cls = in the query;  t1 = fq, fok = fr (behind the match) */ \
            "s_add_i32 %[t1], %[cls], %[blen]\n\t" \
            "s_add_i32 %[t2], %[t1], 64\n\t" \
            "s_sub_i32 %[kb], %[qend], %[C40M]\n\t"              /* Lq */ \
            "s_cmp_gt_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_add_i32 %[fok], %[rec], %[blen]\n\t" \
            "s_add_i32 %[t2], %[fok], 64\n\t"               /* the reference side: [rend, fr + 64) on one strand */ \
            "s_sub_i32 %[kb], %[rlim], %[C41M]\n\t" \
            "s_lshr_b32 %[kb], %[kb], 1\n\t"                /* L */ \
            "s_cmp_le_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lnc_hull_%=\n\t" \
            "s_add_i32 %[kc], %[kb], %[C2MRD]\n\t"                /* rc0 */ \
            "s_cmp_lt_i32 %[rend], %[kc]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_add_i32 %[kc], %[kc], %[kb]\n\t" \
            "s_cmp_gt_i32 %[t2], %[kc]\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n" \
            "Lnc_hull_%=:\n\t" \
            /* to_scan = min(fr - rend, l), shift = l - to_scan; left diagonal (rend, i), right (fr - to_scan, i + shift) */ \
            "s_sub_i32 %[kb], %[fok], %[rend]\n\t" \
            "s_min_i32 %[kb], %[kb], %[t0]\n\t"             /* to_scan (0 when the seed sits at the first step) */ \
            "s_sub_i32 %[kc], %[t0], %[kb]\n\t" \
            "s_sub_i32 %[gap], %[fok], %[kb]\n\t"           /* right diagonal: reference start */ \
            "s_add_i32 %[kc], %[kc], %[i]\n\t"              /*                 query start */ \
            /* six words in flight: the word's byte address ((pos >> 4) << 2) worked out in the register the word lands in */ \
            "v_add_u32_e32 %[rk0], %[rend], %[lane]\n\t" \
            "v_add_u32_e32 %[rk1], %[i], %[lane]\n\t" \
            "v_add_u32_e32 %[qk], %[gap], %[lane]\n\t" \
            "v_add_u32_e32 %[a0], %[kc], %[lane]\n\t" \
            "v_add_u32_e32 %[a1], %[fok], %[lane]\n\t" \
            "v_add_u32_e32 %[aq], %[t1], %[lane]\n\t" \
            "v_lshrrev_b32_e32 %[rk0], 4, %[rk0]\n\t" \
            "v_lshrrev_b32_e32 %[rk1], 4, %[rk1]\n\t" \
            "v_lshrrev_b32_e32 %[qk], 4, %[qk]\n\t" \
            "v_lshrrev_b32_e32 %[a0], 4, %[a0]\n\t" \
            "v_lshrrev_b32_e32 %[a1], 4, %[a1]\n\t" \
            "v_lshrrev_b32_e32 %[aq], 4, %[aq]\n\t" \
            "v_lshlrev_b32_e32 %[rk0], 2, %[rk0]\n\t" \
            "v_lshlrev_b32_e32 %[rk1], 2, %[rk1]\n\t" \
            "v_lshlrev_b32_e32 %[qk], 2, %[qk]\n\t" \
            "v_lshlrev_b32_e32 %[a0], 2, %[a0]\n\t" \
            "v_lshlrev_b32_e32 %[a1], 2, %[a1]\n\t" \
            "v_lshlrev_b32_e32 %[aq], 2, %[aq]\n\t" \
            "s_nop 0\n\t" \
            "global_load_dword %[rk0], %[rk0], %[rt2]\n\t" \
            "global_load_dword %[rk1], %[rk1], %[qt2]\n\t" \
            "global_load_dword %[qk], %[qk], %[rt2]\n\t" \
            "global_load_dword %[a0], %[a0], %[qt2]\n\t" \
            "global_load_dword %[a1], %[a1], %[rt2]\n\t" \
            "global_load_dword %[aq], %[aq], %[qt2]\n\t" \
            "s_bfm_b64 %[seed], %[kb], 0\n\t"               /* the to_scan lanes of the gap's diagonals (to_scan <= 40) */ \
            "s_waitcnt vmcnt(0)\n\t" \
            /* left diagonal -> m = its matches */ \
            "v_add_u32_e32 %[t], %[rend], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[i], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[rk0], %[t], %[rk0]\n\t" \
            "v_lshrrev_b32_e32 %[rk1], %[bq], %[rk1]\n\t" \
            "v_xor_b32_e32 %[rk0], %[rk0], %[rk1]\n\t" \
            "v_and_b32_e32 %[rk0], 3, %[rk0]\n\t" \
            "v_cmp_eq_u32_e64 %[m], 0, %[rk0]\n\t" \
            /* right diagonal -> m2 */ \
            "v_add_u32_e32 %[t], %[gap], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[kc], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[qk], %[t], %[qk]\n\t" \
            "v_lshrrev_b32_e32 %[a0], %[bq], %[a0]\n\t" \
            "v_xor_b32_e32 %[qk], %[qk], %[a0]\n\t" \
            "v_and_b32_e32 %[qk], 3, %[qk]\n\t" \
            "v_cmp_eq_u32_e64 %[m2], 0, %[qk]\n\t" \
            /* forward chunk -> vcc = its MISmatches */ \
            "v_add_u32_e32 %[t], %[fok], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[t1], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[a1], %[t], %[a1]\n\t" \
            "v_lshrrev_b32_e32 %[aq], %[bq], %[aq]\n\t" \
            "v_xor_b32_e32 %[a1], %[a1], %[aq]\n\t" \
            "v_and_b32_e32 %[a1], 3, %[a1]\n\t" \
            "v_cmp_ne_u32_e32 vcc, 0, %[a1]\n\t" \
            "s_and_b64 %[m], %[m], %[seed]\n\t"             /* Lm */ \
            "s_and_b64 %[m2], %[m2], %[seed]\n\t"           /* Rm */ \
            "s_mov_b64 %[seed], vcc\n\t"                    /* Bf */ \
            /* best split: lane s scores popc(Lm below s) + popc(Rm from s on) for s <= to_scan; the maximum is what counts */ \
            "s_bcnt1_i32_b64 %[t2], %[m2]\n\t" \
            "s_mov_b64 vcc, %[m]\n\t" \
            "s_nop 0\n\t" \
            "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t" \
            "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t" \
            "s_mov_b64 vcc, %[m2]\n\t" \
            "s_nop 0\n\t" \
            "v_mbcnt_lo_u32_b32 %[bq], vcc_lo, 0\n\t" \
            "v_mbcnt_hi_u32_b32 %[bq], vcc_hi, %[bq]\n\t" \
            "v_add_u32_e32 %[t], %[t2], %[t]\n\t" \
            "v_sub_u32_e32 %[t], %[t], %[bq]\n\t" \
            "v_cmp_ge_u32_e32 vcc, %[kb], %[lane]\n\t" \
            "s_nop 1\n\t" \
            "v_cndmask_b32_e32 %[t], 0, %[t], vcc\n\t"      /* (a split beyond to_scan scores nothing) */ \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_readlane_b32 %[t2], %[t], 63\n\t"            /* score */ \
            /* the forward extension's first chunk (try_extend_forward, parser.cpp:377-409; ext_lane): symbol j breaks the scan */ \
            /* if the 15 symbols ending at it hold more than 7 mismatches, and qualifies if it and the two before it match. */ \
            /* The window of lane j = bits [j, j + 15) of Bf << 14, as three words e0 (gap), e1 (kc), e2 (kb) */ \
            "s_mov_b64 vcc, %[seed]\n\t" \
            "s_lshl_b32 %[gap], vcc_lo, %[AW1]\n\t" \
            "s_lshr_b32 %[kb], vcc_lo, %[AW1C]\n\t" \
            "s_lshl_b32 %[kc], vcc_hi, %[AW1]\n\t" \
            "s_or_b32 %[kc], %[kc], %[kb]\n\t" \
            "s_lshr_b32 %[kb], vcc_hi, %[AW1C]\n\t" \
            "v_mov_b32_e32 %[rk0], %[gap]\n\t" \
            "v_mov_b32_e32 %[rk1], %[kc]\n\t" \
            "v_mov_b32_e32 %[qk], %[kb]\n\t" \
            "v_cmp_gt_u32_e32 vcc, 32, %[lane]\n\t" \
            "v_alignbit_b32 %[a0], %[rk1], %[rk0], %[lane]\n\t" \
            "v_alignbit_b32 %[a1], %[qk], %[rk1], %[lane]\n\t" \
            "v_cndmask_b32_e32 %[a0], %[a1], %[a0], vcc\n\t" \
            "v_and_b32_e32 %[a0], %[AWM], %[a0]\n\t" \
            "v_bcnt_u32_b32 %[a1], %[a0], 0\n\t" \
            "v_and_b32_e32 %[a0], %[ARM], %[a0]\n\t" \
            "v_cmp_lt_u32_e32 vcc, %[AM], %[a1]\n\t"            /* brk */ \
            "v_cmp_eq_u32_e64 %[m], 0, %[a0]\n\t"           /* qual */ \
            "s_nop 0\n\t" \
            "s_cmp_eq_u64 vcc, 0\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t"                 /* no break inside the chunk: the extension runs on (the general path) */ \
            "s_ff1_i32_b64 %[kb], vcc\n\t" \
            "s_cmp_ge_u32 %[kb], 63\n\t" \
            "s_cbranch_scc1 Lnc_end_%=\n\t" \
            "s_add_i32 %[kb], %[kb], 1\n\t" \
            "s_bfm_b64 %[m2], %[kb], 0\n\t" \
            "s_and_b64 %[m], %[m], %[m2]\n\t"               /* qualifying symbols up to the break */ \
            "s_mov_b32 %[kb], 0\n\t"                         /* e */ \
            "s_mov_b32 %[kc], 0\n\t"                         /* mm */ \
            "s_cbranch_scc0 Lnc_sext_%=\n\t" \
            "s_flbit_i32_b64 %[kb], %[m]\n\t" \
            "s_sub_i32 %[kb], 64, %[kb]\n\t"                /* e = the last qualifying symbol + 1 */ \
            "s_bfm_b64 %[m2], %[kb], 0\n\t" \
            "s_and_b64 %[m2], %[m2], %[seed]\n\t" \
            "s_bcnt1_i32_b64 %[kc], %[m2]\n"                 /* mm: the mismatches among its e symbols */ \
            "Lnc_sext_%=:\n\t" \
            /* commit: the open region (the last committed event's, else the machine's) grows by the gap, the match, the extension */ \
            "s_cmp_lg_u32 %[lastb], 0\n\t" \
            "s_cselect_b32 %[gap], %[lastb], %[ocl]\n\t" \
            "s_cselect_b32 %[rec], %[lastlit], %[oclit]\n\t" \
            "s_add_i32 %[gap], %[gap], %[t2]\n\t"           /* + score */ \
            "s_add_i32 %[gap], %[gap], %[blen]\n\t" \
            "s_add_i32 %[gap], %[gap], %[kb]\n\t" \
            "s_sub_i32 %[lastb], %[gap], %[kc]\n\t"         /* cl += score + blen + e - mm */ \
            "s_add_i32 %[rec], %[rec], %[t0]\n\t" \
            "s_sub_i32 %[rec], %[rec], %[t2]\n\t" \
            "s_add_i32 %[lastlit], %[rec], %[kc]\n\t"       /* clit += l - score + mm */ \
            "s_add_i32 %[i], %[t1], %[kb]\n\t" \
            "s_add_i32 %[rend], %[fok], %[kb]\n\t" \
            "s_mov_b32 %[pre], %[i]\n\t" \
            "s_mov_b32 %[code], 0\n\t" \
            "s_mov_b64 %[m2], -1\n\t" \
            LZ_NC_COUNT \
            "s_branch Lnc_top_%=\n" \
            /* a false seed candidate (nine in ten of the rounds that have one): the next one; none left = the round has no seed */ \
            /* candidate after all -- the fast turn goes on where it stood, the general turn is taken again with the flag set */ \
            "Lnc_sfalse_%=:\n\t" \
            "s_bitset0_b64 %[m], %[t0]\n\t" \
            "s_cmp_eq_u64 %[m], 0\n\t" \
            "s_cbranch_scc1 Lnc_snone_%=\n\t" \
            "s_ff1_i32_b64 %[t0], %[m]\n\t" \
            "s_branch Lnc_sdl_%=\n" \
            "Lnc_snone_%=:\n\t" \
            "s_mov_b32 %[code], 0\n\t" \
            "s_mov_b64 %[m2], -1\n\t" \
            "s_bitcmp1_b32 %[cls], 15\n\t" \
            "s_cbranch_scc1 Lnc_fcont_%=\n\t" \
            "s_mov_b64 %[m2], 0\n\t" \
            "s_branch Lnc_top_%=\n" \
            "Lnc_sseed_%=:\n\t"                            /* the general turn's seed candidates: known to be false if this is that second pass */ \
            "s_cmp_eq_u64 %[m2], 0\n\t" \
            "s_mov_b64 %[m2], -1\n\t" \
            "s_cbranch_scc1 Lnc_noseed_%=\n\t" \
            "s_branch Lnc_fseed_%=\n" \
            "Lnc_kept_%=:\n\t"                              /* no region to drop: the candidate may look back over the literals since */ \
            "s_mov_b32 %[cls], 1\n\t"                        /* the last match only (avail = lit), the rest is the same */ \
            "s_mov_b32 %[t1], %[gap]\n\t" \
            "s_branch Lnc_avail_%=\n" \
            "Lnc_close_%=:\n\t"                             /* a tracking step: close to the predicted position = not ours */ \
            "s_add_i32 %[t1], %[rend], %[gap]\n\t" \
            "s_sub_i32 %[t1], %[bpos], %[t1]\n\t" \
            "s_abs_i32 %[t1], %[t1]\n\t" \
            "s_cmp_le_i32 %[t1], %[MRD]\n\t" \
            "s_cbranch_scc1 Lnc_chk_%=\n\t" \
            "s_branch Lnc_distant_%=\n" \
            "Lnc_npl_%=:\n\t" \
            LZ_NC_FIX \
            "Lnc_end_%=:\n\t" \
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
            LZ_NC_PRIO_OFF \
            "s_nop 4" \
            : LZ_NC_COUNT_OPERAND [i] "+s"(i), [rend] "+s"(r_end), [qh] "+s"(qh), [prs] "+s"(prev_rs), [pre] "+s"(prev_re), [plit] "+s"(pre_lit), \
              [code] "=&s"(code), [ap] "=&s"(ap), [bpos] "=&s"(bpos), [blen] "=&s"(blen), [lastb] "=&s"(last_cl), [lastlit] "=&s"(last_clit), [rec] "=&s"(rec), \
              [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [kb] "=&s"(kb), [kc] "=&s"(kc), [m] "=&s"(m), [seed] "=&s"(seed), \
              [gap] "=&s"(gap), [cls] "=&s"(cls), [fok] "=&s"(fok), [atm] "=&s"(add_tm), [atl] "=&s"(add_tl), [atc] "=&s"(add_tc), [m2] "=&s"(m2), \
              [rk0] "=&v"(rk0), [rk1] "=&v"(rk1), [qk] "=&v"(qk), [a0] "=&v"(a0), [a1] "=&v"(a1), [aq] "=&v"(aq), [t] "=&v"(t), [bq] "=&v"(bq), [w1] "=&v"(w1), [dumv] "=&v"(dumv), [qkb] "=&v"(qkb), [rk0b] "=&v"(rk0b), [rk1b] "=&v"(rk1b) \
            : [qc] "s"(qc_u), [ilim] "s"(ilim_u), [rlim] "s"(rlim_u), [qks] "s"(qks), [rks] "s"(rks), [ocl] "s"(ocl_u), [oclit] "s"(oclit_u), \
              [rt2] "s"(rt2), [qt2] "s"(qt2), [qend] "s"(qend), [wdum] "s"((int)SEED_BM_WORDS) SPLIN, \
              [apos] "v"(a_pos), [alen] "v"(a_len), [aref] "v"(a_ref), [aext] "v"(a_ext), [lane] "v"(lane), \
              [ldsb] "v"(ldsb), [zero] "v"(zero), [one] "v"(one), \
              [MQD] "n"(MQD), [MRD] "n"(MRD), [REG] "n"(REG), [AW] "n"(AW), [NT] "n"(NT), [WIN] "s"((int)WIN), [NR1] "n"(WIN - 64), [MSL] "n"(MSL), [MSL64] "n"(MSL + 64), [C41M] "n"(3 * MRD + 1 - WIN - MSL), [C40M] "n"(MRD - MSL), [CQ64] "n"(MRD + 64), [C2MRD] "n"(2 * MRD), \
              [AW1] "n"(AW - 1), [AW1C] "n"(33 - AW), [AWM] "n"((1 << AW) - 1), [ARM] "n"(((1 << AR) - 1) << (AW - AR)), [AM] "n"(AM), [KS5] "n"(2 * MSL + 5), [KS] "n"(2 * MSL) \
            : "vcc", "scc", "memory");
#ifdef LZANI_PHASE_TIME
        const unsigned long long pt_t0 = pt_now();
#endif
        if constexpr (SPLITW) {
            if constexpr (MSL == 9) { LZ_NC_ASM_X(LZ_NC_WORD9, LZ_NC_WORD9, LZ_NC_SPLITCHK, LZ_NC_SPLITGEN, LZ_NC_SPLITIN) }
            else if constexpr (MSL == 8) { LZ_NC_ASM_X(LZ_NC_WORD8, LZ_NC_WORD8, LZ_NC_SPLITCHK, LZ_NC_SPLITGEN, LZ_NC_SPLITIN) }
            else if constexpr (CP::NF) { LZ_NC_ASM_X(LZ_NC_WORD7, LZ_NC_WORD7N, LZ_NC_SPLITCHK, LZ_NC_SPLITGEN, LZ_NC_SPLITIN) }
            else { LZ_NC_ASM_X(LZ_NC_WORD7, LZ_NC_WORD7, LZ_NC_SPLITCHK, LZ_NC_SPLITGEN, LZ_NC_SPLITIN) }
        } else
        if constexpr (MSL == 9) { LZ_NC_ASM(LZ_NC_WORD9, LZ_NC_WORD9) }
        else if constexpr (MSL == 8) { LZ_NC_ASM(LZ_NC_WORD8, LZ_NC_WORD8) }
        else if constexpr (CP::NF) { LZ_NC_ASM(LZ_NC_WORD7, LZ_NC_WORD7N) }          // (N-free by instantiation: see chain_classes)
        else { LZ_NC_ASM(LZ_NC_WORD7, LZ_NC_WORD7) }
#ifdef LZANI_PHASE_TIME
        pt_chain += pt_now() - pt_t0;
#endif
        q_head = qh;
        if constexpr (JOIN) {
            // Nothing in hand because the queue ran out (or its tail does not cover the 41 steps behind i) while the pair's
            // bitmap has more: refill first and come back -- the round is then this loop's, not the compiler's, and a
            // queue tail costs no wave-wide detection of its own (find_event's light round stays for the scan that
            // has jumped over the queue: related stretches).
            if (code == 0 && scan_pos >= i && scan_pos < iend && (qh >= q_cnt || (i > ilim_q && restart_at != i))) refill_only = true;
        }
        pre_round = code == 1;
        if (code == 0) LZ_PS(16); else if (code == 1) LZ_PS(17); else if (code == 2) LZ_PS(18); else LZ_PS(19);
        pre_seed = seed; pre_rk0 = rk0; pre_rk1 = rk1; pre_qk = qk;
        if (code == 2) { adv = ap - i; last_src = q_head++; }
        if (code == 4) { adv = ap; last_src = -1; }          // the seed event: bpos, blen are the loop's
#ifdef LZANI_CHAIN_STATS
        st[0] += 1; st[1] += ncnt; st[2] += code == 0; st[3] += code == 1 && seed != 0; st[4] += code == 1 && seed == 0; st[5] += code == 2;
        if (code == 2) {                      // why the event found is not a null event (first reason that applies)
            const int g2 = ap - i;
            const bool close = g2 <= MQD && iabs(bpos - (r_end + g2)) <= MRD;
            const bool kept = prev_rs < 0 || prev_re - prev_rs >= REG;
            const bool nofwd = !((u32)rec & EXT_REC_FWDK);
            sr[0] += close; sr[1] += !close && kept; sr[2] += !close && !kept && nofwd; sr[3] += !close && !kept && !nofwd;
        }
        if (code == 1 && seed != 0) {         // why the loop's seed event handed the round back
            sw[0] += why == 1; sw[2] += why == 2; sw[3] += why == 3; sw[4] += why == 4; sw[5] += why == 0;
        }
        if (code == 4) sw[6] += 1;            // the event known, its commit left to the machine
#endif
        return code;
    }

    // The stretch chain (probe form; CHAIN parameters): runs of CLOSE matches behind each other -- what a related stretch is
    // made of once an approximate extension has carried the scan beyond everything the queue has detected (scan_pos < i)
    // -- as a second hand-scheduled loop next to the null chain, built from its parts: the tracking round (the three k-mer
    // loads + the LDS bitmap, LZ_NC_LOADS / LZ_NC_ROUND), the seed event (one window position carries the first seed
    // step's msl-mer, the 64 symbols behind it are real symbols of one strand and hold a mismatch) and the close match
    // (gap fill by the best split's score, the match, the first chunk of the forward extension: parser.cpp:630-635,
    // 251-374, 687-697) -- plus what the queue told the null chain and nothing tells this loop: whether a step up to the
    // seed's has an ANCHOR (parser.cpp:585-602).  The steps' mixed mal-mer hashes ride with the round's loads, their tag
    // words (one 4-byte probe per step up to the seed's) with the seed's verification; a tag hit at an earlier step, an
    // overflowing bucket or a tag in two slots ends the run (the general path), a single hit at the seed's own step must
    // be the seed itself (its bucket entry, one scalar load that flies with the close match's six: the arbitration of
    // parser.cpp:604-623 then has nothing to decide).  State: i, r_end and the open region's accumulators (cl, clit; nl = 0
    // before and after every event).  Stops with nothing half-done at the first event that is not this cycle; returns
    // the number of events it committed.  ~125 vector + ~100 scalar instructions per event against ~330 + ~390 by the
    // compiler's path (profiles/r4_related_*: the related kernel is bound by instruction issue, the scalar unit first).
    // (in the kernels where related pairs are what the time goes into: filtered rows -- the probe form -- and the bitmap forms of
    // the long-genome parameter sets; the dense viral kernel, 999 unrelated pairs in 1,000, stays lean: the loop costs it 2.5 %)
    static constexpr bool HAS_STRETCH_CHAIN = CHAIN != 0 && (ChainP<CHAIN>::MQD <= ChainP<CHAIN>::MRD) &&
                                              (!JOIN || ChainP<CHAIN>::MAL >= 13 || LZANI_STRETCH_DENSE);
    // kind: 0 = nothing in hand; 1 = the last event's gap and match are committed, its forward extension is not: Bf = the
    // mismatches of its first chunk (no break inside it); 2 = a step before the first seed step has an anchor candidate:
    // step (adv) and its mixed hash (hq); 3 = the tracking steps of i hold no seed (their anchors decide: find_event)
    __device__ __forceinline__ int stretch_chain(int& i, int& r_end, int& cl, int& clit, int& kind, u64& Bf, int& adv, u32& hqa)
    {
        typedef ChainP<CHAIN> CP;
        enum { MQD = CP::MQD, MRD = CP::MRD, MSL = CP::MSL, AW = CP::AW, AM = CP::AM, AR = CP::AR, NT = CP::NT, WIN = CP::WIN };
        const int ilim = SPLITW ? imin(iend - NT, stop_i - (int)SPLIT_MARGIN) : iend - NT, rlim = R.len - MSL + 1 - WIN;
        const u32 ldsb = (u32)(size_t)bitmap;
        const u32 zero = 0, one = 1;
        const u32* const qks = uniform_ptr(qkS);
        const u32* const rks = uniform_ptr(rkS);
        const u32* const qkl = uniform_ptr(qkL);
        const u32* const rt2 = uniform_ptr(reinterpret_cast<const u32*>(R.t2));
        const u32* const qt2 = uniform_ptr(reinterpret_cast<const u32*>(Q.t2));
        const u32* const twp = uniform_ptr(I.tw);
        const u32* const bkp = uniform_ptr(I.bk);
        const int qend = __builtin_amdgcn_readfirstlane((R.nfree && Q.nfree) ? iend : -(1 << 30));
        const int tbits = __builtin_amdgcn_readfirstlane(I.kb - I.dirbits), pbits = __builtin_amdgcn_readfirstlane(I.posbits);
        const u32 tagm = (u32)__builtin_amdgcn_readfirstlane((int)I.tagmask);
        int ncm, t0, t1, t2, kb, kc, gap, rec, cls, fok, blen, anc, aent;
        u64 m, seed, m2, pmk;
        u32 rk0, rk1, qk, hq, a0, a1, aq, t, bq, w1, dumv;
        [[maybe_unused]] u64 cwa, cwb;                     // bitmap form: two words of the pair's candidate bitmap
        [[maybe_unused]] u32 e0, e1, e2, e3, tagl;         //              the seed step's bucket, its tag
        [[maybe_unused]] const unsigned long long* const cbp = JOIN ? reinterpret_cast<const unsigned long long*>(uniform_ptr(reinterpret_cast<const u32*>(cand_bits))) : nullptr;
#if defined(LZANI_STAMPS) || defined(LZANI_PATH_STATS)
        constexpr bool unify = true;
#else
        constexpr bool unify = SPLITW;
#endif
        int ilim_u = ilim, rlim_u = rlim;
        if constexpr (unify) {
            i = __builtin_amdgcn_readfirstlane(i); r_end = __builtin_amdgcn_readfirstlane(r_end);
            cl = __builtin_amdgcn_readfirstlane(cl); clit = __builtin_amdgcn_readfirstlane(clit);
            ilim_u = __builtin_amdgcn_readfirstlane(ilim); rlim_u = __builtin_amdgcn_readfirstlane(rlim);
        }
#ifdef LZANI_PATH_STATS
        int why = 0;
#define LZ_SC_WHY(n) "s_mov_b32 %[why], " #n "\n\t"
#define LZ_SC_WHY_OPERAND [why] "=&s"(why),
#else
#define LZ_SC_WHY(n)
#define LZ_SC_WHY_OPERAND
#endif
        // ---- the two forms of "which steps up to the seed's have an anchor candidate" ----
        // probe form: the steps' tag words (a zero byte in w ^ rep4(0x80 | tag) = a slot of the bucket carrying the step's tag)
#define LZ_SC_P_CANDLOAD
#define LZ_SC_P_TWADDR \
            "v_lshrrev_b32_e32 %[bq], %[TB], %[hq]\n\t"     /* the step's bucket */ \
            "v_cmp_ge_u32_e32 vcc, %[t0], %[lane]\n\t"      /* the steps up to the seed's */
#define LZ_SC_P_TWADDR2 \
            "v_lshlrev_b32_e32 %[bq], 2, %[bq]\n\t" \
            "v_cndmask_b32_e32 %[bq], 0, %[bq], vcc\n\t"
#define LZ_SC_P_TWLOAD \
            "s_mov_b64 %[pmk], vcc\n\t" \
            "s_nop 0\n\t" \
            "global_load_dword %[bq], %[bq], %[twp]\n\t"
#define LZ_SC_P_VWAIT "s_waitcnt vmcnt(0)\n\t"
#define LZ_SC_P_TWSWAR \
            "v_and_b32_e32 %[a0], %[TAGM], %[hq]\n\t" \
            "v_or_b32_e32 %[a0], 0x80, %[a0]\n\t" \
            "v_perm_b32 %[a0], %[a0], %[a0], %[zero]\n\t" \
            "v_xor_b32_e32 %[a0], %[a0], %[bq]\n\t"         /* x */ \
            "v_subrev_u32_e32 %[a1], 0x01010101, %[a0]\n\t" \
            "v_not_b32_e32 %[a0], %[a0]\n\t" \
            "v_and_b32_e32 %[a1], %[a1], %[a0]\n\t" \
            "v_and_b32_e32 %[a1], 0x80808080, %[a1]\n\t"    /* z: its lowest flag is exact */
#define LZ_SC_P_CANDMASK \
            "v_cmp_ne_u32_e32 vcc, 0, %[a1]\n\t" \
            "s_mov_b64 %[m2], vcc\n\t" \
            "v_cmp_eq_u32_e32 vcc, 0x808080ff, %[bq]\n\t"   /* an overflowing bucket (TW_OVERFLOW) */ \
            "s_or_b64 %[m2], %[m2], vcc\n\t" \
            "s_and_b64 %[m2], %[m2], %[pmk]\n\t"            /* the steps <= l with an anchor candidate */
#define LZ_SC_P_ANCENTRY \
            /* at the seed's own step: the entry behind the tag must be the seed itself */ \
            "v_readlane_b32 %[kc], %[a1], %[t0]\n\t"        /* z */ \
            "v_readlane_b32 %[t1], %[bq], %[t0]\n\t"        /* w */ \
            "v_readlane_b32 %[t2], %[hq], %[t0]\n\t" \
            "s_cmp_eq_u32 %[t1], 0x808080ff\n\t" \
            LZ_SC_WHY(7) "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_sub_u32 %[t1], %[kc], 1\n\t" \
            "s_and_b32 %[t1], %[t1], %[kc]\n\t" \
            "s_cmp_lg_u32 %[t1], 0\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t"                 /* the tag in two slots */ \
            "s_ff1_i32_b32 %[kc], %[kc]\n\t" \
            "s_lshr_b32 %[kc], %[kc], 3\n\t"                /* the slot */ \
            "s_lshr_b32 %[t2], %[t2], %[TB]\n\t" \
            "s_lshl_b32 %[t2], %[t2], 2\n\t" \
            "s_add_u32 %[t2], %[t2], %[kc]\n\t" \
            "s_lshl_b32 %[t2], %[t2], 2\n\t" \
            "s_load_dword %[aent], %[bkp], %[t2]\n\t" \
            "s_mov_b32 %[anc], 1\n"
#define LZ_SC_P_ANCCHECK \
            "s_bfm_b32 %[anc], %[PB], 0\n\t" \
            "s_and_b32 %[aent], %[aent], %[anc]\n\t" \
            LZ_SC_WHY(8) "s_cmp_lg_u32 %[aent], %[rec]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n"
#define LZ_SC_P_XOUT
#define LZ_SC_P_XIN , [twp] "s"(twp)
        // bitmap form: two words of the pair's candidate bitmap (scalar loads: issued behind the round's LDS operations, which
        // share the counter), the four entries of the seed step's bucket by four more
#define LZ_SC_J_CANDLOAD \
            "s_lshr_b32 %[t1], %[i], 6\n\t" \
            "s_lshl_b32 %[t1], %[t1], 3\n\t" \
            "s_add_u32 %[t2], %[t1], 8\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "s_load_dwordx2 %[cwa], %[cbp], %[t1]\n\t" \
            "s_load_dwordx2 %[cwb], %[cbp], %[t2]\n\t"
#define LZ_SC_J_TWADDR
#define LZ_SC_J_TWADDR2
#define LZ_SC_J_TWLOAD
#define LZ_SC_J_VWAIT "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define LZ_SC_J_TWSWAR
#define LZ_SC_J_CANDMASK \
            "s_and_b32 %[t1], %[i], 63\n\t" \
            "s_lshr_b64 %[m2], %[cwa], %[t1]\n\t" \
            "s_sub_i32 %[t1], 63, %[t1]\n\t" \
            "s_lshl_b64 %[pmk], %[cwb], 1\n\t" \
            "s_lshl_b64 %[pmk], %[pmk], %[t1]\n\t" \
            "s_or_b64 %[m2], %[m2], %[pmk]\n\t"             /* bit j = step i + j has an anchor candidate */ \
            "s_add_i32 %[t1], %[t0], 1\n\t" \
            "s_bfm_b64 %[pmk], %[t1], 0\n\t" \
            "s_and_b64 %[m2], %[m2], %[pmk]\n\t"            /* the steps <= l */
#define LZ_SC_J_ANCENTRY \
            "v_readlane_b32 %[t2], %[hq], %[t0]\n\t" \
            "s_lshr_b32 %[t1], %[t2], %[TB]\n\t" \
            "s_lshl_b32 %[t1], %[t1], 4\n\t" \
            "s_and_b32 %[tagl], %[t2], %[TAGM]\n\t"         /* the step's tag */ \
            "s_load_dword %[e0], %[bkp], %[t1]\n\t" \
            "s_add_u32 %[t2], %[t1], 4\n\t" \
            "s_load_dword %[e1], %[bkp], %[t2]\n\t" \
            "s_add_u32 %[anc], %[t1], 8\n\t" \
            "s_load_dword %[e2], %[bkp], %[anc]\n\t" \
            "s_add_u32 %[aent], %[t1], 12\n\t" \
            "s_load_dword %[e3], %[bkp], %[aent]\n\t" \
            "s_mov_b32 %[anc], 1\n"
#define LZ_SC_J_ANCCHECK \
            /* exactly one of the bucket's four entries carries the tag, no overflow, and it sits at the seed's position */ \
            "s_cmp_eq_u32 %[e3], 0xfffffffe\n\t" \
            LZ_SC_WHY(7) "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_mov_b32 %[anc], 0\n\t" \
            "s_mov_b32 %[aent], 0\n\t" \
            "s_lshr_b32 %[t2], %[e0], %[PB]\n\t" \
            "s_cmp_eq_u32 %[t2], %[tagl]\n\t" \
            "s_cselect_b32 %[aent], %[e0], %[aent]\n\t" \
            "s_addc_u32 %[anc], %[anc], 0\n\t" \
            "s_lshr_b32 %[t2], %[e1], %[PB]\n\t" \
            "s_cmp_eq_u32 %[t2], %[tagl]\n\t" \
            "s_cselect_b32 %[aent], %[e1], %[aent]\n\t" \
            "s_addc_u32 %[anc], %[anc], 0\n\t" \
            "s_lshr_b32 %[t2], %[e2], %[PB]\n\t" \
            "s_cmp_eq_u32 %[t2], %[tagl]\n\t" \
            "s_cselect_b32 %[aent], %[e2], %[aent]\n\t" \
            "s_addc_u32 %[anc], %[anc], 0\n\t" \
            "s_lshr_b32 %[t2], %[e3], %[PB]\n\t" \
            "s_cmp_eq_u32 %[t2], %[tagl]\n\t" \
            "s_cselect_b32 %[aent], %[e3], %[aent]\n\t" \
            "s_addc_u32 %[anc], %[anc], 0\n\t" \
            "s_cmp_lg_u32 %[anc], 1\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_bfm_b32 %[anc], %[PB], 0\n\t" \
            "s_and_b32 %[aent], %[aent], %[anc]\n\t" \
            LZ_SC_WHY(8) "s_cmp_lg_u32 %[aent], %[rec]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n"
#define LZ_SC_J_XOUT , [cwa] "=&s"(cwa), [cwb] "=&s"(cwb), [e0] "=&s"(e0), [e1] "=&s"(e1), [e2] "=&s"(e2), [e3] "=&s"(e3), [tagl] "=&s"(tagl)
#define LZ_SC_J_XIN , [cbp] "s"(cbp)
#define LZ_SC_CAT(F, X) LZ_SC_##F##_##X
#define LZ_SC_ASM(WORD, F) \
        asm volatile( \
            "s_mov_b32 %[ncm], 0\n\t" "s_mov_b32 %[kind], 0\n\t" LZ_SC_WHY(10) \
            "v_add_u32_e32 %[w1], 64, %[lane]\n\t" \
            "v_subrev_u32_e32 %[dumv], %[WIN], %[w1]\n\t" \
            "v_min_u32_e32 %[w1], %[w1], %[dumv]\n\t" \
            "v_add_u32_e32 %[dumv], %[wdum], %[lane]\n\t" \
            "s_nop 3\n" \
            "Lsc_top_%=:\n\t" \
            LZ_SC_WHY(1) "s_cmp_gt_i32 %[i], %[ilim]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            LZ_SC_WHY(1) "s_cmp_gt_i32 %[rend], %[rlim]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            LZ_NC_LOADS_F \
            "global_load_dword %[hq], %[a0], %[qkl]\n\t"    /* the steps' mixed mal-mer hashes */ \
            "s_mov_b32 %[gap], %[NT]\n\t"                   /* (every tracking step counts: LZ_NC_SEEDS) */ \
            LZ_NC_ROUND_F(WORD) \
            LZ_NC_SEEDS \
            LZ_SC_WHY(2) "s_cbranch_scc0 Lsc_noseed_%=\n\t"              /* no seed candidate at all: the anchors of all the steps (find_event, told so) */ \
            LZ_NC_FIX \
            LZ_SC_WHY(2) "s_cmp_eq_u64 %[m], 0\n\t" \
            "s_cbranch_scc1 Lsc_noseed_%=\n\t" \
            LZ_SC_CAT(F, CANDLOAD) \
            "s_ff1_i32_b64 %[t0], %[m]\n"                   /* l: the first step with a seed candidate */ \
            "Lsc_sdl_%=:\n\t" \
            "v_readlane_b32 %[t1], %[qk], %[t0]\n\t"        /* the step's msl-mer */ \
            "s_add_i32 %[t2], %[t0], %[MRD]\n\t"            /* its window: the positions idx < l + mrd */ \
            "s_sub_i32 %[kc], %[t2], 64\n\t" \
            "s_max_i32 %[kc], %[kc], 0\n\t" \
            "s_min_u32 %[kb], %[t2], 64\n\t" \
            "s_sub_i32 %[kb], 64, %[kb]\n\t" \
            "v_cmp_eq_u32_e32 vcc, %[t1], %[rk1]\n\t" \
            "s_bfm_b64 %[m2], %[kc], 0\n\t" \
            "s_and_b64 %[m2], %[m2], vcc\n\t" \
            "v_cmp_eq_u32_e32 vcc, %[t1], %[rk0]\n\t" \
            "s_bcnt1_i32_b64 %[kc], %[m2]\n\t" \
            "s_lshl_b64 vcc, vcc, %[kb]\n\t" \
            "s_bcnt1_i32_b64 %[t1], vcc\n\t" \
            "s_add_i32 %[t2], %[t1], %[kc]\n\t" \
            "s_cmp_eq_u32 %[t2], 0\n\t" \
            "s_cbranch_scc1 Lsc_sfalse_%=\n\t"              /* no window position: the prefilter's false candidate */ \
            "s_cmp_lg_u32 %[t2], 1\n\t" \
            LZ_SC_WHY(3) "s_cbranch_scc1 Lsc_end_%=\n\t"                 /* several window positions: the longest, then the nearest (general path) */ \
            "s_ff1_i32_b64 %[t2], vcc\n\t" \
            "s_sub_i32 %[t2], %[t2], %[kb]\n\t" \
            "s_ff1_i32_b64 %[kc], %[m2]\n\t" \
            "s_add_i32 %[kc], %[kc], 64\n\t" \
            "s_cmp_lg_u32 %[t1], 0\n\t" \
            "s_cselect_b32 %[t2], %[t2], %[kc]\n\t"         /* idx */ \
            "s_add_i32 %[rec], %[rend], %[t2]\n\t"          /* the seed in the reference ... */ \
            "s_add_i32 %[cls], %[i], %[t0]\n\t"             /* ... and in the query */ \
            /* bounds as in the null chain's seed event: the 64 symbols behind the msl-mer inside the query and one strand */ \
            "s_sub_i32 %[t1], %[qend], %[CQ64]\n\t" \
            LZ_SC_WHY(4) "s_cmp_gt_i32 %[cls], %[t1]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_sub_i32 %[t1], %[rlim], %[C41M]\n\t" \
            "s_lshr_b32 %[t1], %[t1], 1\n\t"                /* L */ \
            "s_add_i32 %[t2], %[rec], %[MSL64]\n\t" \
            "s_cmp_le_i32 %[t2], %[t1]\n\t" \
            "s_cbranch_scc1 Lsc_sok_%=\n\t" \
            "s_add_i32 %[kb], %[t1], %[C2MRD]\n\t"          /* rc0 */ \
            "s_cmp_lt_i32 %[rec], %[kb]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_add_i32 %[kb], %[kb], %[t1]\n\t" \
            "s_cmp_gt_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n" \
            "Lsc_sok_%=:\n\t" \
            /* the seed's verification window and, with it, the tag words of the steps 0 .. l */ \
            "s_add_i32 %[t1], %[rec], %[MSL]\n\t" \
            "s_add_i32 %[t2], %[cls], %[MSL]\n\t" \
            "v_add_u32_e32 %[a0], %[t1], %[lane]\n\t" \
            "v_add_u32_e32 %[a1], %[t2], %[lane]\n\t" \
            "v_lshrrev_b32_e32 %[aq], 4, %[a0]\n\t" \
            "v_lshrrev_b32_e32 %[t], 4, %[a1]\n\t" \
            LZ_SC_CAT(F, TWADDR) \
            "v_lshlrev_b32_e32 %[aq], 2, %[aq]\n\t" \
            "v_lshlrev_b32_e32 %[t], 2, %[t]\n\t" \
            LZ_SC_CAT(F, TWADDR2) \
            "global_load_dword %[aq], %[aq], %[rt2]\n\t" \
            "global_load_dword %[t], %[t], %[qt2]\n\t" \
            LZ_SC_CAT(F, TWLOAD) \
            "v_and_b32_e32 %[a0], 15, %[a0]\n\t" \
            "v_and_b32_e32 %[a1], 15, %[a1]\n\t" \
            "v_lshlrev_b32_e32 %[a0], 1, %[a0]\n\t" \
            "v_lshlrev_b32_e32 %[a1], 1, %[a1]\n\t" \
            LZ_SC_CAT(F, VWAIT) \
            "v_lshrrev_b32_e32 %[aq], %[a0], %[aq]\n\t" \
            "v_lshrrev_b32_e32 %[t], %[a1], %[t]\n\t" \
            "v_xor_b32_e32 %[aq], %[aq], %[t]\n\t" \
            "v_and_b32_e32 %[aq], 3, %[aq]\n\t" \
            "v_cmp_ne_u32_e64 %[m], 0, %[aq]\n\t" \
            LZ_SC_CAT(F, TWSWAR) \
            "s_cmp_eq_u64 %[m], 0\n\t" \
            LZ_SC_WHY(5) "s_cbranch_scc1 Lsc_end_%=\n\t"                 /* 64 more symbols match: the general path measures on */ \
            "s_ff1_i32_b64 %[blen], %[m]\n\t" \
            "s_add_i32 %[blen], %[blen], %[MSL]\n\t" \
            LZ_SC_CAT(F, CANDMASK) \
            "s_bfm_b64 %[seed], %[t0], 0\n\t" \
            "s_and_b64 %[seed], %[seed], %[m2]\n\t" \
            "s_cmp_lg_u64 %[seed], 0\n\t" \
            LZ_SC_WHY(6) "s_cbranch_scc1 Lsc_anchor_%=\n\t"              /* ... at a step before the seed's: that anchor is the event (handed over) */ \
            "s_mov_b32 %[anc], 0\n\t" \
            "s_cmp_eq_u64 %[m2], 0\n\t" \
            "s_cbranch_scc1 Lsc_noanc_%=\n\t" \
            LZ_SC_CAT(F, ANCENTRY) \
            "Lsc_noanc_%=:\n\t" \
            /* The close match (as in the null chain): t0 = l, rec / cls = the seed in the reference / query, t1 = fq, fok = fr */ \
            "s_add_i32 %[t1], %[cls], %[blen]\n\t" \
            "s_add_i32 %[t2], %[t1], 64\n\t" \
            "s_sub_i32 %[kb], %[qend], %[C40M]\n\t"         /* Lq */ \
            "s_cmp_gt_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_add_i32 %[fok], %[rec], %[blen]\n\t" \
            "s_add_i32 %[t2], %[fok], 64\n\t"               /* the reference side: [rend, fr + 64) on one strand */ \
            "s_sub_i32 %[kb], %[rlim], %[C41M]\n\t" \
            "s_lshr_b32 %[kb], %[kb], 1\n\t"                /* L */ \
            "s_cmp_le_i32 %[t2], %[kb]\n\t" \
            "s_cbranch_scc1 Lsc_hull_%=\n\t" \
            "s_add_i32 %[kc], %[kb], %[C2MRD]\n\t"          /* rc0 */ \
            "s_cmp_lt_i32 %[rend], %[kc]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n\t" \
            "s_add_i32 %[kc], %[kc], %[kb]\n\t" \
            "s_cmp_gt_i32 %[t2], %[kc]\n\t" \
            "s_cbranch_scc1 Lsc_end_%=\n" \
            "Lsc_hull_%=:\n\t" \
            "s_sub_i32 %[kb], %[fok], %[rend]\n\t" \
            "s_min_i32 %[kb], %[kb], %[t0]\n\t"             /* to_scan */ \
            "s_sub_i32 %[kc], %[t0], %[kb]\n\t" \
            "s_sub_i32 %[gap], %[fok], %[kb]\n\t"           /* right diagonal: reference start */ \
            "s_add_i32 %[kc], %[kc], %[i]\n\t"              /*                 query start */ \
            "v_add_u32_e32 %[rk0], %[rend], %[lane]\n\t" \
            "v_add_u32_e32 %[rk1], %[i], %[lane]\n\t" \
            "v_add_u32_e32 %[qk], %[gap], %[lane]\n\t" \
            "v_add_u32_e32 %[a0], %[kc], %[lane]\n\t" \
            "v_add_u32_e32 %[a1], %[fok], %[lane]\n\t" \
            "v_add_u32_e32 %[aq], %[t1], %[lane]\n\t" \
            "v_lshrrev_b32_e32 %[rk0], 4, %[rk0]\n\t" \
            "v_lshrrev_b32_e32 %[rk1], 4, %[rk1]\n\t" \
            "v_lshrrev_b32_e32 %[qk], 4, %[qk]\n\t" \
            "v_lshrrev_b32_e32 %[a0], 4, %[a0]\n\t" \
            "v_lshrrev_b32_e32 %[a1], 4, %[a1]\n\t" \
            "v_lshrrev_b32_e32 %[aq], 4, %[aq]\n\t" \
            "v_lshlrev_b32_e32 %[rk0], 2, %[rk0]\n\t" \
            "v_lshlrev_b32_e32 %[rk1], 2, %[rk1]\n\t" \
            "v_lshlrev_b32_e32 %[qk], 2, %[qk]\n\t" \
            "v_lshlrev_b32_e32 %[a0], 2, %[a0]\n\t" \
            "v_lshlrev_b32_e32 %[a1], 2, %[a1]\n\t" \
            "v_lshlrev_b32_e32 %[aq], 2, %[aq]\n\t" \
            "s_nop 0\n\t" \
            "global_load_dword %[rk0], %[rk0], %[rt2]\n\t" \
            "global_load_dword %[rk1], %[rk1], %[qt2]\n\t" \
            "global_load_dword %[qk], %[qk], %[rt2]\n\t" \
            "global_load_dword %[a0], %[a0], %[qt2]\n\t" \
            "global_load_dword %[a1], %[a1], %[rt2]\n\t" \
            "global_load_dword %[aq], %[aq], %[qt2]\n\t" \
            "s_bfm_b64 %[seed], %[kb], 0\n\t"               /* the to_scan lanes of the gap's diagonals */ \
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
            /* the anchor at the seed's step, if there is one, must sit at the seed's position */ \
            "s_cmp_eq_u32 %[anc], 0\n\t" \
            "s_cbranch_scc1 Lsc_anok_%=\n\t" \
            LZ_SC_CAT(F, ANCCHECK) \
            "Lsc_anok_%=:\n\t" \
            /* left diagonal -> m = its matches */ \
            "v_add_u32_e32 %[t], %[rend], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[i], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[rk0], %[t], %[rk0]\n\t" \
            "v_lshrrev_b32_e32 %[rk1], %[bq], %[rk1]\n\t" \
            "v_xor_b32_e32 %[rk0], %[rk0], %[rk1]\n\t" \
            "v_and_b32_e32 %[rk0], 3, %[rk0]\n\t" \
            "v_cmp_eq_u32_e64 %[m], 0, %[rk0]\n\t" \
            /* right diagonal -> m2 */ \
            "v_add_u32_e32 %[t], %[gap], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[kc], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[qk], %[t], %[qk]\n\t" \
            "v_lshrrev_b32_e32 %[a0], %[bq], %[a0]\n\t" \
            "v_xor_b32_e32 %[qk], %[qk], %[a0]\n\t" \
            "v_and_b32_e32 %[qk], 3, %[qk]\n\t" \
            "v_cmp_eq_u32_e64 %[m2], 0, %[qk]\n\t" \
            /* forward chunk -> vcc = its MISmatches */ \
            "v_add_u32_e32 %[t], %[fok], %[lane]\n\t" \
            "v_add_u32_e32 %[bq], %[t1], %[lane]\n\t" \
            "v_lshlrev_b32_e32 %[t], 1, %[t]\n\t" \
            "v_lshlrev_b32_e32 %[bq], 1, %[bq]\n\t" \
            "v_and_b32_e32 %[t], 30, %[t]\n\t" \
            "v_and_b32_e32 %[bq], 30, %[bq]\n\t" \
            "v_lshrrev_b32_e32 %[a1], %[t], %[a1]\n\t" \
            "v_lshrrev_b32_e32 %[aq], %[bq], %[aq]\n\t" \
            "v_xor_b32_e32 %[a1], %[a1], %[aq]\n\t" \
            "v_and_b32_e32 %[a1], 3, %[a1]\n\t" \
            "v_cmp_ne_u32_e32 vcc, 0, %[a1]\n\t" \
            "s_and_b64 %[m], %[m], %[seed]\n\t"             /* Lm */ \
            "s_and_b64 %[m2], %[m2], %[seed]\n\t"           /* Rm */ \
            "s_mov_b64 %[seed], vcc\n\t"                    /* Bf */ \
            /* best split: lane s scores popc(Lm below s) + popc(Rm from s on) for s <= to_scan; the maximum is what counts */ \
            "s_bcnt1_i32_b64 %[t2], %[m2]\n\t" \
            "s_mov_b64 vcc, %[m]\n\t" \
            "s_nop 0\n\t" \
            "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t" \
            "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t" \
            "s_mov_b64 vcc, %[m2]\n\t" \
            "s_nop 0\n\t" \
            "v_mbcnt_lo_u32_b32 %[bq], vcc_lo, 0\n\t" \
            "v_mbcnt_hi_u32_b32 %[bq], vcc_hi, %[bq]\n\t" \
            "v_add_u32_e32 %[t], %[t2], %[t]\n\t" \
            "v_sub_u32_e32 %[t], %[t], %[bq]\n\t" \
            "v_cmp_ge_u32_e32 vcc, %[kb], %[lane]\n\t" \
            "s_nop 1\n\t" \
            "v_cndmask_b32_e32 %[t], 0, %[t], vcc\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_max_u32_dpp %[t], %[t], %[t] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_readlane_b32 %[t2], %[t], 63\n\t"            /* score */ \
            /* the forward extension's first chunk (try_extend_forward, parser.cpp:377-409) */ \
            "s_mov_b64 vcc, %[seed]\n\t" \
            "s_lshl_b32 %[gap], vcc_lo, %[AW1]\n\t" \
            "s_lshr_b32 %[kb], vcc_lo, %[AW1C]\n\t" \
            "s_lshl_b32 %[kc], vcc_hi, %[AW1]\n\t" \
            "s_or_b32 %[kc], %[kc], %[kb]\n\t" \
            "s_lshr_b32 %[kb], vcc_hi, %[AW1C]\n\t" \
            "v_mov_b32_e32 %[rk0], %[gap]\n\t" \
            "v_mov_b32_e32 %[rk1], %[kc]\n\t" \
            "v_mov_b32_e32 %[qk], %[kb]\n\t" \
            "v_cmp_gt_u32_e32 vcc, 32, %[lane]\n\t" \
            "v_alignbit_b32 %[a0], %[rk1], %[rk0], %[lane]\n\t" \
            "v_alignbit_b32 %[a1], %[qk], %[rk1], %[lane]\n\t" \
            "v_cndmask_b32_e32 %[a0], %[a1], %[a0], vcc\n\t" \
            "v_and_b32_e32 %[a0], %[AWM], %[a0]\n\t" \
            "v_bcnt_u32_b32 %[a1], %[a0], 0\n\t" \
            "v_and_b32_e32 %[a0], %[ARM], %[a0]\n\t" \
            "v_cmp_lt_u32_e32 vcc, %[AM], %[a1]\n\t"            /* brk */ \
            "v_cmp_eq_u32_e64 %[m], 0, %[a0]\n\t"           /* qual */ \
            "s_nop 0\n\t" \
            "s_cmp_eq_u64 vcc, 0\n\t" \
            LZ_SC_WHY(9) "s_cbranch_scc1 Lsc_extgo_%=\n\t"               /* no break inside the chunk: gap and match committed, the extension handed over */ \
            "s_ff1_i32_b64 %[kb], vcc\n\t" \
            LZ_SC_WHY(9) "s_cmp_ge_u32 %[kb], 63\n\t" \
            "s_cbranch_scc1 Lsc_extgo_%=\n\t" \
            "s_add_i32 %[kb], %[kb], 1\n\t" \
            "s_bfm_b64 %[m2], %[kb], 0\n\t" \
            "s_and_b64 %[m], %[m], %[m2]\n\t"               /* qualifying symbols up to the break */ \
            "s_mov_b32 %[kb], 0\n\t"                         /* e */ \
            "s_mov_b32 %[kc], 0\n\t"                         /* mm */ \
            "s_cbranch_scc0 Lsc_sext_%=\n\t" \
            "s_flbit_i32_b64 %[kb], %[m]\n\t" \
            "s_sub_i32 %[kb], 64, %[kb]\n\t"                /* e = the last qualifying symbol + 1 */ \
            "s_bfm_b64 %[m2], %[kb], 0\n\t" \
            "s_and_b64 %[m2], %[m2], %[seed]\n\t" \
            "s_bcnt1_i32_b64 %[kc], %[m2]\n"                 /* mm: the mismatches among its e symbols */ \
            "Lsc_sext_%=:\n\t" \
            /* commit: the open region grows by the gap (score matches, l - score literals), the match, the extension */ \
            "s_add_i32 %[cl], %[cl], %[t2]\n\t" \
            "s_add_i32 %[cl], %[cl], %[blen]\n\t" \
            "s_add_i32 %[cl], %[cl], %[kb]\n\t" \
            "s_sub_i32 %[cl], %[cl], %[kc]\n\t" \
            "s_add_i32 %[clit], %[clit], %[t0]\n\t" \
            "s_sub_i32 %[clit], %[clit], %[t2]\n\t" \
            "s_add_i32 %[clit], %[clit], %[kc]\n\t" \
            "s_add_i32 %[i], %[t1], %[kb]\n\t" \
            "s_add_i32 %[rend], %[fok], %[kb]\n\t" \
            "s_add_i32 %[ncm], %[ncm], 1\n\t" \
            "s_branch Lsc_top_%=\n" \
            "Lsc_sfalse_%=:\n\t" \
            LZ_SC_WHY(2) "s_bitset0_b64 %[m], %[t0]\n\t" \
            "s_cmp_eq_u64 %[m], 0\n\t" \
            "s_cbranch_scc1 Lsc_noseed_%=\n\t" \
            "s_ff1_i32_b64 %[t0], %[m]\n\t" \
            "s_branch Lsc_sdl_%=\n" \
            "Lsc_noseed_%=:\n\t" \
            "s_mov_b32 %[kind], 3\n\t" \
            "s_branch Lsc_end_%=\n" \
            "Lsc_anchor_%=:\n\t"                           /* the first step with an anchor candidate and its hash */ \
            "s_ff1_i32_b64 %[t0], %[seed]\n\t" \
            "v_readlane_b32 %[t1], %[hq], %[t0]\n\t" \
            "s_mov_b32 %[kind], 2\n\t" \
            "s_branch Lsc_end_%=\n" \
            "Lsc_extgo_%=:\n\t"                            /* the gap (score matches, l - score literals) and the match; Bf = seed */ \
            "s_add_i32 %[cl], %[cl], %[t2]\n\t" \
            "s_add_i32 %[cl], %[cl], %[blen]\n\t" \
            "s_add_i32 %[clit], %[clit], %[t0]\n\t" \
            "s_sub_i32 %[clit], %[clit], %[t2]\n\t" \
            "s_mov_b32 %[i], %[t1]\n\t" \
            "s_mov_b32 %[rend], %[fok]\n\t" \
            "s_mov_b32 %[kind], 1\n" \
            "Lsc_end_%=:\n\t" \
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
            "s_nop 4" \
            : LZ_SC_WHY_OPERAND [i] "+s"(i), [rend] "+s"(r_end), [cl] "+s"(cl), [clit] "+s"(clit), [ncm] "=&s"(ncm), \
              [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [kb] "=&s"(kb), [kc] "=&s"(kc), [gap] "=&s"(gap), [rec] "=&s"(rec), [cls] "=&s"(cls), \
              [fok] "=&s"(fok), [blen] "=&s"(blen), [anc] "=&s"(anc), [aent] "=&s"(aent), [kind] "=&s"(kind), [m] "=&s"(m), [seed] "=&s"(seed), [m2] "=&s"(m2), [pmk] "=&s"(pmk), \
              [rk0] "=&v"(rk0), [rk1] "=&v"(rk1), [qk] "=&v"(qk), [hq] "=&v"(hq), [a0] "=&v"(a0), [a1] "=&v"(a1), [aq] "=&v"(aq), [t] "=&v"(t), [bq] "=&v"(bq), \
              [w1] "=&v"(w1), [dumv] "=&v"(dumv) LZ_SC_CAT(F, XOUT) \
            : [ilim] "s"(ilim_u), [rlim] "s"(rlim_u), [qks] "s"(qks), [rks] "s"(rks), [qkl] "s"(qkl), [rt2] "s"(rt2), [qt2] "s"(qt2), [bkp] "s"(bkp), \
              [qend] "s"(qend), [wdum] "s"((int)SEED_BM_WORDS), [TB] "s"(tbits), [PB] "s"(pbits), [TAGM] "s"(tagm) LZ_SC_CAT(F, XIN), \
              [lane] "v"(lane), [ldsb] "v"(ldsb), [zero] "v"(zero), [one] "v"(one), \
              [MRD] "n"(MRD), [NT] "n"(NT), [WIN] "s"((int)WIN), [NR1] "n"(WIN - 64), [MSL] "n"(MSL), [MSL64] "n"(MSL + 64), \
              [C41M] "n"(3 * MRD + 1 - WIN - MSL), [C40M] "n"(MRD - MSL), [CQ64] "n"(MRD + 64), [C2MRD] "n"(2 * MRD), \
              [AW1] "n"(AW - 1), [AW1C] "n"(33 - AW), [AWM] "n"((1 << AW) - 1), [ARM] "n"(((1 << AR) - 1) << (AW - AR)), [AM] "n"(AM), [KS5] "n"(2 * MSL + 5), [KS] "n"(2 * MSL) \
            : "vcc", "scc", "memory");
        if constexpr (JOIN) {
            if constexpr (MSL == 9) { LZ_SC_ASM(LZ_NC_WORD9, J) }
            else if constexpr (MSL == 8) { LZ_SC_ASM(LZ_NC_WORD8, J) }
            else { LZ_SC_ASM(LZ_NC_WORD7, J) }
        } else {
            if constexpr (MSL == 9) { LZ_SC_ASM(LZ_NC_WORD9, P) }
            else if constexpr (MSL == 8) { LZ_SC_ASM(LZ_NC_WORD8, P) }
            else { LZ_SC_ASM(LZ_NC_WORD7, P) }
        }
#undef LZ_SC_ASM
        last_src = -1;
        Bf = seed; adv = t0; hqa = (u32)t1;
        round_no_seed = kind == 3;
        LZ_PSN(23, ncm);
#ifdef LZANI_PATH_STATS
        pw[why < 12 ? why : 0] += 1;
#endif
#undef LZ_SC_WHY
#undef LZ_SC_WHY_OPERAND
        return ncm;
    }
#undef LZ_NC_LOADS_F
#undef LZ_NC_LOADS_X
#undef LZ_NC_ROUND_X
#undef LZ_NC_ROUND_F
#undef LZ_NC_ASM
#undef LZ_NC_ASM_X
#undef LZ_NC_SPLITCHK
#undef LZ_NC_SPLITGEN
#undef LZ_NC_SPLITIN
#undef LZ_NC_WORD7
#undef LZ_NC_WORD7N
#undef LZ_NC_WORD9
#undef LZ_NC_WORD8
#undef LZ_NC_FTURN
#undef LZ_NC_FIX
#undef LZ_NC_SEEDS
#undef LZ_NC_COMMIT

    __device__ __forceinline__ bool find_event(int i, int n, bool trk, int r_end, int lit, int& adv, int& bpos, int& blen)
    {
        if (!FAST || !BK || P.mqd + P.mrd > 128)                     // other index forms, wide seed windows: rounds
            return find_event_round(i, n, trk, r_end, lit, adv, bpos, blen);
        last_src = -1;
        LZ_PS(0);
        const bool no_seed = round_no_seed;
        round_no_seed = false;
        // tracking steps of this call (the machine clears trk once lit > mqd); one lane per tracking step
        const int nt = trk ? imin(imin(n, P.mqd - lit + 1), 64) : 0;
        if (q_head < q_cnt && __builtin_amdgcn_readlane(a_pos, q_head) < i) drop_before(i);
        const bool only = CHAIN && JOIN && refill_only;              // (see null_chain)
        refill_only = false;
        if (only && q_head < q_cnt) { scan_pos = i; restart_at = i; }        // the queue's tail: detected again, with what follows it
        bool refill_now = only || scan_pos < i + nt;                 // the queue must cover the tracking steps
        bool round_done = !trk;
        u32 rk0 = KM_INVALID, rk1 = KM_INVALID, qk = KM_INVALID;
        u64 seedmask = 0;
        const bool pre = CHAIN && pre_round;                         // null_chain has done this call's tracking round (and found
        pre_round = false;                                           // a seed candidate or a candidate that is not plain)
        if (pre) {
            seedmask = pre_seed; rk0 = pre_rk0; rk1 = pre_rk1; qk = pre_qk; round_done = true;
            // The chain's usual reason to hand a round back: a seed candidate at a step before the next queued anchor, one
            // window position carrying the step's k-mer -- that step is the event (no anchor there to arbitrate with, and
            // a seed is at least msl long): straight down, without the loop below.
            if (seedmask) {
                const int ls = ctz64(seedmask);
                const int la = q_head < q_cnt ? __builtin_amdgcn_readlane(a_pos, q_head) - i : 64;
                if (ls < la) {
                    u64 d0, d1;
                    seed_candidates((u32)__builtin_amdgcn_readlane((int)qk, ls), lit + ls + P.mrd, rk0, rk1, d0, d1);
                    if (popc64(d0) + popc64(d1) == 1) {
                        const int idx = d0 ? ctz64(d0) : 64 + ctz64(d1);
                        adv = ls; bpos = r_end + idx; blen = wave_equal_len(bpos, i + ls, P.msl);
                        LZ_PS(1);
                        return true;
                    }
                }
            }
        }
        else if (__builtin_expect(!refill_now, 1)) {
            // The common call, straight down, no loop: one round over the tracking steps (close seeds as in
            // find_event_round) and, when none of them has a seed candidate (four rounds out of five of an unrelated
            // pair), the next queued candidate.  A tracking step can then only hit through its anchor, which wins the
            // arbitration unopposed unless it sits at reference position 0 (quirk Q1): a PLAIN candidate is the event
            // whether it is still a tracking step or already a lost one.
            if (trk) { seedmask = track_round(i, nt, r_end, lit, rk0, rk1, qk); round_done = true; }
            if (__builtin_expect((seedmask == 0) & (lit + nt > P.mqd) & (q_head < q_cnt), 1)) {
                const int plen = len_at(q_head);
                if (__builtin_expect(plen > 0, 1)) {
                    adv = __builtin_amdgcn_readlane(a_pos, q_head) - i;
                    bpos = (int)((u32)__builtin_amdgcn_readlane((int)a_ref, q_head)); blen = plen;
                    last_src = q_head++;
                    LZ_PS(2);
                    return true;
                }
            }
        }
        int pos = i, guard = 0;
        bool merge_done = false;
        if (trk & refill_now & !only) {
            // A tracking round the queue does not cover (the scan has jumped over it: an extension moved, a related
            // stretch) runs LIGHT, on its own: its anchors are detected for the tracking steps alone and verified by the
            // wave one at a time, only at the steps the round really reaches (as find_event_round does) -- resolving 64
            // candidates per event where every position is one would be wasted.  Kept apart from the loop below so that
            // neither shapes the other's registers.
            // (the steps' mal-mer hashes with the round's own loads: the anchor of a step then needs no fetch of its own
            // before its bucket -- one memory round trip less per event of a related stretch)
            const u32 hqv = qkL[(u32)(i + lane)];
            LZ_PS(3);
            seedmask = no_seed ? 0 : track_round(i, nt, r_end, lit, rk0, rk1, qk);
            // The anchors of the steps are probed only as far as they can matter: up to the first step with a seed
            // candidate (a seed is an event: nothing behind it is reached), else the first eight steps, the rest only if
            // none of those hits.  In a related stretch the event is at the first steps, and each probe spared is a
            // 128-byte line of a table that 20 references per XCD share an L2 with (this kernel waits for memory).
            int known = JOIN ? nt : imin(nt, seedmask ? ctz64(seedmask) + 1 : 8);
            u64 lightmask = detect_steps(i, known);
            for (int it = 0; it < 130; ++it) {
                const int ls = seedmask ? ctz64(seedmask) : 64;
                const int la = lightmask ? ctz64(lightmask) : 64;
                const int l = imin(ls, la);
                if (l >= known && known < nt) { lightmask |= detect_steps(i, nt, known); known = nt; continue; }
                if (l >= 64) break;
                const int qp = i + l;
                int ap = 0, al = 0;
                if (la == l) {
                    lightmask &= lightmask - 1;
                    anchor_by_wave((u32)__builtin_amdgcn_readlane((int)hqv, l), qp, ap, al);
                }
                int sp = 0, sl = 0;
                if (ls == l) {
                    seedmask &= seedmask - 1;
                    const int ref_pred = r_end + lit + l;
                    u64 d0 = 0, d1 = 0;
                    const u32 qkl = (u32)__builtin_amdgcn_readlane((int)qk, l);
                    if (qkl != KM_INVALID) seed_candidates(qkl, lit + l + P.mrd, rk0, rk1, d0, d1);
                    while (d0 | d1) {
                        int idx;
                        if (d0) { idx = ctz64(d0); d0 &= d0 - 1; }
                        else { idx = 64 + ctz64(d1); d1 &= d1 - 1; }
                        seed_consider(r_end + idx, wave_equal_len(r_end + idx, qp, P.msl), ref_pred, sp, sl);
                    }
                }
                arbitrate(P, R.len, lit + l, ap, al, sp, sl);
                if (sl >= P.msl) { adv = l; bpos = sp; blen = sl; LZ_PS(4); return true; }
            }
            LZ_PS(5);
            if (lit + nt <= P.mqd) { adv = nt; return false; }       // mqd = 64: one more tracking step in the next call
            // no tracking step hits: on in lost mode; whatever the queue still holds of these steps goes
            pos = i + nt;
            if (q_head < q_cnt) drop_before(pos);
            if (scan_pos < pos) scan_pos = pos;
            refill_now = false;
            round_done = true; merge_done = true;
        }
        // everything else: refills, seed candidates to verify, candidates the lanes could not settle, the end of the query
        for (;;) {
            if (refill_now) {                                        // (the one call site of refill: it is big)
                if (++guard > (1 << 24)) { LZ_GUARD_TRIP(7); break; }
                refill(pos);
                refill_now = false;
                if (only) { adv = 0; return false; }                 // (nothing looked at: the machine comes back at once)
            }
            if (!round_done) { seedmask = track_round(i, nt, r_end, lit, rk0, rk1, qk); round_done = true; }
            if (!merge_done) {
                merge_done = true;
                if (seedmask != 0 || (trk && lit + nt <= P.mqd)) {
                    // tracking steps with a seed candidate and / or a queued anchor, in step order
                    for (int it = 0; it < 130; ++it) {
                        const int ls = seedmask ? ctz64(seedmask) : 64;
                        int la = 64;
                        if (q_head < q_cnt) la = imin(64, __builtin_amdgcn_readlane(a_pos, q_head) - i);
                        if (la >= nt) la = 64;
                        const int l = imin(ls, la);
                        if (l >= 64) break;
                        const int qp = i + l;
                        int ap = 0, al = 0, src = -1;
                        if (la == l) { anchor_of(q_head, qp, ap, al); src = q_head; ++q_head; }
                        int sp = 0, sl = 0;
                        if (ls == l) {
                            seedmask &= seedmask - 1;
                            const int ref_pred = r_end + lit + l;
                            u64 d0 = 0, d1 = 0;
                            const u32 qkl = (u32)__builtin_amdgcn_readlane((int)qk, l);
                            if (qkl != KM_INVALID) seed_candidates(qkl, lit + l + P.mrd, rk0, rk1, d0, d1);
                            while (d0 | d1) {
                                int idx;
                                if (d0) { idx = ctz64(d0); d0 &= d0 - 1; }
                                else { idx = 64 + ctz64(d1); d1 &= d1 - 1; }
                                seed_consider(r_end + idx, wave_equal_len(r_end + idx, qp, P.msl), ref_pred, sp, sl);
                            }
                        }
                        arbitrate(P, R.len, lit + l, ap, al, sp, sl);
                        if (sl >= P.msl) {
                            adv = l; bpos = sp; blen = sl;
                            if (src >= 0) note_src(src, ap, al, sp, sl);
                            LZ_PS(6);
                            return true;
                        }
                    }
                    if (trk && lit + nt <= P.mqd) { adv = nt; return false; }   // mqd = 64: one more tracking step in the next call
                    pos = i + nt;
                }
            }
            // jump to the next queued candidate (lost steps; tracking steps of a round without seed candidates);
            // every queued position is a step (the scan stops at iend)
            if (__builtin_expect(q_head >= q_cnt, 0)) {
                if (pos >= i + n || scan_pos >= iend) break;
                refill_now = true;
                continue;
            }
            const int qp = __builtin_amdgcn_readlane(a_pos, q_head);
            const int plen = len_at(q_head);
            if (__builtin_expect(plen > 0, 1)) {                     // a plain candidate: the event, whatever kind of step it is
                adv = qp - i; bpos = (int)((u32)__builtin_amdgcn_readlane((int)a_ref, q_head)); blen = plen;
                last_src = q_head++;
                LZ_PS(7);
                return true;
            }
            int ap, al;
            anchor_of(q_head, qp, ap, al);
            ++q_head;
            if (__builtin_expect((al >= P.msl) & ((qp - i >= nt) | (ap != 0)), 1)) {
                adv = qp - i; bpos = ap; blen = al;
                note_src(q_head - 1, ap, al, ap, al);
                return true;
            }
            pos = qp + 1;
        }
        adv = n;
        return false;
    }
    __device__ __forceinline__ ExtMasks ext_scan(u64 prevB, u64 B, int n) const
    {
        bool b, q;
        ext_lane(prevB, B, lane, n, P.aw, P.am, P.ar, b, q);
        ExtMasks m;
        m.brk = wballot(b);
        m.qual = wballot(q);
        return m;
    }
    __device__ __forceinline__ int best_split(u64 Lm, u64 Rm, int to_scan) const
    {
        // lane s scores split s; to_scan can be 64, so split 64 is scored by every lane too
        int key = -1;
        if (lane <= to_scan) key = (popc64(Lm & lowmask(lane)) + popc64(Rm >> lane)) * 128 + lane;
        if (to_scan == 64) key = imax(key, popc64(Lm) * 128 + 64);
        // the wave's maximum by DPP (prefix maxima inside the rows of 16, then the row broadcasts: lane 63 holds it) -- six
        // vector instructions instead of six round trips through the LDS crossbar (ds_bpermute)
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x111, 0xF, 0xF, false));
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x112, 0xF, 0xF, false));
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x114, 0xF, 0xF, false));
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x118, 0xF, 0xF, false));
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x142, 0xA, 0xF, false));
        key = imax(key, __builtin_amdgcn_update_dpp(-1, key, 0x143, 0xC, 0xF, false));
        return __builtin_amdgcn_readlane(key, 63) & 127;
    }
};

// The candidates of a whole pair by a JOIN (long genomes; k_join_keys).  The query's k-mer list is sorted by bucket, so
// the 64 lanes of a step read 64 consecutive keys and probe tag words that lie within a few hundred buckets of each
// other: a handful of consecutive 128-byte lines per step, each line of the reference's table fetched once per pair,
// against one line per query position for the probe form (5 Mbp: 81 G random probes per launch = 5.5 TB/s of line
// fetches, the whole cost of the step).  A hit sets the position's bit in the wave's candidate bitmap.  Runs before
// anything else of the pair is set up, so that its registers are its own.
__device__ __forceinline__ void join_candidates(const u32* __restrict__ tw, int kb, int dirbits, int posbits, u32 tagmask,
                                                const unsigned long long* __restrict__ keys, u32 n_keys,
                                                unsigned long long* __restrict__ cand_bits, int words, int lane)
{
    for (int k = lane; k < words; k += 64) cand_bits[k] = 0;
    const int tb = kb - dirbits;
    const u32 hmask = (u32)lowmask(kb), pmask = (u32)lowmask(posbits);
    u32* const bits32 = reinterpret_cast<u32*>(cand_bits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // the zeroes before the atomics of this wave
    for (u32 k0 = 0; k0 < n_keys; k0 += 128) {
        unsigned long long e[2];
        u32 w[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) { const u32 k = k0 + 64 * c + lane; e[c] = keys[k < n_keys ? k : n_keys - 1]; }
#pragma unroll
        for (int c = 0; c < 2; ++c) w[c] = tw[(((u32)(e[c] >> posbits)) & hmask) >> tb];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32 hq = ((u32)(e[c] >> posbits)) & hmask, pos = (u32)e[c] & pmask;
            const u32 x = w[c] ^ rep4(0x80u | (hq & tagmask));
            const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;
            if ((k0 + 64 * c + lane < n_keys) & ((z != 0) | (w[c] == TW_OVERFLOW))) atomicOr(&bits32[pos >> 5], 1u << (pos & 31));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");                 // the bits before the scan reads them
}

struct PairArgs {
    GenomeTab G;
    Params P;
    IndexGeom geo;
    const u32* dirz;
    const u32* ent;
    u64 dir_stride, ent_stride;
    const u32* bk;           // bucket tables (4 entries per bucket) or nullptr
    u64 bk_stride;
    const u32* tw;           // tag words (one per bucket) or nullptr
    u64 tw_stride;
    const u32* fl;           // presence filters (probe form with tag words; fl_stride 0 = one all-ones word for all)
    u64 fl_stride;
    u32 fmask;
    const u32* ref_ids;      // device, batch-relative rows
    const u64* row_off;      // device, batch-relative rows (+1), absolute pair offsets
    const u32* query_ids;    // device, absolute pair offsets, or nullptr for dense rows
    int* out;                // 3 ints per pair, absolute pair offsets
    // Work queues, one per XCD: queue x owns the batch rows qorder[qb[x] .. qb[x+1]); qcum is the
    // running pair count over qorder.  All waves of an XCD pull from that XCD's queue, so the 32 CUs
    // sharing one 4 MiB L2 work on the same reference (its 0.8 MB index stays L2-resident); an XCD
    // whose queue runs dry steals from the next one.  Placement is a speed matter only.
    const u32* qorder;
    const u64* qcum;
    u32 qb[NQUEUES + 1];
    unsigned long long* cursor;   // NQUEUES tickets counters
    // join form of candidate detection (long genomes; nullptr = probe form): per-genome k-mer lists sorted by bucket
    // (keys of genome g: skeys[soff[g] .. soff[g+1])) and one candidate bitmap of cbits_stride words per wave
    const unsigned long long* skeys;
    const u64* soff;                 // begin of genome g's sorted keys
    const u32* scnt;                 // their number
    unsigned long long* cbits;       // CAND 1: one bitmap per resident wave; CAND 2: one per pair of the batch, from pair cb_e0 on
    u64 cbits_stride;
    u64 cb_e0;
    lzani_region* reg_out;        // alignment instantiation only
    unsigned long long* reg_count;
    unsigned long long reg_cap;
    // batches of few, long pairs: the order a queue hands its tickets out in -- ticket t of the batch (counted through the
    // queues) stands for ticket (u32)torder[t] of the same queue, the pairs with the most anchor candidates first
    // (k_lpt_keys, lzani_kernels_cand.h); nullptr = as they come
    const unsigned long long* torder;
};

__device__ __forceinline__ u32 xcc_id()
{
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// Instantiations: FAST = per-genome k-mer words exist (mal, msl <= 15); NFREE = no genome of the
// context holds an N (the N mask is never consulted); DEFP = the LZ parameters are the reference's
// defaults (params.h:34-48), folded into the code as constants.
// ALN = also emit the regions of every pair (--out-alignment).
// JOIN = candidates by a join with sorted k-mer lists (long genomes; needs FAST and BK).
// One pair: row `lo` of the batch's queue order, its j-th query.  `flt` = the block's LDS copy of the reference's
// presence filter (LFLT instantiations).
// CAND: where the anchor candidates of a pair come from -- 0 = a probe per query position (refill), 1 = the wave's own
// join with the query's sorted k-mer list (long genomes), 2 = the pair's candidate bitmap made ahead by k_pm_cand
// (dense rows, lzani_kernels_cand.h); 1 and 2 share the bitmap form of refill (DevWave's JOIN).
template <bool FAST, bool NFREE, int DEFP, bool ALN, bool BK, int CAND, bool LFLT>
__device__ __forceinline__ void pair_body(const PairArgs& a, u32 lo, u32 j, int lane, u32* lds, const u32* flt)
{
    constexpr bool JOIN = CAND != 0;
    const Params Pk = DEFP ? folded_params(DEFP) : a.P;
    const u32 slot = a.qorder[lo];
    const u32 r = a.ref_ids[slot];
    const u64 e = a.row_off[slot] + j;
    const u32 q = a.query_ids ? a.query_ids[e] : j + (j >= r ? 1u : 0u);

    unsigned long long* cand_bits = nullptr;
    if (CAND == 2) cand_bits = a.cbits + (e - a.cb_e0) * a.cbits_stride;
    if (CAND == 1) {
        cand_bits = a.cbits + (u64)(blockIdx.x * 4 + (threadIdx.x >> 6)) * a.cbits_stride;
        join_candidates(a.tw + slot * a.tw_stride, a.geo.kb, a.geo.dirbits, a.geo.posbits, a.geo.tagmask,
                        a.skeys + a.soff[q], a.scnt[q], cand_bits, ((a.G.L[q] + Pk.mrd) >> 6) + 8, lane);
    }
    const int Lr = a.G.L[r], Lq = a.G.L[q];
    const u64 ro = a.G.nmoff[r], qo = a.G.nmoff[q];
    const int T = ref_text_len(Lr, Pk.mrd), D = Lq + Pk.mrd;
    IndexView iv;
    iv.dirz = a.dirz + slot * a.dir_stride;
    iv.ent = a.ent + slot * a.ent_stride;
    iv.kb = a.geo.kb; iv.dirbits = a.geo.dirbits; iv.posbits = a.geo.posbits; iv.tagmask = a.geo.tagmask;
    iv.bk = a.bk ? a.bk + slot * a.bk_stride : nullptr;
    iv.tw = a.tw ? a.tw + slot * a.tw_stride : nullptr;
    const bool nfree = NFREE ? true : !(a.G.hasN[r] | a.G.hasN[q]);
#if defined(LZANI_STAMPS) || defined(LZANI_PATH_STATS)
    constexpr int CHAIN = (FAST && BK && !ALN && !LFLT) ? chain_of(DEFP, NFREE) : 0;     // (diagnostic build: the block kernel's stamps do not stay scalar around the hand-written loop)
#else
    constexpr int CHAIN = (FAST && BK && !ALN) ? chain_of(DEFP, NFREE) : 0;
#endif
    DevWave<FAST, BK, JOIN, CHAIN, LFLT> w{Pk, ref_view(a.G.t2 + 2 * ro, a.G.nm + ro, Lr, Pk.mrd, nfree),
                    qry_view(a.G.t2 + 2 * qo, a.G.nm + qo, Lq, Pk.mrd, nfree), iv, lane,
                    lds,
                    FAST ? a.G.kmS + 64 * ro : nullptr, FAST ? a.G.kmL + 64 * qo : nullptr,
                    FAST ? a.G.kmS + 64 * qo : nullptr, a.reg_out, a.reg_count, a.reg_cap, e};
    w.iend = D - Pk.msl;
    w.cand_bits = cand_bits;
    w.flt = flt; w.fmask = a.fmask;                      // (the block kernel passes the mask of its LDS copy)
    PairMachine<DevWave<FAST, BK, JOIN, CHAIN, LFLT>, ALN> m(w, Pk, T, D);
    int res[3];
#ifdef LZANI_STAMPS
    for (int k = 0; k < 8; ++k) w.acc[k] = 0;
    w.cur = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w.t0) :: "memory");
#endif
#if defined(LZANI_EXP) && LZANI_EXP == 3                      // diagnostic build: the join alone, no scan
    res[0] = res[1] = res[2] = 0;
    if (!JOIN)
#endif
#ifdef LZANI_EXP_REFILLS                                      // diagnostic build: the refills of the pair alone (queue after queue), no scan
    {
        res[0] = res[1] = res[2] = 0;
        if constexpr (JOIN) {
            int guard = 0;
            while (w.scan_pos < w.iend && ++guard < 100000) { w.refill(w.scan_pos); res[0] += w.q_cnt; }
        }
    }
    if (!JOIN)
#endif
#ifdef LZANI_PHASE_TIME
    const unsigned long long pt_p0 = w.pt_now();
#endif
    m.run(res);
#ifdef LZANI_PHASE_TIME
    {
        const unsigned long long pt_all = w.pt_now() - pt_p0;
        atomicAdd(&g_phase_time[0], lane == 0 ? pt_all : 0ULL);
        atomicAdd(&g_phase_time[1], lane == 0 ? w.pt_chain : 0ULL);
        atomicAdd(&g_phase_time[2], lane == 0 ? w.pt_refill : 0ULL);
        atomicAdd(&g_phase_time[3], lane == 0 ? 1ULL : 0ULL);
    }
#endif
#ifdef LZANI_PATH_STATS
    for (int k = 0; k < 24; ++k) atomicAdd(&g_path_stats[k], lane == 0 ? (unsigned long long)w.ps[k] : 0ULL);
    for (int k = 0; k < 12; ++k) atomicAdd(&g_path_stats[24 + k], lane == 0 ? (unsigned long long)w.pw[k] : 0ULL);
#endif
#ifdef LZANI_CHAIN_STATS
    for (int k = 0; k < 8; ++k) atomicAdd(&g_chain_stats[k], lane == 0 ? (unsigned long long)w.st[k] : 0ULL);
    for (int k = 0; k < 4; ++k) atomicAdd(&g_chain_stats[8 + k], lane == 0 ? w.stc[k] : 0ULL);
    for (int k = 0; k < 4; ++k) atomicAdd(&g_chain_stats[12 + k], lane == 0 ? (unsigned long long)w.sr[k] : 0ULL);
    for (int k = 0; k < 8; ++k) atomicAdd(&g_chain_stats[16 + k], lane == 0 ? (unsigned long long)w.sw[k] : 0ULL);
#endif
#ifdef LZANI_STAMPS
    w.stamp(0);
    for (int k = 0; k < 8; ++k) atomicAdd(&g_stamp_acc[k], lane == 0 ? w.acc[k] : 0ULL);      // (no lane-dependent branch in this loop)
#endif
    int* o = a.out + 3 * e;          // every lane stores the same wave-uniform values
    o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
}

// row of the queue's pair ticket tk (absolute in qcum): the last row of [rb, re) with qcum[row] <= tk
__device__ __forceinline__ u32 row_of_ticket(const u64* __restrict__ qcum, u32 rb, u32 re, u64 tk)
{
    u32 lo = rb, hi = re;
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (qcum[mid] <= tk) lo = mid; else hi = mid;
    }
    return lo;
}

// Instantiations: FAST = per-genome k-mer words exist (mal, msl <= 15); NFREE = no genome of the
// context holds an N (the N mask is never consulted); DEFP = the LZ parameters are the reference's
// defaults (params.h:34-48), folded into the code as constants.
// ALN = also emit the regions of every pair (--out-alignment).
// JOIN = candidates by a join with sorted k-mer lists (long genomes; needs FAST and BK).
#ifndef LZANI_WAVES_PER_SIMD
#define LZANI_WAVES_PER_SIMD 8                 // (occupancy experiments: the register budget of the pair kernel)
#endif
template <bool FAST, bool NFREE, int DEFP, bool ALN, bool BK, int CAND>
__device__ __forceinline__ void pairs_loop(const PairArgs& a)
{
    const int lane = threadIdx.x & 63;
    __shared__ u32 s_seed[4][SEED_LDS_WORDS];
    u32* const lds = s_seed[threadIdx.x >> 6];
    for (int k = lane; k < SEED_LDS_WORDS; k += 64) lds[k] = 0;
    u32 qx = xcc_id() % NQUEUES, dry = 0;
    for (;;) {
        // One ticket per wave.  NB: this is the only lane-dependent branch of the persistent loop.
        // A second `if (lane == 0)` at the loop tail (the result store) let the compiler thread the
        // two branches across the back-edge and split lane 0 from lanes 1..63, which then spun on a
        // dead lane's ticket; the store below is therefore done by every lane.
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(&a.cursor[qx], 1ULL);
        const u32 tlo = __builtin_amdgcn_readfirstlane((u32)t);
        const u32 thi = __builtin_amdgcn_readfirstlane((u32)(t >> 32));
        const u32 rb = a.qb[qx], re = a.qb[qx + 1];
        const u64 tk = a.qcum[rb] + (((u64)thi << 32) | tlo);
        if (tk >= a.qcum[re]) {                   // this queue is dry: move on, leave after NQUEUES dry queues
            if (++dry >= NQUEUES) break;
            qx = (qx + 1) % NQUEUES;
            continue;
        }
        u64 tk2 = tk;
        if (CAND == 2 && a.torder) tk2 = a.qcum[0] + (u32)a.torder[tk - a.qcum[0]];
        const u32 lo = row_of_ticket(a.qcum, rb, re, tk2);
        pair_body<FAST, NFREE, DEFP, ALN, BK, CAND, false>(a, lo, (u32)(tk2 - a.qcum[lo]), lane, lds, nullptr);
    }
}
template <bool FAST, bool NFREE, int DEFP, bool ALN = false, bool BK = false, int CAND = 0>
__global__ void __launch_bounds__(256, LZANI_WAVES_PER_SIMD) k_pairs(PairArgs a)
{
    pairs_loop<FAST, NFREE, DEFP, ALN, BK, CAND>(a);
}

// The same pairs by BLOCKS of 16 waves that stay on one reference at a time (probe form with tag words, rows of
// hundreds of pairs): a block draws a chunk of BLK_CHUNK pair tickets from its XCD's queue and walks it row by row;
// per row segment it copies the reference's presence filter (k_idx_filter) into LDS, its waves pull the segment's
// pairs one by one, and a barrier ends the segment.  The filter answers "no such mal-mer in the reference" for three
// query positions out of four without leaving the CU: the tag-word probes -- random 4-byte reads that cost a
// 128-byte L2 line each, the traffic that bounds the viral pair kernel at the L2 -- are made for the rest only.
// Every barrier is reached by all 16 waves the same number of times: chunk and segment bounds are block-uniform.
#ifndef LZANI_BLK_CHUNK
#define LZANI_BLK_CHUNK 256
#endif
enum { BLK_WAVES = 16, BLK_CHUNK = LZANI_BLK_CHUNK, BLK_CHUNK_MIN = 32 };   // chunk of pair tickets: at most / at least
template <bool NFREE, bool DEFP>
__global__ void __launch_bounds__(64 * BLK_WAVES, 8) k_pairs_blk(PairArgs a, u32 fwords, u32 fold, u32* __restrict__ blkctr)
{
    extern __shared__ u32 s_dyn[];                 // BLK_WAVES x SEED_LDS_WORDS, then the filter (fwords)
    const int lane = threadIdx.x & 63;
    u32* const lds = s_dyn + (threadIdx.x >> 6) * SEED_LDS_WORDS;
    u32* const flt = s_dyn + BLK_WAVES * SEED_LDS_WORDS;
    u32* const ctl = s_dyn;                        // three words of wave 0's seed bitmap while the block is between segments (zeroed before wave 0 goes on)
    for (int k = lane; k < SEED_LDS_WORDS; k += 64) lds[k] = 0;
    u32 qx = xcc_id() % NQUEUES, dry = 0, have_slot = 0xFFFFFFFFu;
    for (;;) {
        __syncthreads();                           // (ctl is free: nobody is inside a pair)
        const u32 rb = a.qb[qx], re = a.qb[qx + 1];
        const u64 base = a.qcum[rb], total = a.qcum[re] - base;
        if (threadIdx.x == 0) {
            // guided chunks: a share of what the queue has left (read without a lock: any size is a valid chunk), so that
            // the barriers are few while the queue is long and the blocks finish together when it ends
            const unsigned long long seen = __hip_atomic_load(&a.cursor[qx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 left = total > seen ? total - seen : 0;
            u64 ch = left / (2 * (gridDim.x / NQUEUES + 1));
            ch = ch < BLK_CHUNK_MIN ? BLK_CHUNK_MIN : ch > BLK_CHUNK ? BLK_CHUNK : ch;
            const unsigned long long t = atomicAdd(&a.cursor[qx], (unsigned long long)ch);
            ctl[0] = (u32)t; ctl[1] = (u32)(t >> 32); ctl[2] = (u32)ch;
        }
        __syncthreads();
        const u64 t0 = ((u64)ctl[1] << 32) | ctl[0];
        const u64 chunk = ctl[2];
        if (t0 >= total) {                         // this queue is dry (block-uniform): move on, leave after NQUEUES dry queues
            if (++dry >= NQUEUES) break;
            qx = (qx + 1) % NQUEUES;
            continue;
        }
        const u64 t1 = t0 + chunk < total ? t0 + chunk : total;
        u32 lo = row_of_ticket(a.qcum, rb, re, base + t0);
        for (u64 cur = t0; cur < t1; ++lo) {       // the rows of the chunk (rows without pairs fall through)
            const u64 row_end = a.qcum[lo + 1] - base;
            const u64 seg_end = row_end < t1 ? row_end : t1;
            if (seg_end <= cur) continue;
            const u32 slot = a.qorder[lo];
            if (slot != have_slot) {                // (block-uniform) the LDS copy still holds the filter of the last segment's reference
                const u32* const gf = a.fl + (u64)slot * a.fl_stride;
                // (a filter of fwords << fold words folded onto fwords: bit b of the copy = OR of the bits b + i * 32 * fwords)
                for (u32 k = threadIdx.x; k < fwords; k += 64 * BLK_WAVES) {
                    u32 v = gf[k];
                    for (u32 i = 1; i < (1u << fold); ++i) v |= gf[k + i * fwords];
                    flt[k] = v;
                }
                have_slot = slot;
            }
            if (threadIdx.x == 0) { blkctr[blockIdx.x] = 0; __threadfence(); }
            __syncthreads();
            if (threadIdx.x == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; }       // (every wave has read them before the barrier)
            const u32 n_seg = (u32)(seg_end - cur);
            const u32 j0 = (u32)(base + cur - a.qcum[lo]);
            for (;;) {
                u32 k = 0;
                if (lane == 0) k = atomicAdd(&blkctr[blockIdx.x], 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                if (k >= n_seg) break;
                pair_body<true, NFREE, DEFP, false, true, 0, true>(a, lo, j0 + k, lane, lds, flt);
            }
            __syncthreads();                       // the filter and the counter are free again
            cur = seg_end;
        }
    }
}

}  // namespace lzani

// Run-time compiled instantiation (lzani_rtc.h): this header as text behind the macros LZANI_P_* (the eight LZ parameters
// of the context), LZANI_RTC_NFREE and LZANI_RTC_CAND; one kernel per code object, found by its C name.
#if defined(LZANI_RTC)
extern "C" __global__ void __launch_bounds__(256, LZANI_WAVES_PER_SIMD) lzani_rtc_pairs(lzani::PairArgs a)
{
    lzani::pairs_loop<true, (LZANI_RTC_NFREE) != 0, 9, false, true, LZANI_RTC_CAND>(a);
}
#endif
