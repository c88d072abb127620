// lzani_core.h -- the pair-path algorithm of the MI355X engine, as wave-structured code.
//
// What it replaces: CParser::{prepare_data, parse, calc_stats}
// (/root/reference/src/parser.cpp:37-50, 482-716, 734-783) for one directed genome pair.
//
// Formulation (not a translation of the reference's loop):
//   * texts are 2-bit packed (32 symbols per u64) with a separate N bitmask (64 per u64);
//     a query is a *prefix view* of its own reference text (R = fwd | N^2mrd | RC | N^mrd,
//     Q = fwd | N^mrd), so one device copy per genome serves both roles;
//   * the greedy scan advances in ROUNDS of up to 64 query positions: every lane looks at one
//     position speculatively (anchor candidates; for the <= mqd+1 "tracking" steps also the
//     close-seed candidates) and the first step that hits (ballot + ctz, verified by the whole
//     wave) is exactly the step the sequential scan would take, because between two hits the
//     state evolves affinely (hard part 1);
//   * no factor list: the calc_stats fold is applied on the fly from mismatch bitmasks
//     (ballot + popcount + clz), with O(1) state per pair;
//   * approximate extension consumes 64 symbols per step from a sliding 128-bit mismatch
//     window instead of a circular flag buffer.
//
// The file is shared by the HIP kernels (wave = 64 lanes, lzani_kernels_pairs.h) and by the host-side
// models used only by the tests (tests/model/): the state machine (PairMachine) is templated on a
// `Wave` policy that supplies the cross-lane primitives -- DevWave on the device (ballots, LDS);
// the tests have two host policies of their own (tests/model/: lane-emulating, and lane-serial).
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LZ_HD __host__ __device__ __forceinline__
#else
#define LZ_HD inline
#endif

// Loop guard: every data-dependent loop has a bound that correct data can never reach; a trip is
// recorded (device: in g_guard_trip, read back by the host) and the loop is left, so that every
// wave of the persistent pair kernel reaches its exit whatever the data.
#if defined(__HIP_DEVICE_COMPILE__)
extern __device__ int g_guard_trip;
#define LZ_GUARD_TRIP(code) (g_guard_trip = (code))
#else
#define LZ_GUARD_TRIP(code) ((void)(code))
#endif

namespace lzani {

typedef uint64_t u64;
typedef uint32_t u32;

struct Params {           // the eight ints CParser reads (params.h:34-48)
    int mal, msl, mrd, mqd, reg, aw, am, ar;
};

struct TextView {         // one packed text: `len` symbols are addressable
    const u64* t2;        // 2 bits per symbol, symbol j at bits 2*(j&31) of word j>>5
    const u64* nm;        // 1 bit per symbol, 1 = N / padding
    int len;
    // Structure of a text whose genome holds no N: the only N symbols are the pads, so the valid
    // symbols are [0, L) and, in a reference text, [rc0, rc0 + L); rc0 = NO_RC (beyond every position) in a
    // query view.
    // With nfree set on both sides the N mask is never loaded (bounds do its job).
    int L, rc0;
    bool nfree;
};

enum { NO_RC = 0x40000000 };
LZ_HD TextView ref_view(const u64* t2, const u64* nm, int L, int mrd, bool nfree)
{ return TextView{t2, nm, 2 * L + 3 * mrd, L, L + 2 * mrd, nfree}; }
LZ_HD TextView qry_view(const u64* t2, const u64* nm, int L, int mrd, bool nfree)
{ return TextView{t2, nm, L + mrd, L, NO_RC, nfree}; }

// N-free texts: is p a real symbol / where does the run of real symbols containing p end
LZ_HD bool pos_valid(const TextView& t, int p)
{
    // p folded onto [0, L): p itself before rc0, p - rc0 from rc0 on (the smaller of the two as unsigned; a
    // negative p or a pad position stays >= L).  One compare, no lane-mask logic (that would be scalar work).
    const u32 a = (u32)p, b = (u32)(p - t.rc0);
    return (a < b ? a : b) < (u32)t.L;
}
LZ_HD int run_end(const TextView& t, int p)
{
    // with mrd = 0 no pad separates the forward part from the reverse complement: one run [0, 2L)
    if (p < t.L) return (t.rc0 == t.L) ? t.rc0 + t.L : t.L;
    return (p >= t.rc0 && p < t.rc0 + t.L) ? t.rc0 + t.L : p;
}

struct IndexView {        // anchor index of one reference (all mal-mers of R)
    const u32* dirz;      // dirz[b] .. dirz[b+1] = entry range of bucket b
    const u32* ent;       // (tag << posbits) | pos; ascending inside a bucket of up to IDX_SORT_MAX entries (larger
                          // buckets - long low-complexity runs - stay in fill order)
    int kb;               // 2*mal key bits
    int dirbits;          // bucket = top dirbits of mix(key)
    int posbits;          // low bits of an entry hold the position
    u32 tagmask;          // stored tag bits
    const u32* bk;        // optional bucket table: 4 entries per bucket (BK_EMPTY padded; entry 3 = BK_OVERFLOW
                          // when the bucket holds more than four), or nullptr
    const u32* tw;        // optional tag words (tag bits <= 7): per bucket one byte 0x80|tag per entry of the bucket
                          // table (slot k in byte k), 0 for an empty slot; TW_OVERFLOW for a bucket of more than
                          // four (no bucket looks like it: the tags of a bucket ascend with the slot)
};
enum : u32 { BK_EMPTY = 0xFFFFFFFFu, BK_OVERFLOW = 0xFFFFFFFEu, TW_OVERFLOW = 0x808080FFu };
enum { IDX_SORT_MAX = 32 };

// ---- bit helpers --------------------------------------------------------------------
LZ_HD u64 lowmask(int n) { return n >= 64 ? ~0ULL : (n <= 0 ? 0ULL : ((1ULL << n) - 1ULL)); }

LZ_HD int popc64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
LZ_HD int ctz64(u64 x)      // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
LZ_HD int clz64(u64 x)      // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
LZ_HD u64 brev64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    u64 r = 0;
    for (int k = 0; k < 64; ++k) r |= ((x >> k) & 1ULL) << (63 - k);
    return r;
#endif
}
LZ_HD int imin(int a, int b) { return a < b ? a : b; }
LZ_HD int imax(int a, int b) { return a > b ? a : b; }
LZ_HD int iabs(int a) { return a < 0 ? -a : a; }

// 32 symbols (64 bits) starting at symbol p >= 0
LZ_HD u64 win2(const u64* t2, int p)
{
    const u32 w = (u32)p >> 5;               // unsigned index: 32-bit offset addressing on the device
    const int s = (p & 31) * 2;
    u64 lo = t2[w];
    if (s == 0) return lo;
    return (lo >> s) | (t2[w + 1] << (64 - s));
}
// 64 N flags starting at symbol p >= 0
LZ_HD u64 winN(const u64* nm, int p)
{
    const u32 w = (u32)p >> 6;
    const int s = p & 63;
    u64 lo = nm[w];
    if (s == 0) return lo;
    return (lo >> s) | (nm[w + 1] << (64 - s));
}
typedef u32 __attribute__((may_alias)) u32a;          // the packed text read as 32-bit words (16 symbols each)
LZ_HD int sym_at(const TextView& t, int p)
{
    return (int)((reinterpret_cast<const u32a*>(t.t2)[(u32)p >> 4] >> ((p & 15) * 2)) & 3u);
}
LZ_HD int isN_at(const TextView& t, int p) { return (int)((t.nm[(u32)p >> 6] >> (p & 63)) & 1ULL); }

// 1 iff both positions exist, neither is N, and the symbols are equal.  N never matches
// anything (reference: code_N_ref = 4 vs code_N_seq = 5, defs.h:28-30).
LZ_HD int sym_match(const TextView& R, int rp, const TextView& Q, int qp)
{
    if (R.nfree && Q.nfree) {
        // both loads are issued unconditionally (from position 0 when the position is not a symbol), so the
        // lane code has no branch: divergent branches cost scalar exec-mask bookkeeping on every wave
        const bool vr = pos_valid(R, rp), vq = pos_valid(Q, qp);
        const int a = sym_at(R, vr ? rp : 0), b = sym_at(Q, vq ? qp : 0);
        return (int)(vr & vq & (a == b));
    }
    if (rp < 0 || rp >= R.len || qp < 0 || qp >= Q.len) return 0;
    if (isN_at(R, rp) | isN_at(Q, qp)) return 0;
    return sym_at(R, rp) == sym_at(Q, qp);
}

// equal_len (parser.cpp:192-207): common prefix of R[rp+start..] and Q[qp+start..], bounded by
// both ends; returns `start` even if the bound is smaller (quirk Q11).
LZ_HD int equal_len(const TextView& R, int rp, const TextView& Q, int qp, int start)
{
    if (R.nfree && Q.nfree) {
        // the first N after rp / qp is the end of the run of real symbols: fold it into the bound
        int bound = imin(run_end(R, rp) - rp, run_end(Q, qp) - qp);
        int n = start;
        while (n < bound) {
            u64 x = win2(R.t2, rp + n) ^ win2(Q.t2, qp + n);
            u64 d = (x | (x >> 1)) & 0x5555555555555555ULL;
            if (d) { n += ctz64(d) >> 1; break; }
            n += 32;
        }
        n = imin(n, bound);
        return n > start ? n : start;
    }
    int bound = imin(R.len - rp, Q.len - qp);
    int n = start;
    int guard = 0;
    while (n < bound) {
        if (++guard > (1 << 26)) { LZ_GUARD_TRIP(1); break; }
        u64 x = win2(R.t2, rp + n) ^ win2(Q.t2, qp + n);
        u64 d = (x | (x >> 1)) & 0x5555555555555555ULL;
        u32 nn = (u32)(winN(R.nm, rp + n) | winN(Q.nm, qp + n));
        int fd = d ? (ctz64(d) >> 1) : 32;
        int fn = nn ? ctz64((u64)nn) : 32;
        int f = imin(fd, fn);
        if (f < 32) { n += f; return n < bound ? n : (bound > start ? bound : start); }
        n += 32;
    }
    return n < bound ? n : (bound > start ? bound : start);
}

// k-mer (k <= 32) starting at p; returns false if it overlaps an N or the end of the text
LZ_HD bool kmer_at(const TextView& t, int p, int k, u64& key)
{
    if (p < 0 || p + k > t.len) return false;
    if (winN(t.nm, p) & lowmask(k)) return false;
    key = win2(t.t2, p) & lowmask(2 * k);
    return true;
}

// Bijective mixer on kb bits: (bucket, tag) = (top dirbits, remaining low bits) identify the key.
LZ_HD u64 mix_key(u64 key, int kb)
{
    if (kb <= 32) {
        u32 m = kb >= 32 ? 0xFFFFFFFFu : ((1u << kb) - 1u);
        u32 x = (u32)key;
        x = (x * 0x9E3779B1u) & m;
        x ^= x >> ((kb + 1) >> 1);
        x = (x * 0x85EBCA6Bu) & m;
        x ^= x >> ((kb + 1) >> 1);
        x = (x * 0xC2B2AE35u) & m;
        return x;
    }
    u64 m = lowmask(kb);
    u64 x = key;
    x = (x * 0x9E3779B97F4A7C15ULL) & m;
    x ^= x >> ((kb + 1) >> 1);
    x = (x * 0xD6E8FEB86659FD93ULL) & m;
    x ^= x >> ((kb + 1) >> 1);
    x = (x * 0xC2B2AE3D27D4EB4FULL) & m;
    return x;
}
LZ_HD void key_slot(const IndexView& I, u64 key, u32& bucket, u32& tag)
{
    u64 h = mix_key(key, I.kb);
    int tb = I.kb - I.dirbits;
    bucket = (u32)(h >> tb);
    tag = (u32)(h & lowmask(tb)) & I.tagmask;
}

// best_anchor (parser.cpp:514-531 == 585-602): over the reference positions holding the same
// mal-mer, ascending, the longest common prefix >= mal; first (smallest position) wins ties.
LZ_HD void anchor_lookup(const Params& P, const TextView& R, const TextView& Q, const IndexView& I,
                         u64 h, int qp, int& ap, int& al)     // h = mix_key(mal-mer at qp)
{
    ap = 0; al = 0;
    int tb = I.kb - I.dirbits;
    u32 b = (u32)(h >> tb);
    u32 tag = (u32)(h & lowmask(tb)) & I.tagmask;
    u32 s = I.dirz[b], e = I.dirz[b + 1];
    u32 pm = (u32)lowmask(I.posbits);
    if (e - s > (u32)R.len || e < s) { LZ_GUARD_TRIP(2); return; }
    for (u32 j = s; j < e; ++j) {
        u32 en = I.ent[j];
        if ((en >> I.posbits) != tag) continue;
        int p = (int)(en & pm);
        int m = equal_len(R, p, Q, qp, 0);
        if (m >= P.mal && (m > al || (m == al && p < ap))) { al = m; ap = p; }   // no reliance on the bucket's order
    }
}

LZ_HD void best_anchor(const Params& P, const TextView& R, const TextView& Q, const IndexView& I,
                       int qp, int& ap, int& al)
{
    ap = 0; al = 0;
    u64 key;
    if (!kmer_at(Q, qp, P.mal, key)) return;
    anchor_lookup(P, R, Q, I, mix_key(key, I.kb), qp, ap, al);
}

// Seed selection rule (parser.cpp:566-578): longer wins; on a tie the one strictly nearer to
// ref_pred; candidates must be offered in ascending position.
LZ_HD void seed_consider(int p, int m, int ref_pred, int& sp, int& sl)
{
    if (m >= sl) {
        if (m == sl) { if (iabs(p - ref_pred) < iabs(sp - ref_pred)) sp = p; }
        else { sl = m; sp = p; }
    }
}

// ipow<double>(1 - 4^-len, e) (parser.h:134-188): square and multiply, IEEE doubles, the same
// multiplication order as the reference so the comparison in `arbitrate` is bit-exact.
LZ_HD double pow_not_chance(int len, u32 e)
{
    // 1 - 4^-len is exactly 1.0 from len = 27 on (quirk Q5), and every power of 1.0 is 1.0: no loop at all for the
    // long matches of related genomes, where this runs at nearly every event
    if (len >= 27) return 1.0;
    // 4^-len by repeated multiplication with 0.25 is exact (a power of two), so it is 2^(-2 len) whichever way it is made
    double base = 1.0 - __builtin_ldexp(1.0, -2 * len);
    double r = 1.0;
    while (e) {
        if (e & 1u) r *= base;
        base *= base;
        e >>= 1;
    }
    return r;
}

// Arbitration between the close seed (sp,sl) and the anchor (ap,al) in tracking mode
// (parser.cpp:604-623), with the 0-position sentinels of quirk Q1 kept.
LZ_HD void arbitrate(const Params& P, int T, int lit, int ap, int al, int& sp, int& sl)
{
    if (!ap) return;
    if (!sp) { sp = ap; sl = al; return; }
    // the anchor IS the close seed (the window holds the anchor's position: the continuation of a related stretch): the
    // comparison of the two probabilities cannot change anything -- and it is ~18 rounds of f64 multiplies per power
    if (sp == ap && sl == al) return;
    u32 ea = (u32)(int)(2 * ((u64)T + 1 - (u64)(int64_t)al));     // (int) cast, then uint32_t parameter (Q4)
    u32 ec = (u32)(lit + P.mrd + 1 - sl);
    double anchor_prob = pow_not_chance(al, ea);
    double close_prob = pow_not_chance(sl, ec);
    if (anchor_prob > close_prob) { sp = ap; sl = al; }
}

// Close-seed search of one tracking step, portable form (parser.cpp:548-580): every window
// position p in [r_end, ref_pred + mrd) whose msl-mer equals the query's, ascending.
LZ_HD void seed_search_window(const Params& P, const TextView& R, const TextView& Q,
                              int qp, int r_end, int lit, int& sp, int& sl)
{
    sp = 0; sl = 0;
    u64 qk;
    if (!kmer_at(Q, qp, P.msl, qk)) return;
    int ref_pred = r_end + lit;
    int hi = imin(ref_pred + P.mrd, R.len - P.msl + 1);
    const int step = 33 - P.msl;             // window positions served by one 32-symbol load
    const u64 km = lowmask(2 * P.msl), nk = lowmask(P.msl);
    for (int p0 = r_end; p0 < hi; p0 += step) {
        u64 w2 = win2(R.t2, p0), wn = winN(R.nm, p0);
        int cnt = imin(step, hi - p0);
        for (int o = 0; o < cnt; ++o) {
            if (((w2 >> (2 * o)) & km) != qk || ((wn >> o) & nk)) continue;
            seed_consider(p0 + o, equal_len(R, p0 + o, Q, qp, P.msl), ref_pred, sp, sl);
        }
    }
}

// One scan step evaluated in isolation (what one lane does in a round).
//   trk   : the step starts in tracking mode (ref_pred >= 0)
//   r_end : reference end of the last match (= ref_pred - lit)
//   lit   : literal run length at the start of this step
// The HIP wave replaces the per-lane window scan by a shared LDS join with the same candidate
// order (lzani_kernels_pairs.h, DevWave::find_event).
LZ_HD void eval_step(const Params& P, const TextView& R, const TextView& Q, const IndexView& I,
                     int qp, bool trk, int r_end, int lit, int& bp, int& bl)
{
    int ap, al;
    best_anchor(P, R, Q, I, qp, ap, al);
    if (!trk) { bp = ap; bl = al; return; }
    int sp, sl;
    seed_search_window(P, R, Q, qp, r_end, lit, sp, sl);
    arbitrate(P, R.len, lit, ap, al, sp, sl);
    bp = sp; bl = sl;
}

// ---- streaming calc_stats state --------------------------------------------------------
struct Regions {
    int cl, clit, nl;      // open region: matches, literals between matches, pending literals
    int tm, tl, tc;        // totals over finalised regions with cl + clit >= reg
    int reg;
    LZ_HD void init(int r) { cl = clit = nl = tm = tl = tc = 0; reg = r; }
    // calc_stats at a match_distant factor (parser.cpp:743-751): close the open region
    LZ_HD void finalize()
    {
        if (cl && cl + clit >= reg) { tm += cl; tl += clit; ++tc; }
        cl = clit = nl = 0;
    }
    LZ_HD void discard() { cl = clit = nl = 0; }
    // fold of `n` symbols whose match flags are the low n bits of M (bit j = symbol j)
    LZ_HD void seg(u64 M, int n)
    {
        if (M) {
            int hi = 63 - clz64(M);
            int c = popc64(M);
            cl += c;
            clit += nl + (hi + 1 - c);
            nl = n - 1 - hi;
        } else nl += n;
    }
    LZ_HD void seg_match_run(int n) { if (n > 0) { cl += n; clit += nl; nl = 0; } }
};

// calc_regions restated as a stream (parser.cpp:786-837): the region_t of the open region, fed with
// the factors in emission order.  Only the alignment instantiation (ALN) of the machine uses it.
struct RegionCoords {
    int ref_start, ref_end, seq_start, seq_end, nm, nmm, buf;
    bool fresh;                       // the next match factor is the region's match_distant
    LZ_HD void clear() { ref_start = ref_end = seq_start = seq_end = -1; nm = nmm = buf = 0; fresh = true; }
    LZ_HD int length() const { return seq_end - seq_start; }
    LZ_HD void touch(int dp, int off, int len)
    {
        if (seq_start < 0 || dp < seq_start) seq_start = dp;
        if (seq_end < 0 || dp + len > seq_end) seq_end = dp + len;
        if (ref_start < 0 || off < ref_start) ref_start = off;
        if (ref_end < 0 || off + len > ref_end) ref_end = off + len;
    }
    LZ_HD void match(int dp, int off, int len)
    {
        if (fresh) { clear(); fresh = false; }
        else { ref_end += buf; seq_end += buf; nmm += buf; }          // extend_region + update_mismatches
        buf = 0;
        touch(dp, off, len);
        nm += len;
    }
    LZ_HD void lit(int len) { buf += len; }
};

// Result of scanning one 64-symbol chunk of an approximate extension (try_extend_*,
// parser.cpp:377-441).  prevB/B are mismatch masks of the previous/current chunk (bit j =
// symbol j of the chunk; symbols before the start of the extension count as matches).
//   brk  : bit j set iff the sliding window ending at j holds more than am mismatches
//   qual : bit j set iff j is a match and the ar-1 symbols before it are matches too
struct ExtMasks { u64 brk, qual; };

LZ_HD void ext_lane(u64 prevB, u64 B, int j, int n, int aw, int am, int ar, bool& brk, bool& qual)
{
    // W = the 64 stream symbols ending at symbol j (bit 63 = symbol j): funnel of prevB:B
    u64 W = (B << (63 - j)) | ((prevB >> 1) >> j);
    int a = ar < 1 ? 1 : ar;
    bool in = j < n;
    brk = in && popc64(W >> (64 - aw)) > am;          // mismatches among the last aw symbols
    qual = in && (W >> (64 - a)) == 0;                // the last max(ar,1) symbols all match
}

// ---- lane-serial forms (one lane works a whole 64-symbol chunk alone) -----------------------------------
// Used where 64 independent items are at hand -- the candidates of the anchor queue, one per lane
// (lzani_kernels_pairs.h: refill) -- so that a wave-wide instruction serves 64 of them instead of one.
LZ_HD u64 compress_even(u64 x)        // bits 0,2,4,.. of x -> bits 0..31
{
    x &= 0x5555555555555555ULL;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return x;
}
LZ_HD u64 bits_below(int n) { return lowmask(n < 0 ? 0 : n); }       // bits j < n, n clamped to [0,64]
LZ_HD u64 bits_range(int a, int b) { return bits_below(b) & ~bits_below(a); }
LZ_HD u64 valid_bits(const TextView& t, int p0)                       // N-free text: bit j = p0 + j is a real symbol
{
    u64 v = bits_range(-p0, t.L - p0);
    if (t.rc0 != NO_RC) v |= bits_range(t.rc0 - p0, t.rc0 + t.L - p0);
    return v;
}
// 32 symbols from p on (p >= 0), branch-free funnel of two words
LZ_HD u64 win2f(const u64* t2, int p)
{
    const u32 w = (u32)p >> 5;
    const int s = (p & 31) * 2;
    return (t2[w] >> s) | ((t2[w + 1] << 1) << (63 - s));
}
LZ_HD u64 winNf(const u64* nm, int p)
{
    const u32 w = (u32)p >> 6;
    const int s = p & 63;
    return (nm[w] >> s) | ((nm[w + 1] << 1) << (63 - s));
}
// bit j (j < 32) = 1 iff Q[q0+j] does not match R[r0+j] (positions beyond a text, pads and N never match);
// q0, r0 >= 0
LZ_HD u32 lane_mism32(const TextView& R, const TextView& Q, int q0, int r0)
{
    const u64 x = win2f(R.t2, r0) ^ win2f(Q.t2, q0);
    const u32 mm = (u32)compress_even(x | (x >> 1));
    u32 valid;
    if (R.nfree && Q.nfree) valid = (u32)(valid_bits(R, r0) & valid_bits(Q, q0));
    else valid = (u32)(~(winNf(R.nm, r0) | winNf(Q.nm, q0)) & bits_below(R.len - r0) & bits_below(Q.len - q0));
    return mm | ~valid;
}
// The same for 16 symbols in 32-bit arithmetic (what a record of aw <= 15 looks at: one 8-byte fetch per text and a
// quarter of the vector instructions of the 64-bit form -- the refill spends a third of its instructions here).
LZ_HD u32 win16f(const u64* t2, int p)
{
    const u32* t32 = reinterpret_cast<const u32*>(t2);               // symbol j at bits 2*(j&15) of dword j>>4
    const u32 w = (u32)p >> 4;
    const u32 s = ((u32)p & 15u) * 2u;
    const u32 d0 = t32[w], d1 = t32[w + 1];
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(d1, d0, s);
#else
    return (u32)((((u64)d1 << 32) | d0) >> s);
#endif
}
LZ_HD u32 compress_even16(u32 x)      // bits 0,2,4,.. of x -> bits 0..15
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
LZ_HD u32 bits_below16(int n) { return n <= 0 ? 0u : n >= 16 ? 0xFFFFu : (1u << n) - 1u; }
LZ_HD u32 valid_bits16(const TextView& t, int p0)
{
    u32 v = bits_below16(t.L - p0) & ~bits_below16(-p0);
    if (t.rc0 != NO_RC) v |= bits_below16(t.rc0 + t.L - p0) & ~bits_below16(t.rc0 - p0);
    return v;
}
// bit j (j < 16) = 1 iff Q[q0+j] does not match R[r0+j]; bits 16..31 set.  Equal to lane_mism32's low 16 bits.
LZ_HD u32 lane_mism16(const TextView& R, const TextView& Q, int q0, int r0)
{
    const u32 x = win16f(R.t2, r0) ^ win16f(Q.t2, q0);
    const u32 mm = compress_even16(x | (x >> 1));
    u32 valid;
    if (R.nfree && Q.nfree) valid = valid_bits16(R, r0) & valid_bits16(Q, q0);
    else valid = (u32)(~(winNf(R.nm, r0) | winNf(Q.nm, q0)) & bits_below(R.len - r0) & bits_below(Q.len - q0)) & 0xFFFFu;
    return mm | ~valid;
}
LZ_HD u32 brev32(u32 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    u32 r = 0;
    for (int k = 0; k < 32; ++k) r |= ((x >> k) & 1u) << (31 - k);
    return r;
#endif
}
LZ_HD int popc32(u32 x) { return popc64((u64)x); }

// "Null extension" record of a queued candidate (query position qp, reference position pos, match length al):
// what a distant event there needs to know to see, WITHOUT touching the texts, that neither approximate
// extension moves (try_extend_*, parser.cpp:377-441) -- true for four events out of five of an unrelated pair,
// whose anchors are chance k-mers with random flanks.  State-free, so computed ahead, by one lane per
// candidate, for a whole batch of the anchor queue (lzani_kernels_pairs.h: refill).
//   bits 0..aw-1  backward qual bits (ext_lane: symbol j back and the ar-1 before it all match), j < aw
//   bit 30        the first aw backward symbols hold more than am mismatches (the scan breaks inside them)
//   bit 31        the forward extension is provably empty: its first aw symbols break the scan and none qualifies up to
//                 the break
//   aw <= 15 and bit 30 set: the backward scan breaks inside its first aw symbols, so -- given that the machine may look
//                 at least aw symbols back -- its result is state-free too; the low bits then hold, instead of the qual
//                 bits: bits 0..3 its length b (0 = empty), bits 4..7 the matches among its b symbols
//   aw <= 15 only (the bits above the qual bits are free):
//   bit 29        the forward extension is KNOWN: its first aw symbols break the scan (at the (am+1)-th mismatch, since
//                 every window up to there starts at the match), so it ends at the last qualifying symbol before the break:
//   bits 24..28   its length e (0 = empty; then bit 31 is set too), bits 20..23 the mismatches among its e symbols
// An extension is empty iff no symbol qualifies up to the break; if the first aw symbols already break the scan
// the later ones do not matter.  A record that proves nothing (aw > 30, a text end nearby) has every qual bit set.
enum : u32 { EXT_REC_NONE = 0x3FFFFFFFu, EXT_REC_FWDK = 0x20000000u, EXT_REC_BRKB = 0x40000000u, EXT_REC_NULLF = 0x80000000u };
LZ_HD u32 ext_rec_none(int aw) { return aw <= 15 ? 0x00007FFFu : (u32)EXT_REC_NONE; }
LZ_HD u32 ext_qual32(u32 B, int ar)            // qual bits of a chunk's first 32 symbols (no earlier chunk)
{
    const int a = ar < 1 ? 1 : ar;
    const u32 Z = ~B;
    u32 acc = Z;
    for (int k = 1; k < a && k < 32; ++k) acc &= (Z << k) | (u32)lowmask(k);      // symbols before the start count as matches
    return acc;
}
LZ_HD u32 null_ext_record(const Params& P, const TextView& R, const TextView& Q, int qp, int pos, int al, bool narrow = true)
{
    if (P.aw > 30) return EXT_REC_NONE;
    const u32 wm = (u32)lowmask(P.aw);
    u32 rec = ext_rec_none(P.aw);
    if (qp >= 32 && pos >= 32) {                                       // all 32 symbols before the match exist
        // bit j = symbol qp-1-j / pos-1-j; only the first aw of them are looked at
        const u32 Bb = (narrow && P.aw <= 15) ? (brev32(lane_mism16(R, Q, qp - 16, pos - 16)) >> 16) | 0xFFFF0000u
                                                : brev32(lane_mism32(R, Q, qp - 32, pos - 32));
        const u32 qb = ext_qual32(Bb, P.ar) & wm;
        if (popc32(Bb & wm) <= P.am) rec = qb;
        else if (P.aw > 15) rec = qb | EXT_REC_BRKB;
        else {                                                          // the scan breaks at the (am+1)-th mismatch
            u32 x = Bb & wm;
            for (int k = 0; k < P.am; ++k) x &= x - 1u;
            const u32 q = qb & (u32)lowmask((int)__builtin_ctz(x) + 1);
            const int b = q ? 32 - (int)__builtin_clz(q) : 0;            // ends at the last qualifying symbol
            rec = (u32)b | ((u32)(b - popc32(Bb & (u32)lowmask(b))) << 4) | EXT_REC_BRKB;
        }
    }
    const int fq = qp + al, fr = pos + al;
    if (imin(Q.len - fq, R.len - fr) >= P.aw) {
        const u32 Bf = (narrow && P.aw <= 15) ? lane_mism16(R, Q, fq, fr) : lane_mism32(R, Q, fq, fr);
        if (popc32(Bf & wm) > P.am) {                                   // the scan breaks at the (am+1)-th mismatch
            u32 x = Bf & wm;
            for (int k = 0; k < P.am; ++k) x &= x - 1u;
            const u32 qf = ext_qual32(Bf, P.ar) & (u32)lowmask((int)__builtin_ctz(x) + 1);
            if (qf == 0) rec |= EXT_REC_NULLF | (P.aw <= 15 ? (u32)EXT_REC_FWDK : 0u);
            else if (P.aw <= 15) {
                const int e = 32 - (int)__builtin_clz(qf);               // ends at the last qualifying symbol
                rec |= EXT_REC_FWDK | ((u32)e << 24) | ((u32)popc32(Bf & (u32)lowmask(e)) << 20);
            }
        }
    }
    return rec;
}
// the backward extension is empty, by the record; reach = how far back the machine may look, uncapped
// (min(avail, i, bpos); the scan itself looks at min(64, reach) symbols, and aw <= 30 < 64)
LZ_HD bool ext_rec_null_bwd(u32 rec, int reach, int aw)
{
    if (reach <= 0) return true;
    if (aw <= 15 && (rec & EXT_REC_BRKB)) return reach >= aw && (rec & 15u) == 0;    // the scan's own result (full first window)
    const int m = imin(imin(reach, aw), 30);                    // 1 .. 30 (aw > 30: the record has every qual bit set)
    return (rec & ((1u << m) - 1u)) == 0 && (reach <= aw || (rec & EXT_REC_BRKB));
}
// the backward extension is in the record (aw <= 15): its length b > 0 and the matches c among its b symbols
LZ_HD bool ext_rec_bwd_known(u32 rec, int reach, int aw, int& b, int& c)
{
    if (aw > 15 || !(rec & EXT_REC_BRKB) || reach < aw) return false;
    b = (int)(rec & 15u); c = (int)((rec >> 4) & 15u);
    return b > 0;
}

// ---- the pair state machine ------------------------------------------------------------
// Wave policy W must provide (all results wave-uniform):
//   u64  mism_fwd(q0, r0, n)       bit j (j<n) = 1 iff Q[q0+j] does not match R[r0+j]
//   u64  mism_bwd(q0, r0, n)       bit j (j<n) = 1 iff Q[q0-1-j] does not match R[r0-1-j]
//   bool find_event(i, n, trk, r_end, lit, adv, bpos, blen)
//                                   looks at the steps i .. i+n-1 (n = all that is left of the query; the policy
//                                   decides how far it looks in one call): true = the first step that hits
//                                   (evaluation gives len >= msl) is step i+adv; false = the adv >= 1 steps it
//                                   looked at do not hit
//   ExtMasks ext_scan(prevB, B, n)
//   int  best_split(Lm, Rm, to_scan) argmax_s popc(Lm & low(s)) + popc(Rm >> s), last max wins
//   void mism2(qa, ra, da, na, qb, rb, db, nb, A, B)   two mismatch masks in one fetch: bit j of A = mismatch of
//                                   Q[qa + da*j] vs R[ra + da*j] for j < na (d = +1 forward, -1 backward), same for B
//   int  null_chain(...)            optional (static constexpr bool NULL_CHAIN = true): see run()
//   bool ext_record(u32&)           the null-extension record (null_ext_record) of the event find_event just
//                                   returned, if the policy has one (the anchor queue of the device)
//   void stamp(section)             profiling hook (no-op outside the LZANI_STAMPS diagnostic build)
//   void emit_region(RegionCoords)  ALN only: one region of calc_regions (length >= reg)
// a policy may bring a hand-scheduled loop over runs of null events (DevWave::null_chain); policies without one say nothing
template <class W, class = void> struct wave_has_null_chain : std::false_type {};
template <class W> struct wave_has_null_chain<W, std::void_t<decltype(W::NULL_CHAIN)>> : std::bool_constant<W::NULL_CHAIN> {};

// a policy may fetch three forward mismatch masks in one go (DevWave::mism3: the two diagonals of a gap fill and the first
// chunk of the forward extension behind the close match -- one memory wait instead of two)
template <class W, class = void> struct wave_has_mism3 : std::false_type {};
template <class W> struct wave_has_mism3<W, std::void_t<decltype(W::HAS_MISM3)>> : std::bool_constant<W::HAS_MISM3> {};

template <class W, class = void> struct wave_has_stretch_chain : std::false_type {};
template <class W> struct wave_has_stretch_chain<W, std::void_t<decltype(W::HAS_STRETCH_CHAIN)>> : std::bool_constant<W::HAS_STRETCH_CHAIN> {};
// ---- one pair by several waves (long genomes, few pairs: a launch lasts as long as its slowest pair) -----------------------
// The scan of parse() is sequential, but where it stands after an event -- (i, r_end), lit = 0, tracking -- is all of its
// state that the EVENTS behind depend on, as long as the region open at that point is later closed by the keep branch (not
// dropped as short: that branch looks back over the region) -- and an approximate extension ends where its window breaks,
// wherever it began.  So a query is cut at fixed positions P_1 < P_2 < ..; for every cut a wave scans from P_j with a fresh
// state through its FIRST event: the state behind it is the cut's CHECKPOINT.  Then segment j runs from checkpoint j
// (segment 0 from the start) until its state EQUALS a later checkpoint -- the true scan passes through it: the segment
// stops, the later one has done the rest -- or the query ends; a checkpoint the true scan does not pass through is skipped
// (its segment's work is void).  What differs between the true scan and a segment started at a checkpoint is bookkeeping:
// the region open at the checkpoint began earlier in the true scan.  A segment therefore keeps its FIRST region apart
// (its counts at the start and at its close), and the stitch (split_stitch) puts the true counts together along the chain
// of segments; a first region that the segment drops, or that the true scan would drop, voids the split (the pair is then
// scanned whole).  Exact by construction: every hand-over is an equality of states.  (parser.cpp:482-716 has no
// counterpart: one thread scans one pair.)
#ifdef LZ_SPLIT_DEBUG
static long g_split_why[8];
#endif
#ifndef LZ_SPLIT_GUESS_LIT
#define LZ_SPLIT_GUESS_LIT 4          // a short first region is kept on a guess from this many half-windows of literals on (see run_impl)
#endif
struct SplitStart {                  // where a segment starts: the state behind the first event of a fresh scan from the cut
    int i, r_end, prev_rs, pre_lit, cl, clit;        // lit = 0, tracking, prev_re = i, nl = 0;  i = -1: no checkpoint (no event behind the cut, or the cut is disabled)
};
enum { NO_CHECKPOINT = 0x7FFFFFFF };                 // (as a limit: no checkpoint ahead)
struct SplitOut {
    int tm, tl, tc;                  // regions this segment closed, other than its first
    int first;                       // the first region: 0 still open at the stop, 1 closed by the keep branch, 2 dropped (void), 3 segment 0 (none apart)
    int first_cl, first_clit;        // first = 1: its counts at the close (with what the segment started from)
    int first_re;                    //            the query position it ended at (prev_re at the close): the stitch tests the true span
    int stop;                        // the checkpoint the segment stopped at (index of the cut), -1 = the end of the query
    int open_cl, open_clit, open_rs; // at the stop: the open region's counts and start (if first = 0: the first region's, as the segment sees them)
    // How far back a distant match may look (try_extend_backward's bound, parser.cpp:636-660) is i - FLOOR: the floor is the end
    // of the last region the scan KEPT (a dropped region leaves it where it was: prev_rs - pre_lit).  A segment knows the true
    // floor from its first keep on; before, its own (the cut) is an upper bound of the true one, so its look-back is a lower
    // bound -- good as long as every backward scan breaks inside it (else first = 2).
    int assumed;                     // at the close of the first region the segment went on as if the true scan 1 kept / 2 dropped it / 3 kept it, on a guess (0: not closed)
    int first_floor;                 // assumed = 2: the highest floor the segment's look-backs since hold for (the stitch wants the true floor at or below it);
                                     // assumed = 3: a backward scan at the close or since (up to the segment's next keep) was cut short
    int synced, floor;               // at the stop: the segment's floor is the true one (it kept a region) / its value
    int stop_i, stop_r;              // the state it stopped in (the checkpoint's i, r_end): with the fields above, all a RESUMED scan needs -- when the
                                     // segment it handed over to turns out void, that cut is disabled and this segment goes on from here
};
// the stitch of one pair: seg[0 .. n) in cut order, starts[j] = the checkpoint of cut j (j >= 1); false = void (scan the pair whole)
// void_at / void_from (optional): the segment whose work is void and the one that handed over to it -- run that one again with
// the void one's cut disabled and the stitch may get through (void_at = -1: nothing to retry)
// why (optional): 0 a look-back cut short, 1 kept by the segment / dropped by the true scan, 2 the other way round, 3 the true
// floor above what the segment's look-backs hold for, 4 kept on a guess and cut short, 5 a broken chain
LZ_HD bool split_stitch(const SplitStart* starts, const SplitOut* seg, int n, int reg, int out[3], int* void_at = nullptr, int* void_from = nullptr,
                        int* why = nullptr)
{
    int from = -1;
    if (void_at) { *void_at = -1; *void_from = -1; }
    if (why) *why = 5;
#ifdef LZ_SPLIT_DEBUG
#define LZ_SPLIT_WHY(k) do { ++g_split_why[k]; if (why) *why = (k); } while (0)
#else
#define LZ_SPLIT_WHY(k) do { if (why) *why = (k); } while (0)
#endif
#define LZ_SPLIT_VOID(at) do { if (void_at) { *void_at = (at); *void_from = from; } return false; } while (0)
    int tm = 0, tl = 0, tc = 0;
    int open_cl = 0, open_clit = 0, open_rs = -1;          // the TRUE open region at the current segment's start
    int floor = 0;                                         // the TRUE floor there
    int j = 0;
    for (int guard = 0; guard <= n; ++guard) {
        const SplitOut& o = seg[j];
        if (o.first == 2) { LZ_SPLIT_WHY(0); LZ_SPLIT_VOID(j > 0 ? j : -1); }
        int t_cl = o.open_cl, t_clit = o.open_clit, t_rs = o.open_rs;      // the true open region at the segment's stop
        if (j > 0) {
            const SplitStart& st = starts[j];
            if (o.first == 1) {                            // the first region closed inside: its true counts, the keep test on the true span
                const int cl = open_cl - st.cl + o.first_cl, clit = open_clit - st.clit + o.first_clit;
                const bool dropped = open_rs >= 0 && o.first_re - open_rs < reg;       // by the true scan: its start is the true one
                if (o.assumed == 1 && dropped) { LZ_SPLIT_WHY(1); LZ_SPLIT_VOID(j); }      // the segment went on as if it were kept
                if (o.assumed == 2 && (!dropped || floor > o.first_floor)) { LZ_SPLIT_WHY(dropped ? 3 : 2); LZ_SPLIT_VOID(j); }   // ... dropped, looking back to its own floor
                if (o.assumed == 3 && dropped && o.first_floor) { LZ_SPLIT_WHY(4); LZ_SPLIT_VOID(j); }     // ... kept on a guess, and its look-back over the literals alone was cut short
                if (!dropped && cl && cl + clit >= reg) { tm += cl; tl += clit; ++tc; }
            } else {                                       // still open at the stop: the true region goes on
                t_cl = open_cl - st.cl + o.open_cl; t_clit = open_clit - st.clit + o.open_clit; t_rs = open_rs;
            }
        }
        tm += o.tm; tl += o.tl; tc += o.tc;
        if (o.stop < 0) {                                  // the end of the query: calc_stats closes what is open
            if (j > 0 && o.first == 0) { if (t_cl && t_cl + t_clit >= reg) { tm += t_cl; tl += t_clit; ++tc; } }
            out[0] = tm; out[1] = tl; out[2] = tc;
            return true;
        }
        if (o.stop <= j || o.stop >= n) LZ_SPLIT_VOID(-1);
        if (o.synced) floor = o.floor;
        from = j;
        open_cl = t_cl; open_clit = t_clit; open_rs = t_rs;
        j = o.stop;
    }
    LZ_SPLIT_VOID(-1);
#undef LZ_SPLIT_VOID
}

template <class W, bool ALN = false>
struct PairMachine {
    W& w;
    const Params& P;
    const int T, D;
    Regions g;
    RegionCoords c;

    LZ_HD PairMachine(W& w_, const Params& P_, int T_, int D_) : w(w_), P(P_), T(T_), D(D_) { g.init(P_.reg); c.clear(); }

    // ---- factor stream for the alignment output (ALN): runs of a match mask along one diagonal
    LZ_HD void runs(u64 M, int n, int q0, int r0, int dp_shift)
    {
        int pos = 0;
        while (pos < n) {
            u64 rest = M >> pos;
            if (!rest) { c.lit(n - pos); break; }
            int z = ctz64(rest);
            if (z) c.lit(z);
            pos += z;
            u64 inv = ~(M >> pos);
            int len = inv ? ctz64(inv) : 64;
            len = imin(len, n - pos);
            c.match(q0 + pos - dp_shift, r0 + pos, len);
            pos += len;
        }
    }
    LZ_HD void region_close() { if (ALN) { if (!c.fresh && c.length() >= P.reg) w.emit_region(c); c.clear(); } }

    // compare_ranges folded (parser.cpp:210-248): any length, forward order
    LZ_HD void seg_range(int q0, int r0, int len)
    {
        for (int base = 0; base < len; base += 64) {
            int n = imin(64, len - base);
            u64 B = w.mism_fwd(q0 + base, r0 + base, n);
            g.seg(~B & lowmask(n), n);
            if (ALN) runs(~B & lowmask(n), n, q0 + base, r0 + base, 0);
        }
    }
    LZ_HD void match_run(int dp, int off, int len)
    {
        g.seg_match_run(len);
        if (ALN && len > 0) c.match(dp, off, len);
    }

    // try_extend_forward (parser.cpp:377-409) fused with the fold of compare_ranges(i, ref_pred, e)
    // have0/B0: the mismatch mask of the first chunk, if the caller fetched it already (mism2)
    LZ_HD int extend_forward(int q0, int r0, bool have0 = false, u64 B0 = 0)
    {
        int maxlen = imin(D - q0, T - r0);
        int last = 0, last_mm = 0, mm_cum = 0;
        u64 prevB = 0, Bn = 0;
        bool haveN = false;                      // the chunk behind the current one is in hand already
        for (int base = 0; base < maxlen; base += 64) {
            int n = imin(64, maxlen - base);
            u64 B;
            if (have0 && base == 0) B = B0;
            else if (haveN) { B = Bn; haveN = false; }
            else {
                // an extension that has not broken inside its first chunk is a long one (a closely related stretch): from
                // here on two chunks a fetch -- four loads in flight, one memory wait per 128 symbols
                const int n2 = imin(64, maxlen - base - 64);
                if (base > 0 && n2 > 0) { w.mism2(q0 + base, r0 + base, 1, n, q0 + base + 64, r0 + base + 64, 1, n2, B, Bn); haveN = true; }
                else B = w.mism_fwd(q0 + base, r0 + base, n);
            }
            ExtMasks m = w.ext_scan(prevB, B, n);
#if defined(LZANI_PATH_STATS) && defined(__HIP_DEVICE_COMPILE__)
            w.ps[22] += 1;
#endif
            u64 qm = m.qual;
            if (m.brk) qm &= lowmask(ctz64(m.brk) + 1);
            if (qm) {
                int jq = 63 - clz64(qm);
                last = base + jq + 1;
                last_mm = mm_cum + popc64(B & lowmask(jq + 1));
            }
            if (m.brk) break;
            mm_cum += popc64(B);
            prevB = B;
        }
        if (__builtin_expect(last > 0, 0)) {   // position last-1 is a match, so every mismatch precedes a match
            g.cl += last - last_mm;
            g.clit += g.nl + last_mm;
            g.nl = 0;
            if (ALN)
                for (int base = 0; base < last; base += 64) {
                    int n = imin(64, last - base);
                    u64 B = w.mism_fwd(q0 + base, r0 + base, n);
                    runs(~B & lowmask(n), n, q0 + base, r0 + base, 0);
                }
        }
        return last;
    }

    // try_extend_backward (parser.cpp:412-441)
    // bounded: set when the scan ended at the look-back bound max_len (no break before it, and the texts go on behind it) -- with a
    // larger bound the result might be another one (the split's segments look back over a lower bound of what the true scan may)
    // looked: the symbols the scan needed (up to its break; everything it was allowed where it did not break)
    LZ_HD int extend_backward(int q0, int r0, int max_len, bool have0 = false, u64 B0 = 0, bool* bounded = nullptr, int* looked = nullptr)
    {
        int maxlen = imin(max_len, imin(q0, r0));
        int last = 0, seen = imax(maxlen, 0);
        u64 prevB = 0;
        bool broke = false;
        for (int base = 0; base < maxlen; base += 64) {
            int n = imin(64, maxlen - base);
            u64 B = (have0 && base == 0) ? B0 : w.mism_bwd(q0 - base, r0 - base, n);
            ExtMasks m = w.ext_scan(prevB, B, n);
            u64 qm = m.qual;
            if (m.brk) qm &= lowmask(ctz64(m.brk) + 1);
            if (qm) last = base + (63 - clz64(qm)) + 1;
            if (m.brk) { broke = true; seen = base + ctz64(m.brk) + 1; break; }
            prevB = B;
        }
        if (bounded) *bounded = !broke && max_len < imin(q0, r0);
        if (looked) *looked = seen;
        return last;
    }

    // compare_ranges_both_ways folded (parser.cpp:251-374); len = literal run <= 64
    // fq / fr / nf / Bf: the first chunk of the forward extension behind the close match, fetched with the gap's two
    // diagonals where the policy can (haveF says whether it was)
    LZ_HD void gap_fill(int ds, int r_left, int r_right_end, int len, int fq = 0, int fr = 0, int nf = 0, u64* Bf = nullptr, bool* haveF = nullptr)
    {
        if (len <= 0) return;
        int to_scan = (r_right_end < r_left) ? len : imin(r_right_end - r_left, len);
        int shift = len - to_scan;
        u64 F = 0;
        if (to_scan > 0) {
            u64 Lm, Rm;
            if constexpr (wave_has_mism3<W>::value && !ALN) {
                if (Bf) { w.mism3(ds, r_left, to_scan, ds + shift, r_right_end - to_scan, to_scan, fq, fr, nf, Lm, Rm, *Bf); *haveF = true; }
                else w.mism2(ds, r_left, 1, to_scan, ds + shift, r_right_end - to_scan, 1, to_scan, Lm, Rm);
            } else
            w.mism2(ds, r_left, 1, to_scan, ds + shift, r_right_end - to_scan, 1, to_scan, Lm, Rm);
            Lm = ~Lm & lowmask(to_scan);
            Rm = ~Rm & lowmask(to_scan);
            int s = w.best_split(Lm, Rm, to_scan);
            u64 right = (s >= 64) ? 0ULL : ((Rm >> s) << s);
            F = (Lm & lowmask(s)) | (right << shift);
            if (ALN) {
                // left part on the left diagonal, the untouched middle, right part on the right diagonal.
                // Quirk of the reference (parser.cpp:353-358): when the first right symbol is a mismatch
                // merged into a preceding literal run, data_p is not advanced, so every later factor of
                // the right part carries a data_pos one too small.
                if (s > 0) runs(Lm & lowmask(s), s, ds, r_left, 0);
                if (shift > 0) c.lit(shift);
                if (s < to_scan) {
                    const bool first_match = (Rm >> s) & 1ULL;
                    const bool prev_lit = shift > 0 || (s > 0 && !((Lm >> (s - 1)) & 1ULL));
                    const int q = (!first_match && prev_lit) ? 1 : 0;
                    runs(Rm >> s, to_scan - s, ds + s + shift, r_right_end - to_scan + s, q);
                }
            }
        } else if (ALN) c.lit(len);
        g.seg(F, len);
    }

    LZ_HD void run(int out[3]) { run_impl<0>(out, 0, nullptr, nullptr, 0, 0, nullptr, nullptr); }
    // one segment of a split pair (see SplitStart): from checkpoint `start` (nullptr: the start of the query) to the first
    // later checkpoint the scan passes through (cuts seg + 1 .. n_cuts - 1), or to the end of the query
    LZ_HD void run_segment(int seg, const SplitStart* start, const SplitStart* cuts, int n_cuts, SplitOut* so)
    { int dummy[3]; run_impl<2>(dummy, seg, start, cuts, n_cuts, 0, so, nullptr); }
    // the same segment on from where it stopped (its record *so is its state), past the cut it stopped at
    LZ_HD void resume_segment(int seg, const SplitStart* cuts, int n_cuts, SplitOut* so)
    { int dummy[3]; run_impl<2>(dummy, seg, nullptr, cuts, n_cuts, -1, so, nullptr); }
    // the checkpoint of a cut at query position p0: a fresh scan from there through its first event
    LZ_HD void run_checkpoint(int p0, SplitStart* cp)
    { int dummy[3]; run_impl<1>(dummy, 0, nullptr, nullptr, 0, p0, nullptr, cp); }

    // SPLIT: 0 = the whole pair; 1 = a checkpoint; 2 = a segment
    template <int SPLIT>
    LZ_HD void run_impl(int out[3], int seg, const SplitStart* start, const SplitStart* cuts, int n_cuts, int p0, SplitOut* so, SplitStart* cp)
    {
        int i = 0, lit = 0, r_end = 0;
        bool trk = false;
        int prev_rs = -1, prev_re = 0, pre_lit = 0;
        const int iend = D - P.msl;              // loop condition i + msl < |Q| (quirk Q10)
        int rounds = 0;                          // every round advances i by >= 1: hard exit bound
        [[maybe_unused]] bool first_open = false;         // SPLIT 2: the region the segment started in is still open
        [[maybe_unused]] bool tainted = false;            //          how far back a distant match may look is known as a lower bound only
        [[maybe_unused]] int next_cut = 0, events = 0, lim_i = -1;
        [[maybe_unused]] bool guess_line = false;         //          ... it kept its first region on a guess and has not kept another since
        [[maybe_unused]] bool floor_own = false;          //          ... it stands on its own floor (first region dropped by its own view): the true one may be HIGHER
        [[maybe_unused]] int need_floor = NO_CHECKPOINT;  //          ... and every look-back since holds for floors up to this one
        if constexpr (SPLIT == 1) { i = p0; cp->i = -1; cp->r_end = cp->prev_rs = cp->pre_lit = cp->cl = cp->clit = 0; }
        if constexpr (SPLIT == 2) {
          if (p0 < 0) {                             // resumed: the record is the state
            i = w.uniform(so->stop_i); r_end = w.uniform(so->stop_r); trk = true;
            prev_rs = w.uniform(so->open_rs); prev_re = i; pre_lit = prev_rs - w.uniform(so->floor);
            g.cl = w.uniform(so->open_cl); g.clit = w.uniform(so->open_clit);
            g.tm = w.uniform(so->tm); g.tl = w.uniform(so->tl); g.tc = w.uniform(so->tc);
            first_open = w.uniform(so->first) == 0; tainted = w.uniform(so->synced) == 0;
            floor_own = tainted && w.uniform(so->assumed) == 2;
            guess_line = tainted && w.uniform(so->assumed) == 3;
            if (floor_own) need_floor = w.uniform(so->first_floor);
            next_cut = w.uniform(so->stop) + 1;
            so->stop = -1;
          } else {
            so->tm = so->tl = so->tc = 0; so->first = 3; so->first_cl = so->first_clit = so->first_re = 0;
            so->stop = -1; so->open_cl = so->open_clit = 0; so->open_rs = -1;
            so->assumed = 0; so->first_floor = 0; so->synced = 1; so->floor = 0; so->stop_i = so->stop_r = 0;
            next_cut = seg + 1;
            if (start) {
                // (w.uniform: a value out of memory, the same in every lane -- the device's policy says so to the compiler)
                i = w.uniform(start->i); r_end = w.uniform(start->r_end); trk = true; prev_rs = w.uniform(start->prev_rs); prev_re = i;
                pre_lit = w.uniform(start->pre_lit);
                g.cl = w.uniform(start->cl); g.clit = w.uniform(start->clit);
                first_open = true; tainted = true; so->first = 0;
            }
          }
        }

        while (i < iend) {
            if (++rounds > D + 8) { LZ_GUARD_TRIP(3); out[0] = -1; out[1] = i; out[2] = lit; return; }
            if constexpr (SPLIT == 1) {
                if (events && trk && lit == 0) {
                    // behind an event of the fresh scan: a checkpoint (an event leaves lit = 0, tracking; where the scan STANDS does not
                    // depend on the region bookkeeping at all, so every such state is on the true scan's way once the two have
                    // met).  The later the better, up to a point: once the open region spans reg positions the segment that starts
                    // here knows how it will be closed (kept) -- a first region that is short in the segment's view is what voids
                    // most segments (kept or dropped by the true scan? a guess either way).  The search gives up after 64 events or
                    // 16 k positions (a zone where nothing is kept) and takes the last state.
                    cp->i = i; cp->r_end = r_end; cp->prev_rs = prev_rs; cp->pre_lit = pre_lit; cp->cl = g.cl; cp->clit = g.clit;
                    if (prev_re - prev_rs >= P.reg || events >= 64 || i - p0 > 16384) return;
                }
            }
            if constexpr (SPLIT == 2) {
                if (i >= lim_i) {                // (i only grows) at or beyond the nearest checkpoint ahead: look at the next few cuts
                    while (next_cut < n_cuts && i > w.uniform(cuts[next_cut].i)) ++next_cut;       // (passed: not on the scan's way)
                    lim_i = NO_CHECKPOINT;
                    for (int k = next_cut; k < n_cuts && k < next_cut + 4; ++k) {           // (a long extension may carry a cut's checkpoint beyond the next cut's)
                        const int ci = w.uniform(cuts[k].i);
                        if (ci == i && trk && lit == 0 && w.uniform(cuts[k].r_end) == r_end) {         // behind an event, in the checkpoint's very state: hand over
                            so->tm = g.tm; so->tl = g.tl; so->tc = g.tc;
                            so->stop = k; so->open_cl = g.cl; so->open_clit = g.clit; so->open_rs = prev_rs;
                            so->synced = !tainted; so->floor = prev_rs - pre_lit; so->stop_i = i; so->stop_r = r_end;
                            return;
                        }
                        if (ci > i) lim_i = imin(lim_i, ci);
                    }
                    w.split_limit(lim_i);        // (the policy's hand-written loops commit several events a call: none across a checkpoint)
                }
            }
            int adv = 0, bpos = 0, blen = 0;
            w.stamp(1);
            int in_hand = 0;
            if constexpr (wave_has_null_chain<W>::value && !ALN && SPLIT != 1) {
                if constexpr (SPLIT == 2) w.split_taint_set(tainted);      // (the chain then keeps to what holds for every look-back)
                if (!(SPLIT == 2 && (first_open || floor_own)))   // (the chain closes regions inside: not while the first one is apart; and its
                                                                  // records want a look-back that is a LOWER bound of the true one)
                // Straight after an event (tracking, nothing skipped yet) the policy may run the whole cycle
                // "tracking round without a seed candidate -> next plain candidate -> distant null event over a dropped
                // short region" for as many events as it lasts: exactly the updates of the null event below, nothing
                // else touched; what it leaves unfinished it hands back (the round done, or the event found).
                if ((trk & (lit == 0)) && w.chain_covers(i)) {      // (not where the scan has jumped over the queue: related stretches)
                    int last_cl = 0, last_clit = 0, add_tm = 0, add_tl = 0, add_tc = 0;
#if defined(LZANI_CHAIN_STATS) && defined(__HIP_DEVICE_COMPILE__)
                    w.cycles_mark(-1);                                 // wave cycles by what the chain handed back: closes the open interval
#endif
                    in_hand = w.null_chain(i, r_end, prev_rs, prev_re, pre_lit, last_cl, last_clit, adv, bpos, blen, g.cl, g.clit, add_tm, add_tl, add_tc);
#if defined(LZANI_CHAIN_STATS) && defined(__HIP_DEVICE_COMPILE__)
                    w.cycles_mark(in_hand);
#endif
                    g.tm += add_tm; g.tl += add_tl; g.tc += add_tc;             // regions the chain closed (kept ones it passed)
                    if (last_cl) { g.cl = last_cl; g.clit = last_clit; g.nl = 0; }  // discard + the match (+ forward extension) of the last event
                }
            }
            w.stamp(4);
            int sc_kind = 0;
            bool sc_hit = false;
            if constexpr (wave_has_stretch_chain<W>::value && !ALN && SPLIT != 1) {
                // a run of close matches behind each other, by the hand-scheduled stretch chain: the events it takes are
                // committed inside (the open region's accumulators, nl = 0 throughout); what ends the run comes back half
                // done -- an extension that runs on, an anchor before the first seed step, a round without a seed
                if (in_hand == 0 && trk && lit == 0 && g.nl == 0 && !w.chain_covers(i)) {
                    u64 sc_Bf = 0;
                    u32 sc_hq = 0;
                    int sc_adv = 0;
                    const int committed = w.stretch_chain(i, r_end, g.cl, g.clit, sc_kind, sc_Bf, sc_adv, sc_hq);
                    if (committed || sc_kind == 1) prev_re = i;
                    if (sc_kind == 1) {                      // gap and match are in: the rest of the forward extension
                        w.stamp(5);
                        const int e = extend_forward(i, r_end, true, sc_Bf);
                        i += e; r_end += e;
                        prev_re = i;
                        continue;
                    }
                    if (sc_kind == 2) {                      // the anchor of that step, by the wave: it is the event if it exists
                        int ap = 0, al = 0;
                        w.anchor_by_wave(sc_hq, i + sc_adv, ap, al);
                        if (ap != 0 && al >= P.msl) { adv = sc_adv; bpos = ap; blen = al; sc_hit = true; }
                    }
                }
            }
            const bool hit = sc_hit || in_hand >= 2 || w.find_event(i, iend - i, trk, r_end, lit, adv, bpos, blen);   // (2, 4: the chain found the event)
#if defined(LZANI_CHAIN_STATS) && defined(__HIP_DEVICE_COMPILE__)
            w.st[6] += hit && in_hand != 2;
#endif
#if defined(LZANI_PATH_STATS) && defined(__HIP_DEVICE_COMPILE__)
            w.ps[20] += hit; w.ps[21] += hit && (trk && lit + adv <= P.mqd && iabs(bpos - (r_end + lit + adv)) <= P.mrd);
#endif
            i += adv; lit += adv;
            if (!hit) {
                if (lit > P.mqd) trk = false;
                continue;
            }
            if constexpr (SPLIT == 1) {
                events += 1;
            }
#if defined(LZANI_EXP) && LZANI_EXP >= 1                     // diagnostic build: events found but not processed
            i += blen; r_end = bpos + blen; lit = 0; trk = true; prev_re = i;
            continue;
#endif
            bool strk = trk && lit <= P.mqd;
            int ref_pred = r_end + lit;
            w.stamp(3);
            int fq = 0, fr = 0;
            u64 Bf = 0, Bb = 0;
            bool haveF = false;
            if (strk && iabs(bpos - ref_pred) <= P.mrd) {
                // close match: fill the gap, then the match itself (parser.cpp:630-635; quirk Q2)
                if constexpr (wave_has_mism3<W>::value && !ALN) {
                    fq = i + blen; fr = bpos + blen;
                    gap_fill(i - lit, r_end, bpos + blen, lit, fq, fr, imax(0, imin(64, imin(D - fq, T - fr))), &Bf, &haveF);
                } else gap_fill(i - lit, r_end, bpos + blen, lit);
                match_run(i, bpos, blen);
            } else {
                // distant match (parser.cpp:636-685)
                int avail;
                // (a segment of a split pair: its FIRST region began elsewhere in the true scan, so its counts go aside for the stitch,
                // and whether the true scan keeps or drops it is the stitch's to say: the segment goes on by its own view of the
                // region's span and says which way it went -- see SplitOut)
                [[maybe_unused]] bool sure = true;         // SPLIT 2: `avail` is the true scan's, not a lower bound of it
                [[maybe_unused]] bool first_kept_guess = false;
                const bool drop = prev_rs >= 0 && prev_re - prev_rs < P.reg;
                if constexpr (SPLIT == 2) {
                    if (first_open) {
                        first_open = false;
                        so->first = 1; so->first_cl = g.cl; so->first_clit = g.clit; so->first_re = prev_re;
                        g.discard();
                        // (short in the segment's own view: the true region began earlier and may well be long enough -- a piece of a
                        // real alignment rather than a chance anchor.  Going on as if KEPT is right whenever the true scan keeps, and
                        // also when it drops but the backward scan below breaks inside the literals; going on as if DROPPED is right
                        // only when the true scan drops.  So: kept, unless the region looks like a chance anchor's and the literals
                        // are too few for a backward scan to break in them)
                        if (!drop) { so->assumed = 1; tainted = false; }
                        else if (lit >= LZ_SPLIT_GUESS_LIT * P.aw / 2 || so->first_cl >= P.reg / 2) { so->assumed = 3; first_kept_guess = true; guess_line = true; sure = false; }
                        else { so->assumed = 2; so->first_floor = prev_rs - pre_lit; sure = false; floor_own = true; }
                    } else if (tainted) { if (drop) sure = false; else { tainted = false; floor_own = false; guess_line = false; } }
                }
                if (__builtin_expect(drop && !first_kept_guess, 1)) {          // drop the short region
                    avail = pre_lit + (i - prev_rs);
                    g.discard();
                    if (ALN) c.clear();
                    prev_rs = -1;
                } else avail = lit;
                fq = i + blen; fr = bpos + blen;
                const int nb = avail > 0 ? imax(0, imin(64, imin(avail, imin(i, bpos)))) : 0;
                // The candidate's record may prove, without touching the texts, that an approximate extension does not
                // move (a chance k-mer with random flanks: four extensions out of five of an unrelated pair).
                u32 rec = EXT_REC_NONE;
                const bool have_rec = !ALN && !(SPLIT == 2 && !sure) && w.ext_record(rec);      // (a record speaks for ONE look-back bound)
                const bool null_f = have_rec && (rec & EXT_REC_NULLF);
                // (aw <= 15) the forward extension itself may be in the record: then neither side of it needs the texts
                const bool fwd_k = have_rec && P.aw <= 15 && (rec & EXT_REC_FWDK);
                const int fe = fwd_k ? (int)((rec >> 24) & 31u) : 0, fmm = fwd_k ? (int)((rec >> 20) & 15u) : 0;
                const bool fwd_free = null_f | fwd_k;
                const int reach = imin(avail, imin(i, bpos));
                const bool null_b = nb == 0 || (have_rec && ext_rec_null_bwd(rec, reach, P.aw));
                int kb = 0, kc = 0;                                     // the backward extension from the record, if it is there
                const bool bwd_k = have_rec && !null_b && ext_rec_bwd_known(rec, reach, P.aw, kb, kc);
                const bool bwd_free = null_b | bwd_k;
                if (null_f & null_b) {
                    // the null event: no text access, no lane work; the match opens a region on its own
                    g.finalize();                                       // a match_distant factor follows
                    pre_lit = avail; prev_rs = i;
                    g.seg_match_run(blen);
                    i += blen; r_end = bpos + blen; lit = 0; trk = true;
                    prev_re = i;
                    continue;
                }
                int b = 0;
                [[maybe_unused]] bool bounded = SPLIT == 2 && nb == 0 && imin(i, bpos) > 0;     // (no look-back at all where the true scan may have one)
                if (bwd_free | fwd_free) {
                    // one side is known (empty, or the extension itself from the record): fetch and scan the other one only
                    if (!bwd_free) { Bb = w.mism_bwd(i, bpos, nb); b = extend_backward(i, bpos, avail, true, Bb); }
                    else b = kb;
                } else {
                    // the first chunks of the backward and of the forward extension are fetched together (one
                    // memory wait); the fold of the backward part comes out of the same mask
                    const int nf = imax(0, imin(64, imin(D - fq, T - fr)));
                    w.mism2(fq, fr, 1, nf, i - 1, bpos - 1, -1, nb, Bf, Bb);
                    haveF = true;
                    if constexpr (SPLIT == 2) {
                        if (nb > 0) {
                            int looked = 0;
                            b = extend_backward(i, bpos, avail, true, Bb, &bounded, &looked);
                            // (a look-back that may be LONGER than the true scan's -- the segment dropped its first region by its own
                            // view and stands on its own floor: the result holds for every floor at or below the scan's break)
                            if (floor_own) need_floor = imin(need_floor, i - looked);
                        }
                    } else
                    b = nb > 0 ? extend_backward(i, bpos, avail, true, Bb) : 0;
                }
                if constexpr (SPLIT == 2) {
                    // (kept on a guess, and every look-back up to the next keep: if the guess is right the segment's floor IS the true
                    // one and a scan that runs to its bound is the true scan's; the stitch voids the segment only if the guess was wrong)
                    if (first_kept_guess) so->first_floor = bounded;
                    else if (!sure && bounded && guess_line) so->first_floor = 1;
                    else if (!sure && bounded) { so->first = 2; return; }     // the lower bound cut the scan short: void
                    if (floor_own) so->first_floor = need_floor;              // (the stitch: the true floor at or below it)
                }
                g.finalize();                                           // a match_distant factor follows
                region_close();
                if (__builtin_expect(b > 0, 0)) {
                    pre_lit = avail - b;
                    if (bwd_k) {
                        // = seg(M, b) below, up to how the b - kc literals split between clit and nl -- and the match run
                        // that follows at once moves nl into clit whatever the split
                        g.cl += kc; g.clit += g.nl + (b - kc); g.nl = 0;
                    } else if (__builtin_expect(b <= nb, 1)) {          // forward order = the b mask bits reversed
                        const u64 M = brev64(~Bb & lowmask(b)) >> (64 - b);
                        g.seg(M, b);
                        if (ALN) runs(M, b, i - b, bpos - b, 0);
                    } else seg_range(i - b, bpos - b, b);
                    prev_rs = i - b;
                } else { pre_lit = avail; prev_rs = i; }
                match_run(i, bpos, blen);
                if (fwd_free) {                                         // (then !ALN) the forward extension is empty or known
                    i += blen; r_end = bpos + blen; lit = 0; trk = true;
                    if (fe > 0) {                                       // = extend_forward: fe symbols, the last one a match
                        g.cl += fe - fmm; g.clit += g.nl + fmm; g.nl = 0;
                        i += fe; r_end += fe;
                    }
                    prev_re = i;
                    continue;
                }
            }
            i += blen;
            r_end = bpos + blen;
            lit = 0;
            trk = true;
            w.stamp(5);
            int e = extend_forward(i, r_end, haveF, Bf);
            i += e; r_end += e;
            prev_re = i;
        }
        w.stamp(6);
        if constexpr (SPLIT == 1) return;          // (no event behind the cut: no checkpoint)
        if (trk)   // tail compare against r_end - msl (parser.cpp:713, quirk Q3)
            seg_range(i - lit, r_end - P.msl, lit + (D - i));
        if constexpr (SPLIT == 2) {
            if (first_open) {                      // the first region is open to the end: the stitch closes it with the true counts
                so->open_cl = g.cl; so->open_clit = g.clit; so->open_rs = prev_rs;
                g.discard();
            }
            g.finalize();
            so->tm = g.tm; so->tl = g.tl; so->tc = g.tc; so->stop = -1;
            return;
        }
        g.finalize();
        region_close();
        out[0] = g.tm; out[1] = g.tl; out[2] = g.tc;
    }
};

}  // namespace lzani
