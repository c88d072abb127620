// lzani_hip.hip -- HIP kernels for gfx950 (MI355X) and the C-ABI (include/lzani.h) above them.
//
// Kernels (all integer / bit work, HBM- and L2-bound; no MFMA by design):
//   k_pack        reservoir codes -> packed reference text  fwd | N^2mrd | RC | N^mrd
//                 (replaces seq_view::unpack + CParser::append/append_rc, parser.h:57-96)
//   k_idx_count / k_idx_scan / k_idx_fill / k_idx_sort
//                 per-reference anchor index of all mal-mers (replaces prepare_kmers +
//                 prepare_ht_long, parser.cpp:53-103, 146-189): bucket directory + (tag|pos)
//                 entries, ascending inside a bucket
//   k_kmers       per-genome k-mer words: mixed mal-mer hash and msl-mer of every text position
//                 (replaces the per-pair prepare_kmers of prepare_data, parser.cpp:46-47)
//   k_pairs       one wavefront per directed genome pair, persistent waves pulling pairs from
//                 per-XCD work queues (replaces prepare_data + parse + calc_stats,
//                 parser.cpp:37-50, 482-716, 734-783, and the worker loop of do_matching,
//                 lz_matcher.cpp:192-269); instantiations FAST/NFREE/DEFP/ALN, see the kernel
//   k_pairs_tpp   thread-per-pair variant of the same machine (opt-in, LZANI_KERNEL=tpp; slower)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/lzani.h"

__device__ int g_guard_trip = 0;   // see LZ_GUARD_TRIP in lzani_core.h
#ifdef LZANI_STAMPS
__device__ unsigned long long g_stamp_acc[8];
#endif

#include "lzani_core.h"
#include "lzani_layout.h"

namespace lzani {

struct GenomeTab {
    const u64* t2;       // all packed texts, concatenated
    const u64* nm;       // all N masks, concatenated
    const u64* nmoff;    // per genome: word offset into nm (t2 offset is twice that, k-mer arrays 64x)
    const int* L;        // per genome: sequence length
    const u32* kmL;      // per text position: mix_key(mal-mer) or KM_INVALID   (fast path: mal, msl <= 15)
    const u32* kmS;      // per text position: msl-mer or KM_INVALID
    const int* hasN;     // per genome: 1 if the sequence holds a non-ACGT symbol
};

enum : u32 { KM_INVALID = 0xFFFFFFFFu };

// k_kmers: one thread per text position of every genome: the two k-mer words the pair kernel and
// the index build read instead of re-extracting k-mers (the reference recomputes them per pair,
// parser.cpp:46-47; here once per genome and run).
__global__ void k_kmers(GenomeTab G, u32* __restrict__ kmL, u32* __restrict__ kmS, int mal, int msl, int mrd, int Tmax)
{
    u32 g = blockIdx.y;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int T = ref_text_len(G.L[g], mrd);
    if (p >= Tmax || p >= T) return;
    u64 o = G.nmoff[g];
    TextView R = ref_view(G.t2 + 2 * o, G.nm + o, G.L[g], mrd, false);
    u64 key;
    u32 a = KM_INVALID, b = KM_INVALID;
    if (kmer_at(R, p, mal, key)) a = (u32)mix_key(key, 2 * mal);
    if (kmer_at(R, p, msl, key)) b = (u32)key;
    kmL[64 * o + p] = a;
    kmS[64 * o + p] = b;
}

// ------------------------------------------------------------------------------------------
// k_pack: one thread per 64-symbol block of a reference text.
// ------------------------------------------------------------------------------------------
__global__ void k_pack(const uint8_t* __restrict__ codes, const u64* __restrict__ codeoff,
                       u64* __restrict__ t2, u64* __restrict__ nm, const u64* __restrict__ nmoff,
                       const int* __restrict__ Ls, int* __restrict__ hasN, int mrd, u32 n)
{
    u32 g = blockIdx.y;
    if (g >= n) return;
    int L = Ls[g];
    int T = ref_text_len(L, mrd);
    size_t nblk = text_wordsN(T);
    size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    const uint8_t* c = codes + codeoff[g];
    u64 w0 = 0, w1 = 0, nw = 0;
    int rc0 = L + 2 * mrd;
    for (int j = 0; j < 64; ++j) {
        long p = (long)blk * 64 + j;
        int s = 4;
        if (p < L) { int v = c[p]; s = v < 4 ? v : 4; if (v >= 4) hasN[g] = 1; }
        else if (p >= rc0 && p < rc0 + L) { int v = c[L - 1 - (p - rc0)]; s = v < 4 ? 3 - v : 4; }
        if (s < 4) {
            if (j < 32) w0 |= (u64)s << (2 * j);
            else w1 |= (u64)s << (2 * (j - 32));
        } else nw |= 1ULL << j;
    }
    size_t o = nmoff[g] + blk;
    nm[o] = nw;
    t2[2 * o] = w0;
    t2[2 * o + 1] = w1;
}

// ------------------------------------------------------------------------------------------
// Anchor index build.  Slot s of the batch holds the index of reference ref_ids[s].
// ------------------------------------------------------------------------------------------
struct IdxArgs {
    GenomeTab G;
    const u32* ref_ids;      // device, batch-relative
    u32* dirz;               // slots * dir_stride
    u32* ent;                // slots * ent_stride
    u64 dir_stride, ent_stride;
    int mal, mrd;            // mal = the k of this index (min_anchor_len, or min_seed_len for the seed index)
    IndexGeom geo;
    int seed;                // 1: index of the msl-mers (key words from kmS, mixed here)
};

__device__ __forceinline__ bool idx_slot_key(const IdxArgs& a, u32 slot, int p, u32& bucket, u32& entry)
{
    u32 g = a.ref_ids[slot];
    int T = ref_text_len(a.G.L[g], a.mrd);
    if (p + a.mal > T) return false;
    u64 o = a.G.nmoff[g];
    u64 h;
    if (a.G.kmL) {
        u32 v = a.seed ? a.G.kmS[64 * o + p] : a.G.kmL[64 * o + p];
        if (v == KM_INVALID) return false;
        h = a.seed ? mix_key(v, a.geo.kb) : (u64)v;
    } else {
        TextView R = ref_view(a.G.t2 + 2 * o, a.G.nm + o, a.G.L[g], a.mrd, false);
        u64 key;
        if (!kmer_at(R, p, a.mal, key)) return false;
        h = mix_key(key, a.geo.kb);
    }
    int tb = a.geo.kb - a.geo.dirbits;
    bucket = (u32)(h >> tb);
    u32 tag = (u32)(h & lowmask(tb)) & a.geo.tagmask;
    entry = (tag << a.geo.posbits) | (u32)p;
    return true;
}

__global__ void k_idx_count(IdxArgs a, int Tmax)
{
    u32 slot = blockIdx.y;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Tmax) return;
    u32 b, e;
    if (idx_slot_key(a, slot, p, b, e)) atomicAdd(&a.dirz[slot * a.dir_stride + 1 + b], 1u);
}

// In-place exclusive scan of the 2^dirbits bucket counts of one slot (one 1024-thread block).
__global__ void __launch_bounds__(1024) k_idx_scan(u32* dirz, u64 dir_stride, u32 nb)
{
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    u32* cnt = dirz + (u64)blockIdx.x * dir_stride + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < nb; base += 1024) {
        u32 idx = base + threadIdx.x;
        u32 v = idx < nb ? cnt[idx] : 0;
        u32 x = v;                                   // inclusive scan inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            u32 y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        u32 woff = 0;
        for (int k = 0; k < wv; ++k) woff += wsum[k];
        u32 carry = carry_s;
        if (idx < nb) cnt[idx] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
}

__global__ void k_idx_fill(IdxArgs a, int Tmax)
{
    u32 slot = blockIdx.y;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Tmax) return;
    u32 b, e;
    if (idx_slot_key(a, slot, p, b, e)) {
        u32 at = atomicAdd(&a.dirz[slot * a.dir_stride + 1 + b], 1u);
        a.ent[slot * a.ent_stride + at] = e;
    }
}

// Ascending order inside every bucket (candidate order = ascending reference position per
// k-mer, the order of the reference's probe chain).  Buckets hold ~1 entry.
__global__ void k_idx_sort(u32* dirz, u32* ent, u64 dir_stride, u64 ent_stride, u32 nb)
{
    u32 slot = blockIdx.y;
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const u32* d = dirz + slot * dir_stride;
    u32 s = d[b], e = d[b + 1];
    u32* v = ent + slot * ent_stride;
    for (u32 i = s + 1; i < e; ++i) {
        u32 x = v[i];
        u32 j = i;
        while (j > s && v[j - 1] > x) { v[j] = v[j - 1]; --j; }
        v[j] = x;
    }
}

// Bucket table: the first four entries of every bucket side by side (16 B), so that a round reaches its
// anchor candidates with one load after the k-mer word instead of directory + entries.
// Tag words (tw != nullptr): the tags of those four entries in one 32-bit word, a byte 0x80|tag each, so that
// a round DETECTS its anchor candidates from a table a quarter the size (it stays in the XCD's L2 while the
// waves of the XCD move from one reference to the next) and only a candidate step reads the 16-byte bucket.
__global__ void k_idx_buckets(const u32* __restrict__ dirz, const u32* __restrict__ ent, u32* __restrict__ bk,
                              u32* __restrict__ tw, u64 dir_stride, u64 ent_stride, u64 bk_stride, u64 tw_stride,
                              u32 nb, int posbits)
{
    u32 slot = blockIdx.y;
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const u32* d = dirz + slot * dir_stride;
    const u32* v = ent + slot * ent_stride;
    const u32 s = d[b], e = d[b + 1];
    uint4 o;
    o.x = s < e ? v[s] : BK_EMPTY;
    o.y = s + 1 < e ? v[s + 1] : BK_EMPTY;
    o.z = s + 2 < e ? v[s + 2] : BK_EMPTY;
    o.w = s + 3 < e ? v[s + 3] : BK_EMPTY;
    if (e - s > 4) o.w = BK_OVERFLOW;
    reinterpret_cast<uint4*>(bk + slot * bk_stride)[b] = o;
    if (tw) {
        u32 w = 0;
        if (s < e) w |= 0x80u | (o.x >> posbits);
        if (s + 1 < e) w |= (0x80u | (o.y >> posbits)) << 8;
        if (s + 2 < e) w |= (0x80u | (o.z >> posbits)) << 16;
        if (s + 3 < e) w |= (0x80u | (o.w >> posbits)) << 24;
        if (e - s > 4) w = TW_OVERFLOW;
        tw[slot * tw_stride + b] = w;
    }
}

// ------------------------------------------------------------------------------------------
// k_pairs: the pair kernel.
// ------------------------------------------------------------------------------------------
enum { SEED_SLOT_BITS = 8, SEED_SLOTS = 1 << SEED_SLOT_BITS, SEED_BM_BITS = 14, SEED_BM_WORDS = 1 << (SEED_BM_BITS - 5),
       SEED_LDS_WORDS = SEED_SLOTS + 256 + SEED_BM_WORDS, NQUEUES = 8 };

// FAST: per-position k-mer words exist;  BK: the bucket table and its tag words exist
template <bool FAST, bool BK = false>
struct DevWave {
    const Params& P;
    TextView R, Q;
    IndexView I;
    int lane;
    u32* heads;      // per-wave LDS: SEED_SLOTS chain heads
    u32* nexts;      // 128 chain links
    u32* keys;       // 128 window msl-mers
    u32* bitmap;     // SEED_BM_WORDS words, all zero between rounds
    const u32* rkS;  // FAST: msl-mers of the reference text, one per position
    const u32* qkL;  // FAST: hashed mal-mers of the query text
    const u32* qkS;  // FAST: msl-mers of the query text
    // alignment instantiation: region sink
    lzani_region* reg_out;
    unsigned long long* reg_count;
    unsigned long long reg_cap, pair_e;
    __device__ __forceinline__ void emit_region(const RegionCoords& c) const
    {
        // one slot per wave without a lane-dependent branch (see the note at the ticket fetch)
        const unsigned long long old = atomicAdd(reg_count, lane == 0 ? 1ULL : 0ULL);
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
        return __ballot((lane < n) & lane_mismatch(r0 + lane, q0 + lane));
    }
    __device__ __forceinline__ u64 mism_bwd(int q0, int r0, int n) const
    {
        return __ballot((lane < n) & lane_mismatch(r0 - 1 - lane, q0 - 1 - lane));
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
        A = __ballot((lane < na) & ma);
        B = __ballot((lane < nb) & mb);
    }
    // Close-seed search of all tracking lanes of a round at once (replaces the ht_short bucket walk,
    // parser.cpp:548-580).  rk0/rk1 = msl-mers of the window positions r_end+lane / r_end+64+lane,
    // qk = msl-mer of this lane's step (KM_INVALID where there is none).
    //  1. prefilter: the window k-mers set bits in a per-wave 16 Kbit LDS bitmap (exact for msl <= 7,
    //     a Bloom filter above), each tracking lane tests its own k-mer; in four rounds out of five no
    //     lane hits and the search ends here (the bits are cleared again, the bitmap is always zero
    //     between rounds);
    //  2. otherwise the window k-mers are chained into a small LDS hash table and every hit lane
    //     walks the chain of its k-mer, collecting the matching positions below its own window limit
    //     into a 128-bit mask; candidates are then taken in ascending position, the order of the
    //     reference's bucket.
    __device__ __forceinline__ u32 bm_hash(u32 k) const
    {
        return P.msl <= 7 ? k : (k * 0x9E3779B1u) >> (32 - SEED_BM_BITS);
    }
    __device__ __forceinline__ void seed_join(int lit, u32 rk0, u32 rk1, u32 qk, u64& c0, u64& c1) const
    {
        c0 = 0; c1 = 0;
        const u32 EMPTY = 0xFFFFFFFFu;
        const u32 b0 = bm_hash(rk0), b1 = bm_hash(rk1), bq = bm_hash(qk);
        if (rk0 != KM_INVALID) atomicOr(&bitmap[b0 >> 5], 1u << (b0 & 31));
        if (rk1 != KM_INVALID) atomicOr(&bitmap[b1 >> 5], 1u << (b1 & 31));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool hit = false;
        if (qk != KM_INVALID) hit = (bitmap[bq >> 5] >> (bq & 31)) & 1u;
        const u64 any = __ballot(hit);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rk0 != KM_INVALID) bitmap[b0 >> 5] = 0;
        if (rk1 != KM_INVALID) bitmap[b1 >> 5] = 0;
        if (!any) return;

        for (int k = 0; k < SEED_SLOTS / 64; ++k) heads[lane + 64 * k] = EMPTY;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rk0 != KM_INVALID) {
            keys[lane] = rk0;
            nexts[lane] = atomicExch(&heads[(rk0 * 0x9E3779B1u) >> (32 - SEED_SLOT_BITS)], (u32)lane);
        }
        if (rk1 != KM_INVALID) {
            keys[lane + 64] = rk1;
            nexts[lane + 64] = atomicExch(&heads[(rk1 * 0x9E3779B1u) >> (32 - SEED_SLOT_BITS)], (u32)lane + 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (hit) {
            const u32 lim = (u32)(lit + lane + P.mrd);          // this step's window is [0, lim)
            int guard = 0;
            for (u32 h = heads[(qk * 0x9E3779B1u) >> (32 - SEED_SLOT_BITS)]; h != EMPTY; h = nexts[h]) {
                if (++guard > 128) { LZ_GUARD_TRIP(4); break; }
                const u64 bit = (u64)(keys[h] == qk && h < lim) << (h & 63);
                c0 |= h < 64 ? bit : 0;
                c1 |= h < 64 ? 0 : bit;
            }
        }
        __builtin_amdgcn_wave_barrier();
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
        u64 hit = __ballot(lane < n && bl >= P.msl);
        if (!hit) return false;
        ev_lane = ctz64(hit);
        bpos = __builtin_amdgcn_readlane(bp, ev_lane);     // ev_lane is wave-uniform (from the ballot)
        blen = __builtin_amdgcn_readlane(bl, ev_lane);
        return true;
    }

    // Fast round (k-mer words available, seed window <= 128): the lanes only DETECT candidates --
    // a bucket entry whose tag equals the step's mal-mer, a window position whose msl-mer equals the
    // step's -- and the wave then verifies the candidates of the first candidate lane together
    // (wave_equal_len), exactly as eval_step would for that step; if that step turns out not to hit
    // (quirk Q1, or mal < msl) the next candidate lane is taken.
    __device__ __forceinline__ bool find_event(int i, int n, bool trk, int r_end, int lit,
                                               int& ev_lane, int& bpos, int& blen) const
    {
        const int nt = trk ? imin(n, P.mqd - lit + 1) : 0;          // lanes [0, nt) are tracking steps
        const int W = nt > 0 ? imin(lit + nt - 1 + P.mrd, R.len - P.msl + 1 - r_end) : 0;
        const int tb = I.kb - I.dirbits;
        if (!FAST || W > 128 || (!BK && I.tagmask != (u32)lowmask(tb))) // the stored tag must identify the k-mer
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
        u32 aj = 0, ac = 0;
        uint4 bkv = {BK_EMPTY, BK_EMPTY, BK_EMPTY, BK_EMPTY};
        bool viadir = !BK && I.bk == nullptr;
        if constexpr (BK) {
            const bool valid = hq != KM_INVALID;
            const u32 w = I.tw[valid ? hq >> tb : 0u];
            const u32 x = w ^ ((0x80u | (hq & I.tagmask)) * 0x01010101u);       // a zero byte = a slot with this tag
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
                for (u32 j = s; j < e; ++j) {
                    const bool m = (I.ent[j] >> I.posbits) == tag;
                    aj = (m && ac == 0) ? j : aj;
                    ac += m;
                }
            }
        }
        stamp(2);
        u64 c0 = 0, c1 = 0;
        if (W > 0) seed_join(lit, rk0, rk1, qk, c0, c1);
        u64 todo = __ballot(ac != 0 || (c0 | c1) != 0);
        stamp(7);
        const u32 pm = (u32)lowmask(I.posbits);
        while (todo) {
            const int l = ctz64(todo);
            todo &= todo - 1;
            const int qp = i + l;
            int ap = 0, al = 0;
            const u32 cnt = __builtin_amdgcn_readlane(ac, l);
            if (BK && cnt) {                                         // candidate step: now its bucket is read, by the wave
                const u32 hql = (u32)__builtin_amdgcn_readlane((int)hq, l), tag = hql & I.tagmask;
                const uint4 bq = reinterpret_cast<const uint4*>(I.bk)[hql >> tb];
                const u32 en[4] = {(u32)__builtin_amdgcn_readfirstlane((int)bq.x), (u32)__builtin_amdgcn_readfirstlane((int)bq.y),
                                   (u32)__builtin_amdgcn_readfirstlane((int)bq.z), (u32)__builtin_amdgcn_readfirstlane((int)bq.w)};
                if (en[3] == BK_OVERFLOW) {                          // the whole bucket, ascending position
                    u32 s = I.dirz[hql >> tb], e = I.dirz[(hql >> tb) + 1];
                    if (e - s > (u32)R.len || e < s) { LZ_GUARD_TRIP(2); e = s; }
                    for (u32 j = s; j < e; ++j) {
                        const u32 x = (u32)__builtin_amdgcn_readfirstlane((int)I.ent[j]);
                        if ((x >> I.posbits) != tag) continue;
                        const int p = (int)(x & pm);
                        const int m = wave_equal_len(p, qp, 0);
                        if (m >= P.mal && m > al) { al = m; ap = p; }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((en[k] >> I.posbits) != tag) continue;
                        const int p = (int)(en[k] & pm);
                        const int m = wave_equal_len(p, qp, 0);
                        if (m >= P.mal && m > al) { al = m; ap = p; }
                    }
                }
            } else if (cnt) {
                if (__builtin_amdgcn_readlane((int)viadir, l)) {
                    const u32 j0 = __builtin_amdgcn_readlane(aj, l);
                    for (u32 k = 0; k < cnt; ++k) {                  // same k-mer, ascending position
                        const int p = (int)(I.ent[j0 + k] & pm);
                        const int m = wave_equal_len(p, qp, 0);
                        if (m >= P.mal && m > al) { al = m; ap = p; }
                    }
                } else {                                             // the step's bucket is still in registers
                    const u32 tag = (u32)__builtin_amdgcn_readlane((int)hq, l) & I.tagmask;
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
                u64 d0 = bcast64(c0, l), d1 = bcast64(c1, l);
                while (d0 | d1) {
                    int idx;
                    if (d0) { idx = ctz64(d0); d0 &= d0 - 1; }
                    else { idx = 64 + ctz64(d1); d1 &= d1 - 1; }
                    seed_consider(r_end + idx, wave_equal_len(r_end + idx, qp, P.msl), ref_pred, sp, sl);
                }
                arbitrate(P, R.len, lit + l, ap, al, sp, sl);
                bp = sp; bl = sl;
            }
            if (bl >= P.msl) { ev_lane = l; bpos = bp; blen = bl; return true; }
        }
        return false;
    }
    __device__ __forceinline__ ExtMasks ext_scan(u64 prevB, u64 B, int n) const
    {
        bool b, q;
        ext_lane(prevB, B, lane, n, P.aw, P.am, P.ar, b, q);
        ExtMasks m;
        m.brk = __ballot(b);
        m.qual = __ballot(q);
        return m;
    }
    __device__ __forceinline__ int best_split(u64 Lm, u64 Rm, int to_scan) const
    {
        // lane s scores split s; to_scan can be 64, so split 64 is scored by every lane too
        int key = -1;
        if (lane <= to_scan) key = (popc64(Lm & lowmask(lane)) + popc64(Rm >> lane)) * 128 + lane;
        if (to_scan == 64) key = imax(key, popc64(Lm) * 128 + 64);
        for (int d = 32; d >= 1; d >>= 1) key = imax(key, __shfl_xor(key, d));
        return key & 127;
    }
};

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
    lzani_region* reg_out;        // alignment instantiation only
    unsigned long long* reg_count;
    unsigned long long reg_cap;
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
template <bool FAST, bool NFREE, bool DEFP, bool ALN = false, bool BK = false>
__global__ void __launch_bounds__(256, 8) k_pairs(PairArgs a)
{
    const Params Pk = DEFP ? Params{11, 7, 40, 40, 35, 15, 7, 3} : a.P;
    const int lane = threadIdx.x & 63;
    __shared__ u32 s_seed[4][SEED_LDS_WORDS];
    u32* const lds = s_seed[threadIdx.x >> 6];
    for (int k = lane; k < SEED_BM_WORDS; k += 64) lds[SEED_SLOTS + 256 + k] = 0;
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
        u32 lo = rb, hi = re;                     // last row of the queue with qcum[row] <= tk
        while (hi - lo > 1) {
            u32 mid = (lo + hi) >> 1;
            if (a.qcum[mid] <= tk) lo = mid; else hi = mid;
        }
        const u32 slot = a.qorder[lo];
        const u32 r = a.ref_ids[slot];
        const u32 j = (u32)(tk - a.qcum[lo]);
        const u64 e = a.row_off[slot] + j;
        const u32 q = a.query_ids ? a.query_ids[e] : j + (j >= r ? 1u : 0u);

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
        DevWave<FAST, BK> w{Pk, ref_view(a.G.t2 + 2 * ro, a.G.nm + ro, Lr, Pk.mrd, nfree),
                        qry_view(a.G.t2 + 2 * qo, a.G.nm + qo, Lq, Pk.mrd, nfree), iv, lane,
                        lds, lds + SEED_SLOTS, lds + SEED_SLOTS + 128, lds + SEED_SLOTS + 256,
                        FAST ? a.G.kmS + 64 * ro : nullptr, FAST ? a.G.kmL + 64 * qo : nullptr,
                        FAST ? a.G.kmS + 64 * qo : nullptr, a.reg_out, a.reg_count, a.reg_cap, e};
        PairMachine<DevWave<FAST, BK>, ALN> m(w, Pk, T, D);
        int res[3];
#ifdef LZANI_STAMPS
        for (int k = 0; k < 8; ++k) w.acc[k] = 0;
        w.cur = 0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w.t0) :: "memory");
#endif
        m.run(res);
#ifdef LZANI_STAMPS
        w.stamp(0);
        if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamp_acc[k], w.acc[k]);
#endif
        int* o = a.out + 3 * e;          // every lane stores the same wave-uniform values
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
    }
}

// ------------------------------------------------------------------------------------------
// k_pairs_tpp: thread-per-pair variant.  Every lane owns one directed pair and runs the same pair
// machine with the lane-serial policy (LaneWave, lzani_core.h): 64 independent pairs per wavefront,
// no cross-lane traffic, the close seeds come from a second (msl) index of the reference.  Lanes pull
// their pairs from the per-XCD queues one ticket each, so a finished lane never waits for its wave.
// ------------------------------------------------------------------------------------------
struct TppArgs {
    PairArgs pa;
    const u32* sdirz;        // seed index slabs, like dirz/ent
    const u32* sent;
    u64 sdir_stride, sent_stride;
    IndexGeom sgeo;
};

template <bool NFREE, bool DEFP>
__global__ void __launch_bounds__(256) k_pairs_tpp(TppArgs ta)
{
    const PairArgs& a = ta.pa;
    const Params Pk = DEFP ? Params{11, 7, 40, 40, 35, 15, 7, 3} : a.P;
    u32 qx = xcc_id() % NQUEUES, dry = 0;
    for (;;) {
        const unsigned long long t = atomicAdd(&a.cursor[qx], 1ULL);     // one ticket per lane
        const u32 rb = a.qb[qx], re = a.qb[qx + 1];
        const u64 tk = a.qcum[rb] + t;
        if (tk >= a.qcum[re]) {
            if (++dry >= NQUEUES) break;
            qx = (qx + 1) % NQUEUES;
            continue;
        }
        u32 lo = rb, hi = re;
        while (hi - lo > 1) {
            u32 mid = (lo + hi) >> 1;
            if (a.qcum[mid] <= tk) lo = mid; else hi = mid;
        }
        const u32 slot = a.qorder[lo];
        const u32 r = a.ref_ids[slot];
        const u32 j = (u32)(tk - a.qcum[lo]);
        const u64 e = a.row_off[slot] + j;
        const u32 q = a.query_ids ? a.query_ids[e] : j + (j >= r ? 1u : 0u);
        const int Lr = a.G.L[r], Lq = a.G.L[q];
        const u64 ro = a.G.nmoff[r], qo = a.G.nmoff[q];
        const int T = ref_text_len(Lr, Pk.mrd), D = Lq + Pk.mrd;
        IndexView iv, sv;
        iv.dirz = a.dirz + slot * a.dir_stride;
        iv.ent = a.ent + slot * a.ent_stride;
        iv.kb = a.geo.kb; iv.dirbits = a.geo.dirbits; iv.posbits = a.geo.posbits; iv.tagmask = a.geo.tagmask;
        iv.bk = nullptr; sv.bk = nullptr;
        iv.tw = nullptr; sv.tw = nullptr;
        sv.dirz = ta.sdirz + slot * ta.sdir_stride;
        sv.ent = ta.sent + slot * ta.sent_stride;
        sv.kb = ta.sgeo.kb; sv.dirbits = ta.sgeo.dirbits; sv.posbits = ta.sgeo.posbits; sv.tagmask = ta.sgeo.tagmask;
        const bool nfree = NFREE ? true : !(a.G.hasN[r] | a.G.hasN[q]);
        LaneWave w{Pk, ref_view(a.G.t2 + 2 * ro, a.G.nm + ro, Lr, Pk.mrd, nfree),
                   qry_view(a.G.t2 + 2 * qo, a.G.nm + qo, Lq, Pk.mrd, nfree), iv, sv,
                   a.G.kmL + 64 * qo, a.G.kmS + 64 * qo};
        PairMachine<LaneWave> m(w, Pk, T, D);
        int res[3];
        m.run(res);
        int* o = a.out + 3 * e;
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
    }
}

}  // namespace lzani

// ============================================================================================
// Host side of the C-ABI
// ============================================================================================
using namespace lzani;

struct lzani_ctx {
    Params P;
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::string err;

    u32 n = 0;
    std::vector<int> L;
    std::vector<u64> nmoff;
    int Tmax = 0;
    IndexGeom geo{};
    u64* d_t2 = nullptr;
    u64* d_nm = nullptr;
    u64* d_nmoff = nullptr;
    int* d_L = nullptr;
    int* d_hasN = nullptr;
    u32* d_kmL = nullptr;     // k-mer arrays (fast path: mal, msl <= 15), 64 entries per nm word
    u32* d_kmS = nullptr;
    u64 total_nm = 0;
    bool kmers_ready = false;
    bool all_nfree = false;       // no genome holds an N: the NFREE kernel instantiation applies

    u32* d_dirz = nullptr;
    u32* d_ent = nullptr;
    u32* d_bk = nullptr;          // bucket tables (viral-size directories only)
    u64 bk_stride = 0;
    u32* d_tw = nullptr;          // tag words of the bucket tables (tag bits <= 7)
    u64 tw_stride = 0;
    u32* d_sdirz = nullptr;       // seed (msl) index slabs, thread-per-pair kernel only
    u32* d_sent = nullptr;
    u64 sdir_stride = 0;
    IndexGeom sgeo{};
    bool use_tpp = false;
    u32 slots = 0;
    u64 dir_stride = 0, ent_stride = 0;
    unsigned long long* d_cursor = nullptr;

    lzani_timing tm{};
};

namespace {

bool trace_on()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("LZANI_TRACE"); on = (e && *e && *e != '0') ? 1 : 0; }
    return on == 1;
}
#define TRACE(...) do { if (trace_on()) { fprintf(stderr, "[lzani] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// Temporary device buffer, released on every exit path of the call that owns it.
template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)); }
    operator T*() const { return p; }
};

int fail(lzani_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(c, e_ == hipErrorOutOfMemory ? LZANI_ERR_NOMEM : LZANI_ERR_DEVICE,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));                           \
    } while (0)

void free_genomes(lzani_ctx* c)
{
    hipFree(c->d_t2); hipFree(c->d_nm); hipFree(c->d_nmoff); hipFree(c->d_L); hipFree(c->d_kmL); hipFree(c->d_kmS); hipFree(c->d_hasN);
    c->d_hasN = nullptr;
    c->d_t2 = c->d_nm = c->d_nmoff = nullptr; c->d_L = nullptr; c->d_kmL = c->d_kmS = nullptr; c->kmers_ready = false;
    c->n = 0;
}
void free_slabs(lzani_ctx* c)
{
    hipFree(c->d_dirz); hipFree(c->d_ent); hipFree(c->d_sdirz); hipFree(c->d_sent); hipFree(c->d_bk); hipFree(c->d_tw);
    c->d_dirz = c->d_ent = c->d_sdirz = c->d_sent = c->d_bk = c->d_tw = nullptr; c->slots = 0;
}

int ensure_slabs(lzani_ctx* c, u32 want_rows)
{
    {   // thread-per-pair kernel: diagnostic opt-in (LZANI_KERNEL=tpp); needs the k-mer words and exact seed tags
        const char* e = getenv("LZANI_KERNEL");
        c->sgeo = index_geometry(c->Tmax, c->P.msl);
        c->sdir_stride = ((u64)1 << c->sgeo.dirbits) + 1;
        c->use_tpp = e && !strcmp(e, "tpp") && c->d_kmL &&
                     c->sgeo.tagmask == (u32)lowmask(c->sgeo.kb - c->sgeo.dirbits);
    }
    {   // bucket table: only where it stays L2-sized (<= 2^18 buckets) and the sentinels cannot be real entries
        int tagbits = 0;
        while (tagbits < 32 && ((c->geo.tagmask >> tagbits) & 1u)) ++tagbits;
        const bool exact = c->geo.tagmask == (u32)lowmask(c->geo.kb - c->geo.dirbits);
        const char* e = getenv("LZANI_NO_BUCKETS");
        c->bk_stride = (c->d_kmL && exact && c->geo.dirbits <= 18 && tagbits + c->geo.posbits <= 30 && !(e && *e == '1'))
                           ? ((u64)4 << c->geo.dirbits) : 0;
        const char* t = getenv("LZANI_NO_TAGWORDS");
        c->tw_stride = (c->bk_stride && tagbits <= 7 && !(t && *t == '1')) ? ((u64)1 << c->geo.dirbits) : 0;
    }
    size_t per_slot = (size_t)4 * (c->dir_stride + c->ent_stride + c->bk_stride + c->tw_stride + (c->use_tpp ? c->sdir_stride + c->ent_stride : 0));
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    size_t have = c->slots * per_slot;
    size_t budget = (size_t)((free_b + have) * 0.6);
    u32 slots = (u32)std::min<size_t>(std::min<u32>(want_rows, 65535u), std::max<size_t>(1, budget / per_slot));  // gridDim.y limit
    if (slots <= c->slots) return LZANI_OK;
    free_slabs(c);
    HIPCHK(c, hipMalloc(&c->d_dirz, (size_t)slots * c->dir_stride * 4));
    HIPCHK(c, hipMalloc(&c->d_ent, (size_t)slots * c->ent_stride * 4));
    if (c->bk_stride) HIPCHK(c, hipMalloc(&c->d_bk, (size_t)slots * c->bk_stride * 4));
    if (c->tw_stride) HIPCHK(c, hipMalloc(&c->d_tw, (size_t)slots * c->tw_stride * 4));
    if (c->use_tpp) {
        HIPCHK(c, hipMalloc(&c->d_sdirz, (size_t)slots * c->sdir_stride * 4));
        HIPCHK(c, hipMalloc(&c->d_sent, (size_t)slots * c->ent_stride * 4));
    }
    c->slots = slots;
    return LZANI_OK;
}

GenomeTab gtab(const lzani_ctx* c) { return GenomeTab{c->d_t2, c->d_nm, c->d_nmoff, c->d_L, c->d_kmL, c->d_kmS, c->d_hasN}; }

// Index build of `rows` references (device list d_ref_ids) into slots 0..rows-1.
int build_indexes(lzani_ctx* c, const u32* d_ref_ids, u32 rows)
{
    IdxArgs ia;
    ia.G = gtab(c);
    ia.ref_ids = d_ref_ids;
    ia.dirz = c->d_dirz; ia.ent = c->d_ent;
    ia.dir_stride = c->dir_stride; ia.ent_stride = c->ent_stride;
    ia.mal = c->P.mal; ia.mrd = c->P.mrd; ia.geo = c->geo; ia.seed = 0;
    const u32 nb = 1u << c->geo.dirbits;
    if (c->d_kmL && !c->kmers_ready) {            // per-genome k-mer words, inside the timed index stage
        for (u32 g0 = 0; g0 < c->n; g0 += 32768) {
            u32 cnt = std::min<u32>(32768, c->n - g0);
            GenomeTab G = gtab(c);
            G.nmoff += g0; G.L += g0;
            hipLaunchKernelGGL(k_kmers, dim3((c->Tmax + 255) / 256, cnt), dim3(256), 0, c->stream,
                               G, c->d_kmL, c->d_kmS, c->P.mal, c->P.msl, c->P.mrd, c->Tmax);
        }
        c->kmers_ready = true;
        c->tm.index_launches += 1;
    }
    HIPCHK(c, hipMemsetAsync(c->d_dirz, 0, (size_t)rows * c->dir_stride * 4, c->stream));
    dim3 gp((c->Tmax + 255) / 256, rows);
    hipLaunchKernelGGL(k_idx_count, gp, dim3(256), 0, c->stream, ia, c->Tmax);
    hipLaunchKernelGGL(k_idx_scan, dim3(rows), dim3(1024), 0, c->stream, c->d_dirz, c->dir_stride, nb);
    hipLaunchKernelGGL(k_idx_fill, gp, dim3(256), 0, c->stream, ia, c->Tmax);
    hipLaunchKernelGGL(k_idx_sort, dim3((nb + 255) / 256, rows), dim3(256), 0, c->stream,
                       c->d_dirz, c->d_ent, c->dir_stride, c->ent_stride, nb);
    if (c->d_bk)
        hipLaunchKernelGGL(k_idx_buckets, dim3((nb + 255) / 256, rows), dim3(256), 0, c->stream,
                           c->d_dirz, c->d_ent, c->d_bk, c->d_tw, c->dir_stride, c->ent_stride, c->bk_stride, c->tw_stride,
                           nb, c->geo.posbits);
    if (c->use_tpp) {                             // second index over the msl-mers
        IdxArgs sa = ia;
        sa.dirz = c->d_sdirz; sa.ent = c->d_sent; sa.dir_stride = c->sdir_stride;
        sa.mal = c->P.msl; sa.geo = c->sgeo; sa.seed = 1;
        const u32 snb = 1u << c->sgeo.dirbits;
        HIPCHK(c, hipMemsetAsync(c->d_sdirz, 0, (size_t)rows * c->sdir_stride * 4, c->stream));
        hipLaunchKernelGGL(k_idx_count, gp, dim3(256), 0, c->stream, sa, c->Tmax);
        hipLaunchKernelGGL(k_idx_scan, dim3(rows), dim3(1024), 0, c->stream, c->d_sdirz, c->sdir_stride, snb);
        hipLaunchKernelGGL(k_idx_fill, gp, dim3(256), 0, c->stream, sa, c->Tmax);
        hipLaunchKernelGGL(k_idx_sort, dim3((snb + 255) / 256, rows), dim3(256), 0, c->stream,
                           c->d_sdirz, c->d_sent, c->sdir_stride, c->ent_stride, snb);
        c->tm.index_launches += 4;
    }
    HIPCHK(c, hipGetLastError());
    c->tm.index_launches += 4;
    return LZANI_OK;
}

struct RegionSink { lzani_region* d_regions; unsigned long long* d_count; unsigned long long capacity; };

int run_rows_impl(lzani_ctx* c, u32 n_rows, const u32* ref_ids, const u64* row_off, const u32* query_ids,
                  int* d_out, const RegionSink* rs = nullptr)
{
    if (!c->n) return fail(c, LZANI_ERR_STATE, "lzani_run_rows: no genomes set");
    c->tm = lzani_timing{};
    c->kmers_ready = false;                       // recomputed inside every run: it is part of the path's work
    if (n_rows == 0) return LZANI_OK;
    const u64 n_pairs = row_off[n_rows];
    for (u32 k = 0; k < n_rows; ++k) {
        if (ref_ids[k] >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: reference id out of range");
        if (row_off[k + 1] < row_off[k]) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: row_off not monotone");
        if (!query_ids && row_off[k + 1] - row_off[k] != (u64)c->n - 1)
            return fail(c, LZANI_ERR_ARG, "lzani_run_rows: dense row must have n-1 queries");
    }
    if (row_off[0] != 0) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: row_off[0] must be 0");
    if (query_ids)
        for (u64 e = 0; e < n_pairs; ++e)
            if (query_ids[e] >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: query id out of range");
    if (n_pairs == 0) return LZANI_OK;

    HIPCHK(c, hipSetDevice(c->dev));
    int rc = ensure_slabs(c, n_rows);
    if (rc) return rc;

    DevBuf<u32> d_ref, d_q, d_qorder;
    DevBuf<u64> d_off, d_qcum;
    HIPCHK(c, d_qorder.alloc(n_rows));
    HIPCHK(c, d_qcum.alloc((size_t)n_rows + 1));
    HIPCHK(c, d_ref.alloc(n_rows));
    HIPCHK(c, d_off.alloc((size_t)n_rows + 1));
    HIPCHK(c, hipMemcpyAsync(d_ref, ref_ids, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_off, row_off, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (query_ids) {
        HIPCHK(c, d_q.alloc(n_pairs));
        HIPCHK(c, hipMemcpyAsync(d_q, query_ids, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    }

    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->dev));
    u32 blocks_per_cu = 8;                                   // 8 blocks x 4 waves = 8 waves per SIMD
    if (const char* e = getenv("LZANI_BLOCKS_PER_CU")) blocks_per_cu = (u32)std::max(1, std::min(8, atoi(e)));   // occupancy experiments
    const u32 max_blocks = (u32)prop.multiProcessorCount * blocks_per_cu;

    for (u32 k0 = 0; k0 < n_rows; k0 += c->slots) {
        u32 rows = std::min(c->slots, n_rows - k0);
        u64 e0 = row_off[k0], e1 = row_off[k0 + rows];
        TRACE("batch rows [%u,%u) pairs [%llu,%llu) slots=%u", k0, k0 + rows, (unsigned long long)e0, (unsigned long long)e1, c->slots);
        HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        rc = build_indexes(c, d_ref + k0, rows);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        if (e1 > e0) {
            PairArgs pa;
            pa.G = gtab(c);
            pa.P = c->P; pa.geo = c->geo;
            pa.dirz = c->d_dirz; pa.ent = c->d_ent;
            pa.dir_stride = c->dir_stride; pa.ent_stride = c->ent_stride;
            pa.bk = c->d_bk; pa.bk_stride = c->bk_stride;
            pa.tw = c->d_tw; pa.tw_stride = c->tw_stride;
            pa.ref_ids = d_ref + k0; pa.row_off = d_off + k0; pa.query_ids = d_q;
            pa.out = d_out; pa.cursor = c->d_cursor;
            // rows -> queues: longest row first onto the least loaded queue (equal rows: round robin)
            std::vector<u32> by_size(rows);
            for (u32 k = 0; k < rows; ++k) by_size[k] = k;
            std::stable_sort(by_size.begin(), by_size.end(), [&](u32 x, u32 y) {
                return row_off[k0 + x + 1] - row_off[k0 + x] > row_off[k0 + y + 1] - row_off[k0 + y]; });
            std::vector<std::vector<u32>> queue(NQUEUES);
            u64 load[NQUEUES] = {0};
            for (u32 k : by_size) {
                u32 best = 0;
                for (u32 x = 1; x < NQUEUES; ++x) if (load[x] < load[best]) best = x;
                queue[best].push_back(k);
                load[best] += row_off[k0 + k + 1] - row_off[k0 + k];
            }
            std::vector<u32> qorder; std::vector<u64> qcum(1, 0);
            for (u32 x = 0; x < NQUEUES; ++x) {
                pa.qb[x] = (u32)qorder.size();
                for (u32 k : queue[x]) { qorder.push_back(k); qcum.push_back(qcum.back() + (row_off[k0 + k + 1] - row_off[k0 + k])); }
            }
            pa.qb[NQUEUES] = (u32)qorder.size();
            HIPCHK(c, hipMemcpyAsync(d_qorder, qorder.data(), qorder.size() * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_qcum, qcum.data(), qcum.size() * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));       // the host vectors die at the end of this scope
            pa.qorder = d_qorder; pa.qcum = d_qcum;
            HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, NQUEUES * sizeof(unsigned long long), c->stream));
            u64 waves = e1 - e0;
            u32 blocks = (u32)std::min<u64>((waves + 3) / 4, max_blocks);
            HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
            const Params& q = c->P;
            const bool defp = q.mal == 11 && q.msl == 7 && q.mrd == 40 && q.mqd == 40 && q.reg == 35 && q.aw == 15 && q.am == 7 && q.ar == 3;
            const dim3 gd(blocks), bd(256);
            pa.reg_out = rs ? rs->d_regions : nullptr; pa.reg_count = rs ? rs->d_count : nullptr; pa.reg_cap = rs ? rs->capacity : 0;
            if (c->use_tpp && !rs) {
                TppArgs ta;
                ta.pa = pa; ta.sdirz = c->d_sdirz; ta.sent = c->d_sent;
                ta.sdir_stride = c->sdir_stride; ta.sent_stride = c->ent_stride; ta.sgeo = c->sgeo;
                const u64 lanes = e1 - e0;
                const dim3 tg((u32)std::min<u64>((lanes + 255) / 256, max_blocks));
                if (c->all_nfree && defp) hipLaunchKernelGGL((k_pairs_tpp<true, true>), tg, bd, 0, c->stream, ta);
                else if (c->all_nfree) hipLaunchKernelGGL((k_pairs_tpp<true, false>), tg, bd, 0, c->stream, ta);
                else if (defp) hipLaunchKernelGGL((k_pairs_tpp<false, true>), tg, bd, 0, c->stream, ta);
                else hipLaunchKernelGGL((k_pairs_tpp<false, false>), tg, bd, 0, c->stream, ta);
            } else {
#define LZ_PAIRS(F, N, D, A, B) hipLaunchKernelGGL((k_pairs<F, N, D, A, B>), gd, bd, 0, c->stream, pa)
                const bool fast = c->d_kmL != nullptr, tw = pa.tw != nullptr, nf = c->all_nfree;
                if (rs) {                                   // alignment output: one generic instantiation per index form
                    if (!fast) LZ_PAIRS(false, false, false, true, false);
                    else if (tw) LZ_PAIRS(true, false, false, true, true);
                    else LZ_PAIRS(true, false, false, true, false);
                } else if (!fast) LZ_PAIRS(false, false, false, false, false);
                else if (tw) {
                    if (nf && defp) LZ_PAIRS(true, true, true, false, true);
                    else if (nf) LZ_PAIRS(true, true, false, false, true);
                    else if (defp) LZ_PAIRS(true, false, true, false, true);
                    else LZ_PAIRS(true, false, false, false, true);
                } else {
                    if (nf && defp) LZ_PAIRS(true, true, true, false, false);
                    else if (nf) LZ_PAIRS(true, true, false, false, false);
                    else if (defp) LZ_PAIRS(true, false, true, false, false);
                    else LZ_PAIRS(true, false, false, false, false);
                }
#undef LZ_PAIRS
            }
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
            c->tm.pair_launches += 1;
        }
        if (trace_on()) { HIPCHK(c, hipEventSynchronize(c->ev[1])); TRACE("index built"); }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        TRACE("pairs done");
#ifdef LZANI_STAMPS
        {
            unsigned long long acc[8];
            HIPCHK(c, hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_stamp_acc), sizeof acc));
            unsigned long long tot = 0;
            for (int k = 0; k < 8; ++k) tot += acc[k];
            fprintf(stderr, "[lzani stamps] pairs=%llu total_cycles/pair=%.0f shares:", (unsigned long long)(e1 - e0), (double)tot / (double)(e1 - e0));
            const char* nm[8] = {"setup", "anchors", "seeds", "event", "-", "ext_fwd", "tail", "post_ballot"};
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)acc[k] / (double)tot);
            fprintf(stderr, "\n");
            unsigned long long z[8] = {0};
            HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), z, sizeof z));
        }
#endif
        int trip = 0;
        HIPCHK(c, hipMemcpyFromSymbol(&trip, HIP_SYMBOL(g_guard_trip), sizeof(int)));
        if (trip) {
            int zero = 0;
            HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_guard_trip), &zero, sizeof(int)));
            return fail(c, LZANI_ERR_DEVICE, "pair kernel: loop guard " + std::to_string(trip) + " tripped (corrupt index or text)");
        }
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->tm.index_ms += ms;
        if (e1 > e0) {
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
            c->tm.pairs_ms += ms;
        }
        c->tm.pairs += e1 - e0;
    }
    return LZANI_OK;
}

}  // namespace

extern "C" {

void lzani_default_params(lzani_params* p)
{
    p->min_anchor_len = 11; p->min_seed_len = 7; p->max_dist_in_ref = 40; p->max_dist_in_query = 40;
    p->min_region_len = 35; p->approx_window = 15; p->approx_mismatches = 7; p->approx_run_len = 3;
}

int lzani_create(const lzani_params* p, int device_id, lzani_ctx** out)
{
    if (!p || !out) return LZANI_ERR_ARG;
    *out = nullptr;
    Params P{p->min_anchor_len, p->min_seed_len, p->max_dist_in_ref, p->max_dist_in_query,
             p->min_region_len, p->approx_window, p->approx_mismatches, p->approx_run_len};
    if (!params_supported(P)) return LZANI_ERR_PARAMS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return LZANI_ERR_DEVICE;
    lzani_ctx* c = new (std::nothrow) lzani_ctx();
    if (!c) return LZANI_ERR_NOMEM;
    c->P = P;
    c->dev = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreate(&c->stream) == hipSuccess &&
              hipMalloc(&c->d_cursor, NQUEUES * sizeof(unsigned long long)) == hipSuccess;
    for (int k = 0; ok && k < 4; ++k) ok = hipEventCreate(&c->ev[k]) == hipSuccess;
    if (!ok) { lzani_destroy(c); return LZANI_ERR_DEVICE; }
    *out = c;
    return LZANI_OK;
}

void lzani_destroy(lzani_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->dev);
    free_genomes(c);
    free_slabs(c);
    hipFree(c->d_cursor);
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char* lzani_last_error(const lzani_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lzani_set_genomes(lzani_ctx* c, uint32_t n, const uint8_t* const* codes, const uint32_t* len)
{
    if (!c) return LZANI_ERR_ARG;
    if (!n || !codes || !len) return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: empty input");
    HIPCHK(c, hipSetDevice(c->dev));
    free_genomes(c);
    free_slabs(c);
    c->L.resize(n);
    c->nmoff.resize(n);
    std::vector<u64> codeoff(n);
    u64 total_codes = 0, total_nm = 0;
    int Lmax = 0;
    for (u32 g = 0; g < n; ++g) {
        if (len[g] > 0x3FFFFFFFu - 3u * (u32)c->P.mrd)
            return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: sequence too long for 32-bit text positions");
        if (len[g] && !codes[g]) return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: null sequence");
        c->L[g] = (int)len[g];
        Lmax = std::max(Lmax, c->L[g]);
        codeoff[g] = total_codes; total_codes += len[g];
        c->nmoff[g] = total_nm; total_nm += text_wordsN(ref_text_len(c->L[g], c->P.mrd));
    }
    c->Tmax = ref_text_len(Lmax, c->P.mrd);
    c->geo = index_geometry(c->Tmax, c->P.mal);
    c->dir_stride = ((u64)1 << c->geo.dirbits) + 1;
    c->ent_stride = (u64)c->Tmax;

    DevBuf<uint8_t> d_codes;
    DevBuf<u64> d_codeoff;
    HIPCHK(c, d_codes.alloc(total_codes));
    HIPCHK(c, d_codeoff.alloc(n));
    HIPCHK(c, hipMalloc(&c->d_t2, total_nm * 16));
    HIPCHK(c, hipMalloc(&c->d_nm, total_nm * 8));
    HIPCHK(c, hipMalloc(&c->d_nmoff, (size_t)n * 8));
    HIPCHK(c, hipMalloc(&c->d_L, (size_t)n * 4));
    HIPCHK(c, hipMalloc(&c->d_hasN, (size_t)n * 4));
    HIPCHK(c, hipMemset(c->d_hasN, 0, (size_t)n * 4));
    c->total_nm = total_nm;
    if (c->P.mal <= 15 && c->P.msl <= 15) {
        HIPCHK(c, hipMalloc(&c->d_kmL, total_nm * 64 * 4));
        HIPCHK(c, hipMalloc(&c->d_kmS, total_nm * 64 * 4));
    }
    // stage the codes through one pinned-size host buffer per chunk of genomes
    {
        std::vector<uint8_t> stage;
        const u64 chunk = 256ull << 20;
        u32 g = 0;
        while (g < n) {
            u32 g1 = g; u64 bytes = 0;
            while (g1 < n && (bytes == 0 || bytes + len[g1] <= chunk)) { bytes += len[g1]; ++g1; }
            stage.resize(bytes);
            u64 o = 0;
            for (u32 k = g; k < g1; ++k) { if (len[k]) memcpy(stage.data() + o, codes[k], len[k]); o += len[k]; }
            if (bytes) HIPCHK(c, hipMemcpy(d_codes.p + codeoff[g], stage.data(), bytes, hipMemcpyHostToDevice));
            g = g1;
        }
    }
    HIPCHK(c, hipMemcpy(d_codeoff, codeoff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_nmoff, c->nmoff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_L, c->L.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    c->n = n;
    size_t maxblk = text_wordsN(c->Tmax);
    // gridDim.y is limited to 65535: pack in slices of genomes
    for (u32 g0 = 0; g0 < n; g0 += 32768) {
        u32 cnt = std::min<u32>(32768, n - g0);
        hipLaunchKernelGGL(k_pack, dim3((u32)((maxblk + 127) / 128), cnt), dim3(128), 0, c->stream,
                           d_codes.p, d_codeoff.p + g0, c->d_t2, c->d_nm, c->d_nmoff + g0, c->d_L + g0, c->d_hasN + g0, c->P.mrd, cnt);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        std::vector<int> hn(n);
        HIPCHK(c, hipMemcpy(hn.data(), c->d_hasN, (size_t)n * 4, hipMemcpyDeviceToHost));
        c->all_nfree = std::all_of(hn.begin(), hn.end(), [](int v) { return v == 0; });
    }
    TRACE("set_genomes: n=%u Tmax=%d dirbits=%d posbits=%d tagmask=%x", n, c->Tmax, c->geo.dirbits, c->geo.posbits, c->geo.tagmask);
    return LZANI_OK;
}

int lzani_run_rows_device(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                          const uint32_t* query_ids, void* d_out)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off || (!d_out && n_rows && row_off[n_rows]))
        return fail(c, LZANI_ERR_ARG, "lzani_run_rows_device: null argument");
    return run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out);
}

int lzani_run_rows(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                   const uint32_t* query_ids, lzani_result* out)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: null argument");
    const u64 n_pairs = n_rows ? row_off[n_rows] : 0;
    if (n_pairs && !out) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: null output");
    HIPCHK(c, hipSetDevice(c->dev));
    DevBuf<lzani_result> d_out;
    if (n_pairs) HIPCHK(c, d_out.alloc(n_pairs));
    int rc = run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out.p);
    if (rc == LZANI_OK && n_pairs) {
        hipError_t e = hipMemcpy(out, d_out.p, n_pairs * sizeof(lzani_result), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, LZANI_ERR_DEVICE, std::string("copy results: ") + hipGetErrorString(e));
    }
    return rc;
}

int lzani_run_rows_regions(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                           const uint32_t* query_ids, lzani_result* out, lzani_region* regions,
                           uint64_t capacity, uint64_t* n_regions)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off || !n_regions || (capacity && !regions))
        return fail(c, LZANI_ERR_ARG, "lzani_run_rows_regions: null argument");
    const u64 n_pairs = n_rows ? row_off[n_rows] : 0;
    if (n_pairs && !out) return fail(c, LZANI_ERR_ARG, "lzani_run_rows_regions: null output");
    *n_regions = 0;
    HIPCHK(c, hipSetDevice(c->dev));
    DevBuf<lzani_result> d_out;
    DevBuf<lzani_region> d_regions;
    DevBuf<unsigned long long> d_count;
    if (n_pairs) HIPCHK(c, d_out.alloc(n_pairs));
    HIPCHK(c, d_regions.alloc(capacity));
    HIPCHK(c, d_count.alloc(1));
    HIPCHK(c, hipMemset(d_count.p, 0, sizeof(unsigned long long)));
    RegionSink rs{d_regions.p, d_count.p, capacity};
    int rc = run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out.p, &rs);
    if (rc == LZANI_OK) {
        unsigned long long cnt = 0;
        hipError_t e = hipMemcpy(&cnt, rs.d_count, sizeof cnt, hipMemcpyDeviceToHost);
        if (e == hipSuccess && n_pairs) e = hipMemcpy(out, d_out.p, n_pairs * sizeof(lzani_result), hipMemcpyDeviceToHost);
        if (e == hipSuccess && cnt && capacity)
            e = hipMemcpy(regions, rs.d_regions, std::min<uint64_t>(cnt, capacity) * sizeof(lzani_region), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, LZANI_ERR_DEVICE, std::string("copy regions: ") + hipGetErrorString(e));
        *n_regions = cnt;
    }
    return rc;
}

int lzani_get_timing(const lzani_ctx* c, lzani_timing* t)
{
    if (!c || !t) return LZANI_ERR_ARG;
    *t = c->tm;
    return LZANI_OK;
}

int lzani_debug_get_index(lzani_ctx* c, uint32_t id, uint64_t* t2, uint64_t* nm, uint32_t* dirz,
                          uint32_t* ent, uint32_t* n_ent, uint32_t* geom)
{
    if (!c) return LZANI_ERR_ARG;
    if (!c->n || id >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_debug_get_index: bad id");
    HIPCHK(c, hipSetDevice(c->dev));
    int rc = ensure_slabs(c, 1);
    if (rc) return rc;
    DevBuf<u32> d_ref;
    HIPCHK(c, d_ref.alloc(1));
    HIPCHK(c, hipMemcpy(d_ref.p, &id, 4, hipMemcpyHostToDevice));
    rc = build_indexes(c, d_ref, 1);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int T = ref_text_len(c->L[id], c->P.mrd);
    size_t wn = text_wordsN(T);
    if (nm) HIPCHK(c, hipMemcpy(nm, c->d_nm + c->nmoff[id], wn * 8, hipMemcpyDeviceToHost));
    if (t2) HIPCHK(c, hipMemcpy(t2, c->d_t2 + 2 * c->nmoff[id], wn * 16, hipMemcpyDeviceToHost));
    std::vector<u32> d(c->dir_stride);
    HIPCHK(c, hipMemcpy(d.data(), c->d_dirz, c->dir_stride * 4, hipMemcpyDeviceToHost));
    u32 ne = d[c->dir_stride - 1];
    if (dirz) memcpy(dirz, d.data(), c->dir_stride * 4);
    if (ent && ne) HIPCHK(c, hipMemcpy(ent, c->d_ent, (size_t)ne * 4, hipMemcpyDeviceToHost));
    if (n_ent) *n_ent = ne;
    if (geom) { geom[0] = c->geo.kb; geom[1] = c->geo.dirbits; geom[2] = c->geo.posbits; geom[3] = c->geo.tagmask; }
    return LZANI_OK;
}

}  // extern "C"
