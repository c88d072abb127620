// lzani_kernels_index.h -- device kernels that prepare the genomes and the per-reference anchor index.
// Included by lzani_hip.hip only (after lzani_core.h / lzani_layout.h).
//   k_pack        reservoir codes -> packed reference text  fwd | N^2mrd | RC | N^mrd
//                 (replaces seq_view::unpack + CParser::append/append_rc, parser.h:57-96)
//   k_kmers       per-genome k-mer words: mixed mal-mer hash and msl-mer of every text position
//                 (replaces the per-pair prepare_kmers of prepare_data, parser.cpp:46-47)
//   k_idx_count / k_idx_scan / k_idx_fill / k_idx_sort / k_idx_buckets
//                 per-reference anchor index of all mal-mers (replaces prepare_kmers +
//                 prepare_ht_long, parser.cpp:53-103, 146-189): bucket directory + (tag|pos)
//                 entries, ascending inside a bucket; bucket table and tag words for viral sizes
#pragma once

namespace lzani {

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
    if (kmer_at(R, p, msl, key)) {
        b = (u32)key;
        // msl 8, 9: 14 hash bits above the msl-mer -- word and bit of the pair kernel's seed bitmap (DevWave::bm_hash,
        // LZ_NC_WORD9) without a multiply per round; 0x3FFF stays KM_INVALID's alone.  Equality of two words is still
        // equality of the msl-mers.
        if (msl == 8 || msl == 9) {
            u32 h = (b * 0x9E3779B1u) >> 18;
            h = h == 0x3FFFu ? 0x3FFEu : h;
            b |= h << (2 * msl);
        }
    }
    kmL[64 * o + p] = a;
    kmS[64 * o + p] = b;
}

// k_join_keys: the k-mer list of every genome AS A QUERY, one 64-bit key per forward position with a mal-mer:
//   key = genome << (kb + posbits) | mixed mal-mer hash << posbits | position
// (invalid positions get all ones and sort behind everything).  Sorted by genome and hash (lzani_sort_keys), genome g's
// keys are contiguous and ascend in bucket: a pair then finds its candidates by a JOIN of the query's sorted list with
// the reference's tag words -- both streamed in bucket order -- instead of one random probe per query position
// (DevWave::join).  Long genomes only: a 64 MB tag-word table answers random probes at one HBM line per probe.
__global__ void __launch_bounds__(256) k_join_keys(GenomeTab G, const u64* __restrict__ koff, unsigned long long* __restrict__ keys,
                                                   u32* __restrict__ valid_cnt, int shift_g, int posbits, int Lmax, u32 g_base)
{
    enum { PER_THREAD = 16 };                    // positions per thread: one atomic per block of 4,096 positions
    __shared__ u32 s_cnt;
    const u32 g = blockIdx.y;                    // index into the (offset) tables; the key carries g_base + g
    const int L = G.L[g];
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const u32* km = G.kmL + 64 * G.nmoff[g];
    unsigned long long* out = keys + koff[g];
    u32 mine = 0;
    for (int k = 0; k < PER_THREAD; ++k) {
        const int p = (blockIdx.x * PER_THREAD + k) * 256 + threadIdx.x;
        if (p >= L) break;
        const u32 h = km[p];
        const bool ok = h != KM_INVALID;
        out[p] = ok ? ((unsigned long long)(g_base + g) << shift_g) | ((unsigned long long)h << posbits) | (unsigned long long)p : ~0ULL;
        mine += ok;
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&valid_cnt[g], s_cnt);
    (void)Lmax;
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
    int mal, mrd;
    IndexGeom geo;
    const u32* todo;         // per slot: nonzero = this slot is (re)built by the global-atomics kernels; nullptr = all
};

__device__ __forceinline__ bool idx_slot_key(const IdxArgs& a, u32 slot, int p, u32& bucket, u32& entry)
{
    u32 g = a.ref_ids[slot];
    int T = ref_text_len(a.G.L[g], a.mrd);
    if (p + a.mal > T) return false;
    u64 o = a.G.nmoff[g];
    u64 h;
    if (a.G.kmL) {
        u32 v = a.G.kmL[64 * o + p];
        if (v == KM_INVALID) return false;
        h = (u64)v;
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

// ---- Sort-based index build (directories beyond what k_idx_build stages in LDS) ---------------------------------
// k_idx_keys -> radix sort (lzani_sort_keys) -> k_idx_base -> k_idx_from_sorted.  One 64-bit key per text position of
// every reference of the batch, slot || mixed mal-mer hash || position; sorted, the keys of a slot ARE its index:
// the entries in order, a bucket wherever the hash's top bits change.  Directory, entries, bucket table and tag words
// come out of one streaming pass; no atomics on the tables (the global-atomics build spends 1.4 ms per 5 Mbp
// reference on 20 M random atomics and as many scattered stores).
__global__ void __launch_bounds__(256) k_idx_keys(IdxArgs a, unsigned long long* __restrict__ keys, u32* __restrict__ cnt,
                                                  int Tmax, int shift_slot)
{
    enum { PER_THREAD = 16 };
    __shared__ u32 s_cnt;
    const u32 slot = blockIdx.y;
    const u32 g = a.ref_ids[slot];
    const int T = ref_text_len(a.G.L[g], a.mrd);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const u32* km = a.G.kmL + 64 * a.G.nmoff[g];
    unsigned long long* out = keys + (u64)slot * (u64)Tmax;
    u32 mine = 0;
    for (int k = 0; k < PER_THREAD; ++k) {
        const int p = (blockIdx.x * PER_THREAD + k) * 256 + threadIdx.x;
        if (p >= Tmax) break;
        const u32 h = p < T ? km[p] : KM_INVALID;
        const bool ok = h != KM_INVALID;
        // (every slot is sorted as a segment of its own: no slot number in the key; the bit above the hash -- shift_slot -- is
        // clear in a key and set in the all-ones filler of a position without a k-mer)
        out[p] = ok ? ((unsigned long long)h << a.geo.posbits) | (unsigned long long)p : ~0ULL;
        mine += ok;
    }
    (void)shift_slot;
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&cnt[slot], s_cnt);
}

// where the sorted keys of every slot begin: the slots of a sort group lie behind each other, the invalid keys of the
// group behind them all
__global__ void k_idx_base(const u32* __restrict__ cnt, u64* __restrict__ base, u32 rows, u32 group, u64 Tmax)
{
    for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += gridDim.x * blockDim.x) {
        const u32 g0 = s / group * group;
        u64 at = (u64)g0 * Tmax;
        for (u32 t = g0; t < s; ++t) at += cnt[t];
        base[s] = at;
    }
}

__global__ void __launch_bounds__(256) k_idx_from_sorted(IdxArgs a, const unsigned long long* __restrict__ sorted,
                                                         const u32* __restrict__ cnt, const u64* __restrict__ base,
                                                         u32* __restrict__ bk, u32* __restrict__ tw, u64 bk_stride, u64 tw_stride)
{
    const u32 slot = blockIdx.y;
    const u32 n = cnt[slot], nb = 1u << a.geo.dirbits;
    const int tb = a.geo.kb - a.geo.dirbits, posbits = a.geo.posbits;
    const u32 hmask = (u32)lowmask(a.geo.kb), pmask = (u32)lowmask(posbits), tagm = (u32)lowmask(tb) & a.geo.tagmask;
    u32* dirz = a.dirz + slot * a.dir_stride;
    u32* ent = a.ent + slot * a.ent_stride;
    uint4* bks = bk ? reinterpret_cast<uint4*>(bk + slot * bk_stride) : nullptr;
    u32* tws = tw ? tw + slot * tw_stride : nullptr;
    const uint4 empty4 = {BK_EMPTY, BK_EMPTY, BK_EMPTY, BK_EMPTY};
    if (n == 0) {                                        // a reference without a single k-mer: everything empty
        for (u32 b = blockIdx.x * blockDim.x + threadIdx.x; b <= nb; b += gridDim.x * blockDim.x) {
            dirz[b] = 0;
            if (b < nb) { if (bks) bks[b] = empty4; if (tws) tws[b] = 0; }
        }
        return;
    }
    const unsigned long long* S = sorted + base[slot];
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const unsigned long long e = S[k];
        const u32 h = ((u32)(e >> posbits)) & hmask, b = h >> tb;
        const u32 en = ((h & tagm) << posbits) | ((u32)e & pmask);
        ent[k] = en;
        const long long pb = k ? (long long)((((u32)(S[k - 1] >> posbits)) & hmask) >> tb) : -1;
        if ((long long)b != pb) {                        // the first entry of bucket b: it owns the buckets pb+1 .. b
            for (long long bb = pb + 1; bb <= (long long)b; ++bb) dirz[bb] = k;
            for (long long bb = pb + 1; bb < (long long)b; ++bb) { if (bks) bks[bb] = empty4; if (tws) tws[bb] = 0; }
            if (bks) {
                u32 o[4] = {en, BK_EMPTY, BK_EMPTY, BK_EMPTY};
                u32 w = 0x80u | (en >> posbits);
                u32 m = 1;
                for (; m < 5 && k + m < n; ++m) {
                    const unsigned long long e2 = S[k + m];
                    const u32 h2 = ((u32)(e2 >> posbits)) & hmask;
                    if ((h2 >> tb) != b) break;
                    if (m < 4) { o[m] = ((h2 & tagm) << posbits) | ((u32)e2 & pmask); w |= (0x80u | (o[m] >> posbits)) << (8 * m); }
                }
                if (m > 4) { o[3] = BK_OVERFLOW; w = TW_OVERFLOW; }
                bks[b] = uint4{o[0], o[1], o[2], o[3]};
                if (tws) tws[b] = w;
            }
        }
        if (k == n - 1) {                                // the last entry also closes the directory
            for (long long bb = (long long)b + 1; bb <= (long long)nb; ++bb) dirz[bb] = n;
            for (long long bb = (long long)b + 1; bb < (long long)nb; ++bb) { if (bks) bks[bb] = empty4; if (tws) tws[bb] = 0; }
        }
    }
}

// (count / fill / zero / sort / buckets walk their range with a grid-stride loop, so that the host can launch
// them with a handful of blocks per slot when they only serve the few slots k_idx_build could not take)
// Presence filter of a reference's mal-mers (probe form of candidate detection, lzani_kernels_pairs.h: refill): bit
// (h & fmask) of a per-slot bitmap of ~1.6 bits per text position -- 16 KB at viral size, resident in the vector L1 of
// the CUs working on that reference.  A query position whose bit is clear has no anchor: its tag word is not fetched.
// The random tag-word probes, one 128-byte L2 line each for 4 bytes used, are what bounds the viral pair kernel (twice
// the probes: 1.64x the time); the filter stops 54 % of them at the L1.
__global__ void k_idx_filter(IdxArgs a, u32* __restrict__ fl, u64 fl_stride, u32 fmask, int Tmax)
{
    const u32 slot = blockIdx.y;
    const u32 g = a.ref_ids[slot];
    const int T = ref_text_len(a.G.L[g], a.mrd);
    const u64 o = a.G.nmoff[g];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p + a.mal <= T && p < Tmax; p += gridDim.x * blockDim.x) {
        const u32 v = a.G.kmL[64 * o + p];
        if (v != KM_INVALID) atomicOr(&fl[slot * fl_stride + ((v & fmask) >> 5)], 1u << (v & 31));
    }
}

__global__ void k_idx_count(IdxArgs a, int Tmax)
{
    u32 slot = blockIdx.y;
    if (a.todo && !a.todo[slot]) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Tmax; p += gridDim.x * blockDim.x) {
        u32 b, e;
        if (idx_slot_key(a, slot, p, b, e)) atomicAdd(&a.dirz[slot * a.dir_stride + 1 + b], 1u);
    }
}

// Zero the directories of the slots the global-atomics kernels have to (re)build.
__global__ void k_idx_zero(u32* dirz, u64 dir_stride, u32 nb, const u32* todo)
{
    u32 slot = blockIdx.y;
    if (!todo[slot]) return;
    for (u32 b = blockIdx.x * blockDim.x + threadIdx.x; b <= nb; b += gridDim.x * blockDim.x) dirz[slot * dir_stride + b] = 0;
}

// In-place exclusive scan of the 2^dirbits bucket counts of one slot (one 1024-thread block).
__global__ void __launch_bounds__(1024) k_idx_scan(u32* dirz, u64 dir_stride, u32 nb, const u32* todo)
{
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    if (todo && !todo[blockIdx.x]) return;
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
    if (a.todo && !a.todo[slot]) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Tmax; p += gridDim.x * blockDim.x) {
        u32 b, e;
        if (idx_slot_key(a, slot, p, b, e)) {
            u32 at = atomicAdd(&a.dirz[slot * a.dir_stride + 1 + b], 1u);
            a.ent[slot * a.ent_stride + at] = e;
        }
    }
}

// k_idx_build: the whole anchor index of ONE reference by ONE 1024-thread block, through LDS (k-mer words
// required; used for directories up to 2^19 buckets - beyond that the 2 x 2^dirbits / 16,384 sweeps over the
// k-mer words cost more than the global atomics).  The global-atomics kernels above cost as much as ~45 pairs of the
// same reference - too much for kmer-db-filtered rows of a few dozen pairs - and four fifths of that is the
// scatter: 2 x T uncoalesced 4-byte accesses.  Here the buckets are taken in ranges of 16,384; per range
//   count    16-bit LDS counters of the range's buckets over a sweep of the reference's k-mer words
//   scan     exclusive prefix -> directory (coalesced); the counters become the buckets' END offsets
//   place    second sweep: entry -> LDS staging at --end[bucket] (afterwards the counters are the STARTS)
//   sort     inside every bucket of 2 .. IDX_SORT_MAX entries (ascending tag, position), one thread per bucket
//   write    staging -> ent (contiguous: the ranges follow each other in ent), bucket table, tag words
// so that every global access is coalesced and all random traffic stays in LDS.  A reference that does not
// fit (a range with more than IDX_STAGE entries or a bucket of 65,535: long low-complexity runs) sets
// status[slot] and is rebuilt by the global-atomics kernels, which skip every other slot.
#ifndef LZANI_IDX_RANGE
#define LZANI_IDX_RANGE 16384
#define LZANI_IDX_STAGE 24576
#endif
enum { IDX_RANGE = LZANI_IDX_RANGE, IDX_STAGE = LZANI_IDX_STAGE };       // 32 KB of counters + 96 KB of staging

__global__ void __launch_bounds__(1024) k_idx_build(IdxArgs a, u32* __restrict__ bk, u32* __restrict__ tw,
                                                    u64 bk_stride, u64 tw_stride, u32* __restrict__ status)
{
    extern __shared__ u32 lds[];
    u32* cnt = lds;                                  // IDX_RANGE / 2 words: two 16-bit counters each
    u32* stage = lds + IDX_RANGE / 2;                // IDX_STAGE entries
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s, ovf_s;
    const u32 slot = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const u32 g = a.ref_ids[slot];
    const int T = ref_text_len(a.G.L[g], a.mrd);
    const u32* km = a.G.kmL + 64 * a.G.nmoff[g];
    const u32 nb = 1u << a.geo.dirbits;
    const u32 rb = nb < (u32)IDX_RANGE ? nb : (u32)IDX_RANGE, rw = rb >> 1;       // buckets / counter words per range
    const int tb = a.geo.kb - a.geo.dirbits, posbits = a.geo.posbits;
    const u32 tagm = (u32)lowmask(tb) & a.geo.tagmask;
    u32* dirz = a.dirz + slot * a.dir_stride;
    u32* ent = a.ent + slot * a.ent_stride;
    u32* bks = bk ? bk + slot * bk_stride : nullptr;
    u32* tws = tw ? tw + slot * tw_stride : nullptr;
    enum { U = 8 };                                  // k-mer words in flight per thread
    if (tid == 0) ovf_s = 0;
    u32 base = 0;                                    // entries of the ranges before this one
    for (u32 r0 = 0; r0 < nb; r0 += rb) {
        for (u32 k = tid; k < rw; k += 1024) cnt[k] = 0;
        if (tid == 0) carry_s = 0;
        __syncthreads();
        // ---- count
        for (int p0 = (int)tid; p0 < T; p0 += U * 1024) {
            u32 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { const int p = p0 + k * 1024; v[k] = p < T ? km[p] : KM_INVALID; }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u32 b = (v[k] >> tb) - r0;                     // KM_INVALID lands far outside every range
                if (v[k] == KM_INVALID || b >= rb) continue;
                const u32 sh = (b & 1) * 16;
                const u32 old = atomicAdd(&cnt[b >> 1], 1u << sh);
                if (((old >> sh) & 0xFFFFu) == 0xFFFFu) ovf_s = 1;
            }
        }
        __syncthreads();
        // ---- scan: directory out, counters -> end offsets inside the staging area
        for (u32 w0 = 0; w0 < rw; w0 += 1024) {
            const u32 idx = w0 + tid;
            const u32 w = idx < rw ? cnt[idx] : 0;
            const u32 c0 = w & 0xFFFFu, c1 = w >> 16;
            const u32 v = c0 + c1;
            u32 x = v;                               // inclusive scan inside the wave
            for (int d = 1; d < 64; d <<= 1) {
                u32 y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            if (lane == 63) wsum[wv] = x;
            __syncthreads();
            u32 woff = 0;
            for (int k = 0; k < wv; ++k) woff += wsum[k];
            const u32 carry = carry_s;
            if (idx < rw) {
                const u32 s0 = carry + woff + x - v;
                dirz[r0 + 2 * idx] = base + s0;
                dirz[r0 + 2 * idx + 1] = base + s0 + c0;
                cnt[idx] = ((s0 + c0) & 0xFFFFu) | ((s0 + c0 + c1) << 16);         // ends (<= IDX_STAGE < 65536)
            }
            __syncthreads();
            if (tid == 1023) carry_s = carry + woff + x;
            __syncthreads();
        }
        const u32 total = carry_s;
        if (total > (u32)IDX_STAGE && tid == 0) ovf_s = 1;
        __syncthreads();
        if (ovf_s) { if (tid == 0) status[slot] = 1; return; }
        // ---- place
        for (int p0 = (int)tid; p0 < T; p0 += U * 1024) {
            u32 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) { const int p = p0 + k * 1024; v[k] = p < T ? km[p] : KM_INVALID; }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const u32 b = (v[k] >> tb) - r0;
                if (v[k] == KM_INVALID || b >= rb) continue;
                const u32 sh = (b & 1) * 16;
                const u32 at = ((atomicSub(&cnt[b >> 1], 1u << sh) >> sh) - 1u) & 0xFFFFu;
                stage[at] = ((v[k] & tagm) << posbits) | (u32)(p0 + k * 1024);
            }
        }
        __syncthreads();
        // ---- sort inside the buckets; bucket table and tag words.  cnt now holds the starts; the end of a
        //      bucket is the start of the next one (the range's total for the last).
        for (u32 b = tid; b < rb; b += 1024) {
            const u32 s = (cnt[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu;
            const u32 e = b + 1 < rb ? (cnt[(b + 1) >> 1] >> (((b + 1) & 1) * 16)) & 0xFFFFu : total;
            for (u32 i = s + 1; i < e && e - s <= (u32)IDX_SORT_MAX; ++i) {
                const u32 x = stage[i];
                u32 j = i;
                while (j > s && stage[j - 1] > x) { stage[j] = stage[j - 1]; --j; }
                stage[j] = x;
            }
            if (bks) {
                uint4 o;
                o.x = s < e ? stage[s] : BK_EMPTY;
                o.y = s + 1 < e ? stage[s + 1] : BK_EMPTY;
                o.z = s + 2 < e ? stage[s + 2] : BK_EMPTY;
                o.w = s + 3 < e ? stage[s + 3] : BK_EMPTY;
                u32 w = 0;
                if (s < e) w |= 0x80u | (o.x >> posbits);
                if (s + 1 < e) w |= (0x80u | (o.y >> posbits)) << 8;
                if (s + 2 < e) w |= (0x80u | (o.z >> posbits)) << 16;
                if (s + 3 < e) w |= (0x80u | (o.w >> posbits)) << 24;
                if (e - s > 4) { o.w = BK_OVERFLOW; w = TW_OVERFLOW; }
                reinterpret_cast<uint4*>(bks)[r0 + b] = o;
                if (tws) tws[r0 + b] = w;
            }
        }
        __syncthreads();
        for (u32 k = tid; k < total; k += 1024) ent[base + k] = stage[k];
        base += total;
        __syncthreads();
    }
    if (tid == 0) dirz[nb] = base;
}

// Ascending order inside every bucket (candidate order = ascending reference position per
// k-mer, the order of the reference's probe chain).  Buckets hold ~1 entry.
// Buckets of more than IDX_SORT_MAX entries (long low-complexity runs: one k-mer hundreds of times) are left in
// fill order unless `all` is set: an insertion sort by one thread is quadratic, and every reader of the anchor
// index picks its candidate by (longest, then smallest position) without relying on the order.
__global__ void k_idx_sort(u32* dirz, u32* ent, u64 dir_stride, u64 ent_stride, u32 nb, const u32* todo, int all)
{
    u32 slot = blockIdx.y;
    if (todo && !todo[slot]) return;
    const u32* d = dirz + slot * dir_stride;
    u32* v = ent + slot * ent_stride;
    for (u32 b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        u32 s = d[b], e = d[b + 1];
        if (e - s > (u32)IDX_SORT_MAX && !all) continue;
        for (u32 i = s + 1; i < e; ++i) {
            u32 x = v[i];
            u32 j = i;
            while (j > s && v[j - 1] > x) { v[j] = v[j - 1]; --j; }
            v[j] = x;
        }
    }
}

// Bucket table: the first four entries of every bucket side by side (16 B), so that a round reaches its
// anchor candidates with one load after the k-mer word instead of directory + entries.
// Tag words (tw != nullptr): the tags of those four entries in one 32-bit word, a byte 0x80|tag each, so that
// a round DETECTS its anchor candidates from a table a quarter the size (it stays in the XCD's L2 while the
// waves of the XCD move from one reference to the next) and only a candidate step reads the 16-byte bucket.
__global__ void k_idx_buckets(const u32* __restrict__ dirz, const u32* __restrict__ ent, u32* __restrict__ bk,
                              u32* __restrict__ tw, u64 dir_stride, u64 ent_stride, u64 bk_stride, u64 tw_stride,
                              u32 nb, int posbits, const u32* todo)
{
    u32 slot = blockIdx.y;
    if (todo && !todo[slot]) return;
    const u32* d = dirz + slot * dir_stride;
    const u32* v = ent + slot * ent_stride;
    for (u32 b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        const u32 s = d[b], e = d[b + 1];
        uint4 o;
        o.x = s < e ? v[s] : BK_EMPTY;
        o.y = s + 1 < e ? v[s + 1] : BK_EMPTY;
        o.z = s + 2 < e ? v[s + 2] : BK_EMPTY;
        o.w = s + 3 < e ? v[s + 3] : BK_EMPTY;
        u32 w = 0;
        if (s < e) w |= 0x80u | (o.x >> posbits);
        if (s + 1 < e) w |= (0x80u | (o.y >> posbits)) << 8;
        if (s + 2 < e) w |= (0x80u | (o.z >> posbits)) << 16;
        if (s + 3 < e) w |= (0x80u | (o.w >> posbits)) << 24;
        if (e - s > 4) { o.w = BK_OVERFLOW; w = TW_OVERFLOW; }
        reinterpret_cast<uint4*>(bk + slot * bk_stride)[b] = o;
        if (tw) tw[slot * tw_stride + b] = w;
    }
}

}  // namespace lzani
