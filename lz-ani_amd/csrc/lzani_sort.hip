// lzani_sort.hip -- the engine's radix sort of 64-bit keys (round 4: hand-written; rounds 2-3 called hipCUB here).
//
// Who sorts: the sort-based index build of long genomes (one key  mixed mal-mer hash || position  per text position of a
// reference, sorted by hash: the sorted keys of a reference ARE its anchor index -- replaces the per-pair hash-table
// fill of CParser::prepare_ht_long, /root/reference/src/parser.cpp:146-189), the per-genome k-mer lists of the join form
// of candidate detection, and the ticket order of batches of few, long pairs.
//
// Form: least-significant-digit radix sort, 8 bits a pass, stable, SEGMENTED -- n_seg segments of seg_len keys each are
// sorted independently in the same launches (the references of a batch: the segment number never has to be sorted on,
// 31 key bits = four passes instead of the five a key with a slot number in front took).  Three kernels a pass, no
// spinning on other blocks' results (nothing here can hang a wave):
//   k_rs_hist     tile of 8,192 keys -> 256 digit counts (LDS atomics) -> counts[seg][digit][tile]
//   k_rs_scan     one block per (segment, digit): exclusive prefix over the tiles, the digit's total aside
//   k_rs_scatter  the tile again: every wave ranks its 1,024 consecutive keys 64 at a time (the lanes holding the same
//                 digit find each other with eight ballots; a per-wave LDS counter carries the count from step to step),
//                 the waves' counts are chained, the keys go through LDS into tile-sorted order and from there to their
//                 places: lanes that follow each other write addresses that follow each other (256-byte runs on average)
// HBM bytes per key and pass: 8 read (hist) + 8 read + 8 written (scatter).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef unsigned long long u64;
typedef uint32_t u32;

#ifndef LZANI_RS_PER_THREAD
#define LZANI_RS_PER_THREAD 16
#endif
enum { RS_BITS = 8, RS_BINS = 1 << RS_BITS, RS_THREADS = 512, RS_WAVES = RS_THREADS / 64, RS_PER_THREAD = LZANI_RS_PER_THREAD,
       RS_TILE = RS_THREADS * RS_PER_THREAD, RS_WAVE_KEYS = 64 * RS_PER_THREAD };

__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const u64* __restrict__ keys, u64 seg_len, u32 tiles, int shift, u32 dmask,
                                                        u32* __restrict__ counts)
{
    __shared__ u32 s_cnt[RS_BINS];
    const u32 tile = blockIdx.x, seg = blockIdx.y;
    for (u32 k = threadIdx.x; k < RS_BINS; k += RS_THREADS) s_cnt[k] = 0;
    __syncthreads();
    const u64* src = keys + (u64)seg * seg_len;
    const u64 i0 = (u64)tile * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; ++k) {
        const u64 i = i0 + (u64)k * RS_THREADS + threadIdx.x;
        if (i < seg_len) atomicAdd(&s_cnt[(u32)(src[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (u32 d = threadIdx.x; d < RS_BINS; d += RS_THREADS) counts[((u64)seg * RS_BINS + d) * tiles + tile] = s_cnt[d];
}

// exclusive prefix of one (segment, digit) row over the tiles, in place; the row's total into totals[seg][digit]
__global__ void __launch_bounds__(256) k_rs_scan(u32* __restrict__ counts, u32 tiles, u32* __restrict__ totals)
{
    __shared__ u32 s_wsum[4];
    __shared__ u32 s_carry;
    u32* row = counts + ((u64)blockIdx.y * RS_BINS + blockIdx.x) * tiles;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < tiles; base += 256) {
        const u32 idx = base + threadIdx.x;
        const u32 v = idx < tiles ? row[idx] : 0;
        u32 x = v;
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) s_wsum[wv] = x;
        __syncthreads();
        u32 woff = 0;
        for (int k = 0; k < wv; ++k) woff += s_wsum[k];
        const u32 carry = s_carry;
        if (idx < tiles) row[idx] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.y * RS_BINS + blockIdx.x] = s_carry;
}

__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(const u64* __restrict__ keys, u64* __restrict__ out, u64 seg_len, u32 tiles,
                                                           int shift, u32 dmask, const u32* __restrict__ counts, const u32* __restrict__ totals)
{
    extern __shared__ u64 s_keys[];                    // RS_TILE keys: the tile in sorted order (64 KB, dynamic)
    __shared__ u32 s_wcnt[RS_WAVES][RS_BINS];          // per wave: keys of each digit (during ranking: so far)
    __shared__ u32 s_start[RS_BINS];                   // where a digit's keys begin inside the sorted tile
    __shared__ u32 s_goff[RS_BINS];                    // where they go: segment-relative position of the digit's first key of this tile
    const u32 tile = blockIdx.x, seg = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (u32 k = threadIdx.x; k < RS_WAVES * RS_BINS; k += RS_THREADS) (&s_wcnt[0][0])[k] = 0;
    __syncthreads();
    const u64* src = keys + (u64)seg * seg_len;
    const u64 i0 = (u64)tile * RS_TILE + (u64)wv * RS_WAVE_KEYS;       // this wave's 1,024 consecutive keys
    u64 key[RS_PER_THREAD];
    u32 rank[RS_PER_THREAD];                           // among the wave's keys of the same digit
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; ++k) {
        const u64 i = i0 + (u64)k * 64 + lane;
        key[k] = i < seg_len ? src[i] : ~0ULL;
    }
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; ++k) {
        const bool valid = i0 + (u64)k * 64 + lane < seg_len;
        const u32 d = (u32)(key[k] >> shift) & dmask;
        u64 peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 m = __builtin_amdgcn_ballot_w64(valid && bit);
            peers &= bit ? m : ~m;
        }
        const u32 below = (u32)__popcll(peers & ((1ULL << lane) - 1ULL));
        // the wave's counter of the digit: read by every peer, advanced by the first of them (LDS operations of one wave
        // execute in order; the barrier keeps the compiler from moving the store above the loads)
        const u32 sofar = valid ? s_wcnt[wv][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) s_wcnt[wv][d] = sofar + (u32)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[k] = sofar + below;
    }
    __syncthreads();
    // chain the waves' counts: s_wcnt[w][d] -> the keys of digit d in the waves before w; the digit's total of the tile
    u32 dtot = 0;
    if (threadIdx.x < RS_BINS) {
        const u32 d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { const u32 c = s_wcnt[w][d]; s_wcnt[w][d] = dtot; dtot += c; }
    }
    // exclusive prefix over the digits (the first 256 threads = four waves), twice: the tile's own counts -> s_start, the
    // segment's totals -> where each digit's run begins in the segment
    {
        __shared__ u32 s_ws[2][4];
        u32 x = dtot, y = threadIdx.x < RS_BINS ? totals[seg * RS_BINS + threadIdx.x] : 0u;
        const u32 x0 = x, y0 = y;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 a = __shfl_up(x, d), b = __shfl_up(y, d);
            if (lane >= d) { x += a; y += b; }
        }
        if (threadIdx.x < RS_BINS && lane == 63) { s_ws[0][wv] = x; s_ws[1][wv] = y; }
        __syncthreads();
        if (threadIdx.x < RS_BINS) {
            u32 ox = 0, oy = 0;
            for (int k = 0; k < wv; ++k) { ox += s_ws[0][k]; oy += s_ws[1][k]; }
            s_start[threadIdx.x] = ox + x - x0;
            s_goff[threadIdx.x] = oy + y - y0 + counts[((u64)seg * RS_BINS + threadIdx.x) * tiles + tile];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; ++k) {
        if (i0 + (u64)k * 64 + lane < seg_len) {
            const u32 d = (u32)(key[k] >> shift) & dmask;
            s_keys[s_start[d] + s_wcnt[wv][d] + rank[k]] = key[k];
        }
    }
    __syncthreads();
    const u64 t0 = (u64)tile * RS_TILE;
    const u32 n_tile = (u32)(seg_len - t0 < (u64)RS_TILE ? seg_len - t0 : (u64)RS_TILE);
    u64* dst = out + (u64)seg * seg_len;
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; ++k) {
        const u32 j = (u32)k * RS_THREADS + threadIdx.x;
        if (j < n_tile) {
            const u64 v = s_keys[j];
            const u32 d = (u32)(v >> shift) & dmask;
            dst[(u64)s_goff[d] + (j - s_start[d])] = v;
        }
    }
}

}  // namespace

// Sorts n_seg segments of seg_len keys each (segment s = keys [s * seg_len, (s + 1) * seg_len)), every one by its own, by the
// key bits [begin_bit, end_bit), stably.  `in` is only read; the result is in `out`.  tmp == nullptr: only reports the
// temporary bytes needed.  Returns 0 or a hipError_t.
int lzani_sort_segments(const unsigned long long* in, unsigned long long* out, size_t seg_len, size_t n_seg, int begin_bit, int end_bit,
                        void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
    if (seg_len > 0xFFFFFFF0ull || n_seg > 65535 || begin_bit < 0 || end_bit > 64 || end_bit < begin_bit) return (int)hipErrorInvalidValue;
    const size_t n = seg_len * n_seg;
    const u32 tiles = (u32)((seg_len + RS_TILE - 1) / RS_TILE);
    const int bits = end_bit - begin_bit, passes = (bits + RS_BITS - 1) / RS_BITS;
    const size_t key_bytes = ((n * 8 + 255) / 256) * 256;
    const size_t cnt_bytes = (((size_t)n_seg * RS_BINS * tiles * 4 + 255) / 256) * 256, tot_bytes = (size_t)n_seg * RS_BINS * 4;
    const size_t need = (passes > 1 ? key_bytes : 0) + cnt_bytes + tot_bytes;
    if (!tmp) { *tmp_bytes = need ? need : 256; return 0; }
    if (*tmp_bytes < need) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    if (passes == 0) return (int)hipMemcpyAsync(out, in, n * 8, hipMemcpyDeviceToDevice, stream);
    static bool attr_set = false;                     // (per process; the attribute belongs to the function, on every device)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rs_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, RS_TILE * 8);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    char* t = static_cast<char*>(tmp);
    u64* alt = passes > 1 ? reinterpret_cast<u64*>(t) : nullptr;
    u32* counts = reinterpret_cast<u32*>(t + (passes > 1 ? key_bytes : 0));
    u32* totals = reinterpret_cast<u32*>(t + (passes > 1 ? key_bytes : 0) + cnt_bytes);
    const u64* src = in;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + p * RS_BITS, w = bits - p * RS_BITS < RS_BITS ? bits - p * RS_BITS : RS_BITS;
        const u32 dmask = (1u << w) - 1u;
        u64* dst = ((passes - 1 - p) & 1) ? alt : out;              // the last pass lands in `out`
        const dim3 gt(tiles, (u32)n_seg);
        hipLaunchKernelGGL(k_rs_hist, gt, dim3(RS_THREADS), 0, stream, src, (u64)seg_len, tiles, shift, dmask, counts);
        hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS, (u32)n_seg), dim3(256), 0, stream, counts, tiles, totals);
        hipLaunchKernelGGL(k_rs_scatter, gt, dim3(RS_THREADS), (size_t)RS_TILE * 8, stream, src, dst, (u64)seg_len, tiles, shift, dmask, counts, totals);
        src = dst;
    }
    return (int)hipGetLastError();
}

// One segment: n keys by their bits [begin_bit, end_bit).
int lzani_sort_keys(const unsigned long long* in, unsigned long long* out, size_t n, int begin_bit, int end_bit,
                    void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
    return lzani_sort_segments(in, out, n, 1, begin_bit, end_bit, tmp, tmp_bytes, stream);
}
