// lzani_tables.h -- the device tables every kernel header reads: the genome table of a context and the k-mer word
// sentinel.  Included by lzani_hip.hip (before the kernel headers) and, as text, by the run-time compiled pair kernels
// (lzani_rtc.h).
#pragma once
#include "lzani_core.h"

namespace lzani {

struct GenomeTab {
    const u64* t2;       // all packed texts, concatenated
    const u64* nm;       // all N masks, concatenated
    const u64* nmoff;    // per genome: word offset into nm (t2 offset is twice that, k-mer arrays 64x)
    const int* L;        // per genome: sequence length
    const u32* kmL;      // per text position: mix_key(mal-mer) or KM_INVALID   (fast path: mal, msl <= 15)
    const u32* kmS;      // per text position: msl-mer (msl 8, 9: with 14 hash bits above it, k_kmers) or KM_INVALID
    const int* hasN;     // per genome: 1 if the sequence holds a non-ACGT symbol
};

enum : u32 { KM_INVALID = 0xFFFFFFFFu };

}  // namespace lzani
