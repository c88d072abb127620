/*
 * lzani.h -- C ABI of the MI355X LZ-ANI pair engine (liblzani_hip.so).
 *
 * This is the drop-in boundary for LZ-ANI's one data-parallel hot path: the worker loop of
 * CLZMatcher::do_matching (/root/reference/src/lz_matcher.cpp:172-277) and everything it calls
 * in CParser (/root/reference/src/parser.h:237-253, parser.cpp:16-783).  A host that owns
 * FASTA ingest, filtering and TSV output (the reference's CLZMatcher, or this repo's own
 * `lz-ani` binary) hands over genomes as reservoir symbol codes and rows of directed pairs,
 * and receives one results_t per pair.  Plain pointers and sizes only; no HIP, torch or C++
 * types cross the boundary.  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions: every function returns LZANI_OK (0) or a negative error code and never calls
 * exit(); lzani_last_error() gives the message.  A context is bound to one GPU, is blocking
 * and not re-entrant (the reference's per-thread CParser has the same property, parser.h:25-56).
 * Outputs are written only on success.
 *
 * Results are bit-identical to the reference's for every parameter set with max_dist_in_query <=
 * max_dist_in_ref (all defaults and published settings).  Above that the reference reads beyond the end
 * of its reference text (undefined behaviour); this engine treats positions outside a text as never
 * matching, deterministically.
 */
#ifndef LZANI_H
#define LZANI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZANI_OK             0
#define LZANI_ERR_ARG       -1   /* NULL / out-of-range argument                           */
#define LZANI_ERR_PARAMS    -2   /* LZ parameters outside the supported envelope           */
#define LZANI_ERR_DEVICE    -3   /* HIP runtime error (message in lzani_last_error)        */
#define LZANI_ERR_STATE     -4   /* call order violated (e.g. run before set_genomes)      */
#define LZANI_ERR_NOMEM     -5   /* host or device allocation failed                       */

/* The eight integers CParser reads from CParams (/root/reference/src/params.h:34-48;
 * CLI: -a/--mal -s/--msl -r/--mrd -q/--mqd -g/--reg --aw --am --ar, lz-ani.cpp:210-249). */
typedef struct lzani_params {
    int32_t min_anchor_len;     /* mal, default 11 */
    int32_t min_seed_len;       /* msl, default 7  */
    int32_t max_dist_in_ref;    /* mrd, default 40 */
    int32_t max_dist_in_query;  /* mqd, default 40 */
    int32_t min_region_len;     /* reg, default 35 */
    int32_t approx_window;      /* aw,  default 15 */
    int32_t approx_mismatches;  /* am,  default 7  */
    int32_t approx_run_len;     /* ar,  default 3  */
} lzani_params;

/* results_t (/root/reference/src/defs.h:48-65): what CParser::calc_stats returns. */
typedef struct lzani_result {
    int32_t sym_in_matches;     /* TSV nt_match    */
    int32_t sym_in_literals;    /* TSV nt_mismatch */
    int32_t no_components;      /* TSV num_alns    */
} lzani_result;

/* Device-side timing of the last lzani_run_rows* call, measured with HIP events on the
 * context's own stream (what bench.py reports as the roofline numerator/denominator). */
typedef struct lzani_timing {
    double   index_ms;          /* sum over batches: per-reference index build kernels      */
    double   pairs_ms;          /* sum over batches: the pair kernel                        */
    uint32_t pair_launches;     /* number of pair-kernel launches (= batches)               */
    uint32_t index_launches;    /* number of index-build kernel launches                    */
    uint64_t pairs;             /* directed pairs processed                                 */
    double   cand_ms;           /* sum over batches: presence matrix + candidate bitmaps of dense rows (0 otherwise) */
    double   kmers_ms;          /* per-genome k-mer words (+ join lists): made by the first run after
                                 * lzani_set_genomes and kept; 0 in the runs that found them ready; not part of index_ms */
    uint32_t cand_launches;     /* kernel launches of the candidate stage                   */
    uint32_t reserved_;
} lzani_timing;

/* One region of CParser::calc_regions / get_parsing (/root/reference/src/parser.cpp:786-837,
 * region_t in defs.h:67-153): what --out-alignment prints (lz_matcher.cpp:102-169).  `pair` is the
 * CSR position of the directed pair the region belongs to. */
typedef struct lzani_region {
    uint64_t pair;
    int32_t  ref_start, ref_end, seq_start, seq_end, num_matches, num_mismatches;
} lzani_region;

typedef struct lzani_ctx lzani_ctx;

void lzani_default_params(lzani_params *p);

/* Replaces CParser::CParser(const CParams&) (parser.h:237-241), once per GPU instead of once
 * per thread.  device_id is the HIP device ordinal. */
int lzani_create(const lzani_params *p, int device_id, lzani_ctx **out);
void lzani_destroy(lzani_ctx *ctx);
const char *lzani_last_error(const lzani_ctx *ctx);

/* Replaces the seq_view hand-over of prepare_reference / prepare_data (parser.cpp:16-50;
 * lz_matcher.cpp:207-221): all n genomes at once, one symbol per byte in the reservoir's
 * codes (A0 C1 G2 T3, anything >= 4 is N; seq_reservoir.h:241-248).  The engine packs them to
 * 2 bit + N mask on the device and keeps them resident; caller buffers may be freed on return.
 * Ids used below are indices into this table (the reference's reordered sequence ids). */
int lzani_set_genomes(lzani_ctx *ctx, uint32_t n, const uint8_t *const *codes, const uint32_t *len);

/* Replaces the body of the do_matching worker (lz_matcher.cpp:196-255) for n_rows reference
 * rows given in CSR form: row k has reference ref_ids[k] and queries
 * query_ids[row_off[k] .. row_off[k+1]).  query_ids == NULL means the dense row "every id !=
 * ref, ascending" (row_off[k+1]-row_off[k] must then be n-1), i.e. lz_matcher.cpp:214-233;
 * a non-NULL list is the filtered case (234-250).  out is CSR-aligned (out[e] belongs to
 * query_ids[e]) in host memory: out[e] = calc_stats() of parse(query = query_ids[e], ref = ref_ids[k]). */
int lzani_run_rows(lzani_ctx *ctx, uint32_t n_rows, const uint32_t *ref_ids, const uint64_t *row_off,
                   const uint32_t *query_ids, lzani_result *out);

/* Same, but the results stay in device memory: d_out is a device pointer (this context's GPU)
 * to row_off[n_rows] lzani_result records, e.g. the shard buffer handed to an RCCL gather. */
int lzani_run_rows_device(lzani_ctx *ctx, uint32_t n_rows, const uint32_t *ref_ids, const uint64_t *row_off,
                          const uint32_t *query_ids, void *d_out);

/* lzani_run_rows plus the per-pair regions (replaces parser.get_parsing() in the worker, lz_matcher.cpp:
 * 222-223, 241-242).  Up to `capacity` regions are written to `regions` (host memory) in no particular
 * order -- sort by (pair, length desc, seq_start) for the reference's per-pair order; *n_regions receives
 * the number found, which may exceed capacity (then call again with a larger buffer). */
int lzani_run_rows_regions(lzani_ctx *ctx, uint32_t n_rows, const uint32_t *ref_ids, const uint64_t *row_off,
                           const uint32_t *query_ids, lzani_result *out, lzani_region *regions,
                           uint64_t capacity, uint64_t *n_regions);

int lzani_get_timing(const lzani_ctx *ctx, lzani_timing *t);

/* What the context laid out in HBM for the current genome set (after lzani_set_genomes), and how the last
 * lzani_run_rows* call was batched.  The reference has no counterpart (its tables are private members of
 * CParser, parser.h:25-56); tests and the bench use it to assert which index form / batch path really ran. */
typedef struct lzani_layout_info {
    int32_t  key_bits, dir_bits, pos_bits;  /* anchor index geometry: 2*mal, log2(buckets), bits of a text position */
    uint32_t tag_mask;                      /* tag bits stored in an entry                                            */
    int32_t  kmer_words;                    /* 1: per-genome k-mer words exist (mal, msl <= 15)                      */
    int32_t  bucket_table, tag_words;       /* 1: the index slabs carry a bucket table / tag words                   */
    int32_t  n_free;                        /* 1: no genome holds an N (NFREE kernel instantiation)                  */
    uint32_t slots;                         /* index slabs allocated = reference rows per batch                      */
    uint32_t batches_last_run;              /* batches of the last run                                               */
    uint64_t bytes_per_slot;                /* HBM bytes of one index slab                                           */
    uint64_t bytes_genomes;                 /* HBM bytes of packed texts + N masks + k-mer words                     */
    int32_t  join_lists;                    /* 1: candidates come from a join with per-genome sorted k-mer lists (long genomes) */
    int32_t  block_launches;                /* pair-kernel launches of the last run by blocks of 16 waves with the
                                             * reference's presence filter in LDS (probe form, rows of >= 128 pairs) */
    int32_t  bitmap_launches;               /* pair-kernel launches of the last run fed by per-pair candidate bitmaps
                                             * (dense rows: presence matrix of the batch's references)                 */
    int32_t  rtc_launches;                  /* pair-kernel launches of the last run by a kernel compiled at run time for
                                             * this context's parameters (lzani_get_rtc_info)                          */
    int32_t  lpt_launches;                  /* pair-kernel launches of the last run that handed their tickets out longest pair
                                             * first (batches of few, long pairs; placement only)                      */
    int32_t  matrix_from_index;             /* presence matrices of the last run made from the batch's anchor indexes
                                             * (long genomes: no global atomics) instead of one atomicOr per text position */
    int32_t  split_launches;                /* batches of the last run whose pairs were scanned by several waves each (few, long
                                             * pairs: checkpoints, segments, stitch -- csrc/lzani_kernels_split.h)           */
    int32_t  reserved_;
    uint64_t split_segments;                /* segments run for them in all, the ones run again included                    */
} lzani_layout_info;
int lzani_get_layout(const lzani_ctx *ctx, lzani_layout_info *info);

/* The reference reads its eight LZ parameters at run time and has one speed for all of them (lz-ani.cpp:205-260,
 * parser.h:31).  Here the pair kernel folds them into its code: ahead of time for the defaults and for
 * --mal 15 --msl 9 --reg 60, and for every other tuple (with mal, msl <= 15) by compiling the same kernel source
 * with hipRTC the first time the context runs rows -- a few seconds once, then a code object cached on disk
 * ($LZANI_RTC_CACHE, else $XDG_CACHE_HOME/lzani_rtc, else ~/.cache/lzani_rtc; LZANI_RTC=0 turns run-time compilation
 * off).  If that fails the generic kernel runs; results are the same either way. */
typedef struct lzani_rtc_info {
    int32_t folded_ahead_of_time;   /* 1: the context's tuple is one of the two compiled ahead of time (nothing to build) */
    int32_t null_chain;             /* 1: the tuple is inside what the hand-scheduled null chain is written for            */
    int32_t kernels_built;          /* kernels of this context made ready at run time so far (compiled or loaded)          */
    int32_t kernels_from_cache;     /* ... of which came from the disk cache                                               */
    int32_t kernels_failed;         /* attempts that fell back to the generic kernel                                       */
    int32_t reserved_;
    double  build_ms;               /* host time spent building / loading them                                             */
} lzani_rtc_info;
int lzani_get_rtc_info(const lzani_ctx *ctx, lzani_rtc_info *info);
/* Test hook (needs no GPU): compiles the pair kernel for a parameter tuple the way a context would (nfree: genomes
 * without N; cand: 0 probe, 1 join, 2 candidate bitmaps) for the gfx target `arch` and returns the size of the code
 * object, or a negative error code with the compiler's messages in `log`. */
int64_t lzani_debug_rtc_compile(const lzani_params *p, int nfree, int cand, const char *arch, char *log, uint64_t log_cap);

/* ---- Sharding over GPUs ---------------------------------------------------------------
 * The unit that shards is the reference's own work unit, one reference ROW (lz_matcher.cpp:196-255: a worker
 * takes a reference, builds its index once and parses every query of the row).  Rows are independent; the
 * genome set is replicated on every GPU; the only exchange is one gather of the per-pair results.
 * The reference itself self-schedules rows over threads with an atomic counter (lz_matcher.cpp:200); over
 * GPUs the rows are dealt up front by the partition below. */

/* cost(row) = sum of the row's query lengths + LZANI_ROW_COST_REF_WEIGHT * reference length: a pair costs
 * time proportional to its query length, the per-row index build that of a few pairs. */
#define LZANI_ROW_COST_REF_WEIGHT 6
int lzani_row_costs(uint32_t n_rows, const uint32_t *ref_ids, const uint64_t *row_off, const uint32_t *query_ids,
                    uint32_t n, const uint32_t *len, uint64_t *cost);

/* part_of_row[k] in [0, n_parts): row_cost == NULL deals the rows cyclically (dense all2all rows in the
 * reordered, length-descending id order cost the same); otherwise greedy longest-processing-time
 * (heaviest row first onto the least loaded shard) for the ragged rows a kmer-db filter leaves
 * (filter.cpp:301-345, lz_matcher.cpp:234-250).  Pure host function: needs no GPU. */
int lzani_partition_rows(uint32_t n_rows, const uint64_t *row_cost, uint32_t n_parts, uint32_t *part_of_row);

/* One process per GPU (torchrun / MPI launchers): an RCCL communicator bound to the context's device and
 * stream.  Rank 0 obtains an id, the caller distributes the LZANI_UNIQUE_ID_BYTES bytes by any means (file,
 * torch.distributed, MPI_Bcast), every rank calls lzani_comm_init.  Buffers are device pointers.
 *   allgather: every rank contributes n_results records (padded equal shards), every rank receives
 *              n_ranks * n_results in rank order;
 *   gatherv:   rank p contributes counts[p] records, the root receives them concatenated in rank order
 *              (grouped ncclSend / ncclRecv; d_recv is ignored elsewhere). */
#define LZANI_UNIQUE_ID_BYTES 128
int lzani_comm_unique_id(uint8_t *id);
int lzani_comm_init(lzani_ctx *ctx, uint32_t n_ranks, uint32_t rank, const uint8_t *id);
int lzani_comm_allgather(lzani_ctx *ctx, const void *d_send, void *d_recv, uint64_t n_results);
int lzani_comm_gatherv(lzani_ctx *ctx, const void *d_send, void *d_recv, const uint64_t *counts, uint32_t root);

/* One process, several GPUs (what `lz-ani --gpus n` uses): a context per listed device, each driven by its
 * own host thread; lzani_group_run_rows has the contract of lzani_run_rows -- it partitions the rows as above,
 * runs every shard on its device, gathers the shards on the first device over RCCL (ncclCommInitAll + grouped
 * ncclSend/ncclRecv), restores the caller's CSR order there and copies the results out once.
 * Listing one device twice is allowed for rehearsals on a single GPU (shards then move by device copies). */
typedef struct lzani_group lzani_group;
int lzani_group_create(const lzani_params *p, uint32_t n_devices, const int *device_ids, lzani_group **out);
void lzani_group_destroy(lzani_group *grp);
const char *lzani_group_last_error(const lzani_group *grp);   /* grp == NULL: why the calling thread's last lzani_group_create failed */
int lzani_group_set_genomes(lzani_group *grp, uint32_t n, const uint8_t *const *codes, const uint32_t *len);
int lzani_group_run_rows(lzani_group *grp, uint32_t n_rows, const uint32_t *ref_ids, const uint64_t *row_off,
                         const uint32_t *query_ids, lzani_result *out);
int lzani_group_get_timing(const lzani_group *grp, uint32_t device_index, lzani_timing *t, double *gather_ms);

/* The shard bookkeeping of lzani_group_run_rows as a pure host function (no GPU): rows keep their order inside
 * their shard, the shards follow each other in the gathered buffer (shard d from result shard_base[d] on,
 * shard_base has n_parts + 1 entries); entry j of the scatter table -- j counts the rows shard by shard,
 * row_of_entry[j] (may be NULL) names the row -- says where the row's results sit in the gathered buffer (src[j]),
 * where they belong in the caller's CSR order (dst[j] = row_off[row]) and how many there are (cnt[j]). */
int lzani_plan_gather(uint32_t n_rows, const uint64_t *row_off, const uint32_t *part_of_row, uint32_t n_parts,
                      uint64_t *shard_base, uint64_t *src, uint64_t *dst, uint64_t *cnt, uint32_t *row_of_entry);

/* Test hook: copies out the device-built packed reference text and anchor index of one genome
 * (any pointer may be NULL).  Sizes: nm = ((T+63)/64+2) u64, t2 = twice that, with
 * T = 2*len+3*mrd; dirz = 2^dirbits+1 u32; ent <= T u32.  geom = {kb, dirbits, posbits, tagmask}.
 * Entries ascend inside every bucket of up to 32 entries; larger buckets (long low-complexity runs) are in
 * fill order, which no reader of the index depends on. */
int lzani_debug_get_index(lzani_ctx *ctx, uint32_t id, uint64_t *t2, uint64_t *nm,
                          uint32_t *dirz, uint32_t *ent, uint32_t *n_ent, uint32_t *geom);

/* Test hook: the engine's own radix sort of 64-bit keys (csrc/lzani_sort.hip; it orders the k-mers of a long reference
 * into its anchor index in place of the hash-table fill of prepare_ht_long, parser.cpp:146-189) on host arrays: n_seg
 * segments of seg_len keys each, every segment sorted by itself, stably, by the key bits [begin_bit, end_bit). */
int lzani_debug_sort_segments(lzani_ctx *ctx, const uint64_t *keys, uint64_t *out, uint64_t seg_len, uint32_t n_seg,
                              int begin_bit, int end_bit);

#ifdef __cplusplus
}
#endif
#endif
