/*
 * lzani_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the LZ-ANI per-pair hot path
 * (CParser::prepare_reference / prepare_data / parse / calc_stats / calc_regions,
 * /root/reference/src/parser.cpp:16-837).  It is the parity checker for the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under lz-ani_amd/ links, imports or executes it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py)
 *  - against the reference's own golden files test/vir61.ani.tsv,
 *    example/output/ani.tsv and example/output/ani.aln.tsv (committed under
 *    tests/golden/ as data), and
 *  - against vectors produced by the reference's CParser itself, compiled from
 *    /root/reference/src/parser.cpp by oracle/Makefile into oracle/_ref/
 *    (generator: oracle/make_goldens.py).
 */
#ifndef LZANI_ORACLE_H
#define LZANI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The eight integers CParser reads from CParams (params.h:34-48). */
typedef struct lzo_params {
    int32_t mal; /* min_anchor_len     (11) */
    int32_t msl; /* min_seed_len       (7)  */
    int32_t mrd; /* max_dist_in_ref    (40) */
    int32_t mqd; /* max_dist_in_query  (40) */
    int32_t reg; /* min_region_len     (35) */
    int32_t aw;  /* approx_window      (15) */
    int32_t am;  /* approx_mismatches  (7)  */
    int32_t ar;  /* approx_run_len     (3)  */
} lzo_params;

/* results_t (defs.h:48-65) */
typedef struct lzo_result {
    int32_t sym_in_matches;
    int32_t sym_in_literals;
    int32_t no_components;
} lzo_result;

/* region_t (defs.h:67-153) */
typedef struct lzo_region {
    int32_t ref_start, ref_end, seq_start, seq_end, num_matches, num_mismatches;
} lzo_region;

/* factor_t (defs.h:37-46); flag: 1 = match_close, 2 = match_distant, 4 = run_literals */
typedef struct lzo_factor {
    int32_t data_pos, flag, offset, len;
} lzo_factor;

typedef struct lzo_ref lzo_ref;

void lzo_default_params(lzo_params *p);

/* prepare_reference (parser.cpp:16-34).  codes: one symbol per byte, 0..3 = ACGT, >=4 = N. */
lzo_ref *lzo_prepare_reference(const uint8_t *codes, uint32_t len, const lzo_params *p);
void lzo_free_reference(lzo_ref *r);

/* prepare_data + parse + calc_stats (parser.cpp:37-50, 482-716, 734-783).
 * If regions != NULL, also runs calc_regions (786-837): writes up to max_regions
 * entries and stores the total count in *n_regions.
 * If factors != NULL, dumps v_parsing the same way.  Returns 0 on success. */
int lzo_query(const lzo_ref *r, const uint8_t *codes, uint32_t len, lzo_result *out,
              lzo_region *regions, uint32_t max_regions, uint32_t *n_regions,
              lzo_factor *factors, uint32_t max_factors, uint32_t *n_factors);

/* Convenience: one directed pair. */
int lzo_pair(const uint8_t *ref, uint32_t ref_len, const uint8_t *qry, uint32_t qry_len,
             const lzo_params *p, lzo_result *out);

/* do_matching restated (lz_matcher.cpp:172-277), dense all2all over n sequences,
 * n_threads pthreads self-scheduling over reference ids.  out[(size_t)r*n + q]
 * receives parse(query=q, ref=r); the diagonal is zeroed. */
int lzo_all2all(uint32_t n, const uint8_t *const *codes, const uint32_t *len,
                const lzo_params *p, uint32_t n_threads, lzo_result *out);

#ifdef __cplusplus
}
#endif
#endif
