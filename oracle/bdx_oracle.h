/*
 * bdx_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of I-Mihara/BioDemuX.jl v1.6.0
 * (src/classification.jl).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped HIP path
 * never links, imports or calls it.
 *
 * Parity pinning: the reference is Julia and cannot run in the build
 * container (no julia binary), so this restatement is pinned by the
 * reference's own known-answer tests (test/unit/{alignment,trimming,hamming,
 * exact}.jl, transcribed to tests/golden/kat.json) and by its byte-exact golden
 * outputs (test/results/{demo1_R1,demo1_R2,demo2}, copied as data to
 * tests/golden/reference/).  See tests/test_oracle_*.py.
 *
 * All positions are 1-based inclusive, exactly as in the Julia source.
 */
#ifndef BDX_ORACLE_H
#define BDX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* classification.jl:7 */
#define ORC_INF_INT (INT64_MAX / 4)

#define ORC_ALG_SEMIGLOBAL 0
#define ORC_ALG_HAMMING 1
#define ORC_ALG_EXACT 2

#define ORC_OUT_SCOREONLY 0  /* classification.jl:124 */
#define ORC_OUT_TRACEBACK 1  /* classification.jl:126 */

/* DynamicRange, classification.jl:9-14 */
typedef struct {
    int64_t start_offset;
    int32_t start_from_end;
    int64_t end_offset;
    int32_t end_from_end;
} orc_range_t;

/* The per-pass slice of DemuxConfig (classification.jl:16-58, selected in :778-792) */
typedef struct {
    orc_range_t ref_search_range;
    orc_range_t barcode_start_range;
    orc_range_t barcode_end_range;
    int32_t trim_side;          /* 0 = nothing, 3, 5 */
    int32_t n_barcodes;
    const uint8_t *bc_bytes;    /* concatenated barcode code units */
    const int64_t *bc_off;      /* n_barcodes + 1 offsets into bc_bytes */
    const int64_t *bc_len_no_N; /* bc_lengths_no_N */
} orc_pass_t;

typedef struct {
    int32_t algorithm;          /* ORC_ALG_* (matching_algorithm) */
    double max_error_rate;
    double min_delta;
    int64_t match, mismatch, indel;
    int32_t has_nindel;
    int64_t nindel;
    int32_t is_dual;
    int32_t summary;            /* stats != nothing -> need_tb (classification.jl:812) */
    orc_pass_t pass[2];
} orc_config_t;

/* Result of one alignment call: Julia returns Float64 or (Float64, Int, Int). */
typedef struct {
    double score;   /* Inf when nothing found */
    int64_t raw;    /* integer numerator (semiglobal raw cost / hamming mismatches), ORC_INF_INT when none */
    int64_t start;
    int64_t end;
} orc_align_t;

/* semiglobal_alignment_core, classification.jl:238-445.
 * scoring: has_nindel==0 -> SimpleScoring, else NScoring. output_mode: ORC_OUT_*. */
orc_align_t orc_semiglobal_core(int64_t *DP, int64_t *origin,
                                const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                                double max_error, int64_t match, int64_t mismatch, int64_t indel,
                                int32_t has_nindel, int64_t nindel,
                                int32_t output_mode, int32_t trim_side,
                                int64_t range_first, int64_t range_last,
                                int64_t max_start_pos, int64_t min_end_pos,
                                int64_t normalization_length);

/* semiglobal_alignment / semiglobal_alignment_N wrappers, classification.jl:447-477
 * (allocates its own workspace of size m). */
orc_align_t orc_semiglobal_alignment(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                                     double max_error, int64_t match, int64_t mismatch, int64_t indel,
                                     int32_t has_nindel, int64_t nindel,
                                     int64_t range_first, int64_t range_last,
                                     int64_t max_start_pos, int64_t min_end_pos,
                                     int64_t non_N_m, int32_t trim_side, int32_t need_traceback);

/* exact_align, classification.jl:485-548 */
orc_align_t orc_exact_align(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                            int64_t range_first, int64_t range_last,
                            int64_t max_start_pos, int64_t min_end_pos, int32_t trim_side);

/* hamming_align, classification.jl:557-625 */
orc_align_t orc_hamming_align(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                              double max_error_rate,
                              int64_t range_first, int64_t range_last,
                              int64_t max_start_pos, int64_t min_end_pos, int32_t trim_side);

/* find_best_matching_bc, classification.jl:722-728 (dispatching to :632 / :669).
 * Returns barcode index (1-based, 0 = none); fills score/delta/start/end. */
typedef struct {
    int64_t bc;       /* min_score_bc */
    double score;     /* min_score */
    double delta;
    int64_t start;
    int64_t end;
} orc_best_t;

orc_best_t orc_find_best_matching_bc(const orc_config_t *cfg, int pass,
                                     const uint8_t *seq, int64_t n,
                                     int64_t *DP, int64_t *origin,
                                     int64_t range_first, int64_t range_last,
                                     int64_t max_start_pos, int64_t min_end_pos,
                                     int32_t trim_side, int32_t need_traceback);

/* resolve(), classification.jl:96-100, including Julia's UnitRange normalisation
 * (an empty a:b has last == a-1). */
void orc_resolve(const orc_range_t *dr, int64_t len, int64_t *first, int64_t *last);

/* Per-read verdict: determine_filename, classification.jl:871-938.
 * bc1: >0 matched (1-based), 0 unknown, -1 ambiguous; bc2 likewise (0 when not dual
 * or when the read is not matched).  keep_start/keep_end: (-1,-1) unknown/ambiguous,
 * (1,0) empty keep range, else 1-based inclusive.  pass_* arrays (length 2) receive the
 * per-pass (start,end,score) of match_barcode_pass (classification.jl:776-868); they
 * may be NULL. */
typedef struct {
    int32_t bc1, bc2;
    int32_t keep_start, keep_end;
    int32_t pass_status[2];   /* 1 match, 0 unknown, -1 ambiguous, 2 not run */
    /* return tuple of find_best_matching_bc for each pass (kept even when ambiguous);
     * (0, Inf, Inf, -1, -1) when the pass did not run or the :805 sanity check failed */
    int32_t pass_bc[2];
    int32_t pass_start[2], pass_end[2];
    double pass_score[2];
    double pass_delta[2];
} orc_verdict_t;

void orc_determine_filename(const orc_config_t *cfg, const uint8_t *seq, int64_t n,
                            int64_t *DP, int64_t *origin, orc_verdict_t *out);

/* worker_task loop (core.jl:226-279) over a packed batch, `nthreads` pthreads over
 * contiguous slices (the reference runs nthreads() identical workers, core.jl:454-466).
 * counts (may be NULL): int64[4 + B1*max(1,B2)] = total, matched, unmatched, ambiguous,
 * sample_counts[(bc1-1)*max(1,B2) + max(bc2,1)-1]  (DemuxStats scalar part,
 * classification.jl:736-744, merged like reporting.jl:1-9). */
int orc_classify_batch(const orc_config_t *cfg, const uint8_t *seq_bytes, const int64_t *seq_off,
                       int64_t n_reads, int32_t *bc1, int32_t *bc2, int32_t *keep_start,
                       int32_t *keep_end, int32_t *pass_start /* 2*n or NULL */,
                       int32_t *pass_end /* 2*n or NULL */, double *pass_score /* 2*n or NULL */,
                       int32_t *pass_bc /* 2*n or NULL */, double *pass_delta /* 2*n or NULL */,
                       int64_t *counts, int32_t nthreads);

/* Plain full-matrix unit-cost semi-global distance (min over read substrings) and the
 * differential self-test of the HIP path's "known-score class" claim — see bdx_oracle.c. */
int64_t orc_unit_distance(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n);
int64_t orc_selftest_known_class(uint64_t seed, int64_t iters, int64_t *first_bad);
int64_t orc_selftest_windowed_exact(uint64_t seed, int64_t iters, int64_t *first_bad);
int64_t orc_selftest_cone(uint64_t seed, int64_t iters, int64_t *first_bad);
int64_t orc_selftest_clean_class(uint64_t seed, int64_t iters, int64_t *first_bad);
int64_t orc_selftest_clean_short_lookback(uint64_t seed, int64_t iters, int64_t *first_bad);
int64_t orc_selftest_band_class(uint64_t seed, int64_t iters, int64_t *first_bad);
/* model of the wave kernel's known-trim class: (d*, end column | start) of one barcode over the 0-based column window [lo, hi) */
int orc_known_trim_positions(const uint8_t *q, int64_t m, const uint8_t *r, int64_t lo, int64_t hi, int32_t trim_side,
                             int64_t *d_out, int64_t *pos_out);
int64_t orc_selftest_known_start(uint64_t seed, int64_t iters, int64_t *first_bad);
/* model of the wave kernel's known-alignment class: the other position of a winner (anchored sweep) */
int64_t orc_known_other_position(const uint8_t *q, int64_t m, const uint8_t *r, int64_t wlo, int64_t whi, int32_t trim_side,
                                 int64_t d, int64_t pos, int64_t kk);
int64_t orc_selftest_known_alignment(uint64_t seed, int64_t iters, int64_t *first_bad);

#ifdef __cplusplus
}
#endif
#endif
