/*
 * bdx_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Line-faithful plain-C restatement of BioDemuX.jl v1.6.0 src/classification.jl.
 * Every function cites the Julia lines it follows.  Loop structure, the fact/lact
 * cut-off bookkeeping, tie-breaks and the Float64 decision logic are kept exactly
 * as written there (Int == int64_t, Float64 == double; build WITHOUT -ffast-math).
 *
 * Parity pinning: see bdx_oracle.h.
 */
#include "bdx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define INF_INT ORC_INF_INT

static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

/* resolve(dr, len) -> max(1, s):min(len, e)            classification.jl:96-100
 * Julia's UnitRange(start, stop) stores stop = start-1 when stop < start, and the
 * callers use last(range) (classification.jl:800-801), so that is reproduced. */
void orc_resolve(const orc_range_t *dr, int64_t len, int64_t *first, int64_t *last) {
    int64_t s = dr->start_from_end ? len + dr->start_offset : dr->start_offset;
    int64_t e = dr->end_from_end ? len + dr->end_offset : dr->end_offset;
    int64_t a = imax(1, s);
    int64_t b = imin(len, e);
    if (b < a) b = a - 1;
    *first = a;
    *last = b;
}

/* ---- output policies, classification.jl:130-168 ---- */
typedef struct {
    int64_t score, start, end;
} res_t;

static inline res_t init_result(void) { /* :130-136 (ScoreOnly keeps only .score) */
    res_t r = {INF_INT, -1, -1};
    return r;
}

static inline res_t update_result_scoreonly(res_t cur, int64_t new_score) { /* :138-140 */
    cur.score = imin(cur.score, new_score);
    return cur;
}

static inline res_t update_result_traceback(int32_t trim_side, res_t cur, int64_t new_score,
                                            int64_t j, int64_t start_pos) { /* :142-153 */
    if (new_score < cur.score) {
        res_t r = {new_score, start_pos, j};
        return r;
    } else if (new_score == cur.score) {
        if (trim_side == 3 && start_pos > cur.start) {
            res_t r = {new_score, start_pos, j};
            return r;
        }
    }
    return cur;
}

static inline orc_align_t finalize_result(int32_t output_mode, res_t result,
                                          int64_t normalization) { /* :155-168 */
    orc_align_t a;
    if (output_mode == ORC_OUT_SCOREONLY) {
        a.start = -1;
        a.end = -1;
    } else {
        a.start = result.start;
        a.end = result.end;
    }
    if (result.score >= INF_INT) {
        a.score = INFINITY;
        a.raw = INF_INT;
    } else {
        a.score = (double)result.score / (double)normalization;
        a.raw = result.score;
    }
    return a;
}

/* max_indel_steps_for, classification.jl:170-176 (Julia div truncates toward zero, like C) */
static inline int64_t max_indel_steps_for(int32_t has_nindel, int64_t indel, int64_t nindel,
                                          int64_t allowed_error) {
    if (!has_nindel) return allowed_error / indel;
    return allowed_error / imin(indel, nindel);
}

/* step_scores_main, classification.jl:178-206.  DP is 1-based (DP[0] unused). */
static inline void step_scores_main(int32_t has_nindel, int64_t match, int64_t mismatch,
                                    int64_t indel, int64_t nindel, const uint8_t *q,
                                    const uint8_t *r, int64_t i, int64_t j,
                                    int64_t previous_score, const int64_t *DP, int64_t *ins,
                                    int64_t *del, int64_t *sub) {
    if (!has_nindel) {
        *ins = DP[i] + indel;
        *del = previous_score + indel;
        *sub = DP[i - 1] + (q[i] == r[j] ? match : mismatch);
    } else {
        int is_N_q = q[i] == (uint8_t)'N';
        int64_t cost = is_N_q ? nindel : indel;
        *ins = DP[i] + cost;
        *del = previous_score + cost;
        int is_match = (q[i] == r[j]) || is_N_q;
        *sub = DP[i - 1] + (is_match ? match : mismatch);
    }
}

/* step_scores (boundary rows i == fact or i == m), classification.jl:208-236 */
static inline void step_scores(int32_t has_nindel, int64_t match, int64_t mismatch,
                               int64_t indel, int64_t nindel, const uint8_t *q,
                               const uint8_t *r, int64_t i, int64_t j, int64_t previous_score,
                               const int64_t *DP, int64_t m, int64_t *ins, int64_t *del,
                               int64_t *sub) {
    if (!has_nindel) {
        *ins = (i == m ? INF_INT : DP[i] + indel);
        *del = previous_score + indel;
        *sub = (i == 1 ? 0 : DP[i - 1]) + (q[i] == r[j] ? match : mismatch);
    } else {
        int is_N_q = q[i] == (uint8_t)'N';
        int64_t cost = is_N_q ? nindel : indel;
        *ins = DP[i] + (i == m ? INF_INT : cost);
        *del = previous_score + cost;
        int is_match = (q[i] == r[j]) || is_N_q;
        *sub = (i == 1 ? 0 : DP[i - 1]) + (is_match ? match : mismatch);
    }
}

/* Origin selection shared by the three cell sites, classification.jl:310-321, :348-359,
 * :384-395: start from deletion, replace by substitution if strictly less, then by
 * insertion if strictly less. */
static inline int64_t pick_origin(int64_t ins, int64_t del, int64_t sub, int64_t ins_o,
                                  int64_t del_o, int64_t sub_o) {
    int64_t best_score = del;
    int64_t best_origin = del_o;
    if (sub < best_score) {
        best_score = sub;
        best_origin = sub_o;
    }
    if (ins < best_score) {
        best_score = ins;
        best_origin = ins_o;
    }
    return best_origin;
}

static inline int64_t min3(int64_t a, int64_t b, int64_t c) { return imin(imin(a, b), c); }

/* semiglobal_alignment_core, classification.jl:238-445.
 * q and r are passed 0-based from the caller; shifted to 1-based here.
 * DP/origin must hold at least m+1 entries (index 0 unused). */
static orc_align_t semiglobal_core_cols(int64_t *DP, int64_t *origin, const uint8_t *q0, int64_t m,
                                        const uint8_t *r0, int64_t n, double max_error, int64_t match,
                                        int64_t mismatch, int64_t indel, int32_t has_nindel,
                                        int64_t nindel, int32_t output_mode, int32_t trim_side,
                                        int64_t range_first, int64_t range_last, int64_t max_start_pos,
                                        int64_t min_end_pos, int64_t normalization_length,
                                        int64_t col_lo, int64_t col_hi);

orc_align_t orc_semiglobal_core(int64_t *DP, int64_t *origin, const uint8_t *q0, int64_t m,
                                const uint8_t *r0, int64_t n, double max_error, int64_t match,
                                int64_t mismatch, int64_t indel, int32_t has_nindel,
                                int64_t nindel, int32_t output_mode, int32_t trim_side,
                                int64_t range_first, int64_t range_last, int64_t max_start_pos,
                                int64_t min_end_pos, int64_t normalization_length) {
    return semiglobal_core_cols(DP, origin, q0, m, r0, n, max_error, match, mismatch, indel, has_nindel, nindel,
                                output_mode, trim_side, range_first, range_last, max_start_pos, min_end_pos,
                                normalization_length, INT64_MIN, INT64_MAX);
}

/* The reference function, plus (test-only) col_lo/col_hi: when given, the column loop :287 runs over
 * max(first, col_lo)..min(last, col_hi) instead of first..last — everything else untouched.  Used by
 * orc_selftest_windowed_exact to check the HIP path's "restricted run" claim; the plain entry point
 * above passes the whole range. */
static orc_align_t semiglobal_core_cols(int64_t *DP, int64_t *origin, const uint8_t *q0, int64_t m,
                                        const uint8_t *r0, int64_t n, double max_error, int64_t match,
                                        int64_t mismatch, int64_t indel, int32_t has_nindel,
                                        int64_t nindel, int32_t output_mode, int32_t trim_side,
                                        int64_t range_first, int64_t range_last, int64_t max_start_pos,
                                        int64_t min_end_pos, int64_t normalization_length,
                                        int64_t col_lo, int64_t col_hi) {
    const uint8_t *q = q0 - 1;
    const uint8_t *r = r0 - 1;
    const int is_traceback = output_mode == ORC_OUT_TRACEBACK; /* :275 */

    if (m == 0 || n == 0) { /* :250-252 */
        return finalize_result(output_mode, init_result(), normalization_length);
    }

    int64_t allowed_error = (int64_t)floor(max_error * (double)normalization_length); /* :254 */
    res_t result = init_result();                                                    /* :255 */

    int64_t max_indel_steps = max_indel_steps_for(has_nindel, indel, nindel, allowed_error); /* :257 */

    int64_t min_valid_start = min_end_pos - (m + max_indel_steps) + 1; /* :259 */

    if (min_valid_start > max_start_pos) { /* :261-263 */
        return finalize_result(output_mode, result, normalization_length);
    }

    /* Shrink ref_search_range start, :266-268 (Julia re-normalises an empty range) */
    if (min_valid_start > range_first) {
        range_first = imax(range_first, min_valid_start);
        if (range_last < range_first) range_last = range_first - 1;
    }

    int64_t band_offset = imax(m - n - max_indel_steps, -max_start_pos - max_indel_steps); /* :270 */

    for (int64_t i = 1; i <= m; i++) { /* :278-283 */
        DP[i] = indel * i;
        if (is_traceback) origin[i] = 1 - i;
    }

    int64_t lact = imin(allowed_error + 1, m); /* :286 */
    if (col_lo > range_first) range_first = col_lo; /* test-only restriction, see above */
    if (col_hi < range_last) range_last = col_hi;
    for (int64_t j = range_first; j <= range_last; j++) { /* :287 */
        int64_t previous_score_origin = j;
        int64_t fact, previous_score;
        int64_t current_origin = 0;
        if (j + band_offset >= 1) { /* :289-295 */
            fact = j + band_offset;
            previous_score = allowed_error;
        } else {
            fact = 1;
            previous_score = 0;
        }

        if (fact > lact) { /* :297-299 */
            return finalize_result(output_mode, result, normalization_length);
        }

        if (fact <= lact) { /* :301 */
            int64_t ins, del, sub;
            /* 1. first iteration (i = fact), :303 */
            step_scores(has_nindel, match, mismatch, indel, nindel, q, r, fact, j, previous_score,
                        DP, m, &ins, &del, &sub);
            if (is_traceback) { /* :306-324 */
                int64_t del_origin = previous_score_origin;
                int64_t sub_origin = (fact == 1 ? j : origin[fact - 1]);
                int64_t ins_origin = origin[fact];
                current_origin = pick_origin(ins, del, sub, ins_origin, del_origin, sub_origin);
            }
            if (fact != 1) { /* :326-331 */
                DP[fact - 1] = previous_score;
                if (is_traceback) origin[fact - 1] = previous_score_origin;
            }
            previous_score = min3(ins, del, sub); /* :332 */
            if (is_traceback) previous_score_origin = current_origin;

            /* 2. main loop, :338-373 */
            int64_t limit = (lact == m) ? m - 1 : lact;
            for (int64_t i = fact + 1; i <= limit; i++) {
                step_scores_main(has_nindel, match, mismatch, indel, nindel, q, r, i, j,
                                 previous_score, DP, &ins, &del, &sub);
                if (is_traceback) {
                    int64_t del_origin = previous_score_origin;
                    int64_t sub_origin = origin[i - 1];
                    int64_t ins_origin = origin[i];
                    current_origin = pick_origin(ins, del, sub, ins_origin, del_origin, sub_origin);
                }
                DP[i - 1] = previous_score;
                if (is_traceback) origin[i - 1] = previous_score_origin;
                previous_score = min3(ins, del, sub);
                if (is_traceback) previous_score_origin = current_origin;
            }

            /* 3. last iteration (i = m) if needed, :376-409 */
            if (lact == m && lact > fact) {
                step_scores(has_nindel, match, mismatch, indel, nindel, q, r, m, j, previous_score,
                            DP, m, &ins, &del, &sub);
                if (is_traceback) {
                    int64_t del_origin = previous_score_origin;
                    int64_t ins_origin = origin[m];
                    int64_t sub_origin = origin[m - 1];
                    current_origin = pick_origin(ins, del, sub, ins_origin, del_origin, sub_origin);
                }
                DP[m - 1] = previous_score;
                if (is_traceback) origin[m - 1] = previous_score_origin;
                previous_score = min3(ins, del, sub);
                if (is_traceback) previous_score_origin = current_origin;
            }
        }

        DP[lact] = previous_score; /* :412-415 */
        if (is_traceback) origin[lact] = previous_score_origin;

        if (lact == m && previous_score <= allowed_error) { /* :417 */
            lact -= 1;
            if (j >= min_end_pos) {
                if (previous_score == 0) { /* :420-430 */
                    int do_early_exit = !is_traceback || (is_traceback && trim_side == 5);
                    if (do_early_exit) {
                        res_t z = {0, -1, -1};
                        if (is_traceback) {
                            z.start = previous_score_origin;
                            z.end = j;
                        }
                        return finalize_result(output_mode, z, normalization_length);
                    }
                }
                if (is_traceback) { /* :432-436 */
                    result = update_result_traceback(trim_side, result, previous_score, j,
                                                     previous_score_origin);
                } else {
                    result = update_result_scoreonly(result, previous_score);
                }
            }
        }
        while (lact > 0 && DP[lact] > allowed_error) lact -= 1; /* :439-441 */
        lact += 1;                                              /* :442 */
    }
    return finalize_result(output_mode, result, normalization_length); /* :444 */
}

/* semiglobal_alignment (:447-461) and semiglobal_alignment_N (:463-477) */
orc_align_t orc_semiglobal_alignment(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                                     double max_error, int64_t match, int64_t mismatch,
                                     int64_t indel, int32_t has_nindel, int64_t nindel,
                                     int64_t range_first, int64_t range_last,
                                     int64_t max_start_pos, int64_t min_end_pos, int64_t non_N_m,
                                     int32_t trim_side, int32_t need_traceback) {
    int64_t *DP = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 2));
    int64_t *origin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 2));
    int32_t output_mode = (trim_side == 0 && !need_traceback) ? ORC_OUT_SCOREONLY : ORC_OUT_TRACEBACK;
    int64_t norm = has_nindel ? non_N_m : m; /* :460 vs :476 */
    orc_align_t a = orc_semiglobal_core(DP, origin, q, m, r, n, max_error, match, mismatch, indel,
                                        has_nindel, nindel, output_mode, trim_side, range_first,
                                        range_last, max_start_pos, min_end_pos, norm);
    free(DP);
    free(origin);
    return a;
}

/* Base.findnext(query::String, ref::String, start): leftmost occurrence whose first code
 * unit index is >= start; 0 when none.  1-based.  (m >= 1) */
static int64_t find_next(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n, int64_t start) {
    if (start < 1) start = 1;
    for (int64_t s = start; s + m - 1 <= n; s++) {
        if (memcmp(r + (s - 1), q, (size_t)m) == 0) return s;
    }
    return 0;
}

/* Base.findprev(query::String, ref::String, k): rightmost occurrence that ENDS at or before
 * code unit k (this is what the reference relies on, classification.jl:498-505); 0 when none. */
static int64_t find_prev(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n, int64_t k) {
    int64_t s = imin(k - m + 1, n - m + 1);
    for (; s >= 1; s--) {
        if (memcmp(r + (s - 1), q, (size_t)m) == 0) return s;
    }
    return 0;
}

static inline orc_align_t align_none(void) {
    orc_align_t a = {INFINITY, INF_INT, -1, -1};
    return a;
}

/* exact_align, classification.jl:485-548 */
orc_align_t orc_exact_align(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n,
                            int64_t range_first, int64_t range_last, int64_t max_start_pos,
                            int64_t min_end_pos, int32_t trim_side) {
    int64_t start_range_first = imax(range_first, 1);                                /* :490 */
    int64_t start_range_last = imin(imin(range_last, max_start_pos), n - m + 1);     /* :491 */

    if (start_range_last < start_range_first) return align_none(); /* :493-495 */

    if (trim_side == 3) { /* :499-515 */
        int64_t s = find_prev(q, m, r, n, start_range_last + m - 1);
        if (s != 0) {
            if (s >= start_range_first) {
                if ((s + m - 1) >= min_end_pos) {
                    orc_align_t a = {0.0, 0, s, s + m - 1};
                    return a;
                }
            }
        }
        return align_none();
    } else { /* :517-547 */
        int64_t s = find_next(q, m, r, n, start_range_first);
        if (s != 0) {
            if (s <= start_range_last) {
                if ((s + m - 1) >= min_end_pos) {
                    orc_align_t a = {0.0, 0, s, s + m - 1};
                    return a;
                } else {
                    int64_t next_search = s + 1;
                    while (next_search <= start_range_last) {
                        s = find_next(q, m, r, n, next_search);
                        if (s == 0) break;
                        if (s > start_range_last) break;
                        if ((s + m - 1) >= min_end_pos) {
                            orc_align_t a = {0.0, 0, s, s + m - 1};
                            return a;
                        }
                        next_search = s + 1;
                    }
                }
            }
        }
        return align_none();
    }
}

/* hamming_align, classification.jl:557-625 */
orc_align_t orc_hamming_align(const uint8_t *q0, int64_t m, const uint8_t *r0, int64_t n,
                              double max_error_rate, int64_t range_first, int64_t range_last,
                              int64_t max_start_pos, int64_t min_end_pos, int32_t trim_side) {
    const uint8_t *q_bytes = q0 - 1;
    const uint8_t *r_bytes = r0 - 1;
    double best_score = INFINITY; /* :562-564 */
    int64_t best_raw = INF_INT;
    int64_t best_start = -1;
    int64_t best_end = -1;

    int64_t allowed_errors = (int64_t)floor(max_error_rate * (double)m); /* :567 */

    int64_t start_range_first = imax(range_first, 1);                            /* :570 */
    int64_t start_range_last = imin(imin(range_last, max_start_pos), n - m + 1); /* :571 */

    if (start_range_last < start_range_first) return align_none(); /* :573-576 */

    for (int64_t j = start_range_first; j <= start_range_last; j++) { /* :581 */
        int64_t end_pos = j + m - 1;
        if (end_pos < min_end_pos) continue; /* :584-586 */

        int64_t current_mismatches = 0;
        int match_failed = 0;

        for (int64_t k = 0; k <= m - 1; k++) { /* :592-604 */
            uint8_t q_char = q_bytes[k + 1];
            uint8_t r_char = r_bytes[j + k];
            if (q_char != r_char && q_char != 0x4E) {
                current_mismatches += 1;
                if (current_mismatches > allowed_errors) {
                    match_failed = 1;
                    break;
                }
            }
        }

        if (!match_failed) { /* :606-621 */
            double score = (double)current_mismatches / (double)m;
            if (score < best_score) {
                best_score = score;
                best_raw = current_mismatches;
                best_start = j;
                best_end = end_pos;
            } else if (score == best_score) {
                if (trim_side == 3) {
                    if (j > best_start) {
                        best_start = j;
                        best_end = end_pos;
                    }
                }
            }
        }
    }
    orc_align_t a = {best_score, best_raw, best_start, best_end};
    return a;
}

/* One barcode's alignment as dispatched inside both reducers, classification.jl:639-656 /
 * :677-694.  Returns (score, s, e) with s = e = -1 for the ScoreOnly semi-global case. */
static orc_align_t align_one(const orc_config_t *cfg, const orc_pass_t *p, int64_t i /*1-based*/,
                             const uint8_t *seq, int64_t n, int64_t *DP, int64_t *origin,
                             double max_error_rate, int64_t range_first, int64_t range_last,
                             int64_t max_start_pos, int64_t min_end_pos, int32_t trim_side,
                             int32_t need_traceback) {
    const uint8_t *bc = p->bc_bytes + p->bc_off[i - 1];
    int64_t m = p->bc_off[i] - p->bc_off[i - 1];
    if (cfg->algorithm == ORC_ALG_HAMMING) {
        return orc_hamming_align(bc, m, seq, n, max_error_rate, range_first, range_last,
                                 max_start_pos, min_end_pos, trim_side);
    } else if (cfg->algorithm == ORC_ALG_EXACT) {
        return orc_exact_align(bc, m, seq, n, range_first, range_last, max_start_pos, min_end_pos,
                               trim_side);
    } else {
        int32_t output_mode =
            (trim_side == 0 && !need_traceback) ? ORC_OUT_SCOREONLY : ORC_OUT_TRACEBACK; /* :454 */
        int64_t norm = cfg->has_nindel ? p->bc_len_no_N[i - 1] : m;
        orc_align_t a = orc_semiglobal_core(DP, origin, bc, m, seq, n, max_error_rate, cfg->match,
                                            cfg->mismatch, cfg->indel, cfg->has_nindel, cfg->nindel,
                                            output_mode, trim_side, range_first, range_last,
                                            max_start_pos, min_end_pos, norm);
        if (output_mode == ORC_OUT_SCOREONLY) { /* :651-653 / :688-690 */
            a.start = -1;
            a.end = -1;
        }
        return a;
    }
}

/* find_best_matching_bc_no_delta, classification.jl:632-667 */
static orc_best_t find_best_no_delta(const orc_config_t *cfg, const orc_pass_t *p,
                                     const uint8_t *seq, int64_t n, int64_t *DP, int64_t *origin,
                                     double max_error_rate, int64_t range_first,
                                     int64_t range_last, int64_t max_start_pos,
                                     int64_t min_end_pos, int32_t trim_side,
                                     int32_t need_traceback) {
    double min_score = INFINITY;
    int64_t min_score_bc = 0;
    int64_t best_start = -1;
    int64_t best_end = -1;

    for (int64_t i = 1; i <= p->n_barcodes; i++) {
        orc_align_t a = align_one(cfg, p, i, seq, n, DP, origin, max_error_rate, range_first,
                                  range_last, max_start_pos, min_end_pos, trim_side, need_traceback);
        double score = a.score;
        if (score <= max_error_rate && score < min_score) { /* :658-664 */
            min_score = score;
            min_score_bc = i;
            max_error_rate = fmin(max_error_rate, min_score);
            best_start = a.start;
            best_end = a.end;
        }
    }
    orc_best_t b = {min_score_bc, min_score, INFINITY, best_start, best_end}; /* :666 */
    return b;
}

/* find_best_matching_bc_with_delta, classification.jl:669-713 */
static orc_best_t find_best_with_delta(const orc_config_t *cfg, const orc_pass_t *p,
                                       const uint8_t *seq, int64_t n, int64_t *DP,
                                       int64_t *origin, double max_error_rate,
                                       int64_t range_first, int64_t range_last,
                                       int64_t max_start_pos, int64_t min_end_pos,
                                       int32_t trim_side, int32_t need_traceback) {
    double min_score = INFINITY;
    double sub_min_score = INFINITY;
    int64_t min_score_bc = 0;
    int64_t best_start = -1;
    int64_t best_end = -1;

    for (int64_t i = 1; i <= p->n_barcodes; i++) {
        orc_align_t a = align_one(cfg, p, i, seq, n, DP, origin, max_error_rate, range_first,
                                  range_last, max_start_pos, min_end_pos, trim_side, need_traceback);
        double score = a.score;
        if (score <= max_error_rate) { /* :696-709 */
            if (score < min_score) {
                sub_min_score = min_score;
                min_score = score;
                min_score_bc = i;
                max_error_rate = fmin(max_error_rate, sub_min_score);
                best_start = a.start;
                best_end = a.end;
            } else if (score < sub_min_score) {
                sub_min_score = score;
                max_error_rate = fmin(max_error_rate, sub_min_score);
            }
        }
    }
    double delta = sub_min_score - min_score; /* :711 */
    orc_best_t b = {min_score_bc, min_score, delta, best_start, best_end};
    return b;
}

/* find_best_matching_bc, classification.jl:722-728 */
orc_best_t orc_find_best_matching_bc(const orc_config_t *cfg, int pass, const uint8_t *seq,
                                     int64_t n, int64_t *DP, int64_t *origin, int64_t range_first,
                                     int64_t range_last, int64_t max_start_pos,
                                     int64_t min_end_pos, int32_t trim_side,
                                     int32_t need_traceback) {
    const orc_pass_t *p = &cfg->pass[pass];
    if (cfg->min_delta == 0.0) {
        return find_best_no_delta(cfg, p, seq, n, DP, origin, cfg->max_error_rate, range_first,
                                  range_last, max_start_pos, min_end_pos, trim_side, need_traceback);
    } else {
        return find_best_with_delta(cfg, p, seq, n, DP, origin, cfg->max_error_rate, range_first,
                                    range_last, max_start_pos, min_end_pos, trim_side,
                                    need_traceback);
    }
}

/* match_barcode_pass, classification.jl:776-868 (stats histograms :827-865 are out of scope).
 * status: 1 match, 0 unknown, -1 ambiguous. */
static void match_barcode_pass(const orc_config_t *cfg, int is_pass2, const uint8_t *seq,
                               int64_t n, int64_t *DP, int64_t *origin, int32_t *status,
                               int64_t *bc_idx, int64_t *s_out, int64_t *e_out,
                               double *score_out, orc_best_t *fb) {
    const orc_pass_t *p = &cfg->pass[is_pass2 ? 1 : 0];
    int32_t trim_side = p->trim_side;

    int64_t rs_f, rs_l, bs_f, bs_l, be_f, be_l; /* :795-797 */
    orc_resolve(&p->ref_search_range, n, &rs_f, &rs_l);
    orc_resolve(&p->barcode_start_range, n, &bs_f, &bs_l);
    orc_resolve(&p->barcode_end_range, n, &be_f, &be_l);

    int64_t start_j = imax(imax(rs_f, bs_f), 1); /* :799 */
    int64_t end_j = imin(imin(rs_l, be_l), n);   /* :800 */
    int64_t max_start_pos = bs_l;                /* :801 */
    int64_t min_end_pos = be_f;                  /* :802 */

    *bc_idx = 0;
    *s_out = -1;
    *e_out = -1;
    *score_out = INFINITY;
    fb->bc = 0;
    fb->score = INFINITY;
    fb->delta = INFINITY;
    fb->start = -1;
    fb->end = -1;

    if (start_j > end_j || start_j > max_start_pos || end_j < min_end_pos) { /* :805-807 */
        *status = 0;
        return;
    }

    int32_t need_tb = (trim_side != 0) || cfg->summary; /* :812 */

    orc_best_t b = orc_find_best_matching_bc(cfg, is_pass2 ? 1 : 0, seq, n, DP, origin, start_j,
                                             end_j, max_start_pos, min_end_pos, trim_side, need_tb);
    *fb = b;

    if (b.bc == 0) { /* :820-824 */
        *status = 0;
        return;
    } else if (b.delta < cfg->min_delta) {
        *status = -1;
        return;
    }
    *status = 1; /* :867 */
    *bc_idx = b.bc;
    *s_out = b.start;
    *e_out = b.end;
    *score_out = b.score;
}

/* determine_filename, classification.jl:871-938 (the filename string itself is formed by the
 * host from bc1/bc2; here the verdict is returned as indices). */
void orc_determine_filename(const orc_config_t *cfg, const uint8_t *seq, int64_t n, int64_t *DP,
                            int64_t *origin, orc_verdict_t *out) {
    int32_t status1, status2 = 2;
    int64_t bc1_idx, start1, end1;
    int64_t idx2 = 0, start2 = -1, end2 = -1;
    double score1, score2 = INFINITY;

    out->bc1 = 0;
    out->bc2 = 0;
    out->keep_start = -1;
    out->keep_end = -1;
    out->pass_status[1] = 2;
    out->pass_bc[1] = 0;
    out->pass_start[1] = -1;
    out->pass_end[1] = -1;
    out->pass_score[1] = INFINITY;
    out->pass_delta[1] = INFINITY;
    orc_best_t fb;

    match_barcode_pass(cfg, 0, seq, n, DP, origin, &status1, &bc1_idx, &start1, &end1, &score1, &fb); /* :875 */
    out->pass_status[0] = status1;
    out->pass_bc[0] = (int32_t)fb.bc;
    out->pass_start[0] = (int32_t)fb.start;
    out->pass_end[0] = (int32_t)fb.end;
    out->pass_score[0] = fb.score;
    out->pass_delta[0] = fb.delta;

    if (status1 == 0) { /* :879-883 */
        out->bc1 = 0;
        return;
    } else if (status1 == -1) {
        out->bc1 = -1;
        return;
    }

    int64_t bc2_idx = 0;
    if (cfg->is_dual) { /* :887-900 */
        match_barcode_pass(cfg, 1, seq, n, DP, origin, &status2, &idx2, &start2, &end2, &score2, &fb);
        out->pass_status[1] = status2;
        out->pass_bc[1] = (int32_t)fb.bc;
        out->pass_start[1] = (int32_t)fb.start;
        out->pass_end[1] = (int32_t)fb.end;
        out->pass_score[1] = fb.score;
        out->pass_delta[1] = fb.delta;
        if (status2 == 0) {
            out->bc1 = 0;
            return;
        } else if (status2 == -1) {
            out->bc1 = -1;
            return;
        }
        bc2_idx = idx2;
    }
    out->bc1 = (int32_t)bc1_idx;
    out->bc2 = (int32_t)bc2_idx;

    int64_t keep_start = 1; /* :907-908 */
    int64_t keep_end = n;

    int32_t ts1 = cfg->pass[0].trim_side;
    int32_t ts2 = cfg->pass[1].trim_side;
    if (ts1 != 0) { /* :910-919 */
        if (ts1 == 3) {
            keep_end = imax(1, start1) - 1;
        } else if (ts1 == 5) {
            keep_start = end1 + 1;
        }
    }
    if (cfg->is_dual && ts2 != 0) { /* :921-929 */
        if (ts2 == 3) {
            keep_end = imin(keep_end, imax(1, start2) - 1);
        } else if (ts2 == 5) {
            keep_start = imax(keep_start, end2 + 1);
        }
    }
    if (keep_start > keep_end) { /* :932-935 */
        out->keep_start = 1;
        out->keep_end = 0;
        return;
    }
    out->keep_start = (int32_t)keep_start;
    out->keep_end = (int32_t)keep_end;
}

/* ---- batch driver: worker_task, core.jl:226-279 ---- */
typedef struct {
    const orc_config_t *cfg;
    const uint8_t *seq_bytes;
    const int64_t *seq_off;
    int64_t lo, hi;
    int32_t *bc1, *bc2, *keep_start, *keep_end, *pass_start, *pass_end, *pass_bc;
    double *pass_score, *pass_delta;
    int64_t *counts; /* thread-local, merged by the caller (merge_stats, reporting.jl:1-9) */
    int64_t n_counts;
    int64_t max_m;
} job_t;

static void *worker(void *arg) {
    job_t *jb = (job_t *)arg;
    const orc_config_t *cfg = jb->cfg;
    /* SemiGlobalWorkspace(max_m, ...), core.jl:229-233 */
    int64_t *DP = (int64_t *)malloc(sizeof(int64_t) * (size_t)(jb->max_m + 2));
    int64_t *origin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(jb->max_m + 2));
    int64_t B2 = cfg->is_dual ? cfg->pass[1].n_barcodes : 0;
    int64_t stride2 = B2 > 1 ? B2 : 1;
    for (int64_t i = jb->lo; i < jb->hi; i++) { /* core.jl:243-267 */
        const uint8_t *seq = jb->seq_bytes + jb->seq_off[i];
        int64_t n = jb->seq_off[i + 1] - jb->seq_off[i];
        orc_verdict_t v;
        orc_determine_filename(cfg, seq, n, DP, origin, &v);
        jb->bc1[i] = v.bc1;
        if (jb->bc2) jb->bc2[i] = v.bc2;
        if (jb->keep_start) jb->keep_start[i] = v.keep_start;
        if (jb->keep_end) jb->keep_end[i] = v.keep_end;
        if (jb->pass_start) {
            jb->pass_start[2 * i] = v.pass_start[0];
            jb->pass_start[2 * i + 1] = v.pass_start[1];
        }
        if (jb->pass_end) {
            jb->pass_end[2 * i] = v.pass_end[0];
            jb->pass_end[2 * i + 1] = v.pass_end[1];
        }
        if (jb->pass_score) {
            jb->pass_score[2 * i] = v.pass_score[0];
            jb->pass_score[2 * i + 1] = v.pass_score[1];
        }
        if (jb->pass_bc) {
            jb->pass_bc[2 * i] = v.pass_bc[0];
            jb->pass_bc[2 * i + 1] = v.pass_bc[1];
        }
        if (jb->pass_delta) {
            jb->pass_delta[2 * i] = v.pass_delta[0];
            jb->pass_delta[2 * i + 1] = v.pass_delta[1];
        }
        if (jb->counts) { /* classification.jl:942, :950, :953, :963, :966, :976-978 */
            jb->counts[0] += 1;
            if (v.bc1 > 0) {
                jb->counts[1] += 1;
                int64_t k = 4 + (int64_t)(v.bc1 - 1) * stride2 + (v.bc2 > 0 ? v.bc2 - 1 : 0);
                jb->counts[k] += 1;
            } else if (v.bc1 == 0) {
                jb->counts[2] += 1;
            } else {
                jb->counts[3] += 1;
            }
        }
    }
    free(DP);
    free(origin);
    return NULL;
}

int orc_classify_batch(const orc_config_t *cfg, const uint8_t *seq_bytes, const int64_t *seq_off,
                       int64_t n_reads, int32_t *bc1, int32_t *bc2, int32_t *keep_start,
                       int32_t *keep_end, int32_t *pass_start, int32_t *pass_end,
                       double *pass_score, int32_t *pass_bc, double *pass_delta, int64_t *counts,
                       int32_t nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    int64_t max_m = 0; /* core.jl:229-232 */
    for (int p = 0; p < (cfg->is_dual ? 2 : 1); p++) {
        for (int64_t i = 0; i < cfg->pass[p].n_barcodes; i++) {
            int64_t m = cfg->pass[p].bc_off[i + 1] - cfg->pass[p].bc_off[i];
            if (m > max_m) max_m = m;
        }
    }
    int64_t B1 = cfg->pass[0].n_barcodes;
    int64_t B2 = cfg->is_dual ? cfg->pass[1].n_barcodes : 0;
    int64_t n_counts = 4 + B1 * (B2 > 1 ? B2 : 1);

    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    int64_t per = (n_reads + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        job_t *jb = &jobs[t];
        jb->cfg = cfg;
        jb->seq_bytes = seq_bytes;
        jb->seq_off = seq_off;
        jb->lo = imin(n_reads, per * t);
        jb->hi = imin(n_reads, per * (t + 1));
        jb->bc1 = bc1;
        jb->bc2 = bc2;
        jb->keep_start = keep_start;
        jb->keep_end = keep_end;
        jb->pass_start = pass_start;
        jb->pass_end = pass_end;
        jb->pass_score = pass_score;
        jb->pass_bc = pass_bc;
        jb->pass_delta = pass_delta;
        jb->max_m = max_m;
        jb->n_counts = n_counts;
        jb->counts = counts ? (int64_t *)calloc((size_t)n_counts, sizeof(int64_t)) : NULL;
    }
    if (nthreads == 1) {
        worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; t++) pthread_create(&tids[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(tids[t], NULL);
    }
    if (counts) { /* merge_stats, reporting.jl:1-9 (adds into the caller's vector) */
        for (int t = 0; t < nthreads; t++) {
            for (int64_t k = 0; k < n_counts; k++) counts[k] += jobs[t].counts[k];
            free(jobs[t].counts);
        }
    }
    free(jobs);
    free(tids);
    return 0;
}

/* ---- differential self-test supporting the "known-score class" of the HIP path ----
 * Claim: with SimpleScoring unit costs (match 0, mismatch 1, indel 1),
 * ScoreOnly output, any column window first:last inside the read, max_start_pos >= n and
 * min_end_pos <= 1, semiglobal_alignment_core returns  d  when d <= floor(max_error*m)  and Inf
 * otherwise, where d = min over substrings of r[first..last] of the unit-cost edit distance
 * (plain full-matrix DP below).
 * Returns the number of disagreeing cases among `iters` random (barcode, read, rate) triples. */
static uint64_t st_next(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int64_t orc_unit_distance(const uint8_t *q, int64_t m, const uint8_t *r, int64_t n) {
    int64_t *prev = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    int64_t best = m;
    for (int64_t i = 0; i <= m; i++) prev[i] = i;
    for (int64_t j = 1; j <= n; j++) {
        cur[0] = 0;
        for (int64_t i = 1; i <= m; i++) {
            int64_t a = prev[i] + 1, b = cur[i - 1] + 1, c = prev[i - 1] + (q[i - 1] != r[j - 1]);
            cur[i] = a < b ? (a < c ? a : c) : (b < c ? b : c);
        }
        if (cur[m] < best) best = cur[m];
        int64_t *t = prev;
        prev = cur;
        cur = t;
    }
    free(prev);
    free(cur);
    return best;
}

int64_t orc_selftest_known_class(uint64_t seed, int64_t iters, int64_t *first_bad /* 6 ints or NULL */) {
    static const char AL[5] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[200];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 1 + (int64_t)(st_next(&s) % 32);
        int64_t n = (int64_t)(st_next(&s) % 160);
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % 4];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % ((st_next(&s) % 50) ? 4 : 5)];
        if (n > 0 && (st_next(&s) % 4)) { /* plant a mutated copy */
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 6) r[pos++] = (uint8_t)AL[st_next(&s) % 4];      /* substitution */
                else if (u < 9) continue;                                  /* deletion */
                else if (u < 12) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = q[i]; } /* insertion */
                else r[pos++] = q[i];
            }
        }
        double rate = RATES[st_next(&s) % 8];
        int64_t max_start = n + (int64_t)(st_next(&s) % 3) * 50; /* n, n+50, n+100: all non-binding */
        int64_t min_end = 1 - (int64_t)(st_next(&s) % 2);         /* 1 or 0 */
        /* column window first:last — the whole read half of the time, else a random sub-range
         * (ref_search_range windows; the start/end ranges stay non-binding) */
        int64_t first = 1, last = n;
        if (n > 0 && (st_next(&s) % 2)) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        orc_align_t a = orc_semiglobal_core(DP, OG, q, m, r, n, rate, 0, 1, 1, 0, 0, ORC_OUT_SCOREONLY, 0,
                                            first, last, max_start, min_end, m);
        int64_t d = n > 0 ? orc_unit_distance(q, m, r + (first - 1), last - first + 1) : m;
        int64_t ae = (int64_t)floor(rate * (double)m);
        int64_t expect = (n > 0 && d <= ae) ? d : INF_INT;
        if (a.raw != expect) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = a.raw;
                first_bad[4] = expect; first_bad[5] = ae;
            }
            bad++;
        }
    }
    return bad;
}


/* ---- differential self-test supporting the "restricted exact run" of the HIP path ----
 * Claim: let kb = floor(floor(max_error*norm) / cmin), cmin = min(mismatch, indel
 * [, nindel]) >= 1, match >= 0, and let E = [e_lo, e_hi] be the first/last column j of the pass
 * window with unit-cost semi-global distance U[m][j] <= kb (barcode N = wildcard under NScoring).
 * Then running the reference's column loop only over  max(first, e_lo - 2(m+kb) - 1) .. min(last,
 * e_hi)  returns the same (score, start, end) as the full run, for every output policy, trim side,
 * range setting and (tightened) threshold <= max_error; with E empty the full run returns Inf. */
static int64_t unit_cols(const uint8_t *q, int64_t m, const uint8_t *r, int64_t first, int64_t last, int wild,
                         int64_t kb, int64_t *e_lo, int64_t *e_hi) {
    int64_t *prev = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    int64_t found = 0;
    for (int64_t i = 0; i <= m; i++) prev[i] = i;
    for (int64_t j = first; j <= last; j++) {
        cur[0] = 0;
        for (int64_t i = 1; i <= m; i++) {
            int eq = (q[i - 1] == r[j - 1]) || (wild && q[i - 1] == 'N');
            int64_t a = prev[i] + 1, b = cur[i - 1] + 1, c = prev[i - 1] + !eq;
            cur[i] = a < b ? (a < c ? a : c) : (b < c ? b : c);
        }
        if (cur[m] <= kb) {
            if (!found) *e_lo = j;
            *e_hi = j;
            found = 1;
        }
        int64_t *t = prev;
        prev = cur;
        cur = t;
    }
    free(prev);
    free(cur);
    return found;
}

/* ---- model of the HIP register DP (csrc/bdx_core.h sg_core_reg), test-only --------------------------
 * Same row-mask formulation as the kernel (all rows visited, state changes predicated on fact..lact),
 * SimpleScoring only, plus the kernel's "reachability cone" for restricted runs: in a run that ends at
 * column E = min(last, col_hi), a cell (i, j) can lie on an alignment that reaches row m by column E with
 * at most kbr = allowed_error / min(mismatch, indel) operations only if  m - i <= (E - j) + kbr;  rows
 * above  bound(j) = m - kbr - (E - j)  are skipped and the deletion input of row bound(j) is infinite.
 * orc_selftest_cone checks the model against the line-faithful core above. */
static orc_align_t kernel_model_core(const uint8_t *q0, int64_t m, const uint8_t *r0, int64_t n, double max_error,
                                     int64_t match, int64_t mismatch, int64_t indel, int32_t output_mode,
                                     int32_t trim_side, int64_t first, int64_t last, int64_t max_start,
                                     int64_t min_end, int64_t norm, int64_t col_lo, int64_t col_hi, int cone) {
    const uint8_t *q = q0 - 1, *r = r0 - 1;
    const int tb = output_mode == ORC_OUT_TRACEBACK;
    const int64_t BIG = (int64_t)1 << 40;
    res_t res = init_result();
    int64_t DP[72], OG[72];
    if (m == 0 || n == 0) return finalize_result(output_mode, res, norm);
    const int64_t ae = (int64_t)floor(max_error * (double)norm);
    const int64_t steps = ae / indel;
    const int64_t min_valid_start = min_end - (m + steps) + 1;
    if (min_valid_start > max_start) return finalize_result(output_mode, res, norm);
    if (min_valid_start > first) first = min_valid_start;
    const int64_t band = imax(m - n - steps, -max_start - steps);
    for (int64_t i = 1; i <= m; i++) {
        DP[i] = indel * i;
        OG[i] = 1 - i;
    }
    int64_t lact = imin(ae + 1, m);
    const int restricted = col_hi != INT64_MAX;
    if (col_lo > first) first = col_lo;
    if (col_hi < last) last = col_hi;
    const int64_t cmin = mismatch < indel ? mismatch : indel;
    const int cone_on = cone && restricted && cmin > 0 && match >= 0 && ae >= 0;
    const int64_t kbr = cone_on ? ae / cmin : 0;
    for (int64_t j = first; j <= last; j++) {
        int64_t prev_o = j, fact, prev;
        if (j + band >= 1) {
            fact = j + band;
            prev = ae;
        } else {
            fact = 1;
            prev = 0;
        }
        if (fact > lact) return finalize_result(output_mode, res, norm);
        const int64_t bound = cone_on ? m - kbr - (last - j) : -BIG;
        const int64_t seed = prev;
        int64_t diag = 0, diag_o = j;
        for (int64_t i = 1; i <= m; i++) {
            const int64_t old = DP[i], old_o = OG[i];
            const int inb = i >= fact && i <= lact && i >= bound;
            const int64_t ins = (i == m) ? BIG : old + indel;
            const int64_t del = (i == bound) ? BIG : prev + indel;
            const int64_t sub = diag + (q[i] == r[j] ? match : mismatch);
            int64_t cur_o = prev_o, best = del;
            if (sub < best) {
                best = sub;
                cur_o = diag_o;
            }
            if (ins < best) cur_o = old_o;
            const int64_t nv = min3(ins, del, sub);
            const int seedrow = (i == fact - 1);
            DP[i] = inb ? nv : (seedrow ? seed : old);
            OG[i] = inb ? cur_o : (seedrow ? j : old_o);
            if (inb) {
                prev = nv;
                prev_o = cur_o;
            }
            diag = old;
            diag_o = old_o;
        }
        if (lact == m && prev <= ae) {
            lact -= 1;
            if (j >= min_end) {
                if (prev == 0 && (!tb || trim_side == 5)) {
                    res_t z = {0, tb ? prev_o : -1, tb ? j : -1};
                    return finalize_result(output_mode, z, norm);
                }
                if (tb) res = update_result_traceback(trim_side, res, prev, j, prev_o);
                else res = update_result_scoreonly(res, prev);
            }
        }
        for (int64_t i = m; i >= 1; --i)
            if (lact == i && DP[i] > ae) lact = i - 1;
        lact += 1;
    }
    return finalize_result(output_mode, res, norm);
}

/* Differential self-test of the kernel model: (a) unrestricted, no cone == the line-faithful core;
 * (b) restricted to e_lo - 2(m+kb) - 1 .. e_hi WITH the cone == the line-faithful core over the whole range.
 * Returns the number of disagreements (first one described in first_bad[0..7]). */
int64_t orc_selftest_cone(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[6] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[260];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 2 + (int64_t)(st_next(&s) % 31);
        int64_t n = (int64_t)(st_next(&s) % 250);
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % 4];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % ((st_next(&s) % 50) ? 4 : 5)];
        int copies = (int)(st_next(&s) % 3);
        for (int cpy = 0; cpy < copies && n > 0; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 5) r[pos++] = (uint8_t)AL[st_next(&s) % 4];
                else if (u < 8) continue;
                else if (u < 11) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        double rate = RATES[st_next(&s) % 8];
        int64_t mismatch = 1 + (int64_t)(st_next(&s) % 3), indel = 1 + (int64_t)(st_next(&s) % 3);
        int64_t match = (st_next(&s) % 8) == 0 ? 1 : 0;
        int32_t mode = (int32_t)(st_next(&s) % 2);
        int32_t trim = mode ? (int32_t)((int[]){0, 3, 5}[st_next(&s) % 3]) : 0;
        int64_t first = 1, last = n, max_start = n, min_end = 1;
        if (n > 0 && (st_next(&s) % 3) == 0) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        if (n > 0 && (st_next(&s) % 4) == 0) max_start = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
        if (n > 0 && (st_next(&s) % 4) == 0) min_end = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
        int64_t ae0 = (int64_t)floor(rate * (double)m);
        int64_t cmin = mismatch < indel ? mismatch : indel;
        int64_t kb = ae0 < 0 ? -1 : ae0 / cmin;
        double used = (st_next(&s) % 3) ? rate : rate * (double)(st_next(&s) % 100) / 100.0;
        orc_align_t full = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, 0, 1, mode, trim, first,
                                                last, max_start, min_end, m, INT64_MIN, INT64_MAX);
        orc_align_t plain = kernel_model_core(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, max_start,
                                              min_end, m, INT64_MIN, INT64_MAX, 0);
        int64_t e_lo = 0, e_hi = 0;
        int64_t f = first < 1 ? 1 : first, l = last > n ? n : last;
        int found = (n > 0 && l >= f && kb >= 0) ? (int)unit_cols(q, m, r, f, l, 0, kb, &e_lo, &e_hi) : 0;
        orc_align_t res;
        if (!found) {
            res.score = INFINITY;
            res.raw = INF_INT;
            res.start = -1;
            res.end = -1;
        } else {
            res = kernel_model_core(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, max_start, min_end,
                                    m, e_lo - 2 * (m + kb) - 1, e_hi, 1);
        }
        const int bad_plain = plain.raw != full.raw || plain.start != full.start || plain.end != full.end;
        const int bad_cone = res.raw != full.raw || res.start != full.start || res.end != full.end;
        if (bad_plain || bad_cone) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = full.raw;
                first_bad[4] = bad_plain ? plain.raw : res.raw; first_bad[5] = full.start;
                first_bad[6] = bad_plain ? plain.start : res.start; first_bad[7] = (bad_plain ? 1000 : 0) + mode * 10 + trim;
            }
            bad++;
        }
    }
    return bad;
}

/* ---- model of the HIP "clean class" DP (csrc/bdx_core.h sg_core_clean), test-only -----------------------
 * Claim: SimpleScoring with match >= 0 and mismatch, indel >= 1, and for this read neither the
 * start nor the end range binds (max_start_pos >= n, min_end_pos <= 1; the column window first:last may be any
 * sub-range, also a restricted one).  Then the reference's banded cut-off loop (:287-442) records exactly what a
 * plain column-by-column semi-global DP over ALL m rows records: a cell <= allowed_error only depends on cells
 * <= allowed_error, those are inside fact..lact with their true values (Ukkonen; D[i][j] >= D[i-1][j-1]), stale
 * cells and cells outside the band are > allowed_error on both sides and never win a comparison, so values,
 * origins (deletion, then substitution if strictly less, then insertion if strictly less, :310-321) and the
 * recorded (score, start, end) agree.  Row m takes no horizontal move (:213/:229): its recorded value is
 * min(del, sub).  orc_selftest_clean_class checks the claim against the line-faithful core. */
static orc_align_t clean_dp(const uint8_t *q0, int64_t m, const uint8_t *r0, int64_t n, double max_error, int64_t match,
                            int64_t mismatch, int64_t indel, int32_t output_mode, int32_t trim_side, int64_t first,
                            int64_t last, int64_t norm, int64_t col_lo, int64_t col_hi) {
    const int tb = output_mode == ORC_OUT_TRACEBACK;
    res_t result = init_result();
    if (m == 0 || n == 0) return finalize_result(output_mode, result, norm);
    const int64_t ae = (int64_t)floor(max_error * (double)norm);
    int64_t D[64], O[64];
    for (int64_t i = 1; i <= m; i++) {
        D[i] = indel * i;
        O[i] = 1 - i;
    }
    if (col_lo > first) first = col_lo;
    if (col_hi < last) last = col_hi;
    for (int64_t j = first; j <= last; j++) {
        int64_t prev = 0, prev_o = j, diag = 0, diag_o = j, vm = INF_INT, om = -1;
        for (int64_t i = 1; i <= m; i++) {
            const int64_t old = D[i], old_o = O[i];
            const int64_t ins = old + indel, del = prev + indel;
            const int64_t sub = diag + (q0[i - 1] == r0[j - 1] ? match : mismatch);
            int64_t b2 = del, o = prev_o;
            if (sub < b2) {
                b2 = sub;
                o = diag_o;
            }
            if (i == m) { /* the last row has no horizontal move */
                vm = b2;
                om = o;
            }
            int64_t nv = b2;
            if (ins < nv) {
                nv = ins;
                o = old_o;
            }
            D[i] = nv;
            O[i] = o;
            prev = nv;
            prev_o = o;
            diag = old;
            diag_o = old_o;
        }
        if (vm <= ae) { /* :417 with j >= min_end_pos always */
            if (vm == 0 && (!tb || trim_side == 5)) { /* :420-430 */
                result.score = 0;
                if (tb) {
                    result.start = om;
                    result.end = j;
                }
                return finalize_result(output_mode, result, norm);
            }
            result = tb ? update_result_traceback(trim_side, result, vm, j, om) : update_result_scoreonly(result, vm);
        }
    }
    return finalize_result(output_mode, result, norm);
}

int64_t orc_selftest_clean_class(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[6] = "ACGTN";
    static const double RATES[9] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5, 1.0};
    uint8_t q[40], r[260];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 1 + (int64_t)(st_next(&s) % 32);
        int64_t n = (int64_t)(st_next(&s) % 250);
        const int lowc = (st_next(&s) % 5) == 0; /* low-complexity pairs: many ties */
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : 4)];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : ((st_next(&s) % 50) ? 4 : 5))];
        int copies = (int)(st_next(&s) % 3);
        for (int cpy = 0; cpy < copies && n > 0; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 5) r[pos++] = (uint8_t)AL[st_next(&s) % 4];
                else if (u < 8) continue;
                else if (u < 11) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        double rate = RATES[st_next(&s) % 9];
        int64_t mismatch = 1 + (int64_t)(st_next(&s) % 3), indel = 1 + (int64_t)(st_next(&s) % 3);
        int64_t match = (st_next(&s) % 8) == 0 ? 1 : 0;
        int32_t mode = (int32_t)(st_next(&s) % 2);
        int32_t trim = mode ? (int32_t)((int[]){0, 3, 5}[st_next(&s) % 3]) : 0;
        int64_t first = 1, last = n;
        if (n > 0 && (st_next(&s) % 3) == 0) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        const int64_t max_start = n + (int64_t)(st_next(&s) % 3) * 40; /* non-binding */
        const int64_t min_end = 1 - (int64_t)(st_next(&s) % 2);
        double used = (st_next(&s) % 3) ? rate : rate * (double)(st_next(&s) % 100) / 100.0;
        /* sometimes a restricted column range: both sides get the same one */
        int64_t clo = INT64_MIN, chi = INT64_MAX;
        if (n > 0 && (st_next(&s) % 4) == 0) {
            clo = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            chi = clo + (int64_t)(st_next(&s) % (uint64_t)(n - clo + 1));
        }
        orc_align_t ref = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, 0, 0, mode, trim, first,
                                               last, max_start, min_end, m, clo, chi);
        orc_align_t got = clean_dp(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, clo, chi);
        if (got.raw != ref.raw || got.start != ref.start || got.end != ref.end) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = ref.raw;
                first_bad[4] = got.raw; first_bad[5] = ref.start; first_bad[6] = got.start; first_bad[7] = mode * 10 + trim;
            }
            bad++;
        }
    }
    return bad;
}

/* Short lookback of the clean class: when only the SCORE and the END column of the best alignment
 * are observable (ScoreOnly, or traceback with trim_side 5 whose start nobody reads), the clean-class DP restricted
 * to  e_lo - (m + kb) .. e_hi  ([e_lo, e_hi] = first / last column with unit distance <= kb) returns the same
 * (score, end) as the reference's full run: a cell <= allowed_error is the end of a path of <= m + kb columns that
 * starts at row 0, so it is exact once the run started m + kb columns earlier; the 2(m + kb) + 1 columns of the
 * general restricted run (§3.2) are only needed for the ORIGINS under ties. */
int64_t orc_selftest_clean_short_lookback(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[6] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[260];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 2 + (int64_t)(st_next(&s) % 31);
        int64_t n = (int64_t)(st_next(&s) % 250);
        const int lowc = (st_next(&s) % 5) == 0;
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : 4)];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : ((st_next(&s) % 50) ? 4 : 5))];
        int copies = (int)(st_next(&s) % 3);
        for (int cpy = 0; cpy < copies && n > 0; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 5) r[pos++] = (uint8_t)AL[st_next(&s) % 4];
                else if (u < 8) continue;
                else if (u < 11) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        double rate = RATES[st_next(&s) % 8];
        int64_t mismatch = 1 + (int64_t)(st_next(&s) % 3), indel = 1 + (int64_t)(st_next(&s) % 3);
        int64_t match = (st_next(&s) % 8) == 0 ? 1 : 0;
        int32_t mode = (int32_t)(st_next(&s) % 2);
        int32_t trim = mode ? 5 : 0; /* ScoreOnly, or traceback with trim_side 5 (only the end is compared) */
        int64_t first = 1, last = n;
        if (n > 0 && (st_next(&s) % 3) == 0) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        int64_t ae0 = (int64_t)floor(rate * (double)m);
        int64_t cmin = mismatch < indel ? mismatch : indel;
        int64_t kb = ae0 < 0 ? -1 : ae0 / cmin;
        if ((st_next(&s) % 3) == 0 && kb > 0) kb = (int64_t)(st_next(&s) % (uint64_t)(kb + 1)); /* a capped budget (tier 1) */
        double used = (st_next(&s) % 3) ? rate : rate * (double)(st_next(&s) % 100) / 100.0;
        orc_align_t full = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, 0, 0, mode, trim, first,
                                                last, n, 1, m, INT64_MIN, INT64_MAX);
        int64_t e_lo = 0, e_hi = 0;
        int64_t f = first < 1 ? 1 : first, l = last > n ? n : last;
        int found = (n > 0 && l >= f && kb >= 0) ? (int)unit_cols(q, m, r, f, l, 0, kb, &e_lo, &e_hi) : 0;
        if (!found) continue; /* (the barcode is not a candidate at this budget) */
        orc_align_t got = clean_dp(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, e_lo - (m + kb), e_hi);
        /* with a capped budget the full run may record an alignment of more than kb operations outside the window:
         * the claim (and the tier settle rule) is about results within the budget */
        int64_t ops_bound = kb * cmin; /* an alignment of <= kb operations may cost up to kb * max cost; compare when the full result is within kb * cmin */
        if (full.raw > ops_bound) continue;
        if (got.raw != full.raw || (mode && got.end != full.end)) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = full.raw;
                first_bad[4] = got.raw; first_bad[5] = full.end; first_bad[6] = got.end; first_bad[7] = mode * 10 + trim;
            }
            bad++;
        }
    }
    return bad;
}

/* ---- model of the HIP "diagonal band" DP (csrc/bdx_core.h sg_core_band), test-only --------------------------
 * Claim: clean class as above; [e_lo, e_hi] = first / last column of the pass window whose unit-cost
 * distance is <= kb (the fused kernel's tracked sweep), H >= (e_hi - e_lo + 1) + 2 kb.  An alignment of <= kb
 * operations that ends in row m at a column je of that window passes only through cells (i, j) with
 * |(je - j) - (m - i)| <= kb, i.e. on the diagonals  j - i  in  [e_lo - m - kb, e_hi - m + kb].  A DP that computes
 * only the H diagonals below  dtop = e_hi + kb - m  (everything else reads as infinite, row 0 as 0 with origin j,
 * the reference's initial column indel*i / 1-i when the band crosses the first column of the pass window) records
 * the same (score, start, end) as the reference's full run whenever that result costs <= kb * cmin: every cell on a
 * recorded path, and every predecessor that attains a minimum on it, lies inside the band with its true value; the
 * other predecessors only grow.  With a capped kb (tier 1) anything else it records costs >= (kb + 1) * cmin.
 * Columns are walked relative to the lane's own anchor (step k <-> column e_hi + kb - (m + H - 2) + k), so the rows
 * of a step are the same for every lane: rows k - H + 2 .. k + 1 (clipped to 1 .. m). */
static orc_align_t band_dp(const uint8_t *q0, int64_t m, const uint8_t *r0, int64_t n, double max_error, int64_t match,
                           int64_t mismatch, int64_t indel, int32_t output_mode, int32_t trim_side, int64_t first,
                           int64_t last, int64_t norm, int64_t e_lo, int64_t e_hi, int64_t kb, int64_t H) {
    const int tb = output_mode == ORC_OUT_TRACEBACK;
    res_t result = init_result();
    if (m == 0 || n == 0) return finalize_result(output_mode, result, norm);
    const int64_t ae = (int64_t)floor(max_error * (double)norm);
    int64_t D[80], O[80];
    for (int64_t i = 0; i < 80; i++) {
        D[i] = INF_INT;
        O[i] = -1;
    }
    if (first < 1) first = 1;
    if (last > n) last = n;
    const int64_t K = m + H - 2, A = e_hi + kb;
    for (int64_t k = 0; k <= K; k++) {
        const int64_t j = A - K + k;
        int64_t ra = k - H + 2, rb = k + 1; /* band rows of this step */
        if (j == first) { /* the reference's initial column (:278-283) on the rows the band held one step earlier */
            for (int64_t i = ra - 1; i <= rb - 1; i++)
                if (i >= 1 && i <= m) {
                    D[i] = indel * i;
                    O[i] = 1 - i;
                }
        }
        if (j < first || j > last || j > e_hi) continue;
        const int64_t lo = ra < 1 ? 1 : ra, hi = rb > m ? m : rb;
        int64_t prev = INF_INT, prev_o = -1, diag = INF_INT, diag_o = -1, vm = INF_INT, om = -1;
        if (lo == 1) {
            prev = 0, prev_o = j, diag = 0, diag_o = j; /* row 0: value 0, origin j (:288, :308) */
        } else {
            diag = D[lo - 1], diag_o = O[lo - 1]; /* (lo-1, j-1) lies on the band's top diagonal */
        }
        for (int64_t i = lo; i <= hi; i++) {
            const int64_t old = D[i], old_o = O[i];
            const int64_t ins = (i == rb) ? INF_INT : old + indel; /* (i, j-1) below the band */
            const int64_t del = prev >= INF_INT ? INF_INT : prev + indel;
            const int64_t sub = diag >= INF_INT ? INF_INT : diag + (q0[i - 1] == r0[j - 1] ? match : mismatch);
            int64_t b2 = del, o = prev_o;
            if (sub < b2) {
                b2 = sub;
                o = diag_o;
            }
            if (i == m) { /* the last row has no horizontal move */
                vm = b2;
                om = o;
            }
            int64_t nv = b2;
            if (ins < nv) {
                nv = ins;
                o = old_o;
            }
            if (nv > INF_INT) nv = INF_INT;
            D[i] = nv;
            O[i] = o;
            prev = nv;
            prev_o = o;
            diag = old;
            diag_o = old_o;
        }
        if (j >= e_lo && vm <= ae) {
            if (vm == 0 && (!tb || trim_side == 5)) { /* :420-430 */
                result.score = 0;
                if (tb) {
                    result.start = om;
                    result.end = j;
                }
                return finalize_result(output_mode, result, norm);
            }
            result = tb ? update_result_traceback(trim_side, result, vm, j, om) : update_result_scoreonly(result, vm);
        }
    }
    return finalize_result(output_mode, result, norm);
}

/* ---- model of the HIP "rolling band" DP (csrc/bdx_core.h sg_band_roll), test-only -----------------------------------
 * The same band as band_dp — the diagonals  j - i  in  [c_lo - m - kb, c_hi - m + kb]  of the alignments that end in row m
 * at a column of [c_lo, c_hi] — walked column by column over a ROLLING window of rows: at column j the band holds the H =
 * (c_hi - c_lo + 1) + 2 kb rows  j - dtop .. j - d0,  row i lives in slot i mod H (the row that enters the band takes the
 * slot of the row that left it): what a barcode of ANY length needs per lane.  The kernel's operations in the kernel's order;
 * the slots start out holding junk (whatever the previous candidate left there) to show that no cell is read before it is
 * written. */
static orc_align_t band_dp_roll(const uint8_t *q0, int64_t m, const uint8_t *r0, int64_t n, double max_error, int64_t match,
                                int64_t mismatch, int64_t indel, int32_t output_mode, int32_t trim_side, int64_t first,
                                int64_t last, int64_t norm, int64_t c_lo, int64_t c_hi, int64_t kb, uint64_t junk) {
    const int tb = output_mode == ORC_OUT_TRACEBACK;
    res_t result = init_result();
    if (m == 0 || n == 0) return finalize_result(output_mode, result, norm);
    const int64_t ae = (int64_t)floor(max_error * (double)norm);
    if (first < 1) first = 1;
    if (last > n) last = n;
    if (c_lo < first) c_lo = first;
    if (c_hi > last) c_hi = last;
    if (c_lo > c_hi) return finalize_result(output_mode, result, norm);
    const int64_t H = (c_hi - c_lo + 1) + 2 * kb, d0 = c_lo - m - kb, dtop = d0 + H - 1;
    int64_t V[600], O[600];
    if (H > 600) return finalize_result(output_mode, result, norm);
    for (int64_t h = 0; h < H; h++) {
        junk = junk * 6364136223846793005ULL + 1442695040888963407ULL;
        V[h] = (int64_t)((junk >> 33) % 7) - 3; /* small values: a stale cell that WAS read would win a comparison */
        O[h] = (int64_t)((junk >> 40) % 400) - 100;
    }
    int64_t j = d0 + 1 > first ? d0 + 1 : first;
    if (j > c_hi) return finalize_result(output_mode, result, norm);
    if (j == first) { /* the reference's initial column (:278-283) on the rows the band held one column earlier */
        int64_t ia = first - 1 - dtop, ib = first - 1 - d0;
        if (ia < 1) ia = 1;
        if (ib > m) ib = m;
        for (int64_t i = ia; i <= ib; i++) {
            V[i % H] = indel * i;
            O[i % H] = 1 - i;
        }
    }
    int64_t lo = j - dtop;
    if (lo < 1) lo = 1;
    for (; j <= c_hi; j++) {
        const int64_t rj = r0[j - 1];
        const int64_t rb = j - d0, hi = rb > m ? m : rb;
        int64_t prev = 0, prev_o = j, diag = 0, diag_o = j; /* row 0: value 0, origin j (:288, :308) */
        if (lo > 1) {
            diag = V[(lo - 1) % H];
            diag_o = O[(lo - 1) % H];
        }
        int64_t vm = INF_INT, om = -1;
        for (int64_t i = lo; i <= hi; i++) {
            const int64_t old = V[i % H], old_o = O[i % H];
            const int64_t sub = diag + (q0[i - 1] == rj ? match : mismatch);
            int64_t b2 = sub, o = diag_o;
            if (i > lo || lo == 1) {
                const int64_t del = prev + indel;
                o = sub < del ? diag_o : prev_o;
                b2 = sub < del ? sub : del;
            }
            if (i == m) {
                vm = b2;
                om = o;
            } else {
                int64_t nv = b2;
                if (i != rb) {
                    const int64_t ins = old + indel;
                    o = ins < b2 ? old_o : o;
                    nv = ins < b2 ? ins : b2;
                }
                V[i % H] = nv;
                O[i % H] = o;
                prev = nv;
                prev_o = o;
            }
            diag = old;
            diag_o = old_o;
        }
        if (hi == m && j >= c_lo && vm <= ae) {
            if (vm == 0 && (!tb || trim_side == 5)) { /* :420-430 */
                result.score = 0;
                if (tb) {
                    result.start = om;
                    result.end = j;
                }
                return finalize_result(output_mode, result, norm);
            }
            result = tb ? update_result_traceback(trim_side, result, vm, j, om) : update_result_scoreonly(result, vm);
        }
        if (j + 1 - dtop > 1) lo += 1;
    }
    return finalize_result(output_mode, result, norm);
}

int64_t orc_selftest_band_class(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL; no mismatch: [1..3] = cases compared exactly, of those with traceback, of those with the band crossing the first column */) {
    int64_t n_exact = 0, n_tb = 0, n_edge = 0, n_chunked = 0;
    static const char AL[6] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[260];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 2 + (int64_t)(st_next(&s) % 31);
        int64_t n = (int64_t)(st_next(&s) % 250);
        const int lowc = (st_next(&s) % 5) == 0;
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : 4)];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % (lowc ? 2 : ((st_next(&s) % 50) ? 4 : 5))];
        int copies = (int)(st_next(&s) % 3);
        for (int cpy = 0; cpy < copies && n > 0; cpy++) {
            /* (also copies hanging over either end of the read: the band then crosses the first column) */
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)(n + 8)) - 4;
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (pos < 0) { pos++; continue; }
                if (u < 5) r[pos++] = (uint8_t)AL[st_next(&s) % 4];
                else if (u < 8) continue;
                else if (u < 11) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        double rate = RATES[st_next(&s) % 8];
        int64_t mismatch = 1 + (int64_t)(st_next(&s) % 3), indel = 1 + (int64_t)(st_next(&s) % 3);
        int64_t match = (st_next(&s) % 8) == 0 ? 1 : 0;
        int32_t mode = (int32_t)(st_next(&s) % 2);
        int32_t trim = mode ? (int32_t)((int[]){0, 3, 5}[st_next(&s) % 3]) : 0;
        int64_t first = 1, last = n;
        if (n > 0 && (st_next(&s) % 3) == 0) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        int64_t ae0 = (int64_t)floor(rate * (double)m);
        int64_t cmin = mismatch < indel ? mismatch : indel;
        int64_t kb = ae0 < 0 ? -1 : ae0 / cmin;
        if ((st_next(&s) % 3) == 0 && kb > 0) kb = (int64_t)(st_next(&s) % (uint64_t)(kb + 1)); /* a capped budget (tier 1) */
        double used = (st_next(&s) % 3) ? rate : rate * (double)(st_next(&s) % 100) / 100.0;
        orc_align_t full = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, 0, 0, mode, trim, first,
                                                last, n, 1, m, INT64_MIN, INT64_MAX);
        int64_t e_lo = 0, e_hi = 0;
        int64_t f = first < 1 ? 1 : first, l = last > n ? n : last;
        int found = (n > 0 && l >= f && kb >= 0) ? (int)unit_cols(q, m, r, f, l, 0, kb, &e_lo, &e_hi) : 0;
        if (!found) continue; /* (the barcode is not a candidate at this budget) */
        int64_t H = (e_hi - e_lo + 1) + 2 * kb + (int64_t)(st_next(&s) % 4);
        if (H > 40) continue; /* (the kernel falls back to the all-rows DP) */
        orc_align_t got = band_dp(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, e_lo, e_hi, kb, H);
        if ((st_next(&s) % 2) == 0 && e_hi > e_lo) {
            /* a wide window walked in chunks of end columns (the kernel's third phase): every chunk is a band of its own,
             * the chunk results are folded in ascending column order with the recording rule (:142-153) and the early
             * exit on a zero (:420-430) */
            const int64_t cw = 1 + (int64_t)(st_next(&s) % (uint64_t)(e_hi - e_lo + 1));
            const int tb = mode == ORC_OUT_TRACEBACK;
            orc_align_t acc = {INFINITY, INF_INT, -1, -1};
            int have = 0;
            for (int64_t c_lo = e_lo; c_lo <= e_hi; c_lo += cw) {
                const int64_t c_hi = c_lo + cw - 1 < e_hi ? c_lo + cw - 1 : e_hi;
                orc_align_t part = band_dp(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, c_lo, c_hi, kb,
                                           cw + 2 * kb + (int64_t)(st_next(&s) % 3));
                if (part.raw >= INF_INT) continue;
                if (!have || part.raw < acc.raw || (part.raw == acc.raw && tb && trim == 3 && part.start > acc.start)) {
                    acc = part;
                    have = 1;
                }
                if (part.raw == 0 && (!tb || trim == 5)) break;
            }
            if (have) got = acc;
            else got.raw = INF_INT, got.start = -1, got.end = -1;
            n_chunked++;
        }
        const int64_t ops_bound = kb * cmin;
        int wrong;
        if (full.raw <= ops_bound) wrong = got.raw != full.raw || (mode && (got.end != full.end || got.start != full.start));
        else wrong = got.raw <= ops_bound; /* beyond the budget: anything, but never a value inside it */
        {
            /* the rolling form of the same band (sg_band_roll): whole range, and folded over chunks of end columns */
            orc_align_t gr = band_dp_roll(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, e_lo, e_hi, kb, st_next(&s));
            if ((st_next(&s) % 2) == 0) {
                const int64_t cw = 1 + (int64_t)(st_next(&s) % (uint64_t)(e_hi - e_lo + 1));
                const int tb = mode == ORC_OUT_TRACEBACK;
                orc_align_t acc = {INFINITY, INF_INT, -1, -1};
                int have = 0;
                for (int64_t c_lo = e_lo; c_lo <= e_hi; c_lo += cw) {
                    const int64_t c_hi = c_lo + cw - 1 < e_hi ? c_lo + cw - 1 : e_hi;
                    orc_align_t part = band_dp_roll(q, m, r, n, used, match, mismatch, indel, mode, trim, first, last, m, c_lo, c_hi, kb, st_next(&s));
                    if (part.raw >= INF_INT) continue;
                    if (part.raw == 0 && (!tb || trim == 5)) {
                        acc = part;
                        have = 1;
                        break;
                    }
                    if (!have || part.raw < acc.raw || (part.raw == acc.raw && tb && trim == 3 && part.start > acc.start)) {
                        acc = part;
                        have = 1;
                    }
                }
                if (have) gr = acc;
                else gr.raw = INF_INT, gr.start = -1, gr.end = -1;
            }
            if (full.raw <= ops_bound) wrong = wrong || gr.raw != full.raw || (mode && (gr.end != full.end || gr.start != full.start));
            else wrong = wrong || gr.raw <= ops_bound;
        }
        n_exact += full.raw <= ops_bound;
        n_tb += full.raw <= ops_bound && mode;
        n_edge += full.raw <= ops_bound && e_hi + kb - (m + H - 2) < first; /* the band crosses the window's first column */
        if (wrong) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = full.raw;
                first_bad[4] = got.raw; first_bad[5] = full.start; first_bad[6] = got.start; first_bad[7] = mode * 10 + trim;
            }
            bad++;
        }
    }
    if (bad == 0 && first_bad) {
        first_bad[1] = n_exact;
        first_bad[2] = n_tb;
        first_bad[3] = n_edge;
        first_bad[4] = n_chunked;
    }
    return bad;
}

int64_t orc_selftest_windowed_exact(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[6] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[260];
    int64_t DP[48], OG[48];
    int64_t bad = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 2 + (int64_t)(st_next(&s) % 31);
        int64_t n = (int64_t)(st_next(&s) % 250);
        int has_n = (st_next(&s) % 4) == 0;
        int64_t non_n = 0;
        for (int64_t i = 0; i < m; i++) {
            q[i] = (uint8_t)AL[st_next(&s) % ((has_n && (st_next(&s) % 6 == 0)) ? 5 : 4)];
            non_n += q[i] != 'N';
        }
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[st_next(&s) % ((st_next(&s) % 50) ? 4 : 5)];
        int copies = (int)(st_next(&s) % 3); /* 0, 1 or 2 mutated copies (ties / repeated occurrences) */
        for (int cpy = 0; cpy < copies && n > 0; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                uint8_t ch = q[i] == 'N' ? (uint8_t)AL[st_next(&s) % 4] : q[i];
                if (u < 5) r[pos++] = (uint8_t)AL[st_next(&s) % 4];
                else if (u < 8) continue;
                else if (u < 11) { r[pos++] = (uint8_t)AL[st_next(&s) % 4]; if (pos < n) r[pos++] = ch; }
                else r[pos++] = ch;
            }
        }
        double rate = RATES[st_next(&s) % 8];
        int64_t mismatch = 1 + (int64_t)(st_next(&s) % 3), indel = 1 + (int64_t)(st_next(&s) % 3);
        int64_t match = (st_next(&s) % 8) == 0 ? 1 : 0;
        int64_t nindel = 1 + (int64_t)(st_next(&s) % 2);
        int64_t norm = has_n ? non_n : m;
        int32_t mode = (int32_t)(st_next(&s) % 2);
        int32_t trim = mode ? (int32_t)((int[]){0, 3, 5}[st_next(&s) % 3]) : 0;
        /* ranges: whole read, a window, and sometimes binding start / end constraints */
        int64_t first = 1, last = n, max_start = n, min_end = 1;
        if (n > 0 && (st_next(&s) % 3) == 0) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        if (n > 0 && (st_next(&s) % 4) == 0) max_start = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
        if (n > 0 && (st_next(&s) % 4) == 0) min_end = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
        int64_t ae0 = (int64_t)floor(rate * (double)norm);
        int64_t cmin = mismatch < indel ? mismatch : indel;
        if (has_n && nindel < cmin) cmin = nindel;
        int64_t kb = ae0 < 0 ? -1 : ae0 / cmin;
        /* the threshold actually used may be tighter than the one kb was derived from */
        double used = (st_next(&s) % 3) ? rate : rate * (double)(st_next(&s) % 100) / 100.0;
        orc_align_t full = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, has_n, nindel, mode,
                                                trim, first, last, max_start, min_end, norm, INT64_MIN, INT64_MAX);
        int64_t e_lo = 0, e_hi = 0;
        int64_t f = first < 1 ? 1 : first, l = last > n ? n : last;
        int found = (n > 0 && l >= f && kb >= 0) ? (int)unit_cols(q, m, r, f, l, has_n, kb, &e_lo, &e_hi) : 0;
        orc_align_t res;
        if (!found) {
            res.score = INFINITY;
            res.raw = INF_INT;
            res.start = -1;
            res.end = -1;
        } else {
            res = semiglobal_core_cols(DP, OG, q, m, r, n, used, match, mismatch, indel, has_n, nindel, mode, trim,
                                       first, last, max_start, min_end, norm, e_lo - 2 * (m + kb) - 1, e_hi);
        }
        if (res.raw != full.raw || res.start != full.start || res.end != full.end) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = full.raw;
                first_bad[4] = res.raw; first_bad[5] = full.start; first_bad[6] = res.start; first_bad[7] = mode * 10 + trim;
            }
            bad++;
        }
    }
    return bad;
}

/* ---- model of the HIP wave kernel's known-trim class (csrc/bdx_wave.hip, KEND), test-only ------------------------------
 * Claim: in the known-score class (SimpleScoring 0 / 1 / 1, start / end ranges that do not bind) the
 * (score, start) a trim_side = 3 pass of the reference reports for one barcode — recording rule classification.jl:142-153
 * ("strictly better, or equal with a larger start"), origin rule :310-321 (deletion, then substitution if strictly less,
 * then insertion if strictly less), no early exit on a zero (:420-430) — is
 *      score = d* = the smallest unit-cost distance of the barcode to a substring of the column window, and
 *      start = the LARGEST origin (column at which row 1 is entered) among all alignments of cost d*,
 * and Myers' bit-vector sweep run RIGHT TO LEFT over the window with the REVERSED barcode delivers that start: after the
 * column of 0-based read position p the score is the smallest cost of an alignment leaving row 0 at node (0, p); the largest
 * such node with score d* is the column that lowered the running minimum LAST in sweep order (p*), and the start is p* + 1
 * when the diagonal move into the sweep's last row (= the barcode's FIRST base) is optimal there — top bit of Eq & Pv before
 * the step — else p* (a vertical first move enters row 1 at column p* itself); at the window's first column such an
 * alignment comes out of the reference's initial column (origins 1 - i <= 0, :278-283): reported as 0, which
 * max(1, start) - 1 (:910-911) cannot tell from the reference's value.
 * Likewise trim_side = 5 (the existing known-end class): end = the FIRST column of a left-to-right sweep that attains d*.
 * orc_known_trim_positions is the 32-bit model (top-aligned patterns, the kernel's operations in the kernel's order);
 * orc_selftest_known_start compares it with the line-faithful core. */
static void known_sweep32(const uint8_t *q, int64_t m, const uint8_t *r, int64_t lo, int64_t hi, int reversed,
                          int64_t *best_out, int64_t *col_out, int *sflag_out) {
    /* columns c = 0 .. hi - lo - 1 of the sweep: forward c <-> 0-based position lo + c, reversed c <-> hi - 1 - c */
    const int shift = (int)(32 - m);
    const uint32_t rows = m == 32 ? 0xFFFFFFFFu : (((1u << m) - 1u) << shift);
    uint32_t peq[256];
    for (int ch = 0; ch < 256; ch++) peq[ch] = ~rows; /* virtual rows below the barcode match everything */
    for (int64_t i = 0; i < m; i++) peq[reversed ? q[m - 1 - i] : q[i]] |= 1u << (shift + i);
    uint32_t Pv = rows, Mv = 0;
    int64_t score = m, best = INT64_MAX, col = -1;
    int sflag = 0;
    for (int64_t c = 0; c < hi - lo; c++) {
        const uint8_t ch = r[reversed ? hi - 1 - c : lo + c];
        const uint32_t Eq = peq[ch];
        const uint32_t Xv = Eq | Mv;
        const uint32_t ep = Eq & Pv;
        const uint32_t Xh = ((ep + Pv) ^ Pv) | Eq;
        uint32_t Ph = Mv | ~(Xh | Pv);
        uint32_t Mh = Pv & Xh;
        score += (int64_t)(Ph >> 31);
        score -= (int64_t)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        if (score < best) { /* this column lowered the running minimum */
            best = score;
            col = c;
            sflag = (int)(ep >> 31);
        }
    }
    *best_out = best;
    *col_out = col;
    *sflag_out = sflag;
}

/* (d*, position) of one barcode over the 0-based column window [lo, hi) of read r: trim_side 5 -> the 1-based end column,
 * trim_side 3 -> the start (0 stands for "<= 0"); returns 0 when the window is empty */
int orc_known_trim_positions(const uint8_t *q, int64_t m, const uint8_t *r, int64_t lo, int64_t hi, int32_t trim_side,
                             int64_t *d_out, int64_t *pos_out) {
    if (hi <= lo || m < 1 || m > 32) return 0;
    int64_t best, col;
    int sflag;
    known_sweep32(q, m, r, lo, hi, trim_side == 3, &best, &col, &sflag);
    *d_out = best;
    if (trim_side == 3) {
        const int64_t pstar = hi - 1 - col;
        *pos_out = (!sflag && pstar <= lo) ? 0 : pstar + sflag;
    } else {
        *pos_out = lo + col + 1;
    }
    return 1;
}

int64_t orc_selftest_known_start(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[5] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[200];
    int64_t DP[48], OG[48];
    int64_t bad = 0, n_start = 0, n_edge = 0, n_end = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 1 + (int64_t)(st_next(&s) % 32);
        int64_t n = 1 + (int64_t)(st_next(&s) % 160);
        const int nal = (st_next(&s) % 5) ? 4 : 2; /* low-complexity pairs: many ties */
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % (uint64_t)nal];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[(st_next(&s) % 50) ? st_next(&s) % (uint64_t)nal : 4];
        int64_t first = 1, last = n;
        if (st_next(&s) % 2) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        int copies = (int)(st_next(&s) % 3); /* 0, 1 or 2 mutated copies; half of them at the window's edges */
        for (int cpy = 0; cpy < copies; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            const uint64_t where = st_next(&s) % 6;
            if (where == 0) pos = first - 1;
            if (where == 1) pos = first - 2 >= 0 ? first - 2 : 0;
            if (where == 2) pos = last - m + (int64_t)(st_next(&s) % 3) - 1;
            if (pos < 0) pos = 0;
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 6) r[pos++] = (uint8_t)AL[st_next(&s) % (uint64_t)nal];
                else if (u < 10) continue;
                else if (u < 14) { r[pos++] = (uint8_t)AL[st_next(&s) % (uint64_t)nal]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        const double rate = RATES[st_next(&s) % 8];
        const int32_t trim = (st_next(&s) % 3) ? 3 : 5;
        const int64_t max_start = n + (int64_t)(st_next(&s) % 2) * 50, min_end = 1 - (int64_t)(st_next(&s) % 2);
        orc_align_t a = orc_semiglobal_core(DP, OG, q, m, r, n, rate, 0, 1, 1, 0, 0, ORC_OUT_TRACEBACK, trim, first, last,
                                            max_start, min_end, m);
        int64_t d = 0, pos = 0;
        const int have = orc_known_trim_positions(q, m, r, first - 1, last, trim, &d, &pos);
        const int64_t ae = (int64_t)floor(rate * (double)m);
        int ok;
        if (!have || d > ae) {
            ok = a.raw >= INF_INT;
        } else if (trim == 3) { /* the start, exactly when it is a column of the read; "<= 0" as a class */
            ok = a.raw == d && ((a.start >= 1 && a.start == pos) || (a.start < 1 && pos == 0));
            n_start += a.start >= 1;
            n_edge += a.start < 1;
        } else {
            ok = a.raw == d && a.end == pos;
            n_end++;
        }
        if (!ok) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = a.raw;
                first_bad[4] = d; first_bad[5] = a.start; first_bad[6] = trim == 3 ? pos : a.end; first_bad[7] = trim == 3 ? trim : pos;
            }
            bad++;
        }
    }
    if (bad == 0 && first_bad) { /* coverage: recorded starts inside the read / "<= 0" starts / recorded ends compared */
        first_bad[1] = n_start; first_bad[2] = n_edge; first_bad[3] = n_end;
    }
    return bad;
}

/* ---- model of the wave kernel's known-ALIGNMENT class (csrc/bdx_wave.hip, KEND = 3: anchored sweeps), test-only ---------
 * With one position of the winner known (orc_known_trim_positions), the OTHER one comes out of an anchored sweep:
 *  * end e known (trim_side 5 / none): the reference's start is origin(m, e) = the largest origin among the alignments of cost d
 *    that end exactly at column e: a right-to-left sweep from e with the reversed barcode whose row 0 is NOT free (the words lose
 *    their "virtual rows match everything" bits, the horizontal delta of row 0 is +1): the first column whose score equals d is
 *    the largest node p* an alignment of cost d ending at e leaves from; start = p* + 1 iff the diagonal move is optimal there
 *    (top bit of Eq & Pv before the step), else p*; p* at the window's first column without the diagonal move: <= 0 (reported 0);
 *  * start s known (trim_side 3): the reference's end is the first column at which an alignment of cost d with origin s ends
 *    (classification.jl:142-153: of equal starts the first column stays): a left-to-right sweep from column s + 1 whose first
 *    column is prepared as "row 1 entered at column s": D[i] = [q1 != r_s] + i - 1.
 * Returns 0 when the sweep does not find the distance (the kernel hands such a read on). */
int64_t orc_known_other_position(const uint8_t *q, int64_t m, const uint8_t *r, int64_t wlo, int64_t whi, int32_t trim_side,
                                 int64_t d, int64_t pos, int64_t kk) {
    if (m < 1 || m > 32) return 0;
    const int shift = (int)(32 - m);
    const uint32_t rows = m == 32 ? 0xFFFFFFFFu : (((1u << m) - 1u) << shift), lowbit = 1u << shift;
    const int rev = trim_side != 3;
    uint32_t peq[256];
    for (int ch = 0; ch < 256; ch++) peq[ch] = rev ? 0u : ~rows;
    for (int64_t i = 0; i < m; i++) peq[rev ? q[m - 1 - i] : q[i]] |= 1u << (shift + i);
    int64_t lo = rev ? pos - m - kk : pos, hi = rev ? pos : pos + m + kk;
    if (lo < wlo) lo = wlo;
    if (hi > whi) hi = whi;
    if (pos <= 0 || (rev && hi <= lo)) return 0;
    uint32_t Pv = rows, Mv = 0;
    int64_t score = m;
    if (!rev) {
        const int match1 = q[0] == r[pos - 1];
        if (match1) Pv &= ~lowbit;
        score = m - match1;
        if (score == d) return pos;  /* (the alignment ends in its first column: every later row deleted) */
    }
    for (int64_t c = 0; c < hi - lo; c++) {
        const uint32_t Eq = peq[r[rev ? hi - 1 - c : lo + c]];
        const uint32_t Xv = Eq | Mv;
        const uint32_t ep = Eq & Pv;
        const uint32_t Xh = ((ep + Pv) ^ Pv) | Eq;
        uint32_t Ph = Mv | ~(Xh | Pv);
        uint32_t Mh = Pv & Xh;
        score += (int64_t)(Ph >> 31);
        score -= (int64_t)(Mh >> 31);
        Ph <<= 1;
        Mh <<= 1;
        if (rev) Ph |= lowbit;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        if (score == d) {
            if (!rev) return lo + c + 1;
            const int64_t pstar = hi - 1 - c;
            const int sfl = (int)(ep >> 31);
            return (!sfl && pstar <= wlo) ? 0 : pstar + sfl;
        }
    }
    return 0;
}

int64_t orc_selftest_known_alignment(uint64_t seed, int64_t iters, int64_t *first_bad /* 8 ints or NULL */) {
    static const char AL[5] = "ACGTN";
    static const double RATES[8] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.34, 0.5};
    uint8_t q[40], r[200];
    int64_t DP[48], OG[48];
    int64_t bad = 0, n_start = 0, n_end = 0, n_lost = 0;
    uint64_t s = seed;
    for (int64_t it = 0; it < iters; it++) {
        int64_t m = 1 + (int64_t)(st_next(&s) % 32);
        int64_t n = 1 + (int64_t)(st_next(&s) % 160);
        const int nal = (st_next(&s) % 5) ? 4 : 2;
        for (int64_t i = 0; i < m; i++) q[i] = (uint8_t)AL[st_next(&s) % (uint64_t)nal];
        for (int64_t j = 0; j < n; j++) r[j] = (uint8_t)AL[(st_next(&s) % 50) ? st_next(&s) % (uint64_t)nal : 4];
        int64_t first = 1, last = n;
        if (st_next(&s) % 2) {
            first = 1 + (int64_t)(st_next(&s) % (uint64_t)n);
            last = first + (int64_t)(st_next(&s) % (uint64_t)(n - first + 1));
        }
        int copies = (int)(st_next(&s) % 3);
        for (int cpy = 0; cpy < copies; cpy++) {
            int64_t pos = (int64_t)(st_next(&s) % (uint64_t)n);
            const uint64_t where = st_next(&s) % 6;
            if (where == 0) pos = first - 1;
            if (where == 1) pos = first - 2 >= 0 ? first - 2 : 0;
            if (where == 2) pos = last - m + (int64_t)(st_next(&s) % 3) - 1;
            if (pos < 0) pos = 0;
            for (int64_t i = 0; i < m && pos < n; i++) {
                uint64_t u = st_next(&s) % 100;
                if (u < 6) r[pos++] = (uint8_t)AL[st_next(&s) % (uint64_t)nal];
                else if (u < 10) continue;
                else if (u < 14) { r[pos++] = (uint8_t)AL[st_next(&s) % (uint64_t)nal]; if (pos < n) r[pos++] = q[i]; }
                else r[pos++] = q[i];
            }
        }
        const double rate = RATES[st_next(&s) % 8];
        const int32_t trim = (int32_t)((int[]){0, 3, 5}[st_next(&s) % 3]);
        orc_align_t a = orc_semiglobal_core(DP, OG, q, m, r, n, rate, 0, 1, 1, 0, 0, ORC_OUT_TRACEBACK, trim, first, last, n, 1, m);
        const int64_t ae = (int64_t)floor(rate * (double)m);
        if (a.raw >= INF_INT) continue;
        /* the kernel's first sweep gives (d, position): the end for trim_side 5 / none, the start for trim_side 3 */
        int64_t d = 0, pos = 0;
        if (!orc_known_trim_positions(q, m, r, first - 1, last, trim == 3 ? 3 : 5, &d, &pos) || d != a.raw || d > ae) {
            bad++;
            continue;
        }
        const int64_t other = orc_known_other_position(q, m, r, first - 1, last, trim, d, pos, ae);
        int ok;
        if (trim == 3) {
            if (a.start < 1) { n_lost += pos == 0; ok = pos == 0; }  /* (start <= 0: the kernel hands the read on) */
            else { ok = pos == a.start && other == a.end; n_end++; }
        } else {
            if (pos != a.end) ok = 0;
            else if (a.start < 1) { ok = other == 0; n_lost++; }
            else { ok = other == a.start; n_start++; }
        }
        if (!ok) {
            if (bad == 0 && first_bad) {
                first_bad[0] = it; first_bad[1] = m; first_bad[2] = n; first_bad[3] = a.raw;
                first_bad[4] = a.start; first_bad[5] = a.end; first_bad[6] = pos; first_bad[7] = other * 10 + trim;
            }
            bad++;
        }
    }
    if (bad == 0 && first_bad) { first_bad[1] = n_start; first_bad[2] = n_end; first_bad[3] = n_lost; }
    return bad;
}
