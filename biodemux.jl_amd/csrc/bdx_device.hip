// bdx_device.hip — gfx950 kernels of the classification hot path (exact evaluation stage).
//
// bdx_generic_kernel evaluates, for one read per lane, exactly what the reference's worker
// does for that read (BioDemuX.jl src/core.jl:243-267 -> src/classification.jl:871
// determine_filename -> :776 match_barcode_pass -> :722 find_best_matching_bc ->
// :238 semiglobal_alignment_core / :557 hamming_align / :485 exact_align): the barcodes are
// visited in file order with the Float64 threshold tightening of :658-664 / :696-707, the
// banded cut-off DP keeps the same fact/lact bookkeeping, the same band seed and the same
// origin tie-breaks, so every quirk listed in SURVEY.md §8(a) Q1-Q12 is reproduced by
// construction rather than by argument.  It is used in two roles:
//   * unfiltered ("generic" path): every barcode is evaluated;
//   * verify stage of the filtered paths: a lossless pre-filter (bdx_filter.hip) hands over a
//     per-read candidate bit mask and only those barcodes are evaluated — barcodes that the
//     filter drops would have returned Inf and cannot change the reducer state.
//
// Data layout (MI355X): one workgroup owns BS consecutive reads.  Their bytes form ONE
// contiguous span of the packed batch, which the workgroup copies HBM -> LDS with 16-byte
// coalesced loads (each byte of the batch is fetched from HBM exactly once); the barcode
// table is staged next to it.  The per-lane DP column (and origin column) live in LDS in a
// lane-interleaved layout  DP[row * BS + lane]  so that any per-lane row index hits bank
// (lane mod 32): conflict-free by construction even when lanes sit on different rows.
// All DP arithmetic is int32 (see bdx_internal.h for the bound that makes that exact); the
// accept/tighten/ambiguity decisions are IEEE doubles exactly as written in the reference.
#include "bdx_internal.h"

#define LDS __attribute__((address_space(3)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Costs {
    int match, mismatch, indel, nindel;
};

struct AlignOut {
    int raw;    // BDX_INF32 when nothing was recorded
    int start;  // 1-based, -1 when not tracked
    int end;
};

// Byte accessors: reads and barcodes either staged in LDS or left in global memory.
template <bool STAGED>
struct Bytes;
template <>
struct Bytes<true> {
    const LDS uint8_t *p;
    __device__ __forceinline__ int operator[](int i) const { return p[i]; }
    __device__ __forceinline__ Bytes<true> at(long long o) const { return Bytes<true>{p + o}; }
};
template <>
struct Bytes<false> {
    const uint8_t *p;
    __device__ __forceinline__ int operator[](int i) const { return p[i]; }
    __device__ __forceinline__ Bytes<false> at(long long o) const { return Bytes<false>{p + o}; }
};

// resolve(), classification.jl:96-100, with Julia's UnitRange normalisation (empty a:b has
// last == a-1; the callers use last(range), :800-801).
__device__ __forceinline__ void resolve_range(const BdxDevRange &dr, long long len, long long &first,
                                              long long &last) {
    long long s = dr.start_from_end ? len + dr.start_offset : dr.start_offset;
    long long e = dr.end_from_end ? len + dr.end_offset : dr.end_offset;
    long long a = s > 1 ? s : 1;
    long long b = e < len ? e : len;
    if (b < a) b = a - 1;
    first = a;
    last = b;
}

// semiglobal_alignment_core, classification.jl:238-445, one (read, barcode) pair per lane.
// DP / OG point at this lane's column: row i lives at DP[i * S].  q, r are 0-based here.
// The three cell sites of the reference (:303 first row via step_scores, :340 main loop via
// step_scores_main, :377 last row via step_scores) are one loop: step_scores differs from
// step_scores_main only for i == 1 (row-0 value 0) and i == m (no horizontal move), and the
// first iteration is the only one that can have i == 1.  DP[i-1] of the previous column is
// carried in `diag` (it is the value read as DP[i] one iteration earlier), likewise origin.
template <bool TB, bool NS, bool STAGED>
__device__ __forceinline__ AlignOut sg_core(LDS int *DP, LDS int *OG, const int S, const Bytes<STAGED> q,
                                            const int m, const Bytes<STAGED> r, const int n, const int ae,
                                            const Costs c, const int trim_side, int first, const int last,
                                            const int max_start, const int min_end) {
    AlignOut res{BDX_INF32, -1, -1};
    if (m == 0 || n == 0) return res;  // :250-252

    const int steps = ae / (NS ? (c.indel < c.nindel ? c.indel : c.nindel) : c.indel);  // :257, :170-176
    const int min_valid_start = min_end - (m + steps) + 1;                            // :259
    if (min_valid_start > max_start) return res;                                      // :261-263
    if (min_valid_start > first) first = min_valid_start;                             // :266-268
    const int b1 = m - n - steps, b2 = -max_start - steps;
    const int band = b1 > b2 ? b1 : b2;  // :270

    for (int i = 1; i <= m; ++i) {  // :278-283
        DP[i * S] = c.indel * i;
        if (TB) OG[i * S] = 1 - i;
    }

    int lact = (ae + 1 < m) ? ae + 1 : m;  // :286
    for (int j = first; j <= last; ++j) {  // :287
        int prev_o = j;                    // :288
        int fact, prev;
        if (j + band >= 1) {  // :289-295
            fact = j + band;
            prev = ae;
        } else {
            fact = 1;
            prev = 0;
        }
        if (fact > lact) return res;  // :297-299

        const int rj = r[j - 1];
        int diag = (fact == 1) ? 0 : DP[(fact - 1) * S];
        int diag_o = 0;
        if (TB) diag_o = (fact == 1) ? j : OG[(fact - 1) * S];
        for (int i = fact; i <= lact; ++i) {
            const int qi = q[i - 1];
            const bool isN = NS && (qi == 'N');
            const int cost = isN ? c.nindel : c.indel;       // :196-197
            const int dpi = DP[i * S];
            const int ins = (i == m) ? BDX_INF32 : dpi + cost;  // :213 / :229, :183
            const int del = prev + cost;                      // :184
            const int sub = diag + ((qi == rj || isN) ? c.match : c.mismatch);  // :185, :202-203, :215
            int cur_o = 0;
            if (TB) {  // :310-321: deletion, then substitution if strictly less, then insertion
                const int ins_o = OG[i * S];
                int best = del;
                cur_o = prev_o;
                if (sub < best) {
                    best = sub;
                    cur_o = diag_o;
                }
                if (ins < best) cur_o = ins_o;
                diag_o = ins_o;
            }
            if (i != 1) {  // :326-331, :364-367
                DP[(i - 1) * S] = prev;
                if (TB) OG[(i - 1) * S] = prev_o;
            }
            int t = del < sub ? del : sub;
            prev = ins < t ? ins : t;  // :332
            if (TB) prev_o = cur_o;
            diag = dpi;
        }
        DP[lact * S] = prev;  // :412-415
        if (TB) OG[lact * S] = prev_o;

        if (lact == m && prev <= ae) {  // :417
            lact -= 1;
            if (j >= min_end) {
                if (prev == 0 && (!TB || trim_side == 5)) {  // :420-430
                    AlignOut z{0, TB ? prev_o : -1, TB ? j : -1};
                    return z;
                }
                if (TB) {  // update_result, :142-153
                    if (prev < res.raw || (prev == res.raw && trim_side == 3 && prev_o > res.start)) {
                        res.raw = prev;
                        res.start = prev_o;
                        res.end = j;
                    }
                } else {  // :138-140
                    res.raw = prev < res.raw ? prev : res.raw;
                }
            }
        }
        while (lact > 0 && DP[lact * S] > ae) --lact;  // :439-441
        ++lact;                                         // :442
    }
    return res;  // :444
}

// hamming_align, classification.jl:557-625.  Scores of one call share the divisor m, so the
// reference's Float64 `score < best_score` / `==` are decided on the integer numerators.
template <bool STAGED>
__device__ __forceinline__ AlignOut hamming_dev(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                                const int n, const int allowed, const int first,
                                                const int last, const int max_start, const int min_end,
                                                const int trim_side) {
    AlignOut best{BDX_INF32, -1, -1};
    const int sf = first > 1 ? first : 1;  // :570
    int sl = last < max_start ? last : max_start;
    if (n - m + 1 < sl) sl = n - m + 1;  // :571
    if (sl < sf) return best;            // :573-576
    if (m == 0) return best;             // 0/0 = NaN never beats Inf (:607-613)
    for (int j = sf; j <= sl; ++j) {     // :581
        const int end_pos = j + m - 1;
        if (end_pos < min_end) continue;  // :584-586
        int mism = 0;
        bool failed = false;
        for (int k = 0; k < m; ++k) {  // :592-604
            const int qc = q[k];
            const int rc = r[j - 1 + k];
            if (qc != rc && qc != 0x4E) {
                if (++mism > allowed) {
                    failed = true;
                    break;
                }
            }
        }
        if (!failed) {  // :606-621
            if (mism < best.raw) {
                best.raw = mism;
                best.start = j;
                best.end = end_pos;
            } else if (mism == best.raw && trim_side == 3 && j > best.start) {
                best.start = j;
                best.end = end_pos;
            }
        }
    }
    return best;
}

template <bool STAGED>
__device__ __forceinline__ bool bytes_equal(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                            const int s /*1-based*/) {
    for (int k = 0; k < m; ++k)
        if (q[k] != r[s - 1 + k]) return false;
    return true;
}

// exact_align, classification.jl:485-548.  findnext(query, ref, i) = leftmost occurrence
// starting at or after i; findprev(query, ref, k) = rightmost occurrence ENDING at or before k.
template <bool STAGED>
__device__ __forceinline__ AlignOut exact_dev(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                              const int n, const int first, const int last,
                                              const int max_start, const int min_end, const int trim_side) {
    AlignOut none{BDX_INF32, -1, -1};
    const int sf = first > 1 ? first : 1;  // :490
    int sl = last < max_start ? last : max_start;
    if (n - m + 1 < sl) sl = n - m + 1;  // :491
    if (sl < sf) return none;            // :493-495
    if (trim_side == 3) {                // :499-515: only the first findprev hit is examined
        for (int s = sl; s >= 1; --s) {
            if (bytes_equal<STAGED>(q, m, r, s)) {
                if (s >= sf && s + m - 1 >= min_end) return AlignOut{0, s, s + m - 1};
                return none;
            }
        }
        return none;
    }
    // :517-547: leftmost hit; hits that end before min_end_pos are skipped and the scan goes on
    for (int s = sf; s <= n - m + 1; ++s) {
        if (bytes_equal<STAGED>(q, m, r, s)) {
            if (s > sl) return none;
            if (s + m - 1 >= min_end) return AlignOut{0, s, s + m - 1};
        }
    }
    return none;
}

// Return tuple of find_best_matching_bc (classification.jl:722) plus match_barcode_pass's status.
struct PassOut {
    int status;  // 1 match, 0 unknown, -1 ambiguous, 2 pass not run
    int bc;      // min_score_bc (kept even when ambiguous)
    int start, end, raw;
    double score;
    double delta;
};

// match_barcode_pass (classification.jl:776-868, minus the histogram block :827-865) with the
// reducers find_best_matching_bc_no_delta (:632-667) / _with_delta (:669-713) inlined.
template <bool STAGED>
__device__ __forceinline__ PassOut run_pass(const BdxDevCfg &cfg, const BdxDevPass &P, const Bytes<STAGED> bcb,
                                            const LDS uint32_t *bc_off, const LDS int *bc_nn,
                                            const Bytes<STAGED> r, const int n, LDS int *DP, LDS int *OG,
                                            const int S, const uint32_t *cand) {
    PassOut po{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    long long first, last, max_start_ll, min_end_ll;
    if (P.explicit_window) {
        first = P.win_first;
        last = P.win_last;
        max_start_ll = P.win_max_start;
        min_end_ll = P.win_min_end;
    } else {
        long long rs_f, rs_l, bs_f, bs_l, be_f, be_l;  // :795-797
        resolve_range(P.ref_search, n, rs_f, rs_l);
        resolve_range(P.bc_start, n, bs_f, bs_l);
        resolve_range(P.bc_end, n, be_f, be_l);
        first = rs_f > bs_f ? rs_f : bs_f;  // :799
        if (first < 1) first = 1;
        last = rs_l < be_l ? rs_l : be_l;  // :800
        if (n < last) last = n;
        max_start_ll = bs_l;  // :801
        min_end_ll = be_f;    // :802
        if (first > last || first > max_start_ll || last < min_end_ll) return po;  // :805-807
    }
    // After the sanity check every bound is within [-(2^30), 2^30]; clamp so int32 arithmetic
    // in the cores cannot overflow for hand-made explicit windows.
    const long long LIM = 1LL << 30;
    auto clampi = [&](long long v) -> int { return (int)(v > LIM ? LIM : (v < -LIM ? -LIM : v)); };
    const int jf = clampi(first), jl = clampi(last), max_start = clampi(max_start_ll), min_end = clampi(min_end_ll);

    const int trim_side = P.trim_side;
    const bool need_tb = (trim_side != 0) || cfg.need_traceback;  // :812
    const Costs c{cfg.match, cfg.mismatch, cfg.indel, cfg.nindel};
    const bool with_delta = !(cfg.min_delta == 0.0);  // :723

    double rate = cfg.max_error_rate;
    double min_score = __builtin_inf(), sub_min = __builtin_inf();
    int best = 0, bs = -1, be = -1, braw = -1;

    const bool align_one = P.explicit_window == BDX_WINDOW_ALIGN_ONE;
    const int B = align_one ? 1 : P.n_barcodes;
    for (int b = 0; b < B; ++b) {  // :638 / :676 — file order, threshold tightens as we go
        if (cand && !((cand[b >> 5] >> (b & 31)) & 1u)) continue;
        const int o = (int)bc_off[b];
        const int m = (int)bc_off[b + 1] - o;
        const Bytes<STAGED> q = bcb.at(o);
        AlignOut a;
        double score;
        if (cfg.algorithm == BDX_ALG_HAMMING) {
            const int allowed = (int)__builtin_floor(rate * (double)m);  // :567
            a = hamming_dev<STAGED>(q, m, r, n, allowed, jf, jl, max_start, min_end, trim_side);
            score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)m;  // :607
        } else if (cfg.algorithm == BDX_ALG_EXACT) {
            a = exact_dev<STAGED>(q, m, r, n, jf, jl, max_start, min_end, trim_side);
            score = a.raw >= BDX_INF32 ? __builtin_inf() : 0.0;
        } else {
            const int norm = cfg.has_nindel ? bc_nn[b] : m;               // :460 / :476
            const int ae = (int)__builtin_floor(rate * (double)norm);     // :254
            if (cfg.has_nindel) {
                a = need_tb ? sg_core<true, true, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end)
                            : sg_core<false, true, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end);
            } else {
                a = need_tb ? sg_core<true, false, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end)
                            : sg_core<false, false, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end);
            }
            score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)norm;  // :155-168
        }
        if (align_one) {  // unit-level API: the direct return of one alignment call
            if (a.raw < BDX_INF32) {
                po.status = 1;
                po.bc = 1;
                po.start = a.start;
                po.end = a.end;
                po.raw = a.raw;
                po.score = score;
            }
            return po;
        }
        if (!with_delta) {  // :658-664
            if (score <= rate && score < min_score) {
                min_score = score;
                best = b + 1;
                rate = rate < min_score ? rate : min_score;
                bs = a.start;
                be = a.end;
                braw = a.raw;
            }
        } else if (score <= rate) {  // :696-708
            if (score < min_score) {
                sub_min = min_score;
                min_score = score;
                best = b + 1;
                rate = rate < sub_min ? rate : sub_min;
                bs = a.start;
                be = a.end;
                braw = a.raw;
            } else if (score < sub_min) {
                sub_min = score;
                rate = rate < sub_min ? rate : sub_min;
            }
        }
    }
    const double delta = with_delta ? (sub_min - min_score) : __builtin_inf();  // :711 / :666
    po.bc = best;
    po.start = bs;
    po.end = be;
    po.raw = braw;
    po.score = min_score;
    po.delta = delta;
    if (best == 0) return po;  // :820-821
    po.status = (delta < cfg.min_delta) ? -1 : 1;  // :822-823, :867
    return po;
}

struct Verdict {
    int bc1, bc2, keep_start, keep_end;
};

template <bool STAGED>
__device__ __forceinline__ void classify_one(const BdxDevCfg &cfg, const Bytes<STAGED> bcb0,
                                             const Bytes<STAGED> bcb1, const LDS uint32_t *off0,
                                             const LDS uint32_t *off1, const LDS int *nn0, const LDS int *nn1,
                                             const Bytes<STAGED> r, const int n, LDS int *DP, LDS int *OG,
                                             const int S, const uint32_t *cand0, const uint32_t *cand1,
                                             Verdict &v, PassOut &p1, PassOut &p2) {
    // determine_filename, classification.jl:871-938
    v = Verdict{0, 0, -1, -1};
    p2 = PassOut{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    p1 = run_pass<STAGED>(cfg, cfg.pass[0], bcb0, off0, nn0, r, n, DP, OG, S, cand0);  // :875
    if (p1.status != 1) {  // :879-883
        v.bc1 = p1.status;
        return;
    }
    if (cfg.is_dual) {  // :887-895
        p2 = run_pass<STAGED>(cfg, cfg.pass[1], bcb1, off1, nn1, r, n, DP, OG, S, cand1);
        if (p2.status != 1) {
            v.bc1 = p2.status;
            return;
        }
        v.bc2 = p2.bc;
    }
    v.bc1 = p1.bc;
    int keep_start = 1, keep_end = n;  // :907-908
    const int ts1 = cfg.pass[0].trim_side, ts2 = cfg.pass[1].trim_side;
    if (ts1 == 3)  // :910-919
        keep_end = (p1.start > 1 ? p1.start : 1) - 1;
    else if (ts1 == 5)
        keep_start = p1.end + 1;
    if (cfg.is_dual) {  // :921-929
        if (ts2 == 3) {
            const int e2 = (p2.start > 1 ? p2.start : 1) - 1;
            keep_end = keep_end < e2 ? keep_end : e2;
        } else if (ts2 == 5) {
            const int s2 = p2.end + 1;
            keep_start = keep_start > s2 ? keep_start : s2;
        }
    }
    if (keep_start > keep_end) {  // :932-935
        v.keep_start = 1;
        v.keep_end = 0;
    } else {
        v.keep_start = keep_start;
        v.keep_end = keep_end;
    }
}

struct GenericArgs {
    BdxDevCfg cfg;
    const uint8_t *seq;
    const long long *off;
    long long n_reads;
    BdxDevOut out;
    unsigned long long *counts;
    const uint32_t *cand0;  // [n_reads][cand_words] or null
    const uint32_t *cand1;
    int dp_rows;
    int stage_bytes;    // capacity of the read staging area (0: never stage)
    int bc_stage_bytes; // bytes of the barcode staging area (both passes; 0: barcodes not staged)
    int hist_entries;
};

// LDS carve-up (all 16-byte aligned):
//   [DP: dp_rows*BS int][OG: dp_rows*BS int if any_traceback][off0|off1: uint32][nn0|nn1: int]
//   [barcode bytes pass0|pass1][hist: int[hist_entries]][read bytes: stage_bytes]
template <int BS>
__global__ __launch_bounds__(BS) void bdx_generic_kernel(const GenericArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LDS unsigned char *smem = (LDS unsigned char *)smem_raw;
    const BdxDevCfg &cfg = a.cfg;
    const int tid = threadIdx.x;
    const int B0 = cfg.pass[0].n_barcodes;
    const int B1 = cfg.is_dual ? cfg.pass[1].n_barcodes : 0;

    size_t o = 0;
    LDS int *DPbase = (LDS int *)(smem + o);
    o += (size_t)a.dp_rows * BS * 4;
    LDS int *OGbase = (LDS int *)(smem + o);
    if (cfg.any_traceback) o += (size_t)a.dp_rows * BS * 4;
    LDS uint32_t *off0 = (LDS uint32_t *)(smem + o);
    o += (size_t)(B0 + 1) * 4;
    LDS uint32_t *off1 = (LDS uint32_t *)(smem + o);
    o += (size_t)(B1 + 1) * 4;
    LDS int *nn0 = (LDS int *)(smem + o);
    o += (size_t)B0 * 4;
    LDS int *nn1 = (LDS int *)(smem + o);
    o += (size_t)B1 * 4;
    o = (o + 15) & ~(size_t)15;
    LDS unsigned char *bcs = smem + o;
    o += (size_t)a.bc_stage_bytes;
    o = (o + 15) & ~(size_t)15;
    LDS int *hist = (LDS int *)(smem + o);
    o += (size_t)a.hist_entries * 4;
    o = (o + 15) & ~(size_t)15;
    LDS unsigned char *rstage = smem + o;

    // barcode tables -> LDS
    for (int i = tid; i <= B0; i += BS) off0[i] = cfg.pass[0].bc_off[i];
    for (int i = tid; i < B0; i += BS) nn0[i] = cfg.pass[0].bc_len_no_N[i];
    if (cfg.is_dual) {
        for (int i = tid; i <= B1; i += BS) off1[i] = cfg.pass[1].bc_off[i];
        for (int i = tid; i < B1; i += BS) nn1[i] = cfg.pass[1].bc_len_no_N[i];
    }
    for (int i = tid; i < a.hist_entries; i += BS) hist[i] = 0;
    __syncthreads();
    const int bytes0 = (int)off0[B0];
    const int bytes1 = cfg.is_dual ? (int)off1[B1] : 0;
    const bool bc_staged = a.bc_stage_bytes > 0;
    if (bc_staged) {
        for (int i = tid; i < bytes0; i += BS) bcs[i] = cfg.pass[0].bc_bytes[i];
        for (int i = tid; i < bytes1; i += BS) bcs[bytes0 + i] = cfg.pass[1].bc_bytes[i];
    }

    // this workgroup's reads: [r0, r1)
    const long long r0 = (long long)blockIdx.x * BS;
    long long r1 = r0 + BS;
    if (r1 > a.n_reads) r1 = a.n_reads;
    const long long span0 = a.off[r0];
    const long long span1 = a.off[r1];
    const uintptr_t g0 = (uintptr_t)(a.seq + span0);
    const uintptr_t g0a = g0 & ~(uintptr_t)15;
    const int head = (int)(g0 - g0a);
    const long long need = (span1 - span0) + head;
    const bool staged = bc_staged && a.stage_bytes > 0 && need + 16 <= (long long)a.stage_bytes;
    if (staged) {  // coalesced 16-B copy of the contiguous span, each HBM byte fetched once
        const int nvec = (int)((need + 15) >> 4);
        const u32x4 *src = (const u32x4 *)g0a;
        LDS u32x4 *dst = (LDS u32x4 *)rstage;
        for (int k = tid; k < nvec; k += BS) dst[k] = __builtin_nontemporal_load(src + k);
    }
    __syncthreads();

    const long long ridx = r0 + tid;
    const bool active = ridx < r1;
    Verdict v{0, 0, -1, -1};
    PassOut p1{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()}, p2{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    if (active) {
        const long long ro = a.off[ridx];
        const long long rn = a.off[ridx + 1] - ro;
        const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
        const uint32_t *c0 = a.cand0 ? a.cand0 + ridx * cfg.pass[0].cand_words : nullptr;
        const uint32_t *c1 = a.cand1 ? a.cand1 + ridx * cfg.pass[1].cand_words : nullptr;
        LDS int *DP = DPbase + tid;
        LDS int *OG = OGbase + tid;
        if (staged) {
            Bytes<true> r{rstage + head + (ro - span0)};
            Bytes<true> q0{bcs}, q1{bcs + bytes0};
            classify_one<true>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, BS, c0, c1, v, p1, p2);
        } else {
            Bytes<false> r{a.seq + ro};
            Bytes<false> q0{cfg.pass[0].bc_bytes}, q1{cfg.pass[1].bc_bytes};
            classify_one<false>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, BS, c0, c1, v, p1, p2);
        }
        // outputs (coalesced: consecutive lanes -> consecutive reads)
        if (a.out.bc1) a.out.bc1[ridx] = v.bc1;
        if (a.out.bc2) a.out.bc2[ridx] = v.bc2;
        if (a.out.keep_start) a.out.keep_start[ridx] = v.keep_start;
        if (a.out.keep_end) a.out.keep_end[ridx] = v.keep_end;
        if (a.out.pass_start) {
            a.out.pass_start[2 * ridx] = p1.start;
            a.out.pass_start[2 * ridx + 1] = p2.start;
        }
        if (a.out.pass_end) {
            a.out.pass_end[2 * ridx] = p1.end;
            a.out.pass_end[2 * ridx + 1] = p2.end;
        }
        if (a.out.pass_raw) {
            a.out.pass_raw[2 * ridx] = p1.raw;
            a.out.pass_raw[2 * ridx + 1] = p2.raw;
        }
        if (a.out.pass_bc) {
            a.out.pass_bc[2 * ridx] = p1.bc;
            a.out.pass_bc[2 * ridx + 1] = p2.bc;
        }
        if (a.out.pass_score) {
            a.out.pass_score[2 * ridx] = p1.score;
            a.out.pass_score[2 * ridx + 1] = p2.score;
        }
        if (a.out.pass_delta) {
            a.out.pass_delta[2 * ridx] = p1.delta;
            a.out.pass_delta[2 * ridx + 1] = p2.delta;
        }
    }

    // DemuxStats scalar counters (classification.jl:942-978), merged like reporting.jl:1-9
    if (a.counts) {
        int slot = -1;
        if (active) {
            if (v.bc1 > 0)
                slot = 4 + (v.bc1 - 1) * cfg.counts_stride2 + (v.bc2 > 0 ? v.bc2 - 1 : 0);
            const int cls = v.bc1 > 0 ? 1 : (v.bc1 == 0 ? 2 : 3);
            if (a.hist_entries > 0) {
                __hip_atomic_fetch_add(&hist[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&hist[cls], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (slot >= 0) __hip_atomic_fetch_add(&hist[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                atomicAdd(&a.counts[0], 1ULL);
                atomicAdd(&a.counts[cls], 1ULL);
                if (slot >= 0) atomicAdd(&a.counts[slot], 1ULL);
            }
        }
        if (a.hist_entries > 0) {
            __syncthreads();
            for (int i = tid; i < a.hist_entries; i += BS) {
                const int h = hist[i];
                if (h) atomicAdd(&a.counts[i], (unsigned long long)h);
            }
        }
    }
}

}  // namespace

hipError_t bdx_generic_set_lds_limit(size_t bytes) {
    hipError_t e;
    e = hipFuncSetAttribute((const void *)bdx_generic_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bdx_generic_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)bdx_generic_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e;
}

hipError_t bdx_launch_generic(const BdxDevCfg &cfg, const BdxGenericPlan &plan, const uint8_t *d_seq,
                              const long long *d_off, long long n_reads, const BdxDevOut &out,
                              unsigned long long *d_counts, const uint32_t *d_cand0, const uint32_t *d_cand1,
                              hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    GenericArgs a;
    a.cfg = cfg;
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.out = out;
    a.counts = d_counts;
    a.cand0 = d_cand0;
    a.cand1 = d_cand1;
    a.dp_rows = plan.dp_rows;
    a.stage_bytes = plan.stage_bytes;
    a.bc_stage_bytes = plan.bc_stage_bytes;
    a.hist_entries = plan.hist_entries;
    const long long blocks = (n_reads + plan.threads - 1) / plan.threads;
    if (blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block((unsigned)plan.threads);
    switch (plan.threads) {
        case 256:
            hipLaunchKernelGGL(bdx_generic_kernel<256>, grid, block, plan.lds_bytes, stream, a);
            break;
        case 128:
            hipLaunchKernelGGL(bdx_generic_kernel<128>, grid, block, plan.lds_bytes, stream, a);
            break;
        case 64:
            hipLaunchKernelGGL(bdx_generic_kernel<64>, grid, block, plan.lds_bytes, stream, a);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
