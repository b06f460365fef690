// bdx_core.h — device-side exact evaluation of one read (shared by the generic kernel and the
// fused filter+verify kernels).  See bdx_device.hip for the design notes.
#pragma once
#include "bdx_internal.h"

#define LDS __attribute__((address_space(3)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// pointers rebuilt from integer arithmetic lose their address space and would compile to flat_load (which
// also counts on lgkmcnt and so stalls every LDS wait): name the global address space explicitly
typedef const u32x4 __attribute__((address_space(1))) *GlobalVec16;

namespace {

struct Costs {
    int match, mismatch, indel, nindel;
    int ncode;  // value that stands for barcode 'N': 0x4E on raw bytes, its symbol code on transcoded data
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
                                            const Costs c, const int trim_side, int first, int last,
                                            const int max_start, const int min_end,
                                            const int jlo = -0x40000000, const int jhi = 0x40000000) {
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
    // restricted run: columns outside jlo..jhi can neither record anything nor
    // influence a cell <= allowed_error inside — the loop bounds are the only thing that changes
    if (jlo > first) first = jlo;
    if (jhi < last) last = jhi;
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
            const bool isN = NS && (qi == c.ncode);
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

// Register-resident variant of sg_core for barcodes of at most M rows: the DP column, the
// origin column and the barcode itself live in VGPRs (statically indexed, rows fully unrolled);
// rows outside fact..lact are predicated off and keep their (stale) values, exactly like the
// array of the reference.  Semantics are identical to sg_core above — same seeds, same stale
// cells, same tie-breaks, same early exits — only the storage differs:
//   * the reference writes row i-1 while visiting row i (:326-331, :364-367) and row lact after the
//     loop (:412-415); net effect per column: rows fact..lact receive their new values, row
//     fact-1 (if any) receives the seed (allowed_error, origin j), all other rows are untouched.
//     Here row i is written in its own iteration and the old value is carried to row i+1 as the
//     diagonal operand, which is the same data flow without the one-row shift.
// Why: lanes of a wave sit in their barcode's match region at different columns, so the wave
// runs ~m rows in every column anyway; without LDS round-trips in the dependent chain the
// statically unrolled form is several times faster.
template <bool TB, bool NS, int M, bool STAGED, bool ENDPOS = false>
__device__ __forceinline__ AlignOut sg_core_reg(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                                const int n, const int ae, const Costs c, const int trim_side,
                                                int first, int last, const int max_start,
                                                const int min_end, const int jlo = -0x40000000,
                                                const int jhi = 0x40000000) {
    AlignOut res{BDX_INF32, -1, -1};
    if (m == 0 || n == 0) return res;  // :250-252

    const int steps = ae / (NS ? (c.indel < c.nindel ? c.indel : c.nindel) : c.indel);  // :257
    const int min_valid_start = min_end - (m + steps) + 1;                            // :259
    if (min_valid_start > max_start) return res;                                      // :261-263
    if (min_valid_start > first) first = min_valid_start;                             // :266-268
    const int b1 = m - n - steps, b2 = -max_start - steps;
    const int band = b1 > b2 ? b1 : b2;  // :270

    int DP[M + 1], OG[M + 1];
    uint32_t QP[(M + 3) / 4];  // barcode code units, four per register (rows beyond m: 0xFF)
#pragma unroll
    for (int i = 1; i <= M; ++i) {  // :278-283 (rows beyond m exist only as dead registers)
        DP[i] = c.indel * i;
        OG[i] = 1 - i;
    }
#pragma unroll
    for (int w = 0; w < (M + 3) / 4; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v |= (uint32_t)((4 * w + k < m) ? q[4 * w + k] : 0xFF) << (8 * k);
        QP[w] = v;
    }

    int lact = (ae + 1 < m) ? ae + 1 : m;  // :286
    if (jlo > first) first = jlo;  // restricted run, DESIGN.md §3.2
    if (jhi < last) last = jhi;
    // The lanes of a wave walk their columns END-ALIGNED: all lanes reach their last column in the same
    // trip (lanes with fewer columns start later).  Each lane still visits first..last in order, so its
    // result is untouched; but a restricted run ends at the alignment's end column, so the
    // lanes now sit on (nearly) the same rows at the same time and the row blocks below can be skipped
    // for the whole wave.  Wave maximum of the column counts by bisection on ballots (safe under divergence).
    int T = 0;
    {
        const int len = last - first + 1;
        for (int bit = 30; bit >= 0; --bit)
            if (__builtin_amdgcn_ballot_w64(len >= (T | (1 << bit)))) T |= 1 << bit;
    }
    for (int t = 0; t < T; ++t) {
        const int j = last - (T - 1 - t);  // :287
        if (j < first) continue;
        int prev_o = j;  // :288
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
        const int seed = prev;
        int diag = 0;    // row-0 value (:215 "i == 1 ? 0")
        int diag_o = j;  // row-0 origin (:308)
        // The reference visits rows fact..lact only (:300); here the rows are static registers, so they are
        // walked in blocks of RB and a block is executed only if SOME lane of the wave has a row of
        // [fact-1, lact] in it (fact-1: the band seed).  For the other lanes the block's code is a no-op
        // (DP, OG, prev unchanged; diag leaves the block as the old value of its last row) — exactly what
        // the skip branch does.  In restricted runs all lanes of a wave sit at about the
        // same distance from their window start, so most blocks are skipped.
        constexpr int RB = 4;
#pragma unroll
        for (int bs = 1; bs <= M; bs += RB) {
            const int be = bs + RB - 1 < M ? bs + RB - 1 : M;
            const bool lane_on = (bs <= lact) && (be >= fact - 1);
            if (__builtin_amdgcn_ballot_w64(lane_on)) {
#pragma unroll
                for (int i = bs; i <= be; ++i) {
                    const int old = DP[i];
                    const int old_o = OG[i];
                    const bool inb = (i >= fact) && (i <= lact);
                    const int qi = (int)((QP[(i - 1) >> 2] >> (8 * ((i - 1) & 3))) & 0xFFu);
                    const bool isN = NS && (qi == c.ncode);
                    const int cost = isN ? c.nindel : c.indel;
                    const int ins = (i == m) ? BDX_INF32 : old + cost;
                    const int del = prev + cost;
                    const int sub = diag + ((qi == rj || isN) ? c.match : c.mismatch);
                    int cur_o = prev_o;
                    if (TB) {  // :310-321
                        int best = del;
                        if (sub < best) {
                            best = sub;
                            cur_o = diag_o;
                        }
                        if (ins < best) cur_o = old_o;
                    }
                    const int t = del < sub ? del : sub;
                    const int nv = ins < t ? ins : t;
                    const bool seedrow = (i == fact - 1);  // :326-331 (fact != 1 is implied by i >= 1)
                    DP[i] = inb ? nv : (seedrow ? seed : old);
                    if (TB) OG[i] = inb ? cur_o : (seedrow ? j : old_o);
                    prev = inb ? nv : prev;
                    if (TB) prev_o = inb ? cur_o : prev_o;
                    diag = old;
                    diag_o = old_o;
                }
            } else {
                diag = DP[be];
                diag_o = OG[be];
            }
        }

        if (lact == m && prev <= ae) {  // :417
            lact -= 1;
            if (j >= min_end) {
                if (prev == 0 && (!TB || trim_side == 5)) {  // :420-430
                    AlignOut z{0, TB ? prev_o : -1, (TB || ENDPOS) ? j : -1};
                    return z;
                }
                if (TB) {  // :142-153
                    if (prev < res.raw || (prev == res.raw && trim_side == 3 && prev_o > res.start)) {
                        res.raw = prev;
                        res.start = prev_o;
                        res.end = j;
                    }
                } else if (prev < res.raw) {
                    res.raw = prev;
                    if (ENDPOS) res.end = j;  // trim_side == 5 needs the end column only (:910-919): no origins
                }
            }
        }
        // :439-442  while lact > 0 && DP[lact] > allowed_error: lact -= 1;  lact += 1
#pragma unroll
        for (int i = M; i >= 1; --i)
            if (lact == i && DP[i] > ae) lact = i - 1;
        ++lact;
    }
    return res;  // :444
}

// "Clean class" variant of the register DP: SimpleScoring with match >= 0 and mismatch, indel
// >= 1, and start / end ranges that bind for no read (the host checks both: BdxDevCfg::clean_ok).  There the
// reference's banded cut-off loop records exactly what a plain column-by-column DP over all m rows records —
// cells <= allowed_error only depend on cells <= allowed_error, which lie inside fact..lact with their true values;
// stale cells and cells outside the band are > allowed_error on both sides and never win a comparison — so no
// fact / lact / band bookkeeping and no per-row predicates are needed: 11 VALU operations per cell with origins,
// 6 without, against ~27 of the predicated form, and no mask registers to spill.  Same origin rule (deletion, then
// substitution if strictly less, then insertion if strictly less, :310-321), same last row without a horizontal
// move (:213/:229: its recorded value is min(del, sub)), same recording and early exit (:417-436).
// Checked against the line-faithful core by orc_selftest_clean_class (oracle; 0 disagreements in 4 M cases).
// UM: every barcode of the config has exactly M rows (the last row is static); else row m is picked per lane.
template <bool TB, int M, bool STAGED, bool ENDPOS, bool UM>
__device__ __forceinline__ AlignOut sg_core_clean(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                                  const int n, const int ae, const Costs c, const int trim_side,
                                                  int first, int last, const int jlo = -0x40000000,
                                                  const int jhi = 0x40000000) {
    AlignOut res{BDX_INF32, -1, -1};
    if (m == 0 || n == 0) return res;  // :250-252
    int DP[M + 1], OG[M + 1], QB[M + 1];
#pragma unroll
    for (int i = 1; i <= M; ++i) {  // :278-283
        DP[i] = c.indel * i;
        OG[i] = 1 - i;
        QB[i] = (UM || i <= m) ? q[i - 1] : 0x100;  // rows beyond m never match and are never looked at
    }
    if (jlo > first) first = jlo;  // restricted run, DESIGN.md §3.2
    if (jhi < last) last = jhi;
    // the read's bytes arrive a 32-bit word at a time (aligned; one load per four columns)
    const uintptr_t rbase = (uintptr_t)r.p;
    uint32_t word = 0;
    int wpos = -1;
    for (int j = first; j <= last; ++j) {  // :287
        const uintptr_t ad = rbase + (uintptr_t)(j - 1);
        const int wi = (int)(ad >> 2);
        if (wi != wpos) {
            wpos = wi;
            if constexpr (STAGED)
                word = *(const LDS uint32_t *)(ad & ~(uintptr_t)3);
            else
                word = *(const uint32_t *)(ad & ~(uintptr_t)3);
        }
        const int rj = (int)((word >> (8 * (int)(ad & 3))) & 0xFFu);
        int prev = 0, prev_o = j, diag = 0, diag_o = j;  // row 0: value 0, origin j (:288, :308)
        int vm = BDX_INF32, om = -1;
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            const int old = DP[i], old_o = OG[i];
            const int ins = old + c.indel;                                    // :183
            const int del = prev + c.indel;                                   // :184
            const int sub = diag + (QB[i] == rj ? c.match : c.mismatch);      // :185
            int o = prev_o;
            if (TB) o = sub < del ? diag_o : o;                               // :310-316
            const int b2 = sub < del ? sub : del;
            if (UM ? (i == M) : (i == m)) {  // the last row: no horizontal move
                vm = b2;
                if (TB) om = o;
            }
            if (TB) o = ins < b2 ? old_o : o;                                 // :317-320
            const int nv = ins < b2 ? ins : b2;                               // :332
            DP[i] = nv;
            if (TB) OG[i] = o;
            prev = nv;
            prev_o = o;
            diag = old;
            diag_o = old_o;
            // rows are a serial chain; keep the scheduler from hoisting every row's independent operations (ins, the
            // comparisons with rj) to the column head: that triples the live registers and spills
            if ((i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        if (vm <= ae) {  // :417 (j >= min_end_pos holds for every column: the end range does not bind)
            if (vm == 0 && (!TB || trim_side == 5)) return AlignOut{0, TB ? om : -1, (TB || ENDPOS) ? j : -1};  // :420-430
            if (TB) {  // :142-153
                if (vm < res.raw || (vm == res.raw && trim_side == 3 && om > res.start)) {
                    res.raw = vm;
                    res.start = om;
                    res.end = j;
                }
            } else if (vm < res.raw) {
                res.raw = vm;
                if (ENDPOS) res.end = j;
            }
        }
    }
    return res;  // :444
}

// Diagonal-band form of the clean-class DP, for candidates whose end columns are known: the fused
// kernel's tracked sweep delivers [e_lo, e_hi] = first / last column with unit distance <= kb.  An alignment of <= kb
// operations ending in row M at a column je of that window only touches cells with |(je - j) - (M - i)| <= kb, i.e.
// the diagonals  j - i  in  [e_lo - M - kb, e_hi - M + kb]; everything the reference records (values, origins, ties:
// every predecessor that attains a minimum on a recorded path lies in the band with its true value, the others only
// grow) is reproduced by a DP over the H >= (e_hi - e_lo + 1) + 2 kb diagonals below  dtop = e_hi + kb - M.
// Columns are walked relative to the lane's own anchor — step k is column  e_hi + kb - (M + H - 2) + k  — so the rows
// of a step are the same for every lane (k - H + 2 .. k + 1, clipped to 1 .. M): the whole DP is straight-line code
// over statically indexed registers, M x H cells instead of (columns x M), no per-row predicates, no infinities
// (cells outside the band are simply not read).  When the band crosses the first column of the pass window the rows it
// held one step earlier take the reference's initial column (indel * i, origin 1 - i, :278-283); steps before that
// compute garbage that nothing reads.  Every barcode has exactly M rows (UM kernels).
// Model and proof by enumeration: oracle band_dp / orc_selftest_band_class.
// One out-of-line copy per (TB, M, H) and kernel — the body is long straight-line code, the call sites are many
// (two passes x staged / unstaged reads): the barcode (4 bytes per word) and the read words travel in vector registers.
typedef uint32_t bdx_u32x4 __attribute__((ext_vector_type(4)));
template <bool TB, int M, int H>
__device__ __attribute__((noinline)) AlignOut sg_core_band(const bdx_u32x4 qa, const bdx_u32x4 qb, const bdx_u32x4 sa,
                                                           const bdx_u32x4 sb, const bdx_u32x4 sc, const bdx_u32x4 sd,
                                                           const int j0, const int ae, const uint32_t costs /* match | mismatch << 8 | indel << 16 | trim_side << 24 */,
                                                           const int first, const int e_lo, const int e_hi) {
    static_assert(M <= 32 && (M + H + 2) / 4 + 1 <= 16, "barcode / read words do not fit the register arguments");
    constexpr int K = M + H - 2;
    const uint32_t QW[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const uint32_t S[16] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w, sc.x, sc.y, sc.z, sc.w, sd.x, sd.y, sd.z, sd.w};
    struct { int match, mismatch, indel; } c{(int)(costs & 255u), (int)((costs >> 8) & 255u), (int)((costs >> 16) & 255u)};
    const int trim_side = (int)(costs >> 24);
    int QB[M + 1];
    QB[0] = 0;
#pragma unroll
    for (int i = 1; i <= M; ++i) QB[i] = (int)((QW[(i - 1) >> 2] >> (8 * ((i - 1) & 3))) & 0xFFu);
    AlignOut res{BDX_INF32, -1, -1};
    int DP[M + 1], OG[M + 1];
#pragma unroll
    for (int i = 0; i <= M; ++i) {
        DP[i] = 0;
        OG[i] = 0;
    }
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        const int j = j0 + k;
        const int ra = k - H + 2, rb = k + 1;
        const int lo = ra < 1 ? 1 : ra, hi = rb > M ? M : rb;
        if (j == first) {
#pragma unroll
            for (int i = ra - 1; i <= rb - 1; ++i)
                if (i >= 1 && i <= M) {
                    DP[i] = c.indel * i;
                    if (TB) OG[i] = 1 - i;
                }
        }
        const int rj = (int)((S[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        int prev = 0, prev_o = j, diag = 0, diag_o = j;  // row 0: value 0, origin j (:288, :308)
        if (lo > 1) {
            diag = DP[lo - 1];  // (lo - 1, j - 1): the band's top diagonal
            diag_o = OG[lo - 1];
        }
        int vm = BDX_INF32, om = -1;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            const int old = DP[i], old_o = OG[i];
            const int sub = diag + (QB[i] == rj ? c.match : c.mismatch);  // :185
            int b2 = sub, o = diag_o;
            if (i > lo || lo == 1) {  // the deletion input lies in the band (or is row 0)
                const int del = prev + c.indel;  // :184
                if (TB) o = sub < del ? diag_o : prev_o;  // :310-316
                b2 = sub < del ? sub : del;
            }
            if (i == M) {  // the last row: no horizontal move; its stored value is never read
                vm = b2;
                om = o;
            } else {
                int nv = b2;
                if (i != rb) {  // (i, j - 1) lies in the band
                    const int ins = old + c.indel;  // :183
                    if (TB) o = ins < b2 ? old_o : o;  // :317-320
                    nv = ins < b2 ? ins : b2;
                }
                DP[i] = nv;
                if (TB) OG[i] = o;
                prev = nv;
                prev_o = o;
            }
            diag = old;
            diag_o = old_o;
        }
        if (hi == M) {
            if (j >= e_lo && j <= e_hi && vm <= ae) {  // :417 (j >= min_end_pos: the end range does not bind)
                if (vm == 0 && (!TB || trim_side == 5)) return AlignOut{0, TB ? om : -1, j};  // :420-430
                if (TB) {  // :142-153
                    if (vm < res.raw || (vm == res.raw && trim_side == 3 && om > res.start)) {
                        res.raw = vm;
                        res.start = om;
                        res.end = j;
                    }
                } else if (vm < res.raw) {
                    res.raw = vm;
                    res.end = j;
                }
            }
        }
    }
    return res;
}

// Rolling form of the diagonal band for barcodes of ANY length: the same cells, the same recurrence and
// the same recording rule as sg_core_band — alignments of at most kbb operations that end in row m at a column of
// [c_lo, c_hi] only touch the H = (c_hi - c_lo + 1) + 2 kbb diagonals  j - i  in  [c_lo - m - kbb, c_hi - m + kbb]  —
// walked column by column like the reference (:287) over a ROLLING window of rows: at column j the band holds the H rows
// j - dtop .. j - d0, row i lives in slot i mod H of V / O (the row that enters the band at a column takes the slot of the
// row that left it), so a lane needs H cells however long the barcode is — against the m + 1 rows per lane of sg_core,
// which leave a 128-lane workgroup per CU for barcodes of 80 nt.  One read byte per column (fetched a column ahead), the
// barcode's bytes from the staged table.  Cells outside the band are not read; row 0 is 0 with origin j (:288, :308); when
// the walk starts at the first column of the pass window the rows the band held one column earlier take the reference's
// initial column (indel * i, origin 1 - i, :278-283); columns beyond c_hi are not computed (nothing a recording at or before
// c_hi depends on lies to its right).  The last row takes no horizontal move and is recorded in column order (:142-153,
// early exit :420-430).  Several calls with consecutive column ranges fold to the whole range (run_pass; enumerated for
// random chunkings by the oracle's orc_selftest_band_class, model band_dp_roll).
// (First version, measured: the same band ROW by row — one read byte per CELL, fetched from L2 when the workgroup's reads do
// not fit its staging area: 4.4 ms instead of the exact kernel's 29 ms for 500 k reads x 48 barcodes of 80 nt, but ~450
// cycles per cell.  Also measured: value and origin of a cell as ONE 64-bit LDS word — no gain, the LDS instruction count is not
// what bounds the walk.)
template <bool TB, bool STAGED>
__device__ __forceinline__ AlignOut sg_band_roll(LDS int *V, LDS int *O, const int S, const Bytes<STAGED> q, const int m,
                                                 const Bytes<STAGED> r, const int ae, const Costs c, const int trim_side,
                                                 const int first, const int c_lo, const int c_hi, const int kbb) {
    AlignOut res{BDX_INF32, -1, -1};
    const int H = (c_hi - c_lo + 1) + 2 * kbb;
    const int d0 = c_lo - m - kbb, dtop = d0 + H - 1;  // lowest / highest diagonal (j - i) of the band
    int j = d0 + 1 > first ? d0 + 1 : first;           // the first column at which the band holds a row >= 1 inside the window
    if (j > c_hi) return res;
    if (j == first) {  // the reference's initial column on the rows the band held one column earlier
        int ia = first - 1 - dtop, ib = first - 1 - d0;
        ia = ia < 1 ? 1 : ia;
        ib = ib > m ? m : ib;
        if (ia <= ib) {
            int s = ia % H;
            for (int i = ia; i <= ib; ++i) {
                V[s * S] = c.indel * i;
                if (TB) O[s * S] = 1 - i;
                s = s + 1 == H ? 0 : s + 1;
            }
        }
    }
    int lo = j - dtop;
    lo = lo < 1 ? 1 : lo;
    int s_lo = lo % H;  // slot of row lo (kept up to date as lo moves down the barcode)
    int rj = r[j - 1];
    for (; j <= c_hi; ++j) {
        const int rj_next = j < c_hi ? r[j] : 0;
        const int rb = j - d0;  // the row that enters the band at this column (its left neighbour lies outside)
        const int hi = rb > m ? m : rb;
        int prev = 0, prev_o = j, diag = 0, diag_o = j;  // row 0: value 0, origin j (:288, :308)
        if (lo > 1) {  // (lo - 1, j - 1): the band's top diagonal — its slot is overwritten by the entering row at the END of this column
            const int sd = s_lo == 0 ? H - 1 : s_lo - 1;
            diag = V[sd * S];
            if (TB) diag_o = O[sd * S];
        }
        int s = s_lo;
        int vm = BDX_INF32, om = -1;
        for (int i = lo; i <= hi; ++i) {
            const int old = V[s * S];
            int old_o = 0;
            if (TB) old_o = O[s * S];
            const int sub = diag + (q[i - 1] == rj ? c.match : c.mismatch);  // :185
            int b2 = sub, o = diag_o;
            if (i > lo || lo == 1) {  // the deletion input lies in the band (or is row 0)
                const int del = prev + c.indel;  // :184
                if (TB) o = sub < del ? diag_o : prev_o;  // :310-316
                b2 = sub < del ? sub : del;
            }
            if (i == m) {  // the last row: no horizontal move; its stored value is never read
                vm = b2;
                om = o;
            } else {
                int nv = b2;
                if (i != rb) {  // (i, j - 1) lies in the band
                    const int ins = old + c.indel;  // :183
                    if (TB) o = ins < b2 ? old_o : o;  // :317-320
                    nv = ins < b2 ? ins : b2;
                }
                V[s * S] = nv;
                if (TB) O[s * S] = o;
                prev = nv;
                prev_o = o;
            }
            diag = old;
            diag_o = old_o;
            s = s + 1 == H ? 0 : s + 1;
        }
        if (hi == m && j >= c_lo && vm <= ae) {  // :417 (the end range does not bind)
            if (vm == 0 && (!TB || trim_side == 5)) return AlignOut{0, TB ? om : -1, j};  // :420-430
            if (TB) {  // :142-153
                if (vm < res.raw || (vm == res.raw && trim_side == 3 && om > res.start)) {
                    res.raw = vm;
                    res.start = om;
                    res.end = j;
                }
            } else if (vm < res.raw) {
                res.raw = vm;
                res.end = j;
            }
        }
        if (j + 1 - dtop > 1) {  // the top row leaves the band
            lo += 1;
            s_lo = s_lo + 1 == H ? 0 : s_lo + 1;
        }
        rj = rj_next;
    }
    return res;
}

// the read's bytes of the band's columns j0 .. j0 + M + H - 2, as words whose byte k is column j0 + k (bytes in front
// of the read are never used: clamped addresses)
template <int NS, bool STAGED>
__device__ __forceinline__ void band_fetch(const Bytes<STAGED> r, const int j0, uint32_t (&S)[NS]) {
    const long long rbase = (long long)(uintptr_t)r.p;
    const long long ad0 = rbase + (long long)(j0 - 1);
    const long long lo_ok = rbase & ~3LL;
    const long long a0 = ad0 & ~3LL;
    const uint32_t sh = (uint32_t)(ad0 & 3LL);
    uint32_t W[NS + 1];
#pragma unroll
    for (int w = 0; w <= NS; ++w) {
        long long aw = a0 + 4 * w;
        aw = aw < lo_ok ? lo_ok : aw;
        if constexpr (STAGED)
            W[w] = *(const LDS uint32_t *)(uintptr_t)aw;
        else
            W[w] = *(const uint32_t *)(uintptr_t)aw;
    }
#pragma unroll
    for (int w = 0; w < NS; ++w) S[w] = __builtin_amdgcn_alignbyte(W[w + 1], W[w], sh);
}

// hamming_align, classification.jl:557-625.  Scores of one call share the divisor m, so the
// reference's Float64 `score < best_score` / `==` are decided on the integer numerators.
template <bool STAGED>
__device__ __forceinline__ AlignOut hamming_dev(const Bytes<STAGED> q, const int m, const Bytes<STAGED> r,
                                                const int n, const int allowed, const int first,
                                                const int last, const int max_start, const int min_end,
                                                const int trim_side, const int ncode,
                                                const int jlo = -0x40000000, const int jhi = 0x40000000) {
    AlignOut best{BDX_INF32, -1, -1};
    int sf = first > 1 ? first : 1;  // :570
    int sl = last < max_start ? last : max_start;
    if (n - m + 1 < sl) sl = n - m + 1;  // :571
    if (sl < sf) return best;            // :573-576
    if (m == 0) return best;             // 0/0 = NaN never beats Inf (:607-613)
    // restricted run: an occurrence with <= allowed mismatches ENDS at a column whose unit distance is within the
    // budget, i.e. inside the hand-over window [jlo + m - 1, jhi] -> only the starts jlo .. jhi - m + 1 can be accepted
    if (jlo > sf) sf = jlo;
    if (jhi - m + 1 < sl) sl = jhi - m + 1;
    for (int j = sf; j <= sl; ++j) {     // :581
        const int end_pos = j + m - 1;
        if (end_pos < min_end) continue;  // :584-586
        int mism = 0;
        bool failed = false;
        for (int k = 0; k < m; ++k) {  // :592-604
            const int qc = q[k];
            const int rc = r[j - 1 + k];
            if (qc != rc && qc != ncode) {
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
                                              const int max_start, const int min_end, const int trim_side,
                                              const int jlo = -0x40000000, const int jhi = 0x40000000) {
    AlignOut none{BDX_INF32, -1, -1};
    const int sf = first > 1 ? first : 1;  // :490
    int sl = last < max_start ? last : max_start;
    if (n - m + 1 < sl) sl = n - m + 1;  // :491
    if (sl < sf) return none;            // :493-495
    // restricted run: every occurrence that starts inside sf..sl starts inside jlo .. jhi - m + 1 (the filter's
    // hand-over window); both scans below only ever ACCEPT such an occurrence, and an occurrence outside sf..sl
    // can only turn the answer into "none" when no occurrence inside precedes it in scan order — in which case
    // the restricted scan finds nothing and says "none" as well.
    const int rlo = jlo > 1 ? jlo : 1, rhi = jhi - m + 1;
    if (trim_side == 3) {                // :499-515: only the first findprev hit is examined
        for (int s = (sl < rhi ? sl : rhi); s >= rlo; --s) {
            if (bytes_equal<STAGED>(q, m, r, s)) {
                if (s >= sf && s + m - 1 >= min_end) return AlignOut{0, s, s + m - 1};
                return none;
            }
        }
        return none;
    }
    // :517-547: leftmost hit; hits that end before min_end_pos are skipped and the scan goes on
    const int top = n - m + 1 < rhi ? n - m + 1 : rhi;
    for (int s = (sf > rlo ? sf : rlo); s <= top; ++s) {
        if (bytes_equal<STAGED>(q, m, r, s)) {
            if (s > sl) return none;
            if (s + m - 1 >= min_end) return AlignOut{0, s, s + m - 1};
        }
    }
    return none;
}

// Column window of one pass for a read of length n (match_barcode_pass, classification.jl
// :795-809): final_search_range = first:last, max_start_pos, min_end_pos; false when the
// :805 sanity check sends the read to :unknown.  Shared by the exact stage and the filters
// (a recorded alignment only ever consumes read positions first..last).
struct PassWindow {
    int first, last, max_start, min_end;
};

__device__ __forceinline__ bool pass_window(const BdxDevPass &P, const int n, PassWindow &w) {
    long long first, last, max_start_ll, min_end_ll;
    if (P.explicit_window) {
        // hand-made windows are clamped to the read: columns outside 1..n index past the
        // sequence in the reference too (undefined there), so this only adds definedness
        first = P.win_first > 1 ? P.win_first : 1;
        last = P.win_last < n ? P.win_last : n;
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
        if (first > last || first > max_start_ll || last < min_end_ll) return false;  // :805-807
    }
    // After the sanity check every bound is within [-(2^30), 2^30]; clamp so int32 arithmetic
    // in the cores cannot overflow for hand-made explicit windows.
    const long long LIM = 1LL << 30;
    auto clampi = [&](long long v) -> int { return (int)(v > LIM ? LIM : (v < -LIM ? -LIM : v)); };
    w.first = clampi(first);
    w.last = clampi(last);
    w.max_start = clampi(max_start_ll);
    w.min_end = clampi(min_end_ll);
    return true;
}

// Return tuple of find_best_matching_bc (classification.jl:722) plus match_barcode_pass's status.
struct PassOut {
    int status;  // 1 match, 0 unknown, -1 ambiguous, 2 pass not run
    int bc;      // min_score_bc (kept even when ambiguous)
    int start, end, raw;
    double score;
    double delta;
    double sub = __builtin_inf();  // the with_delta reducer's sub_min_score (internal: tier settle rule)
};

// The two sequential reducers of the reference as one state machine:
// find_best_matching_bc_no_delta (classification.jl:632-667) when min_delta == 0.0 (:723),
// find_best_matching_bc_with_delta (:669-713) otherwise.  Every comparison is the Float64
// comparison written there; `rate` is the tightening max_error_rate.
struct Reducer {
    double rate, min_score, sub_min;
    int best, bs, be, braw;
    bool with_delta;

    __device__ __forceinline__ void init(const BdxDevCfg &cfg) {
        rate = cfg.max_error_rate;
        min_score = __builtin_inf();
        sub_min = __builtin_inf();
        best = 0;
        bs = -1;
        be = -1;
        braw = -1;
        with_delta = !(cfg.min_delta == 0.0);  // :723
    }
    __device__ __forceinline__ void feed(const int b /*0-based*/, const AlignOut &a, const double score) {
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
    __device__ __forceinline__ PassOut finish(const BdxDevCfg &cfg) const {
        PassOut po{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
        const double delta = with_delta ? (sub_min - min_score) : __builtin_inf();  // :711 / :666
        po.bc = best;
        po.start = bs;
        po.end = be;
        po.raw = braw;
        po.score = min_score;
        po.delta = delta;
        po.sub = sub_min;
        if (best == 0) return po;                        // :820-821
        po.status = (delta < cfg.min_delta) ? -1 : 1;    // :822-823, :867
        return po;
    }
};

// One candidate through the band bodies of barcode length MM (WIDE17: a 17-diagonal body exists for it).  false: the
// wave needs the wide body and there is none — the caller takes the all-rows DP.
template <int MM, bool WIDE17, bool STAGED>
__device__ __forceinline__ bool band_call(const Bytes<STAGED> q, const Bytes<STAGED> r, AlignOut &a, const bool tbf,
                                          const bool wide, const int kbb, const int ae, const Costs c, const int trim_side,
                                          const int jf, const int e_lo, const int e_hi) {
    if (wide && !WIDE17) return false;
    // the barcode's bytes, four per word (the barcode tables are word-aligned only by chance)
    uint32_t QW[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MM; ++i) QW[i >> 2] |= (uint32_t)q[i] << (8 * (i & 3));
    const bdx_u32x4 qa = {QW[0], QW[1], QW[2], QW[3]}, qb4 = {QW[4], QW[5], QW[6], QW[7]};
    const uint32_t costs = (uint32_t)c.match | ((uint32_t)c.mismatch << 8) | ((uint32_t)c.indel << 16) | ((uint32_t)trim_side << 24);
    uint32_t S[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) S[w] = 0;
    if (!wide) {
        constexpr int HH = 9, NS = (MM + HH + 2) / 4 + 1;
        const int j0 = e_hi + kbb - (MM + HH - 2);
        uint32_t T[NS];
        band_fetch<NS, STAGED>(r, j0, T);
#pragma unroll
        for (int w = 0; w < NS; ++w) S[w] = T[w];
        const bdx_u32x4 sa = {S[0], S[1], S[2], S[3]}, sb = {S[4], S[5], S[6], S[7]}, sc = {S[8], S[9], S[10], S[11]}, sd = {S[12], S[13], S[14], S[15]};
        a = tbf ? sg_core_band<true, MM, HH>(qa, qb4, sa, sb, sc, sd, j0, ae, costs, jf, e_lo, e_hi)
                : sg_core_band<false, MM, HH>(qa, qb4, sa, sb, sc, sd, j0, ae, costs, jf, e_lo, e_hi);
    } else if constexpr (WIDE17) {
        constexpr int HH = 17, NS = (MM + HH + 2) / 4 + 1;
        const int j0 = e_hi + kbb - (MM + HH - 2);
        uint32_t T[NS];
        band_fetch<NS, STAGED>(r, j0, T);
#pragma unroll
        for (int w = 0; w < NS; ++w) S[w] = T[w];
        const bdx_u32x4 sa = {S[0], S[1], S[2], S[3]}, sb = {S[4], S[5], S[6], S[7]}, sc = {S[8], S[9], S[10], S[11]}, sd = {S[12], S[13], S[14], S[15]};
        a = tbf ? sg_core_band<true, MM, HH>(qa, qb4, sa, sb, sc, sd, j0, ae, costs, jf, e_lo, e_hi)
                : sg_core_band<false, MM, HH>(qa, qb4, sa, sb, sc, sd, j0, ae, costs, jf, e_lo, e_hi);
    }
    return true;
}

// band bodies by barcode length (every barcode of the config has cfg.band_m bases); false: no body for this case
template <bool STAGED, int REGM>
__device__ __forceinline__ bool band_dispatch(const int band_m, const Bytes<STAGED> q, const Bytes<STAGED> r, AlignOut &a,
                                              const bool tbf, const bool wide, const int kbb, const int ae, const Costs c,
                                              const int trim_side, const int jf, const int e_lo, const int e_hi) {
    switch (band_m) {
        case 8:
            if constexpr (REGM == 24) return band_call<8, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 10:
            if constexpr (REGM == 24) return band_call<10, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 12:
            if constexpr (REGM == 24) return band_call<12, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 16:
            if constexpr (REGM == 24) return band_call<16, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 20:
            if constexpr (REGM == 24) return band_call<20, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 24:
            if constexpr (REGM == 24) return band_call<24, true, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        case 32:
            if constexpr (REGM == 32) return band_call<32, false, STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
            break;
        default:
            break;
    }
    return false;
}

// One pass of a read with MANY candidates (short barcodes: dense window table, wcount == 254) in the band's domain —
// clean class, every barcode with the same number of bases m, no N-scoring.  There the reference's reducers are
// order-independent: an alignment's result does not depend on the tightened threshold beyond being accepted (clean
// class), all scores share the divisor m, so both reducers (:632-713) end with  min = the smallest raw score (first
// barcode in file order among ties), sub_min = the second smallest  — whatever the order of evaluation, and a
// barcode that ties the running sub_min changes nothing whether floor(sub_min * m) lets it through or not.
// That frees the evaluation order: lane-per-read evaluation in file order makes a wave pay the most expensive body
// of any of its lanes at EVERY step (one candidate in eight has two hits in the read and a window of many columns);
// here each lane walks its candidates class by class — 9-diagonal band, 17-diagonal band, then the wide windows in
// chunks of end columns, each chunk a band of its own (results folded in column order with the recording rule
// :142-153 and the early exit :420-430; oracle: orc_selftest_band_class, chunked cases) — so a wave runs one body per
// phase.  The winner and the runner-up are handed to the ordinary Reducer at the end.
template <bool STAGED, int REGM>
__device__ __forceinline__ PassOut run_pass_phased(const BdxDevCfg &cfg, const BdxDevPass &P, const int pidx,
                                                   const Bytes<STAGED> bcb, const LDS uint32_t *bc_off,
                                                   const Bytes<STAGED> r, const uint32_t *cand, const uint32_t *went,
                                                   const int jf, const int jl_last, const int n_read, const int trim_side,
                                                   const bool need_tb) {
    const int m = cfg.band_m;
    const int kbb = cfg.band_kb[pidx], lb = cfg.band_lb[pidx];
    const int B = P.n_barcodes;
    const Costs c{cfg.match, cfg.mismatch, cfg.indel, cfg.nindel, 0x4E};
    const int ae = (int)__builtin_floor(cfg.max_error_rate * (double)m);  // :254 at the initial threshold
    const bool end_only = need_tb && trim_side == 5 && !cfg.need_traceback && cfg.end_only_ok;
    const bool tbf = need_tb && !end_only;
    const bool has17 = m <= 24;  // (no 17-diagonal body for 32-row barcodes: their medium windows are walked in chunks)
    int best_raw = BDX_INF32, best_b = -1, best_s = -1, best_e = -1, sub_raw = BDX_INF32;
    const auto fold = [&](const int b, const AlignOut &a) {
        if (a.raw >= BDX_INF32) return;
        if (a.raw < best_raw || (a.raw == best_raw && b < best_b)) {
            sub_raw = best_raw < sub_raw ? best_raw : sub_raw;
            best_raw = a.raw;
            best_b = b;
            best_s = a.start;
            best_e = a.end;
        } else {
            sub_raw = a.raw < sub_raw ? a.raw : sub_raw;
        }
    };
    // the lane's candidates of one class, in ascending barcode order
    struct Cursor { int word; uint32_t bits; };
    const auto next_of_class = [&](Cursor &cu, const int cls, int &b, int &e_lo, int &e_hi) {
        for (;;) {
            while (cu.bits == 0u) {
                if (++cu.word >= P.cand_words) return false;
                cu.bits = cand[cu.word];
            }
            b = cu.word * 32 + __builtin_ctz(cu.bits);
            cu.bits &= cu.bits - 1u;
            if (b >= B) return false;
            const uint32_t e = went[b];
            e_lo = (int)(e & 0xFFFFu) - 1024 + lb;
            e_hi = (int)(e >> 16);
            if (e_hi < 1 || e_hi > n_read || e_lo > e_hi) {  // not a window (defence in depth): the whole pass window, in chunks
                if (cfg.dbg_rejected) atomicAdd(cfg.dbg_rejected, 1u);
                e_lo = jf;
                e_hi = jl_last;
            }
            const int need = (e_hi - e_lo + 1) + 2 * kbb;
            const int k = need <= 9 ? 0 : (need <= 17 && has17) ? 1 : 2;
            if (k == cls) return true;
        }
    };
    for (int cls = 0; cls < 2; ++cls) {
        Cursor cu{0, cand[0]};
        for (;;) {
            int b = 0, e_lo = 0, e_hi = 0;
            const bool have = next_of_class(cu, cls, b, e_lo, e_hi);
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
            if (have) {
                AlignOut a{BDX_INF32, -1, -1};
                band_dispatch<STAGED, REGM>(m, bcb.at((int)bc_off[b]), r, a, tbf, cls == 1, kbb, ae, c, trim_side, jf, e_lo, e_hi);
                if (!need_tb) a.end = -1;
                fold(b, a);
            }
        }
    }
    {   // wide windows: chunks of end columns
        const bool wide_body = has17;
        const int cw = (wide_body ? 17 : 9) - 2 * kbb;  // end columns per chunk (kb <= 4: >= 1)
        Cursor cu{0, cand[0]};
        int cur_b = -1, c_lo = 0, cur_hi = -1;
        AlignOut cur{BDX_INF32, -1, -1};
        bool stop = false;
        for (;;) {
            bool have = cur_b >= 0 && !stop && c_lo <= cur_hi;
            if (!have) {
                if (cur_b >= 0) fold(cur_b, cur);
                cur_b = -1;
                int b = 0, e_lo = 0, e_hi = 0;
                if (next_of_class(cu, 2, b, e_lo, e_hi)) {
                    cur_b = b;
                    c_lo = e_lo;
                    cur_hi = e_hi;
                    cur = AlignOut{BDX_INF32, -1, -1};
                    stop = false;
                    have = true;
                }
            }
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
            if (have) {
                const int c_hi = c_lo + cw - 1 < cur_hi ? c_lo + cw - 1 : cur_hi;
                AlignOut a{BDX_INF32, -1, -1};
                band_dispatch<STAGED, REGM>(m, bcb.at((int)bc_off[cur_b]), r, a, tbf, wide_body, kbb, ae, c, trim_side, jf, c_lo, c_hi);
                if (a.raw < BDX_INF32) {
                    if (a.raw < cur.raw || (a.raw == cur.raw && tbf && trim_side == 3 && a.start > cur.start)) cur = a;  // :142-153
                    if (a.raw == 0 && (!tbf || trim_side == 5)) stop = true;                                             // :420-430
                }
                c_lo += cw;
            }
        }
    }
    Reducer red;
    red.init(cfg);
    if (best_b >= 0) {
        red.feed(best_b, AlignOut{best_raw, best_s, need_tb ? best_e : -1}, (double)best_raw / (double)m);  // :155-168
        if (sub_raw < BDX_INF32) red.feed(best_b + 1, AlignOut{sub_raw, -1, -1}, (double)sub_raw / (double)m);
    }
    return red.finish(cfg);
}

// match_barcode_pass (classification.jl:776-868, minus the histogram block :827-865) with the
// reducers find_best_matching_bc_no_delta (:632-667) / _with_delta (:669-713) inlined.
template <bool STAGED, int REGM = 0, bool CLEAN = false, bool UM = false>
__device__ __forceinline__ PassOut run_pass(const BdxDevCfg &cfg, const BdxDevPass &P, const Bytes<STAGED> bcb,
                                            const LDS uint32_t *bc_off, const LDS int *bc_nn,
                                            const Bytes<STAGED> r, const int n, LDS int *DP, LDS int *OG,
                                            const int S, const uint32_t *cand, const int ncode = 0x4E,
                                            const uint32_t *went = nullptr, const int wcount = 255) {
    PassOut po{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    PassWindow w;
    if (!pass_window(P, n, w)) return po;  // :805-807
    const int jf = w.first, jl = w.last, max_start = w.max_start, min_end = w.min_end;

    const int trim_side = P.trim_side;
    const bool need_tb = (trim_side != 0) || cfg.need_traceback;  // :812
    const Costs c{cfg.match, cfg.mismatch, cfg.indel, cfg.nindel, ncode};
    Reducer red;
    red.init(cfg);

    const bool align_one = P.explicit_window == BDX_WINDOW_ALIGN_ONE;
    const int B = align_one ? 1 : P.n_barcodes;
    if constexpr (REGM > 0 && CLEAN && !UM) {
        // many candidates per read (dense window table) inside the band's domain: class-phased evaluation.  (Not in the
        // kernels whose barcodes all have REGM rows: 24- and 32-base barcodes never have that many genuine candidates,
        // and the extra code costs the ordinary path registers — C4 853 -> 814 M reads/s with it.)
        const int pidx0 = (&P == &cfg.pass[1]) ? 1 : 0;
        if (wcount == 254 && cand && !align_one && n > 0 && cfg.band_m > 0 && cfg.band_kb[pidx0] >= 0 && !cfg.has_nindel &&
            cfg.algorithm == BDX_ALG_SEMIGLOBAL && cfg.match < 256 && cfg.mismatch < 256 && cfg.indel < 256)
            return run_pass_phased<STAGED, REGM>(cfg, P, pidx0, bcb, bc_off, r, cand, went, jf, jl, n, trim_side, need_tb);
    }
    // :638 / :676 — barcodes in file order, the threshold tightens as we go.  With a candidate
    // mask each lane walks its OWN set bits (ascending = file order), so the lanes of a wave
    // evaluate their k-th candidate together instead of serialising on the barcode index.
    int b = -1;
    int cword = 0;
    uint32_t cbits = cand ? cand[0] : 0u;
    for (;;) {
        if (cand) {
            while (cbits == 0u) {
                if (++cword >= P.cand_words) break;
                cbits = cand[cword];
            }
            if (cbits == 0u) break;
            b = cword * 32 + __builtin_ctz(cbits);
            cbits &= cbits - 1u;
            if (b >= B) break;
        } else if (++b >= B) {
            break;
        }
        const int o = (int)bc_off[b];
        const int m = (int)bc_off[b + 1] - o;
        const Bytes<STAGED> q = bcb.at(o);
        AlignOut a;
        double score;
        // column restriction of this candidate (split mode hands over up to BDX_WCAP entries
        // {barcode, first column, last column} per read and pass; none -> the whole window)
        int cjlo = -0x40000000, cjhi = 0x40000000;
        if (wcount == 254) {  // dense table: one entry per barcode (plain-sweep kernels, every pair swept once)
            const uint32_t e = went[b];
            cjlo = (int)(e & 0xFFFFu) - 1024;
            cjhi = (int)(e >> 16);
        } else if (wcount <= BDX_WCAP) {
            // several entries of one barcode (separately swept occurrences) are united
            bool any_entry = false;
            for (int e = 0; e < wcount; ++e)
                if ((int)went[3 * e] == b) {
                    const int lo_e = (int)went[3 * e + 1], hi_e = (int)went[3 * e + 2];
                    cjlo = any_entry ? (lo_e < cjlo ? lo_e : cjlo) : lo_e;
                    cjhi = any_entry ? (hi_e > cjhi ? hi_e : cjhi) : hi_e;
                    any_entry = true;
                }
        }
        // a hand-over window ends inside the read; anything else is not a window (defence in depth: columns drive
        // addresses in the band form) -> no restriction
        if (cjhi < 0x40000000 && (cjhi < 1 || cjhi > n || cjlo > cjhi)) {
            if (cfg.dbg_rejected) atomicAdd(cfg.dbg_rejected, 1u);
            cjlo = -0x40000000;
            cjhi = 0x40000000;
        }
        if (cfg.algorithm == BDX_ALG_HAMMING) {
            const int allowed = (int)__builtin_floor(red.rate * (double)m);  // :567
            a = hamming_dev<STAGED>(q, m, r, n, allowed, jf, jl, max_start, min_end, trim_side, ncode, cjlo, cjhi);
            score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)m;  // :607
        } else if (cfg.algorithm == BDX_ALG_EXACT) {
            a = exact_dev<STAGED>(q, m, r, n, jf, jl, max_start, min_end, trim_side, cjlo, cjhi);
            score = a.raw >= BDX_INF32 ? __builtin_inf() : 0.0;
        } else {
            const int norm = cfg.has_nindel ? bc_nn[b] : m;               // :460 / :476
            const int ae = (int)__builtin_floor(red.rate * (double)norm);  // :254
            if (cfg.has_nindel) {
                a = need_tb ? sg_core<true, true, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi)
                            : sg_core<false, true, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi);
            } else if (REGM > 0 && CLEAN) {
                // clean class (see sg_core_clean): same three output forms as below
                const bool end_only = need_tb && trim_side == 5 && !cfg.need_traceback && cfg.end_only_ok;
                {
                    // diagonal band (sg_core_band): the tracked sweep's end columns are known and the band fits
                    const int pidx = (&P == &cfg.pass[1]) ? 1 : 0;
                    const int kbb = cfg.band_kb[pidx];
                    const int e_lo = cjlo + cfg.band_lb[pidx], e_hi = cjhi;
                    const int need = kbb >= 0 && (wcount <= BDX_WCAP || wcount == 254) && cjhi < 0x40000000 ? (e_hi - e_lo + 1) + 2 * kbb : 0x7FFF;
                    if (need <= 17 && n > 0 && cfg.match < 256 && cfg.mismatch < 256 && cfg.indel < 256) {
                        const bool tbf = need_tb && !end_only;
                        bool ran = false;
                        // one band width per wave: the straight-line bodies are long, a wave should run only one
                        const bool wide = __builtin_amdgcn_ballot_w64(need > 9) != 0ull;
                        if constexpr (UM)  // (every barcode has REGM rows: one body set, no dispatch)
                            ran = band_call<(REGM > 0 ? REGM : 4), (REGM <= 24), STAGED>(q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
                        else
                            ran = band_dispatch<STAGED, REGM>(cfg.band_m, q, r, a, tbf, wide, kbb, ae, c, trim_side, jf, e_lo, e_hi);
                        if (ran) {
                            if (!need_tb) a.end = -1;
                            goto band_done;
                        }
                    }
                }
                a = !need_tb ? sg_core_clean<false, (REGM > 0 ? REGM : 4), STAGED, false, UM>(q, m, r, n, ae, c, trim_side, jf, jl, cjlo, cjhi)
                    : end_only ? sg_core_clean<false, (REGM > 0 ? REGM : 4), STAGED, true, UM>(q, m, r, n, ae, c, trim_side, jf, jl, cjlo, cjhi)
                               : sg_core_clean<true, (REGM > 0 ? REGM : 4), STAGED, false, UM>(q, m, r, n, ae, c, trim_side, jf, jl, cjlo, cjhi);
            } else if (REGM == 0 && CLEAN) {
                // Barcodes beyond the register DP's 32 rows inside the clean class: the rolling diagonal band (sg_band_roll) over
                // the end columns the filter handed over — or, without a hand-over (filter off, more entries than a row holds),
                // over the whole pass window — in chunks of as many end columns as the lane's H cells allow; the chunks are
                // folded in column order with the recording rule (:142-153) and the early exit on a zero (:420-430).  The budget
                // of the band is what a recorded alignment can spend at the CURRENT threshold: ae / min(mismatch, indel)
                // operations (the threshold only tightens, so the planned H — two budgets at the initial rate + 9 — always
                // leaves room for at least nine end columns per chunk).
                const int pidx = (&P == &cfg.pass[1]) ? 1 : 0;
                const int cmin = c.mismatch < c.indel ? c.mismatch : c.indel;
                const int kbb = ae / cmin;
                int e_lo = jf, e_hi = jl;
                if (cjhi < 0x40000000) {
                    // (uniform budgets: the first end column itself; else the restricted run's first column — a superset)
                    const int lo_e = cfg.band_kb[pidx] >= 0 ? cjlo + cfg.band_lb[pidx] : cjlo;
                    e_lo = lo_e > e_lo ? lo_e : e_lo;
                    e_hi = cjhi < e_hi ? cjhi : e_hi;
                }
                a = AlignOut{BDX_INF32, -1, -1};
                const int wc = cfg.band_hcap - 2 * kbb;
                if (m > 0 && n > 0 && wc >= 1) {
                    const bool end_only = need_tb && trim_side == 5 && !cfg.need_traceback && cfg.end_only_ok;
                    const bool tbf = need_tb && !end_only;
                    for (int clo = e_lo; clo <= e_hi; clo += wc) {
                        const int chi = clo + wc - 1 < e_hi ? clo + wc - 1 : e_hi;
                        const AlignOut x = tbf ? sg_band_roll<true, STAGED>(DP, OG, S, q, m, r, ae, c, trim_side, jf, clo, chi, kbb)
                                               : sg_band_roll<false, STAGED>(DP, OG, S, q, m, r, ae, c, trim_side, jf, clo, chi, kbb);
                        if (x.raw >= BDX_INF32) continue;
                        if (x.raw == 0 && (!tbf || trim_side == 5)) {  // the walk ends at the first zero
                            a = x;
                            break;
                        }
                        if (tbf ? (x.raw < a.raw || (x.raw == a.raw && trim_side == 3 && x.start > a.start)) : x.raw < a.raw) a = x;
                    }
                    if (!need_tb) a.end = -1;
                    if (end_only) a.start = -1;
                }
            } else if (REGM > 0) {
                // With trim_side == 5 and nobody asking for the start position, the alignment's END is all that is
                // observable (keep_start = end + 1, :914): the origin half of the DP is dropped.  Same values, same
                // early exit (:420) and the same strict-improvement rule as the traceback form with trim_side 5.
                const bool end_only = need_tb && trim_side == 5 && !cfg.need_traceback && cfg.end_only_ok;
                a = !need_tb ? sg_core_reg<false, false, (REGM > 0 ? REGM : 4), STAGED>(q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi)
                    : end_only ? sg_core_reg<false, false, (REGM > 0 ? REGM : 4), STAGED, true>(q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi)
                               : sg_core_reg<true, false, (REGM > 0 ? REGM : 4), STAGED>(q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi);
            } else {
                a = need_tb ? sg_core<true, false, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi)
                            : sg_core<false, false, STAGED>(DP, OG, S, q, m, r, n, ae, c, trim_side, jf, jl, max_start, min_end, cjlo, cjhi);
            }
        band_done:
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
        red.feed(b, a, score);
    }
    return red.finish(cfg);
}

// The same pass when the per-barcode alignment result is already known as a function of the
// threshold: for the class delimited in bdx_bitpar.hip ("known-score class") the reference's
// semiglobal_alignment returns  d  if d <= floor(max_error * m)  else Inf, d being the exact
// unit-cost semi-global distance delivered by the bit-vector sweep.  `entries` holds up to four
// (barcode << 8 | d) words of this read in arbitrary order; they are replayed in ascending
// barcode (= file) order through the very same reducer.
// The reference evaluates every barcode ONCE over the whole window (:638, :676).  The seeded variants
// may sweep one (read, barcode) pair over several column windows (one per cluster of seed diagonals:
// a barcode that occurs twice in a read, or overlapping clusters around one occurrence) and so push
// several entries of one barcode; each is the minimum over its window, the window holding the best
// occurrence delivers the whole-window minimum d*, so the barcode's value is the SMALLEST of its
// entries.  Ascending (barcode << 8 | d) order visits that one first; the others are skipped.
// KEND (known-trim class, bdx_wave.hip): entries are barcode << 22 | d << 16 | position key — the same ascending order; the key
// is the 1-based end column for a trim_side = 5 pass (the leftmost end first among equal distances of a barcode), 0xFFFF - start
// for a trim_side = 3 pass (the largest start first); the winner's key goes through the reducer in AlignOut::end and
// classify_known turns it into the pass's start or end.
template <bool KEND = false, class MLen>
__device__ __forceinline__ PassOut run_pass_known(const BdxDevCfg &cfg, const MLen mlen,
                                                  const uint32_t e0, const uint32_t e1, const uint32_t e2,
                                                  const uint32_t e3, const int count) {
    Reducer red;
    red.init(cfg);
    uint32_t last = 0;  // entries are > 0 only if barcode > 0 or d > 0; use +1 bias below
    int fed_b = -1;     // barcode of the entry fed last
    for (int k = 0; k < count; ++k) {
        // smallest biased entry greater than `last` (static scan: no dynamic register indexing)
        uint32_t pick = 0xFFFFFFFFu;
        const uint32_t c0 = e0 + 1u, c1 = e1 + 1u, c2 = e2 + 1u, c3 = e3 + 1u;
        if (count > 0 && c0 > last && c0 < pick) pick = c0;
        if (count > 1 && c1 > last && c1 < pick) pick = c1;
        if (count > 2 && c2 > last && c2 < pick) pick = c2;
        if (count > 3 && c3 > last && c3 < pick) pick = c3;
        if (pick == 0xFFFFFFFFu) break;  // equal duplicates: fewer distinct entries than `count`
        last = pick;
        const uint32_t e = pick - 1u;
        const int b = KEND ? (int)(e >> 22) : (int)(e >> 8);
        const int d = KEND ? (int)((e >> 16) & 63u) : (int)(e & 255u);
        if (b == fed_b) continue;  // a larger entry of the barcode just fed (another window of the same pair)
        fed_b = b;
        const int m = mlen(b);  // barcode length
        const int ae = (int)__builtin_floor(red.rate * (double)m);  // :254 with the tightened rate
        AlignOut a{d <= ae ? d : BDX_INF32, -1, KEND ? (int)(e & 0xFFFFu) : -1};
        const double score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)m;  // :155-160
        red.feed(b, a, score);
    }
    return red.finish(cfg);
}

struct Verdict {
    int bc1, bc2, keep_start, keep_end;
};

// The same replay over `count` entries in LDS (short barcodes at high rates have many genuine candidates per read:
// a 10-mer within two edits of a random 150-base read is no rarity).  Selection of the next entry in ascending
// (barcode << 8 | d) order is a scan per step — count is small (<= 32).
template <class MLen>
__device__ __forceinline__ PassOut run_pass_known_ent(const BdxDevCfg &cfg, const MLen mlen, const LDS uint32_t *ent,
                                                      const int count) {
    Reducer red;
    red.init(cfg);
    uint32_t last = 0;
    int fed_b = -1;
    for (int k = 0; k < count; ++k) {
        uint32_t pick = 0xFFFFFFFFu;
        for (int t = 0; t < count; ++t) {
            const uint32_t c = ent[t] + 1u;
            if (c > last && c < pick) pick = c;
        }
        if (pick == 0xFFFFFFFFu) break;
        last = pick;
        const uint32_t e = pick - 1u;
        const int b = (int)(e >> 8);
        const int d = (int)(e & 255u);
        if (b == fed_b) continue;
        fed_b = b;
        const int m = mlen(b);
        const int ae = (int)__builtin_floor(red.rate * (double)m);  // :254 with the tightened rate
        AlignOut a{d <= ae ? d : BDX_INF32, -1, -1};
        const double score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)m;  // :155-160
        red.feed(b, a, score);
    }
    return red.finish(cfg);
}

template <class MLen>
__device__ __forceinline__ PassOut run_pass_known_dense(const BdxDevCfg &cfg, const MLen mlen, const LDS unsigned char *dt,
                                                        const LDS uint32_t *cbits, const int cwords) {
    Reducer red;
    red.init(cfg);
    for (int w = 0; w < cwords; ++w) {
        uint32_t bits = cbits[w];
        while (bits) {
            const int b = w * 32 + __builtin_ctz(bits);
            bits &= bits - 1u;
            const int d = dt[b];
            const int m = mlen(b);
            const int ae = (int)__builtin_floor(red.rate * (double)m);  // :254 with the tightened rate
            AlignOut a{d <= ae ? d : BDX_INF32, -1, -1};
            const double score = a.raw >= BDX_INF32 ? __builtin_inf() : (double)a.raw / (double)m;  // :155-160
            red.feed(b, a, score);
        }
    }
    return red.finish(cfg);
}

// Per-pass hand-over from the bit-vector sweep for reads of the known-score class.
struct KnownPass {
    bool use;
    uint32_t e0, e1, e2, e3;
    int count;
    const LDS uint32_t *ent;  // != nullptr: the entries live in LDS (count may exceed four: short barcodes, see slot_cap)
    // dense form (plain-sweep kernels with few barcodes): d of EVERY candidate in a byte table, walked through the
    // read's candidate mask in ascending barcode (= file) order — any number of survivors, no selection scan
    const LDS unsigned char *dt;
    const LDS uint32_t *cbits;
    int cwords;
};

template <bool STAGED, int REGM = 0, bool CLEAN = false, bool UM = false>
__device__ __forceinline__ void classify_one(const BdxDevCfg &cfg, const Bytes<STAGED> bcb0,
                                             const Bytes<STAGED> bcb1, const LDS uint32_t *off0,
                                             const LDS uint32_t *off1, const LDS int *nn0, const LDS int *nn1,
                                             const Bytes<STAGED> r, const int n, LDS int *DP, LDS int *OG,
                                             const int S, const uint32_t *cand0, const uint32_t *cand1,
                                             Verdict &v, PassOut &p1, PassOut &p2,
                                             const KnownPass kn0 = KnownPass{false, 0, 0, 0, 0, 0},
                                             const KnownPass kn1 = KnownPass{false, 0, 0, 0, 0, 0},
                                             const int ncode = 0x4E, const uint32_t *went0 = nullptr,
                                             const int wcount0 = 255, const uint32_t *went1 = nullptr,
                                             const int wcount1 = 255) {
    // determine_filename, classification.jl:871-938
    v = Verdict{0, 0, -1, -1};
    p2 = PassOut{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    const auto m0 = [&](const int b) { return (int)off0[b + 1] - (int)off0[b]; };
    const auto m1 = [&](const int b) { return (int)off1[b + 1] - (int)off1[b]; };
    p1 = kn0.use ? run_pass_known(cfg, m0, kn0.e0, kn0.e1, kn0.e2, kn0.e3, kn0.count)
                 : run_pass<STAGED, REGM, CLEAN, UM>(cfg, cfg.pass[0], bcb0, off0, nn0, r, n, DP, OG, S, cand0, ncode, went0, wcount0);  // :875
    if (p1.status != 1) {  // :879-883
        v.bc1 = p1.status;
        return;
    }
    if (cfg.is_dual) {  // :887-895
        p2 = kn1.use ? run_pass_known(cfg, m1, kn1.e0, kn1.e1, kn1.e2, kn1.e3, kn1.count)
                     : run_pass<STAGED, REGM, CLEAN, UM>(cfg, cfg.pass[1], bcb1, off1, nn1, r, n, DP, OG, S, cand1, ncode, went1, wcount1);
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


// determine_filename (classification.jl:871-938) for a read whose passes all sit in the known-score
// class: both passes are reducer replays, nothing is aligned, nothing is trimmed (ScoreOnly configs
// have no trim side; the keep range is the whole read as in :907-908, :932-935).
// KEND: the known-trim class — the passes' entries carry position keys (run_pass_known): the winner's start (trim_side = 3) or end
// (trim_side = 5) trims the keep range as in :910-929.
// KEND = 2: the known-alignment class — a pass without a trim side keeps its end column too (the kernel then finds the other
// position of every pass's winner with an anchored sweep).
template <int KEND>
__device__ __forceinline__ void known_positions(PassOut &po, const int trim_side, const bool need_tb) {
    if (!KEND) return;
    const int key = po.end;  // (-1: no winner)
    po.start = -1;
    po.end = -1;
    if (key < 0) return;
    if (trim_side == 3) po.start = 0xFFFF - key;
    if (trim_side == 5 || (KEND == 2 && trim_side == 0 && need_tb)) po.end = key;
}

template <int KEND = 0, class MLen0, class MLen1>
__device__ __forceinline__ void classify_known(const BdxDevCfg &cfg, const MLen0 m0, const MLen1 m1, const int n,
                                               const KnownPass kn0, const KnownPass kn1, Verdict &v, PassOut &p1,
                                               PassOut &p2) {
    v = Verdict{0, 0, -1, -1};
    p2 = PassOut{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    p1 = kn0.dt ? run_pass_known_dense(cfg, m0, kn0.dt, kn0.cbits, kn0.cwords)
         : kn0.ent ? run_pass_known_ent(cfg, m0, kn0.ent, kn0.count)
                   : run_pass_known<(KEND != 0)>(cfg, m0, kn0.e0, kn0.e1, kn0.e2, kn0.e3, kn0.count);  // :875
    known_positions<KEND>(p1, cfg.pass[0].trim_side, cfg.need_traceback != 0);
    if (p1.status != 1) {  // :879-883
        v.bc1 = p1.status;
        return;
    }
    if (cfg.is_dual) {  // :887-895
        p2 = kn1.dt ? run_pass_known_dense(cfg, m1, kn1.dt, kn1.cbits, kn1.cwords)
             : kn1.ent ? run_pass_known_ent(cfg, m1, kn1.ent, kn1.count)
                       : run_pass_known<(KEND != 0)>(cfg, m1, kn1.e0, kn1.e1, kn1.e2, kn1.e3, kn1.count);
        known_positions<KEND>(p2, cfg.pass[1].trim_side, cfg.need_traceback != 0);
        if (p2.status != 1) {
            v.bc1 = p2.status;
            return;
        }
        v.bc2 = p2.bc;
    }
    v.bc1 = p1.bc;
    int keep_start = 1, keep_end = n;  // :907-908
    if (KEND) {
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
    }
    if (keep_start > keep_end) {  // :932-935
        v.keep_start = 1;
        v.keep_end = 0;
    } else {
        v.keep_start = keep_start;
        v.keep_end = keep_end;
    }
}

// match_barcode_pass's statistics block (classification.jl:827-865): runs iff the pass returned :match
__device__ __forceinline__ void stats_update(const BdxDevStats &st, const int p, const int B, const PassOut &po) {
    if (po.status != 1) return;
    const long long b = po.bc - 1;
    const long long prow = (long long)po.start - 1 + st.pos_bias;
    const long long lrow = (long long)po.end - po.start + 1;
    const long long rrow = po.raw;
    if (prow < 0 || prow >= st.rows || lrow < 0 || lrow >= st.len_rows || rrow < 0 || rrow >= st.raw_rows) {
        atomicOr(st.overflow, 1u);
        return;
    }
    atomicAdd(&st.pos[p][prow * B + b], 1ULL);
    atomicAdd(&st.len[p][b * st.len_stride + lrow], 1ULL);
    atomicAdd(&st.raw[p][b * st.raw_stride + rrow], 1ULL);
}

}  // namespace
