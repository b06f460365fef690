// bdx_wave.hip — wave-autonomous seeded filter + reducer replay for gfx950 (the C2 headline path).
//
// Same lossless filter and the same verdict logic as bdx_bitpar.hip's single-seed variant — pigeonhole
// q-gram seeds decide which (read, barcode) pairs are swept and where, Myers' bit-vector sweep gives the
// unit distance d* of every seeded pair, and for reads of the known-score class the
// verdict is a replay of the reference's reducers (classification.jl:632-713) on those distances — but
// laid out for the CDNA4 execution model instead of for a workgroup:
//
//   * every WAVE owns its own tile of RW reads and walks the phases of a tile on its own; there is no
//     workgroup barrier after the tables are loaded, so no wave ever parks behind another one's phase
//     (bdx_bitpar.hip's waves issue during 26 % of their life and wait at barriers for most of the rest);
//     a workgroup is just the unit that shares one copy of the tables in LDS;
//   * read bytes go HBM -> registers -> LDS and are transcoded ARITHMETICALLY on the way
//     ((byte >> 1) & 7 indexes two 8-entry v_perm tables: symbol code and expected byte): the LDS only
//     ever holds a 2-bit image (seed keys) and a 4-bit image (symbol code | "not ACGT" flag) of the tile —
//     no byte image, no 256-byte lookup table;
//   * the seed scan probes a DIRECT bitmap over the 4^q key space at LDS address 0 (4 VALU operations
//     and one LDS byte read per read position);
//   * a sweep fetches its 32 columns as four aligned dwords of the 4-bit image up front; per column the
//     recurrence then needs one bit-field extract, one address add and one LDS read besides its own 10-13
//     operations, and the score is only tracked once a column can end an alignment within the budget
//     (before that it is recovered as popcount(Pv) - popcount(Mv)).
//
// Whatever this kernel cannot answer itself — reads outside the known-score class, reads with more seeded
// barcodes or survivors than its small per-read tables hold, tiles whose bytes do not fit the staging
// area, and (tier 1) reads the settle rule of DESIGN.md §3.4 leaves open — is appended to a list; the
// general kernel (bdx_bitpar.hip, list mode) then evaluates exactly those reads.  Nothing is decided
// differently here: an Inf result never changes the reducer state (classification.jl:658, :696), so
// dropping pairs whose unit distance exceeds the budget is lossless, and the replay is the reference's
// own Float64 code.
#include <atomic>

#include <cstdio>
#include <cstdlib>

#include "bdx_core.h"

// a launcher that refuses its plan says where (stderr, only with BDX_TRACE_LAUNCH set: developer aid)
#define BDX_BAD_PLAN() (getenv("BDX_TRACE_LAUNCH") ? (void)fprintf(stderr, "[bdx] launch refused at %s:%d\n", __FILE__, __LINE__) : (void)0, hipErrorInvalidValue)

namespace {

struct WaveArgs {
    double max_error_rate, min_delta;  // the two doubles of the reducers (classification.jl:632-713)
    int counts_stride2;
    const uint8_t *seq;
    const long long *off;
    long long n_reads;
    BdxDevOut out;
    unsigned long long *counts;
    int hist_entries;
    const uint8_t *bitmap;   // direct bitmap over the 4^q keys
    int bm_bytes;
    const uint16_t *rank;    // [bm_bytes / 4]: keys present below each 32-bit word of the bitmap
    const uint32_t *ent;     // [n_ent]: barcode + 1 | piece start << 11 | next entry with the same key << 16 (0: none); entry i < keys present belongs to the i-th key
    int n_ent;
    const uint32_t *peq8;    // [B][9]: sweep word of barcode b for symbol code c (4..7: "other"; the ninth word pads the stride)
    const uint32_t *peq8r;   // [B][9]: the same for the REVERSED barcode (known-trim class: the sweeps of trim_side = 3 passes run right to left)
    int trim0, trim1;        // known-trim class: the passes' trim sides (0 / 3 / 5)
    int need_tb;             // known-alignment class: the config collects statistics (summary = true): passes without a trim side report positions too (:812)
    BdxDevStats stats;       // known-alignment class: the DemuxStats histograms (rows == 0: none), updated for every pass that returns :match
    const uint32_t *meta;    // [B]: m | kb << 8 | (largest distance the reducer accepts for a lone survivor, 255: none) << 16
    const uint32_t *settle;  // [B]: tier 1, lone survivor: bit d = a read whose only survivor has distance d is settled (no_delta: low half, with_delta: high half)
    int B;
    int q;
    int span_cap;            // bytes of one tile's span the images hold
    int per_wave;            // LDS bytes of one wave's work area
    int tier;                // 1: tier 1 of the tiered budgets (settle rule applies)
    double tier_slo;
    double tier_slo1;        // ... of pass 1 (dual configs)
    int dual;                // two passes (known-score form: the survivors of pass 1 sit in the candidate-word area)
    uint32_t *carry_ent;     // dual tiered configs with min_delta = 0 (else null): see BdxWavePlan::d_carry
    uint32_t *list;          // reads this kernel does not answer ...
    unsigned int *list_count;  // ... and how many
    // split mode (trimming / summary / weighted costs: the exact kernel gives every verdict; this kernel only filters):
    // candidate masks and column windows per pass, as bdx_bitpar.hip's split mode writes them
    int B0;                  // barcodes of pass 0 (barcode numbers of pass 1 follow them: g = B0 + b)
    int cw[2];               // candidate words per pass
    uint32_t *cand_out[2];   // [n_reads][cw]
    uint32_t *wins_out[2];   // [n_reads][BDX_WCAP][3] = {barcode, first column of the restricted run, last column}
    uint8_t *wcnt_out[2];    // [n_reads] entries valid (255: none -> whole window)
    int short_lb[2];         // lookback m + kb instead of 2 (m + kb) + 1
    int sg;                  // :semiglobal (else :hamming / :exact: a window entry's first field is the first START position)
    // pairs mode (KB > 0; two-intact-pieces filter over the reads an earlier tier listed) reads SCATTERED tiles: read k of the
    // launch is read idmap[k] of the batch (NULL: read k), its bytes are fetched straight from the batch into a slot of
    // `slot` flat positions of the tile's images (16-byte multiple; the read starts `head` = its address mod 16 positions into
    // its slot, so that every load is an aligned 16-byte vector)
    int slot;                // flat positions per slot
    int vps, vps_inv;        // slot / 16 and ceil(2^16 / vps)
    int max_len;             // the read length the scan was planned for: longer reads are handed on
    int win_sfe, win_so, win_efe, win_eo;  // window mode: the pass's ref_search_range (start / end: from the read's end?, offset)
    int cpr;                 // 16-diagonal chunks scanned per read
    int cpr_inv;             // ceil(2^16 / cpr)
    int hq_cap, sq_cap;      // entries of the hit queue / the sweep list of a wave's tile
    int ngroups;             // pairs mode: groups of 128 barcodes (one set of piece tables each)
    int scan_gpr, scan_gpr_inv;  // ranged single-pass configs: groups of sixteen positions scanned per read (0: the flat image) and ceil(2^16 / it)
    int cand_area;           // words per read of the area behind the sweep list (candidate masks / survivors of pass 1)
    int ranged;              // some pass has a ref_search_range: per read the column window [first, last] of each pass (classification.jl:795-807)
    BdxDevPass dpass[2];     // the passes' ranges (ranged only)
    const uint32_t *idmap;   // [count] batch read numbers (= the list an earlier tier wrote; NULL: every read of the batch)
    const unsigned int *n_dev;  // the number of listed reads lives on the device (NULL: n_reads)
    int dbg;  // timing experiments (env BDX_DEBUG), compiled in ONLY with -DBDX_TUNING — results are wrong when a skip bit
              // is set: 1 skip verdicts, 2 skip sweeps, 4 skip resolve + emit, 8 skip seed scan, 32 skip transcode, 64 skip loads
};

// The product library has no phase-skip switches: BDX_DBG folds to 0 and the branches disappear.
#ifdef BDX_TUNING
#define BDX_DBG(bit) (a.dbg & (bit))
#else
#define BDX_DBG(bit) 0
#endif

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// 16 raw bytes -> 16 x 2 bits (p2), 16 x 4 bits (nlo: bases 0..7, nhi: bases 8..15) and the sum of absolute
// differences between the bytes and the bytes their 3-bit index stands for (0 <=> every byte is A, C, G, T or N).
// idx = (byte >> 1) & 7:  A 0, C 1, T 2, G 3, N 7;  code = idx for ACGT, 4 ("other") for everything else.
// EXACT: bytes that alias an index (any byte that is not the index's own letter) get code 4 as well.
template <bool EXACT>
__device__ __forceinline__ void pack16(const u32x4 v, uint32_t &p2, uint32_t &nlo, uint32_t &nhi, uint32_t &sad) {
    constexpr uint32_t CODE_LO = 0x03020100u, CODE_HI = 0x04040404u;  // idx 0..3 -> 0..3, 4..7 -> 4
    constexpr uint32_t EXP_LO = 0x47544341u /* G T C A */, EXP_HI = 0x4E000000u /* idx 7: N */;
    uint32_t u[4], t2[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = v[w];
        const uint32_t sel = (x >> 1) & 0x07070707u;
        uint32_t n4 = __builtin_amdgcn_perm(CODE_HI, CODE_LO, sel);
        const uint32_t e = __builtin_amdgcn_perm(EXP_HI, EXP_LO, sel);
        if (EXACT) {
            const uint32_t d = x ^ e;
            const uint32_t y = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;  // 0x80 in every byte that differs
            const uint32_t m = (y >> 7) * 0xFFu;
            n4 = (n4 & ~m) | (m & 0x04040404u);
        } else {
            sad = __builtin_amdgcn_sad_u8(x, e, sad);
        }
        u[w] = n4 | (n4 >> 4);  // bytes 0 and 2: two 4-bit codes each
        // 2-bit codes (x >> 1) & 3 of the four bytes gathered into the top byte
        const uint32_t t6 = (x & 0x06060606u) << 5;
        const uint32_t a = t6 | (t6 << 6);
        t2[w] = a | (a << 12);
    }
    nlo = __builtin_amdgcn_perm(u[1], u[0], 0x06040200u);
    nhi = __builtin_amdgcn_perm(u[3], u[2], 0x06040200u);
    const uint32_t lo = __builtin_amdgcn_perm(t2[1], t2[0], 0x0C0C0703u);
    const uint32_t hi = __builtin_amdgcn_perm(t2[3], t2[2], 0x07030C0Cu);
    p2 = lo | hi;
}

// One column of Myers' recurrence on a top-aligned pattern (bdx_bitpar.hip `step`); TRACK: the horizontal delta
// of the barcode's last row is the carry-out of the shift and updates the score.
template <bool TRACK>
__device__ __forceinline__ uint32_t sweep_step(const uint32_t Eq, uint32_t &Pv, uint32_t &Mv, int &score, int &best) {
    const uint32_t Xv = Eq | Mv;
    const uint32_t ep = Eq & Pv;  // (returned: its top bit is the known-start class's "the diagonal move into the last row is optimal")
    const uint32_t Xh = ((ep + Pv) ^ Pv) | Eq;
    uint32_t Ph = Mv | ~(Xh | Pv);
    uint32_t Mh = Pv & Xh;
    if (TRACK) {
        uint32_t cp, cm;
        Ph = __builtin_addc(Ph, Ph, 0u, &cp);
        Mh = __builtin_addc(Mh, Mh, 0u, &cm);
        score += (int)cp;
        score -= (int)cm;
    } else {
        Ph = Ph + Ph;
        Mh = Mh + Mh;
    }
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    if (TRACK) best = score < best ? score : best;
    return ep;
}

// 32 columns of one sweep; columns [0, TF) cannot end an alignment within any barcode's budget (the score after
// j + 1 columns is >= m - (j + 1)), so the score is first needed at column TF, where it is popcount(Pv) -
// popcount(Mv): D[m][j] = D[0][j] + the vertical deltas, D[0][j] = 0 (free start), the virtual rows below the
// barcode carry no delta.
// Eight LDS reads issued back to back (the compiler, left alone, keeps one or two in flight and waits in front of
// every use): inline asm for the loads and for the wait, which names the destinations so that nothing that uses
// them moves above it.  Waits are always lgkmcnt(0), which is correct whatever else is outstanding.
__device__ __forceinline__ void lds_read8(uint32_t (&d)[8], const uint32_t (&addr)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(d[i]) : "v"(addr[i]) : "memory");
}
__device__ __forceinline__ void lds_wait8(uint32_t (&d)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                 :
                 : "memory");
}

// TRACKW: also note which columns have a score within the budget (split mode): bit 31 - j of `inm` for column j of the
// block (kk1 = budget + 1: the sign bit of score - kk1 is shifted in).
// TRACKW >= 2 (known-trim class): bit 31 - j instead says "column j lowered the running minimum" — the last such column of a
// sweep is the FIRST column that attains its minimum (the reference keeps the leftmost end of the best score,
// classification.jl:142-153 with trim_side = 5: strict `<`).  TRACKW = 3: `inm2` (same bit numbering) also notes, per column, the top
// bit of Eq & Pv BEFORE the step: "the barcode's last row matches this column and its vertical delta was +1", i.e. the
// diagonal move into the last row attains the column's value (needed by the reversed sweeps of trim_side = 3 passes, see
// sweep_lane: there the last row is the barcode's FIRST base).
template <int TF, int TRACKW>
__device__ __forceinline__ void sweep_block(const uint32_t A0, const uint32_t A1, const uint32_t A2, const uint32_t A3,
                                            const uint32_t pbase, uint32_t &Pv, uint32_t &Mv, int &score, int &best, const int kk1,
                                            uint32_t &inm, uint32_t &inm2, const int ngr) {
    const uint32_t A[4] = {A0, A1, A2, A3};
    uint32_t Eq[2][8];
    const auto issue = [&](const int h) __attribute__((always_inline)) {
        uint32_t addr[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) addr[jj] = pbase + (__builtin_amdgcn_ubfe(A[h], 4 * jj, 3) << 2);
        lds_read8(Eq[h & 1], addr);
    };
    issue(0);
    inm = 0u;
    inm2 = 0u;
#pragma unroll
    for (int h = 0; h < 4; ++h) {  // the Eq words of the next eight columns fly while these eight are worked on
        if (h >= ngr) break;       // (wave-uniform: no lane has a column in the remaining groups of eight)
        lds_wait8(Eq[h & 1]);
        if (h < 3 && h + 1 < ngr) issue(h + 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = 8 * h + jj;
            if (j < TF) {
                sweep_step<false>(Eq[h & 1][jj], Pv, Mv, score, best);
            } else {
                if (j == TF && TF > 0) score = __builtin_popcount(Pv) - __builtin_popcount(Mv);
                const int best_before = best;
                const uint32_t ep = sweep_step<true>(Eq[h & 1][jj], Pv, Mv, score, best);
                if (TRACKW == 1) inm = __builtin_amdgcn_alignbit(inm, (uint32_t)(score - kk1), 31);  // (inm << 1) | (score <= budget)
                if (TRACKW >= 2) inm = __builtin_amdgcn_alignbit(inm, (uint32_t)(score - best_before), 31);  // (inm << 1) | (score < minimum so far)
                if (TRACKW >= 3) inm2 = __builtin_amdgcn_alignbit(inm2, ep, 31);
            }
        }
    }
    if (TRACKW && ngr < 4) inm <<= 32 - 8 * ngr;  // (bit 31 - j stands for column j also when the block stopped early)
    if (TRACKW >= 3 && ngr < 4) inm2 <<= 32 - 8 * ngr;
}

// 32 columns of an ANCHORED sweep (known-alignment class, KEND = 3): one end of the alignment is fixed, the sweep looks for the
// first column whose score EQUALS the known distance d: bit 31 - j of `eqm` for column j; `inm2` as in sweep_block.
// REVA: the sweep runs right to left from the alignment's end column with the reversed barcode and its row 0 is NOT free — the
// barcode's words are stripped of the "virtual rows below the barcode match everything" bits (rows) and the horizontal delta of
// row 0 is +1 (lowbit is shifted in): the score of a column is the cost of aligning the whole barcode with exactly the positions
// from there to the anchor.  !REVA: an ordinary left-to-right sweep from a prepared first column.
template <bool REVA>
__device__ __forceinline__ void anchored_block(const uint32_t A0, const uint32_t A1, const uint32_t A2, const uint32_t A3, const uint32_t pbase,
                                               const uint32_t rows, const uint32_t lowbit, uint32_t &Pv, uint32_t &Mv, int &score, const int d,
                                               uint32_t &eqm, uint32_t &inm2, const int ngr) {
    const uint32_t A[4] = {A0, A1, A2, A3};
    uint32_t Eq[2][8];
    const auto issue = [&](const int h) __attribute__((always_inline)) {
        uint32_t addr[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) addr[jj] = pbase + (__builtin_amdgcn_ubfe(A[h], 4 * jj, 3) << 2);
        lds_read8(Eq[h & 1], addr);
    };
    issue(0);
    eqm = 0u;
    inm2 = 0u;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (h >= ngr) break;
        lds_wait8(Eq[h & 1]);
        if (h < 3 && h + 1 < ngr) issue(h + 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const uint32_t E = REVA ? (Eq[h & 1][jj] & rows) : Eq[h & 1][jj];
            const uint32_t Xv = E | Mv;
            const uint32_t ep = E & Pv;
            const uint32_t Xh = ((ep + Pv) ^ Pv) | E;
            uint32_t Ph = Mv | ~(Xh | Pv);
            uint32_t Mh = Pv & Xh;
            uint32_t cp, cm;
            Ph = __builtin_addc(Ph, Ph, 0u, &cp);
            Mh = __builtin_addc(Mh, Mh, 0u, &cm);
            score += (int)cp;
            score -= (int)cm;
            if (REVA) Ph |= lowbit;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
            eqm = __builtin_amdgcn_alignbit(eqm, score == d ? 0x80000000u : 0u, 31);
            inm2 = __builtin_amdgcn_alignbit(inm2, ep, 31);
        }
    }
    if (ngr < 4) {
        eqm <<= 32 - 8 * ngr;
        inm2 <<= 32 - 8 * ngr;
    }
}

// NV: 16-byte vectors of a tile's span per lane (the next tile's bytes wait in 4 NV registers while this tile is worked
// on); Q: seed length.
// KB > 0: PAIRS mode — the filter is the two-intact-pieces lemma instead of single seeds (budgets up to KB, see the
// scan below), the input is a gathered slot buffer, every flagged (barcode, diagonal run) is one sweep (no record
// tables); NW: words of a barcode mask (table entries of 8 bytes for NW <= 2, else 16).
// MG: pairs mode with more than 128 barcodes (groups of 128; the queue is drained inside the scan).
// KEND: known-trim class (ScoreOnly conditions with any trim side per pass, no per-pass start positions wanted): survivors carry
// what their pass's trim side makes observable — trim_side = 5: the first column of the minimum (the reference's end,
// classification.jl:142-153, :912-914); trim_side = 3: the LARGEST origin among the optimal alignments (the reference's start,
// :142-153 tie rule + :310-321 origin order, :910-911), delivered by sweeping the window right to left with the reversed barcode
// (sweep_lane) — and the replay trims with them.
// GEN: the general form — dual configs and ref_search_range windows; false: single pass over whole reads (the headline
// configuration: those checks are compiled out).
// WINM: WINDOW mode of the seeded kernel — a single-pass config whose ref_search_range window is much shorter than its reads
// (BASELINE config 5: 10 kbp reads, window 1:200): the tile is scattered like a pairs-mode tile, only each read's resolved
// column window (classification.jl:795-807) is fetched into its slot, and everything downstream sees the window as the read
// (what the verdict needs of the real read — its length, its number — rides along).
#ifndef BDX_WAVE_BOUNDS  // (tuning: the occupancy experiment of DESIGN §4 compiles the kernels for more waves per SIMD)
#define BDX_WAVE_BOUNDS __launch_bounds__(1024)
#endif
template <int RW, int TF, int NV, int Q, bool SPLIT, int KB, int NW, bool MG = false, int KEND = 0, bool GEN = true, bool WINM = false>
__global__ BDX_WAVE_BOUNDS void bdx_wave_kernel(const WaveArgs a) {
    constexpr bool PAIRS = KB > 0;
    constexpr bool KREV = KEND >= 2;  // known-trim class with a trim_side = 3 pass: reversed sweeps (1: trim sides 5 / none only — the sweeps of round 3's known-end class)
    constexpr bool KALN = KEND == 3;  // known-alignment class: start AND end of every pass's winner (anchored sweeps), for per-pass outputs and the statistics tables
    constexpr int RCAP = 8;       // sweep records (seeded barcode x diagonal cluster) per read
    // seed hits per tile (pairs mode: flagged (barcode, run of diagonals)s = sweeps) / sweeps (= records) per tile: sized per
    // config from the expected chance hits (size_wave) — the two queues sit behind the images, at run-time offsets
    const int HQ = a.hq_cap, SQ = PAIRS ? 0 : a.sq_cap;
    constexpr int NREC = PAIRS ? 0 : RW * RCAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LDS unsigned char *smem = (LDS unsigned char *)smem_raw;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: tile numbers and their geometry live in scalar registers)
    const int B = a.B;
    constexpr int q = Q;  // seed length
    // (pairs mode: the number of gathered reads is only known on the device)
    const bool ranged = GEN && a.ranged != 0, dual = GEN && a.dual != 0;
    static_assert(!WINM || (!PAIRS && !SPLIT && KEND == 0 && GEN), "window mode: the non-split single-seed kernel");
    constexpr bool SCAT = PAIRS || WINM;  // scattered tiles: every read of a tile is fetched on its own (by list index) into a slot of the images
    // Carried passes (dual tiered known-class configs with min_delta = 0): tier 1 lists a read when ONE of its passes is open; the pass
    // it did settle travels with the read — two state bits on the list entry (1: pass 0 settled and matched, 2: pass 1 settled
    // on its own) and the pass's winning survivor entry in carry_ent[read] — and the pairs mode drops that pass's barcodes from
    // its flags and replays the carried survivor instead: about half of its sweeps for C4.
    constexpr bool CARRY = PAIRS && !SPLIT && !MG && KB <= 4;
    const uint32_t idmask = (CARRY && a.carry_ent != nullptr) ? 0x3FFFFFFFu : 0xFFFFFFFFu;
    const long long n_reads = (SCAT && a.n_dev) ? (long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)*a.n_dev) : a.n_reads;

    // ---- LDS carve-up: shared tables, then one work area per wave ----
    size_t o = 0;
    auto take = [&](size_t bytes) -> LDS unsigned char * {
        LDS unsigned char *p = smem + o;
        o = (o + bytes + 31) & ~(size_t)31;
        return p;
    };
    LDS unsigned char *bm = take((size_t)a.bm_bytes);  // LDS address 0: a probe's address is its byte index
    LDS uint16_t *rnk = (LDS uint16_t *)take(PAIRS ? 0 : (size_t)a.bm_bytes / 2);
    LDS uint32_t *ent = (LDS uint32_t *)take((size_t)a.n_ent * 4);
    LDS uint32_t *peq = (LDS uint32_t *)take((size_t)B * 36);  // 9 dwords per barcode: (9 b + code) mod 32 spreads over every bank
    LDS uint32_t *peqr = (LDS uint32_t *)take(KREV ? (size_t)B * 36 : 0);  // the reversed barcodes' words (known-trim class)
    LDS uint32_t *meta = (LDS uint32_t *)take((size_t)B * 4);
    LDS uint32_t *settle = (LDS uint32_t *)take((size_t)B * 4);
    LDS int *hist = (LDS int *)take((size_t)a.hist_entries * 4);
    // per-wave work area: the arrays whose size only depends on RW come first, at compile-time offsets from the area's
    // base (one base register + immediate offsets in the DS instructions), the two images after them
    LDS unsigned char *wbase = smem + o + (size_t)wv * (size_t)a.per_wave;
    constexpr int O_FB = 0;                                  // int[RW + 1]: flat index of every read's first base
    constexpr int O_RID = O_FB + ((RW + 1) * 4 + 15) / 16 * 16;  // u32[RW * RCAP]: sweep records: barcode + 1 | (first diagonal + 64) << 16
    constexpr int O_RMK = O_RID + NREC * 4;                  // u32[RW * RCAP]: diagonals seen, as bits: diagonal - first + kb
    constexpr int O_SLOTS = O_RMK + NREC * 4;                // u32[RW * 4]: survivors: barcode << 8 | d
    constexpr int O_SCNT = O_SLOTS + RW * 16;                // int[RW]
    constexpr int O_FLAG = O_SCNT + RW * 4;                  // int[RW]: read goes to the list
    constexpr int O_WCL1 = O_FLAG + RW * 4;                  // int[RW]: split mode: window entries written for pass 1 (pass 0: scnt)
    constexpr int O_LBUF = O_WCL1 + RW * 4;                  // u32[64]: reads for the list, flushed in batches
    constexpr int O_RL = O_LBUF + 64 * 4;                    // int[RW]: scattered tiles: read (window) lengths (the slots are longer)
    constexpr int O_GID = O_RL + (SCAT ? RW * 4 : 0);        // u32[RW]: scattered tiles: batch read numbers of the tile's reads
    constexpr int O_TN = O_GID + (SCAT ? RW * 4 : 0);        // int[RW]: window mode: the reads' true lengths
    // scattered tiles: slot geometry of this tile and the next (double-buffered: written when a tile's bytes are requested, one
    // tile ahead) — u32x4 {aligned address lo, hi, vectors to fetch | head << 8, length (-1: handed on)}, read number, true length
    constexpr int O_SG4 = O_TN + ((WINM || PAIRS) ? RW * 4 : 0);  // u32x4[2][RW]   (pairs mode: tn = which pass of the read tier 1 settled, stn = that pass's winning survivor)
    constexpr int O_SGID = O_SG4 + (SCAT ? 2 * RW * 16 : 0); // u32[2][RW]
    constexpr int O_STN = O_SGID + (SCAT ? 2 * RW * 4 : 0);  // int[2][RW] (window mode)
    constexpr int O_IMG2 = O_STN + ((WINM || PAIRS) ? 2 * RW * 4 : 0) + (SCAT ? 16 : 0);  // u32[nvec_cap + 2]: 2-bit image (scattered tiles: four guard words in front)
    const int nvec_cap = a.span_cap >> 4;
    LDS int *fb = (LDS int *)(wbase + O_FB);
    LDS uint32_t *rid = (LDS uint32_t *)(wbase + O_RID);
    LDS uint32_t *rmk = (LDS uint32_t *)(wbase + O_RMK);
    LDS uint32_t *slots = (LDS uint32_t *)(wbase + O_SLOTS);
    LDS int *scnt = (LDS int *)(wbase + O_SCNT);
    LDS int *flag = (LDS int *)(wbase + O_FLAG);
    LDS int *wcl1 = (LDS int *)(wbase + O_WCL1);
    LDS uint32_t *lbuf = (LDS uint32_t *)(wbase + O_LBUF);
    LDS int *rl = (LDS int *)(wbase + O_RL);
    LDS uint32_t *gid = (LDS uint32_t *)(wbase + O_GID);
    LDS int *tn = (LDS int *)(wbase + O_TN);
    LDS u32x4 *sg4 = (LDS u32x4 *)(wbase + O_SG4);
    LDS uint32_t *sgid = (LDS uint32_t *)(wbase + O_SGID);
    LDS int *stn = (LDS int *)(wbase + O_STN);
    LDS uint32_t *img2 = (LDS uint32_t *)(wbase + O_IMG2);
    LDS uint32_t *img4 = img2 + ((nvec_cap + 2 + 3) & ~3);  // u32[2 nvec_cap + 6]: 4-bit image (16-byte aligned)
    LDS uint32_t *hq = img4 + ((2 * nvec_cap + 6 + 3) & ~3);  // u32[HQ]: seed hits: flat position << 16 | key (pairs mode: sweep entries)
    LDS uint32_t *recq = hq + HQ;                             // u32[SQ]: the tile's records in use (slot numbers) = its sweeps
    const int cwt = a.cw[0] + a.cw[1];                        // split mode: candidate mask words per read (pass 0 then pass 1)
    LDS uint32_t *cand = recq + SQ;                           // u32[RW][cwt] (split mode)
    // ranged configs: per read and pass the 0-based first column and the last column (1-based = exclusive end) of the window
    LDS int *wwin = (LDS int *)(cand + RW * a.cand_area);     // int[4][RW]: first0, last0, first1, last1
    const auto win_lo = [&](const int t, const bool second) -> int { return ranged ? wwin[(second ? 2 : 0) * RW + t] : 0; };
    const auto win_hi = [&](const int t, const bool second, const int n) -> int { return ranged ? wwin[(second ? 3 : 1) * RW + t] : n; };

    // ---- tables -> LDS (the only workgroup barrier of the kernel besides the final histogram flush) ----
    for (int i = tid; i < a.bm_bytes / 4; i += blockDim.x) ((LDS uint32_t *)bm)[i] = ((const uint32_t *)a.bitmap)[i];
    if (!PAIRS) {  // (pairs mode: `bm` holds the piece tables)
        for (int i = tid; i < a.bm_bytes / 8; i += blockDim.x) ((LDS uint32_t *)rnk)[i] = ((const uint32_t *)a.rank)[i];
        for (int i = tid; i < a.n_ent; i += blockDim.x) ent[i] = a.ent[i];
    }
    if (SCAT && lane < 4) img2[lane - 4] = 0u;  // guard words in front of the first slot
    for (int i = tid; i < B * 9; i += blockDim.x) peq[i] = a.peq8[i];
    if (KREV)
        for (int i = tid; i < B * 9; i += blockDim.x) peqr[i] = a.peq8r[i];
    for (int i = tid; i < B; i += blockDim.x) {
        meta[i] = a.meta[i];
        settle[i] = a.settle[i];
    }
    for (int i = tid; i < a.hist_entries; i += blockDim.x) hist[i] = 0;
    __syncthreads();

    const uint32_t peq_base = (uint32_t)(uintptr_t)peq;
    const uint32_t peqr_base = (uint32_t)(uintptr_t)peqr;
    const int ntiles = (int)((n_reads + RW - 1) / RW);  // (< 2^29: a batch holds fewer than 2^32 reads)

    // Tiles are dealt round robin over all waves of the grid (tile = wave + k x waves): no queue, no atomics.  The
    // bytes of tile k + 1 are requested while tile k is worked on, its offsets one tile earlier still, so the HBM
    // latency of neither is ever at the head of a tile.
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));  // (32-bit: tile numbers stay in scalar registers)
    int tile = (int)(blockIdx.x * (blockDim.x >> 6)) + wv;
    // geometry of a tile's span from its offsets (lanes 0 .. nr hold off[r0 + lane]); everything wave-uniform
    struct Geo {
        long long span0;
        uintptr_t g0a;
        int head, total, nvec, nr;
        bool ok;
    };
    const auto geometry = [&](const int t, const long long ov) -> Geo {
        Geo g;
        const long long r0 = (long long)t * RW;
        g.nr = (int)(n_reads - r0 < RW ? n_reads - r0 : RW);
        const uint32_t ov_lo = (uint32_t)ov, ov_hi = (uint32_t)(ov >> 32);
        g.span0 = (long long)(((unsigned long long)__builtin_amdgcn_readlane(ov_hi, 0) << 32) | __builtin_amdgcn_readlane(ov_lo, 0));
        const long long span1 = (long long)(((unsigned long long)__builtin_amdgcn_readlane(ov_hi, g.nr) << 32) | __builtin_amdgcn_readlane(ov_lo, g.nr));
        const uintptr_t g0 = (uintptr_t)(a.seq + g.span0);
        g.g0a = g0 & ~(uintptr_t)15;
        g.head = (int)(g0 - g.g0a);
        const long long need = (span1 - g.span0) + g.head;
        g.ok = need + 16 <= (long long)a.span_cap && need <= (long long)NV * 1024;  // wave-uniform
        g.total = g.ok ? (int)need : 0;  // flat bases of the tile (head included)
        g.nvec = (g.total + 15) >> 4;
        return g;
    };
    // Scattered tiles (SCAT): lane t < RW owns read t of the tile: its batch number (from the list, three tiles ahead), its
    // offset and length (two tiles ahead), and from those the 16-byte aligned address its slot is filled from; the bytes are
    // requested one tile ahead like those of a contiguous tile, every lane fetching vector j of read t (k = t * vps + j).
    typedef long long ll2a __attribute__((ext_vector_type(2), aligned(8)));
    struct ScatOff {
        long long o0;
        int len;
    };
    const auto load_scat = [&](const int t, const uint32_t id) -> ScatOff {
        const long long r0 = (long long)t * RW;
        ScatOff so{0, -1};
        if (SCAT && t < ntiles && r0 + lane < n_reads && lane < RW) {
            const ll2a o = *(const ll2a __attribute__((address_space(1))) *)(a.off + (CARRY ? (id & idmask) : id));
            so.o0 = o[0];
            const long long l = o[1] - o[0];
            so.len = (l >= 0 && l < (1LL << 30)) ? (int)l : -1;
        }
        return so;
    };
    // per-lane slot geometry of read `lane` of a scattered tile: aligned base address, head, vectors to fetch (0: none — the
    // read is longer than its slot, or there is no such read)
    struct ScatGeo {
        uintptr_t abase;
        int head, nv, len;  // len: bases of the slot's read (window mode: of its column window); -1: not answered by this kernel
    };
    const auto scat_geo = [&](const ScatOff &so) -> ScatGeo {
        ScatGeo g{0, 0, 0, -1};
        if (so.len >= 0) {
            int wlo = 0, wlen = so.len;
            bool ok = true;
            if (WINM) {
                // the read's column window (classification.jl:795-807) in 32-bit arithmetic: this kernel only runs for configs whose
                // barcode_start / barcode_end ranges are the whole read (build_wave_tables) and whose ref_search_range offsets are
                // small (the launcher checks), so  first = max(s, 1), last = min(e, n) (empty: last = first - 1, Julia's
                // normalisation, resolve :96-100), max_start_pos = n, min_end_pos = 1, and the :805 sanity check is
                // first <= last (first <= n and last >= 1 follow).  An empty read is not in the known-score class.
                const int n = so.len;
                const int sx = a.win_sfe ? n + a.win_so : a.win_so, ex = a.win_efe ? n + a.win_eo : a.win_eo;
                const int first = sx > 1 ? sx : 1;
                int last = ex < n ? ex : n;
                if (last < first) last = first - 1;
                ok = n > 0 && first <= last;
                wlo = ok ? first - 1 : 0;
                wlen = ok ? last - first + 1 : 0;
            }
            const uintptr_t ad = (uintptr_t)a.seq + (uintptr_t)so.o0 + (uintptr_t)wlo;
            g.head = (int)(ad & 15);
            g.abase = ad - (uintptr_t)g.head;
            const int nv = (g.head + wlen + 15) >> 4;
            if (ok && nv <= a.vps && (WINM || so.len <= a.max_len)) {  // (pairs mode: the scan covers the diagonals of reads up to max_len)
                g.nv = nv;
                g.len = wlen;
            }
        }
        return g;
    };
    const auto load_offsets = [&](const int t) -> long long {  // (t < 2^30 also when it runs past the last tile)
        const long long r0 = (long long)t * RW;
        const long long left = n_reads - r0;
        const int cnt = t < ntiles ? (int)(left < RW ? left : RW) : -1;  // lanes 0 .. cnt load
        if (SCAT) return 0;  // (scattered tiles: load_scat)
        // wave-uniform base in scalar registers + a 32-bit lane offset
        const uintptr_t bp = (uintptr_t)(a.off + r0);
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)bp), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
        const long long *base = (const long long *)(((uintptr_t)bhi << 32) | blo);
        return lane <= cnt ? base[lane] : 0;
    };
    // scattered tiles: batch read number of the tile's reads (lanes 0 .. cnt - 1)
    const auto load_gid = [&](const int t) -> uint32_t {
        const long long r0 = (long long)t * RW;
        return (SCAT && t < ntiles && r0 + lane < n_reads && lane < RW) ? (a.idmap ? a.idmap[r0 + lane] : (uint32_t)(r0 + lane)) : 0u;
    };
    const auto rlen = [&](const int t) -> int { return SCAT ? rl[t] : fb[t + 1] - fb[t]; };
    u32x4 v[NV];
    LDS uint32_t *const img2_lane = img2 + lane;      // (one base register each: the unrolled stores differ by immediates)
    LDS uint32_t *const img4_lane = img4 + 2 * lane;
    const auto load_bytes = [&](const Geo &g) {
        const GlobalVec16 src = (GlobalVec16)g.g0a;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int k = 64 * u + lane;
            if (k < g.nvec && !BDX_DBG(64)) v[u] = __builtin_nontemporal_load(src + k);
        }
    };
    // scattered tile: vector k of the images is vector j = k - t vps of read t = k / vps; its address comes from lane t
    // (ds_bpermute); vectors a read does not reach are filled with 'N' (no barcode symbol, no seed of interest)
    // The slot geometry of a tile is worked out ONCE, by lane t for read t, when the tile's bytes are requested, and left in LDS
    // (buffer `par`): the loads below read it from there (every lane its own read's entry), the tile itself one stage later.
    const auto stash_scat = [&](const int par, const ScatOff &so, const uint32_t id) {
        if (lane < RW) {
            const ScatGeo sg = scat_geo(so);
            sg4[par * RW + lane] = u32x4{(uint32_t)sg.abase, (uint32_t)((unsigned long long)sg.abase >> 32), (uint32_t)sg.nv | ((uint32_t)sg.head << 8), (uint32_t)sg.len};
            sgid[par * RW + lane] = id;
            if (WINM) stn[par * RW + lane] = so.len;
            if (CARRY) stn[par * RW + lane] = (a.carry_ent != nullptr && (id >> 30) != 0u) ? (int)a.carry_ent[id & idmask] : 0;
        }
        WAVE_SYNC();
    };
    const auto load_bytes_scat = [&](const int par, const int nr_t) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int k = 64 * u + lane;
            const int t = (int)(((uint32_t)k * (uint32_t)a.vps_inv) >> 16);
            const int j = k - t * a.vps;
            const u32x4 e = sg4[par * RW + (t < RW ? t : 0)];
            u32x4 x = {0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu};
            if (t < nr_t && j < (int)(e[2] & 255u) && !BDX_DBG(64)) x = __builtin_nontemporal_load((GlobalVec16)((((unsigned long long)e[1] << 32) | e[0]) + 16ull * (unsigned)j));
            v[u] = x;
        }
    };
    for (int i = lane; i < RW * RCAP; i += 64) {  // (a sweep clears its record: the tables are empty at the top of every tile)
        rid[i] = 0u;
        rmk[i] = 0u;
    }
    int lcnt = 0;  // entries waiting in lbuf (wave-uniform)
    const auto flush_list = [&]() {
        if (lcnt > 0) {
            WAVE_SYNC();
            unsigned int basek = 0;
            if (lane == 0) basek = atomicAdd(a.list_count, (unsigned int)lcnt);
            basek = (unsigned int)__builtin_amdgcn_readfirstlane((int)basek);
            if (lane < lcnt) a.list[basek + lane] = lbuf[lane];
            WAVE_SYNC();
            lcnt = 0;
        }
    };
    long long ov = load_offsets(tile);
    Geo geo{};
    // scattered tiles: read numbers of this tile and the next three, offsets of this tile and the next two
    uint32_t iv_next = load_gid(tile + nwaves), iv_after = load_gid(tile + 2 * nwaves);
    ScatOff so_next = load_scat(tile + nwaves, iv_next);
    int par = 0;  // the stash buffer of the current tile (wave-uniform)
    const auto scat_tile = [&](const int t) -> Geo {  // the wave-uniform part of a scattered tile's geometry
        Geo g{};
        const long long r0 = (long long)t * RW;
        g.nr = (int)(n_reads - r0 < RW ? n_reads - r0 : RW);
        g.ok = true;
        g.nvec = g.nr * a.vps;
        g.total = g.nvec << 4;
        return g;
    };
    if (tile < ntiles) {
        if (SCAT) {
            geo = scat_tile(tile);
            const uint32_t iv0 = load_gid(tile);
            stash_scat(0, load_scat(tile, iv0), iv0);
            load_bytes_scat(0, geo.nr);
        } else {
            geo = geometry(tile, ov);
            load_bytes(geo);
        }
    }
    long long ov_next = load_offsets(tile + nwaves);

    while (tile < ntiles) {
        const long long r0 = (long long)tile * RW;
        const int nr = geo.nr;
        const bool tile_ok = geo.ok;
        const int total = geo.total, nvec = geo.nvec;

        // ---- per-tile tables ----
        if (!SCAT && lane <= nr) fb[lane] = tile_ok ? geo.head + (int)(ov - geo.span0) : 0;
        if (lane < RW) {
            scnt[lane] = 0;
            flag[lane] = 0;
            if (SCAT) {
                const u32x4 e = sg4[par * RW + lane];
                const int slen = (int)e[3];
                fb[lane] = lane * a.slot + (int)(e[2] >> 8);  // the read's first base within the tile's flat images
                rl[lane] = lane < nr ? slen : 0;              // (-1: longer than its slot — handed on)
                gid[lane] = sgid[par * RW + lane] & idmask;
                if (CARRY) tn[lane] = a.carry_ent != nullptr ? (int)(sgid[par * RW + lane] >> 30) : 0;
                if (WINM) {
                    tn[lane] = stn[par * RW + lane];
                    if (lane < nr && slen < 0) flag[lane] = 1;  // (not in the known-score class, or a window longer than planned: listed)
                }
            }
            wcl1[lane] = 0;  // (split mode: window entries of pass 1; known-score dual configs: survivors of pass 1)
            if (CARRY && lane < nr) {  // a pass tier 1 settled: its winning survivor is the pass's only entry (its barcodes are dropped from the flags below)
                const int st = tn[lane];
                if (st == 1) {
                    slots[lane * 4] = (uint32_t)stn[par * RW + lane];
                    scnt[lane] = 1;
                } else if (st == 2) {
                    cand[lane * 4] = (uint32_t)stn[par * RW + lane];
                    wcl1[lane] = 1;
                }
            }
            if (SPLIT)
                for (int w = 0; w < cwt; ++w) cand[lane * cwt + w] = 0u;
        }

        // ---- bytes (requested one tile ago): registers -> 2-bit / 4-bit images ----
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int k = 64 * u + lane;
            if (64 * u < nvec) {  // wave-uniform
                uint32_t p2 = 0, nlo = 0, nhi = 0, sad = 0;
                if (k < nvec && !BDX_DBG(32)) pack16<false>(v[u], p2, nlo, nhi, sad);
                if (__builtin_amdgcn_ballot_w64(sad != 0)) {  // some byte is neither A, C, G, T nor N (rare)
                    if (sad != 0) pack16<true>(v[u], p2, nlo, nhi, sad);
                }
                if (k < nvec) {
                    img2_lane[64 * u] = p2;
                    *(LDS u32x2 *)(img4_lane + 128 * u) = u32x2{nlo, nhi};
                }
            }
        }
        // ---- request the next tile's bytes and the offsets of the one after ----
        const int tile_next = tile + nwaves;
        Geo geo_next{};
        if (tile_next < ntiles) {
            if (SCAT) {
                geo_next = scat_tile(tile_next);
                stash_scat(par ^ 1, so_next, iv_next);
                load_bytes_scat(par ^ 1, geo_next.nr);
            } else {
                geo_next = geometry(tile_next, ov_next);
                load_bytes(geo_next);
            }
        }
        const long long ov_after = load_offsets(tile_next + nwaves);
        const ScatOff so_after = load_scat(tile_next + nwaves, iv_after);
        const uint32_t iv_after2 = load_gid(tile_next + 2 * nwaves);
        WAVE_SYNC();

        if (ranged && lane < RW) {
            // the passes' column windows for this read; a read outside the known-score class (a binding start / end range,
            // the :805 sanity check) goes to the list — in split mode the exact kernel decides anyway, an empty window
            // just leaves the read without candidates
            const int n = lane < nr ? rlen(lane) : 0;
            bool known = true;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                PassWindow w{1, 0, 0, 0};
                const bool ok = (p == 0 || a.B0 < B) && lane < nr && pass_window(a.dpass[p], n, w);
                wwin[(2 * p) * RW + lane] = ok ? w.first - 1 : 0;
                wwin[(2 * p + 1) * RW + lane] = ok ? w.last : 0;
                if (p == 0 || a.B0 < B) known = known && ok && n > 0 && w.max_start >= n && w.min_end <= 1;
            }
            if (!SPLIT && !known && lane < nr) flag[lane] = 1;
        }
        if (ranged) WAVE_SYNC();

        // uniform read length of the tile (0: mixed) for the hit -> read mapping
        int ulen = 0;
        {
            const int my = (!SCAT && lane < nr) ? fb[lane + 1] - fb[lane] : 0;
            const int l0 = __builtin_amdgcn_readfirstlane(my);
            ulen = (l0 > 0 && !__builtin_amdgcn_ballot_w64(lane < nr && my != l0)) ? l0 : 0;
        }

        // ---- sweeps: lane = one record = one (read, barcode, window) ----
        // (a lambda: split mode runs it a second time for the reads whose tables overflowed, below)
        const auto sweep_lane = [&](bool valid, const int t, const int b, const int lo, const int hi) __attribute__((always_inline)) {
            const uint32_t mt = meta[b];
            const int mm = (int)(mt & 255u), kk = (int)((mt >> 8) & 255u);
            valid = valid && hi > lo && !BDX_DBG(2);
            const int ncol = valid ? hi - lo : 0;
            uint32_t Pv = mm >= 32 ? 0xFFFFFFFFu : (((1u << mm) - 1u) << (32 - mm));
            uint32_t Mv = 0;
            int score = mm, best = 0x7FFFFFFF;
            // Known-trim class, trim_side = 3 passes: the sweep runs RIGHT TO LEFT over the window with the reversed barcode.  After
            // the column of 0-based read position p its score is the smallest cost of an alignment that leaves row 0 at node (0, p)
            // (consumes the read from position p on); the reference's start is the largest origin — the column at which row 1 is
            // entered — among the alignments of the best score: its origin rule (deletion, then substitution if strictly less, then
            // insertion, classification.jl:310-321) walks back along the rightmost optimal path, and its recording rule keeps the best
            // score's largest start (:142-153 with trim_side = 3; no early exit on a zero, :420-430).  That is  p* + s :  p* = the
            // largest node an optimal alignment leaves from = the column that lowered the running minimum LAST in sweep order, s = 1
            // iff the diagonal move is optimal there (top bit of Eq & Pv before the step) — a vertical first move enters row 1 at
            // column p* itself; if p* is the pass window's first column the alignment comes out of the reference's initial column,
            // whose origins are 1 - i <= 0 (:278-283): reported as 0 (keep_end = max(1, start) - 1 = 0 either way, :910-911).
            // Model + proof by enumeration: oracle orc_known_start / orc_selftest_known_start.
            const bool second_b = GEN && b >= a.B0;
            const int trim_b = KEND ? (second_b ? a.trim1 : a.trim0) : 0;
            const bool rev = KREV && trim_b == 3;
            const uint32_t pbase = (rev ? peqr_base : peq_base) + (uint32_t)b * 36u;
            const int sb0 = fb[t] + lo;  // flat index of the window's first base
            const int se0 = fb[t] + hi - 32;  // reversed sweeps: flat index of the lowest of the first block's 32 positions
            int e_lo = 0, e_hi = -1;     // split mode: first / last column (of the sweep) with a distance within the budget
            uint32_t sflag = 0u;         // known-trim class, reversed sweeps: s of the column e_hi
            for (int blk = 0;; ++blk) {
                const int rem = ncol - 32 * blk;
                if (!__builtin_amdgcn_ballot_w64(rem > 0)) break;
                const int sb = rev ? se0 - 32 * blk : sb0 + 32 * blk;
                const int d0 = sb >> 3, shb = (sb & 7) * 4;  // (reversed: sb >= -31 while rem > 0, the words in front of the 4-bit image belong to the 2-bit image's padding)
                uint32_t W[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) W[u] = (valid && (!KREV || rem > 0)) ? img4[d0 + u] : 0u;
                uint32_t A[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) A[u] = __builtin_amdgcn_alignbit(W[u + 1], W[u], shb);
                if (KREV) {
                    // column c of a reversed sweep is flat position sb + 31 - c: the 32 four-bit codes in reverse order
                    uint32_t R[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t y = __builtin_amdgcn_perm(0u, A[3 - u], 0x00010203u);
                        R[u] = ((y & 0x0F0F0F0Fu) << 4) | ((y >> 4) & 0x0F0F0F0Fu);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) A[u] = rev ? R[u] : A[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // columns beyond the window become "other" symbols: they match no barcode row, and a column that
                    // matches nothing never lowers the running minimum (D[i][j] >= D[i][j-1] for every row)
                    const int nv = rem - 8 * u;
                    const uint32_t junk = nv >= 8 ? 0u : (nv <= 0 ? 0x44444444u : (0x44444444u << (4 * nv)));
                    A[u] |= junk;
                }
                uint32_t inm = 0u, inm2 = 0u;
                // groups of eight columns some lane still needs (the tail block of a 33..48-column window is mostly junk)
                const int ngr = __builtin_amdgcn_ballot_w64(rem > 24) ? 4 : (__builtin_amdgcn_ballot_w64(rem > 16) ? 3 : (__builtin_amdgcn_ballot_w64(rem > 8) ? 2 : 1));
                if (blk == 0)
                    sweep_block<TF, (SPLIT ? 1 : KEND ? 1 + KEND : 0)>(A[0], A[1], A[2], A[3], pbase, Pv, Mv, score, best, kk + 1, inm, inm2, ngr);
                else
                    sweep_block<0, (SPLIT ? 1 : KEND ? 1 + KEND : 0)>(A[0], A[1], A[2], A[3], pbase, Pv, Mv, score, best, kk + 1, inm, inm2, ngr);
                if (KEND && !SPLIT) {
                    // (junk columns never lower the minimum, §3.0; masked all the same)
                    inm &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ~((1u << (32 - rem)) - 1u));
                    if (inm) {  // the last column that lowered the minimum
                        const int tz = (int)__builtin_ctz(inm);
                        e_hi = 32 * blk + 31 - tz;
                        sflag = (inm2 >> tz) & 1u;
                    }
                }
                if (SPLIT) {
                    // first / last column of the window whose unit distance is within the budget; the
                    // junk columns behind the window are not columns
                    inm &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ~((1u << (32 - rem)) - 1u));
                    if (inm) {
                        if (e_hi < 0) e_lo = 32 * blk + (int)__builtin_clz(inm);
                        e_hi = 32 * blk + 31 - (int)__builtin_ctz(inm);
                    }
                }
            }
            if (valid && best <= kk) {
                if (SPLIT) {
                    // candidate bit + one window entry for the exact kernel, exactly as bdx_bitpar.hip's tracked sweeps
                    // hand them over: {barcode, first column of the restricted run, last column}, 1-based columns
                    const int pass = b >= a.B0 ? 1 : 0;
                    const int bl = b - (pass ? a.B0 : 0);
                    __hip_atomic_fetch_or(&cand[t * cwt + (pass ? a.cw[0] : 0) + (bl >> 5)], 1u << (bl & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int kx = __hip_atomic_fetch_add(pass ? &wcl1[t] : &scnt[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (kx < BDX_WCAP && e_hi >= 0) {
                        const int jf_abs = lo + 1;  // 1-based column of sweep column 0
                        // :semiglobal: first column of the restricted run; :hamming / :exact: first start position
                        const int lb = !a.sg ? mm - 1 : ((pass ? a.short_lb[1] : a.short_lb[0]) ? mm + kk : 2 * (mm + kk) + 1);  // (no run-time index into the argument arrays: that puts them in scratch)
                        const size_t rg = PAIRS ? (size_t)gid[t] : (size_t)(r0 + t);
                        uint32_t *dst = (pass ? a.wins_out[1] : a.wins_out[0]) + (rg * BDX_WCAP + kx) * 3;
                        dst[0] = (uint32_t)bl;
                        dst[1] = (uint32_t)(jf_abs + e_lo - lb);
                        dst[2] = (uint32_t)(jf_abs + e_hi);
                    }
                } else {
                    // (dual known-score configs: the survivors of pass 1 — barcodes numbered behind those of pass 0 — have their own
                    // four slots, in the candidate-word area, and their own count)
                    const bool second = GEN && b >= a.B0;
                    const int bl = second ? b - a.B0 : b;
                    const int ks = __hip_atomic_fetch_add(second ? &wcl1[t] : &scnt[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // known-trim class: barcode << 22 | d << 16 | position key; ascending order = the replay's order: per barcode the
                    // smallest distance first, and of equal ones — trim_side = 5: key = 1-based end column, the leftmost end first;
                    // trim_side = 3: key = 0xFFFF - start, the largest start first (no trim side: 0)
                    uint32_t pkey = 0u;
                    if (KEND && (trim_b == 5 || (KALN && trim_b == 0 && a.need_tb))) pkey = (uint32_t)(lo + e_hi + 1);  // (known-alignment class: a pass without a trim side records like trim_side = 5, :142-153)
                    if (KREV && rev) {
                        const int pstar = hi - 1 - e_hi;  // 0-based read position = node the last lowering column stands for
                        const int origin = (sflag == 0u && pstar <= win_lo(t, second)) ? 0 : pstar + (int)sflag;
                        pkey = 0xFFFFu - (uint32_t)origin;
                    }
                    if (ks < 4) (second ? cand : slots)[t * 4 + ks] = KEND ? (((uint32_t)bl << 22) | ((uint32_t)best << 16) | pkey) : (((uint32_t)bl << 8) | (uint32_t)best);
                }
            }
        };
        int nhq = 0;  // seed hits of the tile so far (wave-uniform)
        bool hq_over = false;  // pairs mode: the queue ran over at some point of the tile
        // pairs mode: every queue entry is a sweep over the columns [first - kb, last + m + kb) of its diagonals; the queue is
        // drained whenever it is nearly full (many barcodes: > 100 flagged pairs per read) and at the end of the scan
        const auto drain = [&]() __attribute__((always_inline)) {
            WAVE_SYNC();
            hq_over = hq_over || nhq > HQ;
            const int nh = BDX_DBG(4) ? 0 : (nhq < HQ ? nhq : HQ);
            for (int s0 = 0; s0 < nh; s0 += 64) {
                const int k = s0 + lane;
                bool valid = k < nh;
                const uint32_t h = valid ? hq[k] : 0u;
                const int b = (int)(h & 511u), t = (int)((h >> 9) & 15u), wd = (int)((h >> 13) & 15u), dlo = (int)(h >> 17) - 64;
                const uint32_t mt = meta[b];
                const int mm = (int)(mt & 255u), kk = (int)((mt >> 8) & 255u);
                valid = valid && kk != 255;
                const int n = rl[t];
                const int sp = KB >= 8 ? (int)(mt >> 24) : kk;  // columns an alignment can lie off the flagged diagonal: its indels
                const int dr = dlo - (fb[t] - t * a.slot);      // (the scan's diagonals are relative to the slot: the read starts `head` positions in)
                int lo = dr - sp, hi = dr + wd + mm + sp;
                const int wlo = win_lo(t, b >= a.B0), whi = win_hi(t, b >= a.B0, n);
                lo = lo < wlo ? wlo : lo;
                hi = hi > whi ? whi : hi;
                sweep_lane(valid, t, b, lo, hi);
            }
            nhq = 0;
            WAVE_SYNC();
        };

        // ---- seed scan: lane = 16 consecutive flat positions, one bitmap probe per position ----
        // The bitmap is read as 32-bit words (word = key >> 5 at LDS address 0 + 4 word, bit = key & 31): the shift
        // that brings the key to bit 0 also is the shift amount of the bit test (the hardware takes its low five bits).
        if constexpr (PAIRS) {
            // ---- pairs scan: lane = 16 consecutive diagonals of one read ----
            // Two-intact-pieces lemma: an alignment of barcode b with at most kb <= KB unit operations leaves at least two
            // of the kb + 2 disjoint 4-base pieces at barcode offsets 0, 4, 8, .. untouched; they occur in the read on
            // diagonals (read position - barcode offset) at most kb apart, and the alignment lies within the columns
            // [d - kb, d + m + kb) of the larger diagonal d.  Per diagonal, the table entry of (piece t, key of the read's
            // four bases at d + 4 t) is the set of barcodes whose piece t has that key, as a bit mask over the barcodes:
            // `twice` = barcodes with two pieces on this diagonal, `once & near` = one here and one on the KB diagonals
            // before.  Flagged barcodes are swept over that window; every other pair has a distance beyond its budget.
            // Keys of positions outside the read are whatever the image holds there: they can only add sweeps.
            // SAME-DIAGONAL variants (KB = 8: six 4-base pieces, KB = 9: eight 3-base pieces; ND = 0): configs whose indels cost
            // more than their mismatches (the reference's own demo2 options: mismatch 1, indel 2, budget 6 of 24).  An alignment
            // with g indels lies on at most g + 1 diagonals and has at most e(g) = g + floor((budget - g indel) / mismatch)
            // operations; when P - e(g) >= g + 2 for every g, two of its intact pieces share a diagonal (build_pair_tables checks
            // it per barcode), and it lies within the columns [d - g_max, d + m + g_max) of that diagonal d (`spread` in meta).
            constexpr int PL = KB == 9 ? 3 : 4;         // bases per piece
            constexpr int P = KB == 8 ? 6 : KB == 9 ? 8 : KB + 2;
            constexpr int ND = KB >= 8 ? 0 : KB;        // the second piece may sit on one of the ND diagonals before
            constexpr int ESTRIDE = NW <= 2 ? 8 : 16;   // bytes per table entry
            constexpr int TSTRIDE = (1 << (2 * PL)) * ESTRIDE;  // bytes per piece table (the tables start at LDS address 0)
            constexpr int XLO = -8 - ND;                // first position, relative to the chunk, whose key is needed
            constexpr int NX = 16 + ND + PL * (P - 1);  // positions
            static_assert(7 + PL * (P - 1) + PL - 1 <= 31, "the keys of a chunk come out of three words of the 2-bit image");
            uint32_t amask = ((1u << (2 * PL)) - 1u) * ESTRIDE;
            asm volatile("" : "+v"(amask));
            const int s16 = a.vps;
            const int items = BDX_DBG(8) ? 0 : nr * a.cpr;
            for (int i0 = 0; i0 < items; i0 += 64) {
                const int i = i0 + lane;
                const bool on = i < items;
                const int t = on ? (int)(((uint32_t)i * (uint32_t)a.cpr_inv) >> 16) : 0;
                const int c = on ? i - t * a.cpr : 0;
                uint32_t ad[NX];  // LDS byte offset of the key's entry within a piece table
                uint32_t Fl[NW], dmw[NW];
                // chunk c holds the diagonals d = 16 c - 8 + j, j = 0 .. 15; position of piece t on diagonal d: d + 4 t.  The
                // address of a position's key is computed when the walk over the diagonals first needs it (piece P - 1 of
                // diagonal j) and dies after piece 0 of diagonal j + 4 (P - 1): ~21 of the 40 live at a time
                const int gword = t * s16 + c;
                uint32_t wm1 = 0u, w0 = 0u, w1 = 0u;
                const auto keys = [&]() __attribute__((always_inline)) {
                    wm1 = img2[gword - 1];
                    w0 = img2[gword];
                    w1 = img2[gword + 1];
                };
                const auto key_addr = [&](const int xi) __attribute__((always_inline)) -> uint32_t {
                    const int bit = 2 * (XLO + xi + 16) - (ESTRIDE == 16 ? 4 : 3);  // the key lands at bit 4 (3): times 16 (8)
                    static_assert(2 * (XLO + 16) - 4 >= 0, "the first key's bits start inside the first word");
                    const int wi = bit >> 5, sh = bit & 31;
                    const uint32_t lo = wi == 0 ? wm1 : (wi == 1 ? w0 : w1);
                    const uint32_t hi = wi == 0 ? w0 : (wi == 1 ? w1 : 0u);
                    return __builtin_amdgcn_alignbit(hi, lo, sh) & amask;
                };
                const auto diagonals = [&](const uint32_t gbase) __attribute__((always_inline)) {
                    uint32_t Ah[ND > 0 ? ND : 1][NW];  // barcodes with any piece on each of the previous ND diagonals
#pragma unroll
                    for (int j = -ND; j < 16; ++j) {
#pragma unroll
                        for (int xi = (j == -ND ? 0 : j + ND + PL * (P - 1)); xi <= j + ND + PL * (P - 1); ++xi) ad[xi] = key_addr(xi);
                        uint32_t H[P][NW];
#pragma unroll
                        for (int tt = 0; tt < P; ++tt) {
                            const uint32_t ea = ad[j - 8 + PL * tt - XLO] + gbase + (uint32_t)(tt * TSTRIDE);
                            if constexpr (NW == 1) {
                                H[tt][0] = *(const LDS uint32_t *)(bm + ea);
                            } else if constexpr (NW == 2) {
                                const u32x2 x = *(const LDS u32x2 *)(bm + ea);
                                H[tt][0] = x[0];
                                H[tt][1] = x[1];
                            } else {
                                const u32x4 x = *(const LDS u32x4 *)(bm + ea);  // (NW = 3: a 12-byte read is slower than reading the padding along: 2.94 -> 2.74 ms for C2d)
                                H[tt][0] = x[0];
                                H[tt][1] = x[1];
                                H[tt][2] = x[2];
                                if constexpr (NW == 4) H[tt][3] = x[3];
                            }
                        }
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            uint32_t once = H[0][w] | H[1][w];
                            uint32_t twice = H[0][w] & H[1][w];
#pragma unroll
                            for (int tt = 2; tt < P; ++tt) {
                                if (j >= 0) twice |= once & H[tt][w];
                                once |= H[tt][w];
                            }
                            if (j >= 0) {
                                uint32_t F = twice;
                                if constexpr (ND > 0) {
                                    uint32_t near = Ah[0][w];
#pragma unroll
                                    for (int u = 1; u < ND; ++u) near |= Ah[u][w];
                                    F |= once & near;
                                }
                                Fl[w] |= F;
                                dmw[w] = (dmw[w] << 1) | (F != 0u ? 1u : 0u);  // bit 15 - j
                            }
                            if constexpr (ND > 0) Ah[(j + ND) % ND][w] = once;  // (replaces the oldest)
                        }
                    }
                };
                // one queue entry per flagged barcode: barcode | read << 9 | (last - first flagged diagonal) << 13 | (first + 64) << 17
                const auto append = [&](const int grp) __attribute__((always_inline)) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        uint32_t bits = Fl[w];
                        const int jlo = 15 - (31 - (int)__builtin_clz(dmw[w] | 1u)), jhi = 15 - (int)__builtin_ctz(dmw[w] | 0x10000u);
                        const uint32_t common = ((uint32_t)t << 9) | ((uint32_t)(jhi - jlo) << 13) | ((uint32_t)(16 * c - 8 + jlo + 64) << 17) |
                                                (uint32_t)(128 * grp + 32 * w);
                        unsigned long long mk = __builtin_amdgcn_ballot_w64(bits != 0u);
                        while (mk) {
                            const int k = nhq + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                            if (bits) {
                                const int bi = __builtin_ctz(bits);
                                bits &= bits - 1u;
                                if (k < HQ) hq[k] = common | (uint32_t)bi;
                            }
                            nhq += (int)__builtin_popcountll(mk);
                            mk = __builtin_amdgcn_ballot_w64(bits != 0u);
                        }
                    }
                };
                if constexpr (KB >= 8) {
                    // Same-diagonal variants: ~90 chance flags per read at 96 barcodes of eight 3-base pieces, i.e. several per lane
                    // and mask word — one run of diagonals per lane would span most of its sixteen and make every sweep 40-odd
                    // columns.  The flags are appended per group of THREE diagonals: a sweep's window is m + 2 spread + 2 <= 32
                    // columns (one block), and the append loops take as many trips in total as one append of all sixteen.
                    // (all lanes walk the diagonals — the appends are wave-wide; lanes without a chunk look at read 0 and drop their flags)
                    keys();
                    const uint32_t onm = on ? 0xFFFFFFFFu : 0u;
#pragma unroll
                    for (int j0 = 0; j0 < 16; j0 += 3) {
                        const int j1 = j0 + 2 < 15 ? j0 + 2 : 15;
#pragma unroll
                        for (int w = 0; w < NW; ++w) Fl[w] = dmw[w] = 0u;
#pragma unroll
                        for (int j = j0; j <= j1; ++j) {
#pragma unroll
                            for (int xi = (j == 0 ? 0 : j + PL * (P - 1)); xi <= j + PL * (P - 1); ++xi) ad[xi] = key_addr(xi);
                            uint32_t H[P][NW];
#pragma unroll
                            for (int tt = 0; tt < P; ++tt) {
                                const uint32_t ea = ad[j - 8 + PL * tt - XLO] + (uint32_t)(tt * TSTRIDE);
                                if constexpr (NW <= 2) {
                                    const u32x2 x = *(const LDS u32x2 *)(bm + ea);
                                    H[tt][0] = x[0];
                                    if constexpr (NW == 2) H[tt][1] = x[1];
                                } else {
                                    const u32x4 x = *(const LDS u32x4 *)(bm + ea);
                                    H[tt][0] = x[0];
                                    H[tt][1] = x[1];
                                    H[tt][2] = x[2];
                                    if constexpr (NW == 4) H[tt][3] = x[3];
                                }
                            }
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                uint32_t once = H[0][w] | H[1][w];
                                uint32_t twice = H[0][w] & H[1][w];
#pragma unroll
                                for (int tt = 2; tt < P; ++tt) {
                                    twice |= once & H[tt][w];
                                    once |= H[tt][w];
                                }
                                const uint32_t F = twice & onm;
                                Fl[w] |= F;
                                dmw[w] = (dmw[w] << 1) | (F != 0u ? 1u : 0u);  // bit j1 - j
                            }
                        }
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            uint32_t bits = Fl[w];
                            const int jlo = j1 - (31 - (int)__builtin_clz(dmw[w] | 1u)), jhi = j1 - (int)__builtin_ctz(dmw[w] | 0x8u);
                            const uint32_t common = ((uint32_t)t << 9) | ((uint32_t)(jhi - jlo) << 13) | ((uint32_t)(16 * c - 8 + jlo + 64) << 17) | (uint32_t)(32 * w);
                            unsigned long long mk = __builtin_amdgcn_ballot_w64(bits != 0u);
                            while (mk) {
                                const int k = nhq + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                                if (bits) {
                                    const int bi = __builtin_ctz(bits);
                                    bits &= bits - 1u;
                                    if (k < HQ) hq[k] = common | (uint32_t)bi;
                                }
                                nhq += (int)__builtin_popcountll(mk);
                                mk = __builtin_amdgcn_ballot_w64(bits != 0u);
                            }
                        }
                    }
                    if (nhq > HQ - 768) drain();  // (the queue is swept whenever a round of flags might not fit any more)
                } else if constexpr (!MG) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) Fl[w] = dmw[w] = 0u;
                    if (on) {
                        keys();
                        diagonals(0u);
                    }
                    if constexpr (CARRY) {
                        const int st = on ? tn[t] : 0;
                        if (st) {  // (the run of flagged diagonals may stay wider than the kept barcodes need: a superset)
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                const int blo = 32 * w;
                                const uint32_t p0 = a.B0 >= blo + 32 ? 0xFFFFFFFFu : (a.B0 <= blo ? 0u : ((1u << (a.B0 - blo)) - 1u));  // barcodes of pass 0 in this word
                                Fl[w] &= st == 1 ? ~p0 : p0;
                            }
                        }
                    }
                    append(0);
                } else {
                    // more than 128 barcodes: groups of 128, each with its own piece tables (the keys' addresses are shared)
                    if (on) keys();
                    for (int grp = 0; grp < a.ngroups; ++grp) {
#pragma unroll
                        for (int w = 0; w < NW; ++w) Fl[w] = dmw[w] = 0u;
                        if (on) diagonals((uint32_t)grp * (uint32_t)(P * TSTRIDE));
                        append(grp);
                        if (nhq > HQ - 384) drain();  // (room for one more round of flags: ~5 per lane and group at 128 barcodes)
                    }
                }
            }
        } else {
            constexpr uint32_t AMASK = ((1u << (2 * Q - 5)) - 1u) << 2;
            constexpr uint32_t KMASK = (1u << (2 * Q)) - 1u;
            uint32_t amask = AMASK;
            asm volatile("" : "+v"(amask));  // (in a vector register: a literal or scalar operand slows the AND down)
            // ranged single-pass configs whose window is much shorter than the read (ref_search_range = "1:60"): only the groups
            // of sixteen positions that overlap each read's window are scanned — lane = (read, group of its window) instead of
            // lane = group of the flat image.  A tile with a window longer than planned (a read beyond the length hint) takes
            // the flat scan.
            // (dual configs: the windows of both passes, one after the other; what the first one's lanes cover is not reported twice)
            int gpr = (GEN && ranged) ? a.scan_gpr : 0;
            const int npw = (GEN && a.B0 < B) ? 2 : 1;  // windows per read
            if (gpr > 0) {
                bool over = false;
                if (lane < nr) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (p < npw) {
                            const int s0 = fb[lane] + wwin[(2 * p) * RW + lane], s1 = fb[lane] + wwin[(2 * p + 1) * RW + lane] - Q;  // first / last flat seed start of the window
                            over = over || (s1 >= s0 && (s1 >> 4) - (s0 >> 4) + 1 > gpr);
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(over)) gpr = 0;
            }
            const int nscan = BDX_DBG(8) ? 0 : (gpr > 0 ? nr * npw * gpr : nvec);
            for (int g0i = 0; g0i < nscan; g0i += 64) {
                int g = g0i + lane;
                bool ong = g < nscan;
                uint32_t keep = 0xFFFFu;  // positions of the group that are this lane's to report
                if (gpr > 0) {
                    const int i = g0i + lane;
                    const int tw = ong ? (int)(((uint32_t)i * (uint32_t)a.scan_gpr_inv) >> 16) : 0;  // (read, window) number
                    const int t = npw == 2 ? tw >> 1 : tw, which = npw == 2 ? tw & 1 : 0;
                    int s0 = fb[t] + wwin[(2 * which) * RW + t];
                    const int f1 = fb[t + 1];
                    g = (s0 >> 4) + (i - tw * gpr);
                    if (which) {  // positions the lanes of the read's first window report already
                        const int c0 = (((fb[t] + wwin[t]) >> 4) + gpr) << 4;
                        const int a0 = fb[t] + wwin[t];
                        if (s0 >= a0 && s0 < c0) s0 = c0;
                    }
                    const int below = s0 - 16 * g, above = f1 - 16 * g;  // bits < below lie in front of the window, bits >= above in the next read
                    ong = ong && above > 0 && g < nvec;
                    keep = (below > 0 ? (0xFFFFu << (below > 16 ? 16 : below)) : 0xFFFFu) & (above < 16 ? ((1u << (above < 0 ? 0 : above)) - 1u) : 0xFFFFu) & 0xFFFFu;
                }
                uint32_t hits = 0;
                uint32_t w0 = 0, w1 = 0;
                if (ong) {
                    w0 = img2[g];
                    w1 = img2[g + 1];
                    const uint32_t wm = __builtin_amdgcn_alignbit(w1, w0, 16);  // bases 8 .. 23 of the group's window
                    uint32_t Wk[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) Wk[i] = i == 0 ? w0 : (i <= 8 ? w0 >> (2 * i) : wm >> (2 * (i - 8)));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t word[8], addr[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) addr[i] = (Wk[8 * h + i] >> 3) & amask;
                        lds_read8(word, addr);
                        lds_wait8(word);
#pragma unroll
                        for (int i = 0; i < 8; ++i) hits = __builtin_amdgcn_alignbit(word[i] >> (Wk[8 * h + i] & 31u), hits, 1);
                    }
                    hits >>= 16;
                    hits &= keep;
                }
                // append: one trip per "layer" of hits (the lowest remaining hit of every lane that has one)
                unsigned long long mk = __builtin_amdgcn_ballot_w64(hits != 0u);
                while (mk) {
                    const int k = nhq + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    if (hits) {
                        const int i = __builtin_ctz(hits);
                        hits &= hits - 1u;
                        if (k < HQ) hq[k] = ((uint32_t)(16 * g + i) << 16) | (__builtin_amdgcn_alignbit(w1, w0, 2 * i) & KMASK);
                    }
                    nhq += (int)__builtin_popcountll(mk);
                    mk = __builtin_amdgcn_ballot_w64(hits != 0u);
                }
            }
        }
        if constexpr (PAIRS) drain();
        WAVE_SYNC();
        const bool hq_ok = PAIRS ? !hq_over : nhq <= HQ;  // else: the whole tile goes to the list
        const int nh = (hq_ok && !BDX_DBG(4) && !PAIRS) ? nhq : 0;

        // ---- resolve: one lane per hit -> (read, barcode, diagonal) -> the read's record table ----
        // A record is one (barcode, cluster of diagonals): hits of the barcode whose diagonal lies within kb of the
        // record's first one are merged into it (the intact pieces of ONE alignment within the budget lie on diagonals
        // at most kb apart, so they always share a record and its window stays within 32 columns); a hit further away
        // opens a record of its own — each record's window alone holds every alignment its own hits can belong to, and
        // the replay takes the smallest of a barcode's entries (bdx_core.h run_pass_known).  The lane that opens a
        // record appends its slot number to the tile's record list: the records ARE the sweeps.
        int ns = 0;  // records of the tile so far (wave-uniform)
        if constexpr (!PAIRS) {
            const int fb0 = fb[0];
            const float rinv = ulen > 0 ? 1.0f / (float)ulen : 0.0f;
            const float ginv = total > 0 ? (float)nr / (float)total : 0.0f;
            for (int k0 = 0; k0 < nh; k0 += 64) {
                const int k = k0 + lane;
                int new0 = -1, new1 = -1;  // record slots this lane opened
                if (k < nh) {
                    const uint32_t h = hq[k];
                    const int pos = (int)(h >> 16);
                    const uint32_t key = h & 0xFFFFu;
                    int t;
                    bool ok = true;
                    if (SCAT) {  // (slots of equal size)
                        t = (int)(((uint32_t)(pos >> 4) * (uint32_t)a.vps_inv) >> 16);
                        ok = t < nr;
                        t = ok ? t : 0;
                    } else if (ulen > 0) {
                        const int x = pos - fb0;
                        t = (int)((float)x * rinv);
                        t = t * ulen > x ? t - 1 : t;
                        t = (t + 1) * ulen <= x ? t + 1 : t;
                        ok = x >= 0 && t < nr;
                        t = ok ? t : 0;
                    } else {
                        t = (int)((float)pos * ginv);
                        t = t > nr - 1 ? nr - 1 : t;
                        while (t > 0 && pos < fb[t]) --t;
                        while (t < nr - 1 && pos >= fb[t + 1]) ++t;
                    }
                    const int f0 = fb[t];
                    const int p = pos - f0, n = SCAT ? rl[t] : fb[t + 1] - f0;
                    // a seed lies inside its read (final_search_range = 1:n for this kernel's configs, classification.jl:795-800)
                    if (ok && p >= 0 && p + q <= n) {
                        // the key is in the bitmap (the bitmap is exact): its entry is the one with the key's rank among the
                        // keys present; further pieces with the same key (rare) hang off it
                        const uint32_t wi = key >> 5;
                        uint32_t idx = (uint32_t)rnk[wi] + (uint32_t)__builtin_popcount(((const LDS uint32_t *)bm)[wi] & ((1u << (key & 31u)) - 1u));
                        do {
                            const uint32_t e = ent[idx];
                            idx = e >> 16;
                            bool inwin = true;  // (ranged configs: the seed must lie inside the pass's column window)
                            if (ranged) {
                                const bool second_b = (int)(e & 2047u) - 1 >= a.B0;
                                inwin = p >= wwin[(second_b ? 2 : 0) * RW + t] && p + q <= wwin[(second_b ? 3 : 1) * RW + t];
                            }
                            if (inwin) {
                                const uint32_t pb = e & 2047u;  // barcode + 1
                                const int kk = (int)((meta[pb - 1u] >> 8) & 255u);
                                const int diag = p - (int)((e >> 11) & 31u);
                                const uint32_t mine = pb | ((uint32_t)(diag + 64) << 16);
                                int rs = (int)((pb + (uint32_t)(diag >> 3)) & (RCAP - 1));
                                bool placed = false;
                                for (int tries = 0; tries < RCAP && !placed; ++tries) {
                                    LDS uint32_t *id = rid + t * RCAP + rs;
                                    uint32_t old = *id;
                                    bool opened = false;
                                    if (old == 0u) {
                                        uint32_t expect = 0u;
                                        __hip_atomic_compare_exchange_strong(id, &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        opened = expect == 0u;
                                        old = opened ? mine : expect;
                                    }
                                    const int dd = diag - ((int)(old >> 16) - 64);
                                    if ((old & 0xFFFFu) == pb && dd >= -kk && dd <= kk) {
                                        __hip_atomic_fetch_or(&rmk[t * RCAP + rs], 1u << (dd + kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        placed = true;
                                        if (opened) {
                                            if (new0 < 0)
                                                new0 = t * RCAP + rs;
                                            else if (new1 < 0)
                                                new1 = t * RCAP + rs;
                                            else {  // a third record opened by one hit: not swept, the read is swept over every barcode below (flag 2)
                                                if (flag[t] != 1) flag[t] = 2;  // (1: outside the known-score class — stays listed)
                                                rid[t * RCAP + rs] = 0u;
                                                rmk[t * RCAP + rs] = 0u;
                                            }
                                        }
                                    }
                                    rs = (rs + 1) & (RCAP - 1);
                                }
                                if (!placed && flag[t] != 1) flag[t] = 2;  // more than RCAP records in this read
                            }
                        } while (idx != 0u);
                    }
                }
                // the slots opened in this round -> the record list (at most two layers)
#pragma unroll
                for (int layer = 0; layer < 2; ++layer) {
                    const int nw = layer == 0 ? new0 : new1;
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(nw >= 0);
                    if (mk) {
                        const int kq = ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                        if (nw >= 0) {
                            if (kq < SQ) {
                                recq[kq] = (uint32_t)nw;
                            } else {  // more records than the tile's sweep list holds: this read is swept over every barcode below
                                if (flag[nw / RCAP] != 1) flag[nw / RCAP] = 2;
                                rid[nw] = 0u;
                                rmk[nw] = 0u;
                            }
                        }
                        ns += (int)__builtin_popcountll(mk);
                    }
                }
            }
        }
        ns = ns < SQ ? ns : SQ;
        WAVE_SYNC();

        if constexpr (!PAIRS)
        for (int s0 = 0; s0 < ns; s0 += 64) {
            const int k = s0 + lane;
            bool valid = k < ns;
            const uint32_t rslot = valid ? recq[k] : 0u;
            const uint32_t id = valid ? rid[rslot] : 0u;
            const uint32_t dmk = valid ? rmk[rslot] : 0u;
            if (valid) {  // the record is consumed
                rid[rslot] = 0u;
                rmk[rslot] = 0u;
            }
            valid = valid && id != 0u && dmk != 0u;
            const int t = (int)(rslot / RCAP), b = valid ? (int)(id & 0xFFFFu) - 1 : 0;
            int lo = 0, hi = 0;  // [lo, hi): 0-based columns of the sweep
            if (valid) {
                const uint32_t mt = meta[b];
                const int mm = (int)(mt & 255u), kk = (int)((mt >> 8) & 255u);
                const int d0 = (int)(id >> 16) - 64;
                const int dmin = d0 + __builtin_ctz(dmk) - kk, dmax = d0 + (31 - __builtin_clz(dmk)) - kk;
                const int n = rlen(t);
                lo = dmin - kk - 1;
                hi = dmax + mm + kk + 1;
                int wlo = 0, whi = n;
                if (ranged) {
                    wlo = wwin[(b >= a.B0 ? 2 : 0) * RW + t];
                    whi = wwin[(b >= a.B0 ? 3 : 1) * RW + t];
                }
                lo = lo < wlo ? wlo : lo;
                hi = hi > whi ? whi : hi;
            }
            sweep_lane(valid, t, b, lo, hi);
        }
        WAVE_SYNC();
        if (SPLIT || !PAIRS) {
            // Reads whose tables overflowed (more records than a read or the tile holds, a hit queue that ran over: low
            // complexity): every barcode is swept over the whole read here, lane = barcode — the exact kernel then still
            // gets a true candidate mask.  (Handing such a read on with "every barcode, no windows" would make ONE lane
            // of the exact kernel run B whole-window DPs one after the other: two such reads in 2 M cost 11 ms.)
            // Known-score forms (round 4): the same for a read whose record tables overflowed (flag 2) — its survivors come out of
            // the all-barcode sweeps and it is replayed like any other read (more than four survivors: still listed).  C2 lists ONE
            // read in 10 M this way, and the general kernel's list launch behind the wave kernel takes 30 us for a one-read list
            // against 5 us for an empty one.  (Reads outside the known-score class — flag 1 — and tiles whose hit queue ran over
            // stay on the list.)
            unsigned long long fm = !tile_ok ? 0ull
                                    : SPLIT  ? __builtin_amdgcn_ballot_w64(lane < nr && (flag[lane] != 0 || !hq_ok))
                                             : ((hq_ok && !(a.dbg & (1 << 30))) ? __builtin_amdgcn_ballot_w64(lane < nr && flag[lane] == 2) : 0ull);  // (bit 30: BDX_NO_WAVE_FALLBACK)
            while (fm) {
                const int t = (int)__builtin_ctzll(fm);
                fm &= fm - 1ull;
                if (lane == 0) {
                    scnt[t] = 0;
                    wcl1[t] = 0;
                    flag[t] = 0;
                    if (SPLIT)
                        for (int w = 0; w < cwt; ++w) cand[t * cwt + w] = 0u;
                }
                WAVE_SYNC();
                const int n = rlen(t);
                for (int b0 = 0; b0 < B; b0 += 64) {
                    const int b = b0 + lane < B ? b0 + lane : 0;
                    const bool valid = b0 + lane < B && ((meta[b] >> 8) & 255u) != 255u;  // (255: the barcode can never be recorded)
                    sweep_lane(valid, t, b, win_lo(t, b >= a.B0), win_hi(t, b >= a.B0, n));
                }
                WAVE_SYNC();
            }
        }

        if (SPLIT) {
            // ---- split mode: hand the candidate masks and the window counts to the exact kernel (lane = read) ----
            if (lane < nr && !BDX_DBG(1)) {
                const long long rid_g = PAIRS ? (long long)gid[lane] : r0 + lane;
                const bool usable = tile_ok && !flag[lane] && (!PAIRS || rl[lane] >= 0);  // (overflows were swept above) else — a tile that does not fit the images, a read longer than its slot: every barcode over its whole window
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    if (pass == 1 && a.cw[1] == 0) break;
                    const int cwp = pass ? a.cw[1] : a.cw[0];
                    uint32_t *dst = (pass ? a.cand_out[1] : a.cand_out[0]) + rid_g * cwp;
                    for (int w = 0; w < cwp; ++w) dst[w] = usable ? cand[lane * cwt + (pass ? a.cw[0] : 0) + w] : 0xFFFFFFFFu;
                    const int c = pass ? wcl1[lane] : scnt[lane];
                    (pass ? a.wcnt_out[1] : a.wcnt_out[0])[rid_g] = (unsigned char)((usable && c <= BDX_WCAP) ? c : 255);
                }
            }
            WAVE_SYNC();  // the next tile reuses the per-read tables
            tile = tile_next;
            geo = geo_next;
            ov = ov_next;
            ov_next = ov_after;
            so_next = so_after;
            iv_next = iv_after;
            iv_after = iv_after2;
            par ^= 1;
            continue;
        }

        // ---- verdicts: lane = read; reducer replay on the survivors' unit distances ----
        const bool active = lane < nr;
        const long long ridx = SCAT ? (long long)gid[lane < RW ? lane : 0] : r0 + lane;
        Verdict vd{0, 0, -1, -1};
        PassOut p1{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()}, p2{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
        bool done = false;
        uint32_t cst = 0u, centry = 0u;  // tier 1 of a dual config: which pass of a listed read is settled (1 / 2) and its winning survivor
        if (active && tile_ok && hq_ok && !BDX_DBG(1)) {
            const int n = WINM ? tn[lane] : rlen(lane);  // (window mode: the keep range is the whole READ, :907-908)
            const int cnt = scnt[lane], cnt1 = dual ? wcl1[lane] : 0;
            // known-score class per read: this kernel only runs for configs whose ranges resolve to
            // 1:n, so n >= 1 is all that is left to check (n = 0: the :805 sanity check sends the read to :unknown)
            const bool simple = a.out.pass_start == nullptr && a.out.pass_end == nullptr && a.out.pass_raw == nullptr && a.out.pass_bc == nullptr &&
                                a.out.pass_score == nullptr && a.out.pass_delta == nullptr;  // (kernel-uniform: only the verdict vectors are wanted)
            if (simple && !KALN && !flag[lane] && cnt <= 1 && cnt1 <= 1 && n >= 1) {
                // No or one survivor per pass and nobody asked for scores: the reducers' answer for a lone survivor with distance d
                // is a per-barcode constant — accepted iff d <= floor(rate * m) and fl(d / m) <= rate (classification.jl:254, :658 /
                // :696; with_delta: delta = Inf - score is never below min_delta) — precomputed on the host with the same
                // IEEE operations (build_wave_tables); likewise tier 1's settle rule.  No Float64 here.  Dual configs: pass 2 runs
                // only behind a matched pass 1, and a pass 2 without a match makes the read unknown (:887-895).
                done = true;
                int pos1 = 0, pos2 = 0;  // known-trim class: position keys of the two passes' survivors (end column / 0xFFFF - start)
                const int sbit = a.min_delta == 0.0 ? 0 : 16;
                bool settled_all = true;
                if (cnt == 1) {
                    const uint32_t e = slots[lane * 4];
                    const int bb = KEND ? (int)(e >> 22) : (int)(e >> 8), d = KEND ? (int)((e >> 16) & 63u) : (int)(e & 255u);
                    pos1 = (int)(e & 0xFFFFu);
                    const int dmax = (int)((meta[bb] >> 16) & 255u);
                    vd.bc1 = (dmax != 255 && d <= dmax) ? bb + 1 : 0;
                    settled_all = vd.bc1 > 0 && ((settle[bb] >> (d + sbit)) & 1u) != 0u;
                } else {
                    vd.bc1 = 0;
                    settled_all = false;  // (nothing within the capped budgets: tier 0 decides)
                }
                const bool s1_settled = settled_all;  // pass 0 settled and matched
                vd.bc2 = 0;
                if (dual && vd.bc1 > 0) {
                    int bc2v = 0;
                    bool s2 = false;
                    if (cnt1 == 1) {
                        const uint32_t e = cand[lane * 4];
                        const int bb = KEND ? (int)(e >> 22) : (int)(e >> 8), d = KEND ? (int)((e >> 16) & 63u) : (int)(e & 255u), g = a.B0 + bb;
                        pos2 = (int)(e & 0xFFFFu);
                        const int dmax = (int)((meta[g] >> 16) & 255u);
                        bc2v = (dmax != 255 && d <= dmax) ? bb + 1 : 0;
                        s2 = bc2v > 0 && ((settle[g] >> (d + sbit)) & 1u) != 0u;
                    }
                    settled_all = settled_all && s2;
                    vd.bc2 = bc2v;
                    if (bc2v == 0) vd.bc1 = 0;  // (:891-894: the verdict is pass 2's status)
                    if (bc2v == 0) vd.bc2 = 0;
                }
                if (a.tier) done = settled_all;
                if (!PAIRS && a.tier && dual && a.carry_ent != nullptr && !settled_all) {
                    // one pass settled, the other open: the settled one travels with the read (see CARRY above)
                    if (s1_settled) {
                        cst = 1u;
                        centry = slots[lane * 4];
                    } else if (cnt1 == 1) {  // pass 1 on its own (the reducers of a pass do not look at the other pass)
                        const uint32_t e = cand[lane * 4];
                        const int bb = KEND ? (int)(e >> 22) : (int)(e >> 8), d = KEND ? (int)((e >> 16) & 63u) : (int)(e & 255u), g = a.B0 + bb;
                        const int dmax = (int)((meta[g] >> 16) & 255u);
                        if (dmax != 255 && d <= dmax && ((settle[g] >> (d + sbit)) & 1u) != 0u) {
                            cst = 2u;
                            centry = e;
                        }
                    }
                }
                vd.keep_start = vd.bc1 > 0 ? 1 : -1;  // :907-908 / :879-883 (ScoreOnly: the whole read, n >= 1)
                vd.keep_end = vd.bc1 > 0 ? n : -1;
                if (KEND && vd.bc1 > 0) {
                    // trim_side = 5: keep what follows the alignment's end (:912-914); trim_side = 3: what precedes its start
                    // (:910-911); a second pass narrows the range (:921-929); (1, 0) if nothing is left (:932-935)
                    int ks = 1, ke = n;
                    if (a.trim0 == 3) {
                        const int st = 0xFFFF - pos1;
                        ke = (st > 1 ? st : 1) - 1;
                    } else if (a.trim0 == 5) {
                        ks = pos1 + 1;
                    }
                    if (dual) {
                        if (a.trim1 == 3) {
                            const int st = 0xFFFF - pos2, e2 = (st > 1 ? st : 1) - 1;
                            ke = ke < e2 ? ke : e2;
                        } else if (a.trim1 == 5) {
                            ks = ks > pos2 + 1 ? ks : pos2 + 1;
                        }
                    }
                    vd.keep_start = ks > ke ? 1 : ks;
                    vd.keep_end = ks > ke ? 0 : ke;
                }
            } else if (!flag[lane] && cnt <= 4 && cnt1 <= 4 && n >= 1) {
                const LDS uint32_t *e0 = slots + lane * 4;
                const LDS uint32_t *e1 = cand + lane * 4;  // (dual only)
                const KnownPass kn0{true, e0[0], e0[1], e0[2], e0[3], cnt, nullptr, nullptr, nullptr, 0};
                const KnownPass kn1 = dual ? KnownPass{true, e1[0], e1[1], e1[2], e1[3], cnt1, nullptr, nullptr, nullptr, 0}
                                             : KnownPass{false, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0};
                const auto m0 = [&](const int bb) { return (int)(meta[bb] & 255u); };
                const auto m1 = [&](const int bb) { return (int)(meta[a.B0 + bb] & 255u); };
                BdxDevCfg cfg;  // (only the fields the replay reads)
                cfg.is_dual = dual ? 1 : 0;
                cfg.max_error_rate = a.max_error_rate;
                cfg.min_delta = a.min_delta;
                cfg.pass[0].trim_side = KEND ? a.trim0 : 0;
                cfg.pass[1].trim_side = KEND ? a.trim1 : 0;
                cfg.need_traceback = KALN ? a.need_tb : 0;
                classify_known<(KALN ? 2 : (KEND != 0 ? 1 : 0))>(cfg, m0, m1, n, kn0, kn1, vd, p1, p2);
                done = true;
                if (a.tier) {
                    // tier settle rule (same code as bdx_bitpar.hip)
                    const bool nd = a.min_delta == 0.0;
                    const auto settled = [&](const PassOut &po, const int c, const double slo) {
                        if (c < 1 || !(po.score < slo)) return false;
                        if (nd) return true;
                        if (c >= 2 && po.sub <= slo) return true;
                        return a.out.pass_delta == nullptr && (slo - po.score) >= a.min_delta && po.status == 1;
                    };
                    bool ok = settled(p1, cnt, a.tier_slo);
                    if (ok && dual && p1.status == 1) ok = settled(p2, cnt1, a.tier_slo1);
                    done = ok;
                }
            }
        }
        if constexpr (KALN) {
            // ---- known-alignment class: the OTHER position of every pass's winner, by an anchored sweep (lane = read) ----
            // The replay knows, per pass, the winner b, its distance d and the position its trim side makes observable: the END of
            // the first column at the minimum (trim_side 5 / none; :142-153) or the START = the largest origin (trim_side 3, §3.0c).
            //  * end known -> start: the reference's start is origin(m, end) = the largest origin among the alignments of cost d that
            //    end exactly there (same exchange argument as §3.0c) — a right-to-left sweep from the end column whose row 0 is NOT
            //    free (anchored_block<true>): its score after the column of position p is the cost of aligning the barcode with
            //    exactly p .. end; the first column (largest p) whose score is d, + 1 iff the diagonal move is optimal there.
            //  * start known -> end: the reference's end is the first column with an alignment of cost d whose origin is that start
            //    (:142-153 with trim_side = 3: of equal starts the first column stays) — a left-to-right sweep from the start's
            //    column, whose first column is prepared as "row 1 entered here" (D[i] = delta(q1, r[start]) + i - 1).
            // Models + enumeration: oracle orc_known_other_position / orc_selftest_known_alignment.
            // A start <= 0 (the alignment comes out of the initial column, :278-283) is not representable here: such a read is handed on.
            const bool want_pos = a.out.pass_start != nullptr || a.out.pass_end != nullptr || a.stats.rows > 0;
            const int tl = lane < RW ? lane : 0;
            const int n_t = rlen(tl);
            bool lost = false;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                if (ps == 1 && !dual) break;
                PassOut &po = ps ? p2 : p1;
                const int tr = ps ? a.trim1 : a.trim0;
                const bool rev = tr != 3;  // (the end is known: sweep back for the start)
                const bool tb = tr != 0 || a.need_tb != 0;  // (a ScoreOnly pass reports no positions, :812 / :124)
                bool valid = want_pos && tb && done && lane < nr && po.bc > 0 && po.status != 2;
                const int b = valid ? (ps ? a.B0 : 0) + po.bc - 1 : 0;
                const int pos = rev ? po.end : po.start;
                if (valid && pos <= 0) {  // (trim_side 3: a start <= 0)
                    lost = true;
                    valid = false;
                }
                const uint32_t mt = meta[b];
                const int mm = (int)(mt & 255u), kk = (int)((mt >> 8) & 255u), d = valid ? po.raw : 0;
                const uint32_t rows = mm >= 32 ? 0xFFFFFFFFu : (((1u << mm) - 1u) << (32 - mm)), lowbit = 1u << (32 - mm);
                const int wlo = win_lo(tl, ps != 0), whi = win_hi(tl, ps != 0, n_t);
                // positions [lo, hi) of the read: rev: hi = end column (exclusive as a 0-based position), walked downwards; else upwards from the start
                int lo = rev ? pos - mm - kk : pos, hi = rev ? pos : pos + mm + kk;
                lo = lo < wlo ? wlo : lo;
                hi = hi > whi ? whi : hi;
                const bool has_cols = hi > lo;
                valid = valid && (has_cols || !rev);  // (a known start at the window's last column: the prepared column may already be the end)
                const int ncol = (valid && has_cols) ? hi - lo : 0;
                uint32_t Pv = rows, Mv = 0u;
                int score = mm;
                const uint32_t pbase = (rev ? peqr_base : peq_base) + (uint32_t)b * 36u;
                if (!rev) {  // the prepared first column: row 1 is entered at column `pos` (read position pos - 1)
                    const int fx = fb[tl] + pos - 1;
                    const uint32_t cw = valid ? img4[fx >> 3] : 0u;
                    const uint32_t code = (cw >> ((fx & 7) * 4)) & 7u;
                    const uint32_t e1 = valid ? peq[b * 9 + (int)code] : 0u;
                    const bool match1 = (e1 & lowbit) != 0u;
                    if (match1) Pv &= ~lowbit;
                    score = mm - (match1 ? 1 : 0);
                }
                const bool ends_at_once = valid && !rev && score == d;  // (the alignment ends in its first column: every later row deleted)
                int found = -1;
                uint32_t sfl = 0u;
                const int sb0 = fb[tl] + lo, se0 = fb[tl] + hi - 32;
                for (int blk = 0;; ++blk) {
                    const int rem = ncol - 32 * blk;
                    if (!__builtin_amdgcn_ballot_w64(valid && !ends_at_once && found < 0 && rem > 0)) break;
                    const int sb = rev ? se0 - 32 * blk : sb0 + 32 * blk;
                    const int d0 = sb >> 3, shb = (sb & 7) * 4;
                    uint32_t W[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) W[u] = (valid && rem > 0) ? img4[d0 + u] : 0u;
                    uint32_t A[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) A[u] = __builtin_amdgcn_alignbit(W[u + 1], W[u], shb);
                    {
                        uint32_t R[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t y = __builtin_amdgcn_perm(0u, A[3 - u], 0x00010203u);
                            R[u] = ((y & 0x0F0F0F0Fu) << 4) | ((y >> 4) & 0x0F0F0F0Fu);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) A[u] = rev ? R[u] : A[u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nv = rem - 8 * u;
                        A[u] |= nv >= 8 ? 0u : (nv <= 0 ? 0x44444444u : (0x44444444u << (4 * nv)));
                    }
                    const int ngr = __builtin_amdgcn_ballot_w64(rem > 24) ? 4 : (__builtin_amdgcn_ballot_w64(rem > 16) ? 3 : (__builtin_amdgcn_ballot_w64(rem > 8) ? 2 : 1));
                    uint32_t eqm = 0u, inm2 = 0u;
                    // (a wave runs both forms when its lanes differ: a dual config with trim sides 5 and 3 has one form per pass)
                    const bool any_rev = __builtin_amdgcn_ballot_w64(valid && rev) != 0ull, any_fwd = __builtin_amdgcn_ballot_w64(valid && !rev) != 0ull;
                    uint32_t Pv2 = Pv, Mv2 = Mv;
                    int score2 = score;
                    uint32_t eqm2 = 0u, inm22 = 0u;
                    if (any_rev) anchored_block<true>(A[0], A[1], A[2], A[3], pbase, rows, lowbit, Pv, Mv, score, d, eqm, inm2, ngr);
                    if (any_fwd) anchored_block<false>(A[0], A[1], A[2], A[3], pbase, rows, lowbit, Pv2, Mv2, score2, d, eqm2, inm22, ngr);
                    if (!rev) {
                        Pv = Pv2;
                        Mv = Mv2;
                        score = score2;
                        eqm = eqm2;
                        inm2 = inm22;
                    }
                    eqm &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ~((1u << (32 - rem)) - 1u));
                    if (valid && found < 0 && eqm) {
                        const int cz = (int)__builtin_clz(eqm);
                        found = 32 * blk + cz;
                        sfl = (inm2 >> (31 - cz)) & 1u;
                    }
                }
                if (ends_at_once) {
                    po.end = pos;
                } else if (valid) {
                    if (found < 0) {
                        lost = true;  // (cannot happen: the distance d was attained by a sweep of this very window)
                    } else if (rev) {
                        const int pstar = hi - 1 - found;
                        if (sfl == 0u && pstar <= wlo)
                            lost = true;  // a start <= 0 (out of the reference's initial column)
                        else
                            po.start = pstar + (int)sfl;
                    } else {
                        po.end = lo + found + 1;
                    }
                } else if (want_pos && tb && done && lane < nr && po.bc > 0 && po.status != 2 && !lost) {
                    lost = true;  // (an empty anchored window: hand the read on rather than guess)
                }
            }
            if (lost) done = false;
            if (done && a.stats.rows > 0) {
                stats_update(a.stats, 0, a.B0, p1);
                if (dual) stats_update(a.stats, 1, B - a.B0, p2);
            }
        }
        {
            // reads for the list: collected in LDS and handed over in batches — one returning global atomic per batch
            // (same-address atomics retire one per ~10 ns chip-wide, and waiting for the returned value also waits for
            // the byte loads of the next tile that are in flight)
            const bool hand = active && !done && !BDX_DBG(1);
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(hand);
            if (mk) {
                const int n_new = (int)__builtin_popcountll(mk);
                if (lcnt + n_new > 64) flush_list();
                if (hand) lbuf[lcnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (uint32_t)ridx | (cst << 30);
                if (hand && cst != 0u) a.carry_ent[ridx] = centry;
                lcnt += n_new;
            }
        }
        if (done) {
            if (a.out.bc1) a.out.bc1[ridx] = vd.bc1;
            if (a.out.bc2) a.out.bc2[ridx] = vd.bc2;
            if (a.out.keep_start) a.out.keep_start[ridx] = vd.keep_start;
            if (a.out.keep_end) a.out.keep_end[ridx] = vd.keep_end;
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
            // DemuxStats scalar counters (classification.jl:942-978), accumulated in LDS across the workgroup's tiles
            if (a.counts) {
                const int slot = vd.bc1 > 0 ? 4 + (vd.bc1 - 1) * a.counts_stride2 + (vd.bc2 > 0 ? vd.bc2 - 1 : 0) : -1;
                const int cls = vd.bc1 > 0 ? 1 : (vd.bc1 == 0 ? 2 : 3);
                __hip_atomic_fetch_add(&hist[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&hist[cls], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (slot >= 0 && slot < a.hist_entries)
                    __hip_atomic_fetch_add(&hist[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (slot >= 0)
                    atomicAdd(&a.counts[slot], 1ULL);
            }
        }
        WAVE_SYNC();  // the next tile reuses the per-read tables
        tile = tile_next;
        geo = geo_next;
        ov = ov_next;
        ov_next = ov_after;
        so_next = so_after;
        iv_next = iv_after;
        iv_after = iv_after2;
        par ^= 1;
    }

    flush_list();
    if (a.counts) {
        __syncthreads();
        for (int i = tid; i < a.hist_entries; i += blockDim.x) {
            const int h = hist[i];
            if (h) atomicAdd(&a.counts[i], (unsigned long long)h);
        }
    }
}

template <int RW, int TF, int NV, int Q, bool SPLIT, int KB, int NW, bool MG = false, int KEND = 0, bool GEN = true, bool WINM = false>
hipError_t launch_wave(const WaveArgs &a, size_t lds, int waves, long long blocks, hipStream_t stream) {
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)bdx_wave_kernel<RW, TF, NV, Q, SPLIT, KB, NW, MG, KEND, GEN, WINM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((bdx_wave_kernel<RW, TF, NV, Q, SPLIT, KB, NW, MG, KEND, GEN, WINM>), dim3((unsigned)blocks), dim3(64 * waves), lds, stream, a);
    return hipGetLastError();
}

void fill_args(WaveArgs &a, const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const BdxDevOut &out, unsigned long long *d_counts,
               uint32_t *list, unsigned int *list_count, int dbg, const BdxWaveSplit *sp) {
    a.max_error_rate = cfg.max_error_rate;
    a.min_delta = cfg.min_delta;
    a.counts_stride2 = cfg.counts_stride2;
    a.out = out;
    a.counts = d_counts;
    a.hist_entries = hist_entries;
    a.bitmap = wp.d_bitmap;
    a.bm_bytes = wp.bm_bytes;
    a.rank = wp.d_rank;
    a.ent = wp.d_ent;
    a.n_ent = wp.n_ent;
    a.peq8 = wp.d_peq8;
    a.peq8r = wp.d_peq8r;
    a.trim0 = cfg.pass[0].trim_side;
    a.trim1 = cfg.is_dual ? cfg.pass[1].trim_side : 0;
    a.stats = BdxDevStats{};
    a.need_tb = (cfg.need_traceback || cfg.algorithm == BDX_ALG_EXACT) ? 1 : 0;  // (:exact always reports the occurrence's positions)
    a.meta = wp.d_meta;
    a.settle = wp.d_settle;
    a.B = wp.n_barcodes;
    a.q = wp.q;
    a.span_cap = wp.span_cap;
    a.per_wave = (int)bdx_wave_area_bytes(wp.rw, wp.span_cap, wp.pairs_kb > 0, wp.hq_cap, wp.sq_cap, wp.cand_words + (wp.ranged ? 4 : 0), wp.winm != 0);
    a.hq_cap = wp.hq_cap;
    a.sq_cap = wp.sq_cap;
    a.list = list;
    a.list_count = list_count;
    a.carry_ent = wp.d_carry;
    a.dbg = dbg;
    a.B0 = wp.b0;
    for (int k = 0; k < 2; ++k) {
        a.cw[k] = sp ? sp->cw[k] : 0;
        a.cand_out[k] = sp ? sp->cand_out[k] : nullptr;
        a.wins_out[k] = sp ? sp->wins_out[k] : nullptr;
        a.wcnt_out[k] = sp ? sp->wcnt_out[k] : nullptr;
        a.short_lb[k] = sp ? sp->short_lb[k] : 0;
    }
    a.sg = cfg.algorithm == BDX_ALG_SEMIGLOBAL ? 1 : 0;
    a.ngroups = wp.groups > 0 ? wp.groups : 1;
    a.cand_area = wp.cand_words;
    a.scan_gpr = wp.scan_gpr;
    a.scan_gpr_inv = wp.scan_gpr > 0 ? (65536 + wp.scan_gpr - 1) / wp.scan_gpr : 0;
    a.ranged = wp.ranged;
    a.dpass[0] = cfg.pass[0];
    a.dpass[1] = cfg.pass[1];
    a.tier_slo1 = 0.0;
    a.dual = 0;
    a.slot = 0;
    a.vps = 1;
    a.vps_inv = 65536;
    a.max_len = 0;
    a.win_sfe = a.win_efe = a.win_so = a.win_eo = 0;
    a.cpr = 1;
    a.cpr_inv = 65536;
    a.idmap = nullptr;
    a.n_dev = nullptr;
}

}  // namespace

// (the argument block crosses translation units as bytes: WaveArgs is this file's, compiled into each of them)
hipError_t bdx_launch_wave_gen(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);
hipError_t bdx_launch_wave_split_gen(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);
// (bdx_wave_rev.hip: the known-trim instantiations with reversed sweeps, KEND = 2)
hipError_t bdx_launch_wave_end_rev(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);
hipError_t bdx_launch_pairs_rev(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);
// (bdx_wave_aln.hip: the known-alignment instantiations, KEND = 3)
hipError_t bdx_launch_wave_end_aln(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);
hipError_t bdx_launch_pairs_aln(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream);

#if defined(BDX_WAVE_TU_WIN)  // the window-mode instantiations (bdx_wave_win.hip)

// Window mode: a single-pass known-score config whose column window is much shorter than its reads; tier 1 or the only
// wave launch of the config, same outputs and the same list as bdx_launch_wave.
hipError_t bdx_launch_wave_win(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                               long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int tier1, double tier_slo, uint32_t *list,
                               unsigned int *list_count, hipStream_t stream, int dbg) {
    if (n_reads <= 0) return hipSuccess;
    if (!wp.winm || wp.pairs_kb > 0 || wp.split || wp.kend || cfg.is_dual || !wp.ranged || wp.slot < 16 || (wp.slot & 15)) return BDX_BAD_PLAN();
    WaveArgs a;
    fill_args(a, cfg, wp, hist_entries, out, d_counts, list, list_count, dbg, nullptr);
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.tier = tier1;
    a.tier_slo = tier_slo;
    a.ranged = 0;     // (the window is resolved by the tile loader: downstream the window IS the read)
    {
        const BdxDevRange &dr = cfg.pass[0].ref_search;
        const long long LIM = 1LL << 28;
        if (dr.start_offset < -LIM || dr.start_offset > LIM || dr.end_offset < -LIM || dr.end_offset > LIM) return BDX_BAD_PLAN();
        a.win_sfe = dr.start_from_end ? 1 : 0;
        a.win_so = (int)dr.start_offset;
        a.win_efe = dr.end_from_end ? 1 : 0;
        a.win_eo = (int)dr.end_offset;
    }
    a.scan_gpr = 0;
    a.slot = wp.slot;
    a.vps = wp.slot >> 4;
    a.vps_inv = (65536 + a.vps - 1) / a.vps;
    a.idmap = nullptr;
    a.n_dev = nullptr;
    a.per_wave = (int)bdx_wave_area_bytes(wp.rw, wp.span_cap, false, wp.hq_cap, wp.sq_cap, 0, true);
    const size_t lds = bdx_wave_table_bytes(wp, hist_entries) + (size_t)wp.waves * (size_t)a.per_wave;
    const long long tiles = (n_reads + wp.rw - 1) / wp.rw;
    long long blocks = (long long)wp.blocks;
    const long long useful = (tiles + wp.waves - 1) / wp.waves;
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    const int tf = wp.track_from;
    const int vecs = wp.rw * a.vps;  // 16-byte vectors of a tile
    if (wp.rw * wp.slot + 16 > wp.span_cap) return BDX_BAD_PLAN();
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 0, true, true>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_TF(RWV, NVV)                                                                              \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_SP(RWV, 20, NVV, 8) : tf >= 12 ? BDX_WAVE_SP(RWV, 12, NVV, 8) : BDX_WAVE_SP(RWV, 0, NVV, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_SP(RWV, 12, NVV, 7) : BDX_WAVE_SP(RWV, 0, NVV, 7))                      \
                       : BDX_WAVE_SP(RWV, 0, NVV, 6)
    if (wp.rw == 32 && vecs <= 64 * 3) BDX_WAVE_TF(32, 3);
    if (wp.rw == 32 && vecs <= 64 * 7) BDX_WAVE_TF(32, 7);
    if (wp.rw == 16 && vecs <= 64 * 4) BDX_WAVE_TF(16, 4);
    return BDX_BAD_PLAN();
#undef BDX_WAVE_TF
#undef BDX_WAVE_SP
}

#elif defined(BDX_WAVE_TU_ALN)  // the known-alignment instantiations (bdx_wave_aln.hip)

hipError_t bdx_launch_wave_end_aln(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    const int tf = wp.track_from;
    if (wp.kend != 3) return BDX_BAD_PLAN();
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 3>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

hipError_t bdx_launch_pairs_aln(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    if (wp.pairs_kb > 4 || wp.nw > 4 || wp.track_from < 12 || wp.groups > 1 || wp.split || wp.kend != 3) return BDX_BAD_PLAN();
#define BDX_PAIRS_SP(RWV, NVV, KBV, NWV) launch_wave<RWV, 12, NVV, 4, false, KBV, NWV, false, 3>(a, lds, wp.waves, blocks, stream)
#define BDX_PAIRS_NW(RWV, NVV, KBV) (wp.nw <= 2 ? BDX_PAIRS_SP(RWV, NVV, KBV, 2) : wp.nw == 3 ? BDX_PAIRS_SP(RWV, NVV, KBV, 3) : BDX_PAIRS_SP(RWV, NVV, KBV, 4))
#define BDX_PAIRS_KB(RWV, NVV) (wp.pairs_kb <= 3 ? BDX_PAIRS_NW(RWV, NVV, 3) : BDX_PAIRS_NW(RWV, NVV, 4))
    if (wp.rw == 16 && wp.span_cap <= 3 * 1024 + 16) return BDX_PAIRS_KB(16, 3);
    if (wp.rw == 16 && wp.span_cap <= 6 * 1024 + 16) return BDX_PAIRS_KB(16, 6);
    return BDX_BAD_PLAN();
#undef BDX_PAIRS_KB
#undef BDX_PAIRS_NW
#undef BDX_PAIRS_SP
}

#elif !defined(BDX_WAVE_TU_PAIRS) && !defined(BDX_WAVE_TU_KEND) && !defined(BDX_WAVE_TU_KREV)
// LDS bytes of the shared tables / of one wave's work area (must mirror the kernel's carve-up)
size_t bdx_wave_table_bytes(const BdxWavePlan &wp, int hist_entries) {
    auto al = [](size_t x) { return (x + 31) & ~(size_t)31; };
    return al((size_t)wp.bm_bytes) + al(wp.pairs_kb > 0 ? 0 : (size_t)wp.bm_bytes / 2) + al((size_t)wp.n_ent * 4) + al((size_t)wp.n_barcodes * 36) +
           al(wp.kend >= 2 ? (size_t)wp.n_barcodes * 36 : 0) + 2 * al((size_t)wp.n_barcodes * 4) + al((size_t)hist_entries * 4);
}

size_t bdx_wave_area_bytes(int rw, int span_cap, bool pairs, int hq_cap, int sq_cap, int cand_words, bool winm) {
    const size_t nvec = (size_t)span_cap >> 4;
    const size_t recs = pairs ? 0 : 2 * (size_t)rw * 8 * 4;  // record tables
    const size_t fixed = (size_t)(((rw + 1) * 4 + 15) / 16 * 16) + recs + (size_t)rw * 16 + 3 * (size_t)rw * 4 + 256 +
                         ((pairs || winm) ? 2 * (size_t)rw * 4 + 16 + 2 * (size_t)rw * 20 : 0) + ((winm || pairs) ? (size_t)rw * 4 + 2 * (size_t)rw * 4 : 0);
    const size_t o = fixed + ((nvec + 2 + 3) & ~(size_t)3) * 4 + ((2 * nvec + 6 + 3) & ~(size_t)3) * 4 + ((size_t)hq_cap + (pairs ? 0 : (size_t)sq_cap) + (size_t)rw * (size_t)cand_words) * 4;
    return (o + 31) & ~(size_t)31;
}

hipError_t bdx_launch_wave(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                           long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int *d_tile_counter, int tier1,
                           double tier_slo, uint32_t *list, unsigned int *list_count, hipStream_t stream, int dbg, const BdxWaveSplit *sp,
                           double tier_slo1) {
    if (n_reads <= 0) return hipSuccess;
    (void)d_tile_counter;  // (tiles are dealt round robin: no queue)
    WaveArgs a;
    fill_args(a, cfg, wp, hist_entries, out, d_counts, list, list_count, dbg, sp);
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.tier = tier1;
    a.tier_slo = tier_slo;
    a.tier_slo1 = tier_slo1;
    a.dual = (!wp.split && cfg.is_dual) ? 1 : 0;
    if (a.dual && wp.cand_words != 4) return BDX_BAD_PLAN();  // (the survivors of pass 1 live in the candidate-word area: four per read)
    if (wp.pairs_kb > 0) return BDX_BAD_PLAN();
    if (wp.split && (!sp || !a.cand_out[0] || !a.wins_out[0] || !a.wcnt_out[0])) return BDX_BAD_PLAN();
    const size_t lds = bdx_wave_table_bytes(wp, hist_entries) + (size_t)wp.waves * (size_t)a.per_wave;
    const long long tiles = (n_reads + wp.rw - 1) / wp.rw;
    long long blocks = (long long)wp.blocks;
    const long long useful = (tiles + wp.waves - 1) / wp.waves;  // one tile per wave at least
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    const int tf = wp.track_from;
    // instantiated: seeds of 8 bases with every score-tracking start, 7 and 6 bases with the plain ones
    // (known-score dual configs and configs with a ref_search_range take the general form of the non-split kernel: it is
    // instantiated in bdx_wave_end.hip)
    if (!wp.split && (a.dual || a.ranged)) return bdx_launch_wave_gen(&a, wp, lds, blocks, stream);
    if (wp.split && a.ranged) return bdx_launch_wave_split_gen(&a, wp, lds, blocks, stream);  // (bdx_pairs.hip)
#define BDX_WAVE_SP(RWV, TFV, NVV, QV)                                                              \
    (wp.split ? launch_wave<RWV, TFV, NVV, QV, true, 0, 0, false, 0, false>(a, lds, wp.waves, blocks, stream) : launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 0, false>(a, lds, wp.waves, blocks, stream))
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

#elif defined(BDX_WAVE_TU_PAIRS)  // the pairs-mode instantiations live in a translation unit of their own (bdx_pairs.hip)

// Pairs mode over the reads of a list (d_idmap[0 .. *d_count), fetched straight from the batch; d_idmap == NULL: every read of
// the batch): final verdicts at the full budgets for the known-score class (what it cannot answer goes to `list`), candidate
// masks + windows in split mode.
hipError_t bdx_launch_pairs(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                            long long n_reads, const uint32_t *d_idmap, const unsigned int *d_count, const BdxDevOut &out,
                            unsigned long long *d_counts, uint32_t *list, unsigned int *list_count, hipStream_t stream, int dbg,
                            const BdxWaveSplit *sp, const BdxDevStats *stats) {
    if (wp.pairs_kb <= 0 || !d_seq || !d_off || n_reads <= 0 || (d_idmap && !d_count)) return BDX_BAD_PLAN();
    WaveArgs a;
    fill_args(a, cfg, wp, hist_entries, out, d_counts, list, list_count, dbg, sp);
    if (stats) {
        if (wp.kend != 3) return BDX_BAD_PLAN();
        a.stats = *stats;
    }
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.tier = 0;
    a.tier_slo = 0.0;
    a.slot = wp.slot;
    a.vps = wp.slot >> 4;
    a.vps_inv = (65536 + a.vps - 1) / a.vps;
    a.max_len = wp.read_len_hint;
    a.cpr = wp.cpr;
    a.cpr_inv = (65536 + wp.cpr - 1) / wp.cpr;
    a.idmap = d_idmap;
    a.n_dev = d_idmap ? d_count : nullptr;
    a.dual = (!wp.split && cfg.is_dual) ? 1 : 0;
    if (a.dual && wp.cand_words != 4) return BDX_BAD_PLAN();
    if (wp.split && (!sp || !a.cand_out[0] || !a.wins_out[0] || !a.wcnt_out[0])) return BDX_BAD_PLAN();
    if (wp.rw * wp.cpr > 32 * 40 || (wp.slot & 15) || wp.slot < 16 || wp.rw * wp.slot + 16 > wp.span_cap) return BDX_BAD_PLAN();
    const size_t lds = bdx_wave_table_bytes(wp, hist_entries) + (size_t)wp.waves * (size_t)a.per_wave;
    const long long blocks = wp.blocks < 1 ? 1 : wp.blocks;
    const int tf = wp.track_from;
#define BDX_PAIRS_SP(RWV, TFV, NVV, KBV, NWV)                                                                                  \
    (wp.split  ? launch_wave<RWV, TFV, NVV, 4, true, KBV, NWV>(a, lds, wp.waves, blocks, stream)                               \
     : wp.kend ? launch_wave<RWV, TFV, NVV, 4, false, KBV, NWV, false, 1>(a, lds, wp.waves, blocks, stream)                    \
               : launch_wave<RWV, TFV, NVV, 4, false, KBV, NWV>(a, lds, wp.waves, blocks, stream))
#define BDX_PAIRS_NW(RWV, TFV, NVV, KBV) (wp.groups > 1 ? launch_wave<RWV, TFV, NVV, 4, false, KBV, 4, true>(a, lds, wp.waves, blocks, stream) : wp.nw <= 2 ? BDX_PAIRS_SP(RWV, TFV, NVV, KBV, 2) : wp.nw == 3 ? BDX_PAIRS_SP(RWV, TFV, NVV, KBV, 3) : BDX_PAIRS_SP(RWV, TFV, NVV, KBV, 4))
// (same-diagonal variants 8 / 9: weighted costs, i.e. always split mode)
#define BDX_PAIRS_SD(RWV, TFV, NVV, KBV) (wp.nw <= 2 ? launch_wave<RWV, TFV, NVV, 4, true, KBV, 2>(a, lds, wp.waves, blocks, stream) : wp.nw == 3 ? launch_wave<RWV, TFV, NVV, 4, true, KBV, 3>(a, lds, wp.waves, blocks, stream) : launch_wave<RWV, TFV, NVV, 4, true, KBV, 4>(a, lds, wp.waves, blocks, stream))
#define BDX_PAIRS_KB(RWV, TFV, NVV) (wp.pairs_kb == 8 ? BDX_PAIRS_SD(RWV, TFV, NVV, 8) : wp.pairs_kb == 9 ? BDX_PAIRS_SD(RWV, TFV, NVV, 9) : wp.pairs_kb <= 3 ? BDX_PAIRS_NW(RWV, TFV, NVV, 3) : BDX_PAIRS_NW(RWV, TFV, NVV, 4))
#define BDX_PAIRS_TF(RWV, NVV) return BDX_PAIRS_KB(RWV, 12, NVV)
    // (4 (kb + 2) <= m makes m - kb - 1 >= 16: the first twelve columns of a sweep never need the score)
    if ((wp.pairs_kb > 4 && wp.pairs_kb != 8 && wp.pairs_kb != 9) || wp.nw > 4 || tf < 12 || wp.n_barcodes > 512 || (wp.groups > 1 && (wp.nw != 4 || wp.split || wp.kend))) return BDX_BAD_PLAN();
    if (wp.pairs_kb >= 8 && (!wp.split || wp.groups > 1)) return BDX_BAD_PLAN();
    if (wp.kend && !wp.d_peq8r) return BDX_BAD_PLAN();
    if (wp.kend == 3) {
        if (wp.pairs_kb > 4 || wp.groups > 1 || wp.split) return BDX_BAD_PLAN();
        return bdx_launch_pairs_aln(&a, wp, lds, blocks, stream);  // (bdx_wave_aln.hip)
    }
    if (wp.kend && out.pass_start != nullptr) return BDX_BAD_PLAN();
    if (wp.kend && out.pass_end != nullptr && (a.trim0 == 3 || a.trim1 == 3)) return BDX_BAD_PLAN();
    if ((wp.kend == 2) != (wp.kend && (a.trim0 == 3 || a.trim1 == 3))) return BDX_BAD_PLAN();
    if (wp.kend == 2) return bdx_launch_pairs_rev(&a, wp, lds, blocks, stream);  // (bdx_wave_rev.hip)
    if (wp.rw == 16 && wp.span_cap <= 3 * 1024 + 16) BDX_PAIRS_TF(16, 3);
    if (wp.rw == 16 && wp.span_cap <= 6 * 1024 + 16) BDX_PAIRS_TF(16, 6);
    return BDX_BAD_PLAN();
#undef BDX_PAIRS_TF
#undef BDX_PAIRS_KB
#undef BDX_PAIRS_SD
#undef BDX_PAIRS_NW
#undef BDX_PAIRS_SP
}

// The split-mode kernel with per-read column windows (ref_search_range), for bdx_launch_wave.
hipError_t bdx_launch_wave_split_gen(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    const int tf = wp.track_from;
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, true, 0, 0, false, 0, true>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

#elif defined(BDX_WAVE_TU_KEND)  // the known-trim instantiations without reversed sweeps + the general non-split form (bdx_wave_end.hip)

// The kernel for configs of the known-end class (ScoreOnly conditions + trim_side = 5, no start positions wanted): same
// launch as bdx_launch_wave for a known-score config, the verdicts carry the trimmed keep range.
hipError_t bdx_launch_wave_end(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                               long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int tier1, double tier_slo, uint32_t *list,
                               unsigned int *list_count, hipStream_t stream, int dbg, double tier_slo1, const BdxDevStats *stats) {
    if (n_reads <= 0) return hipSuccess;
    if (wp.pairs_kb > 0 || wp.split || !wp.kend || !wp.d_peq8r) return BDX_BAD_PLAN();
    if (wp.kend != 3 && (out.pass_start != nullptr || stats != nullptr)) return BDX_BAD_PLAN();  // (only the known-alignment class knows both positions)
    WaveArgs a;
    fill_args(a, cfg, wp, hist_entries, out, d_counts, list, list_count, dbg, nullptr);
    if (wp.kend != 3 && out.pass_end != nullptr && (a.trim0 == 3 || a.trim1 == 3)) return BDX_BAD_PLAN();  // (a trim_side = 3 pass knows its start only)
    if (stats) a.stats = *stats;
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.tier = tier1;
    a.tier_slo = tier_slo;
    a.tier_slo1 = tier_slo1;
    a.dual = cfg.is_dual ? 1 : 0;
    if (a.dual && wp.cand_words != 4) return BDX_BAD_PLAN();  // (the survivors of pass 1 live in the candidate-word area: four per read)
    const size_t lds = bdx_wave_table_bytes(wp, hist_entries) + (size_t)wp.waves * (size_t)a.per_wave;
    const long long tiles = (n_reads + wp.rw - 1) / wp.rw;
    long long blocks = (long long)wp.blocks;
    const long long useful = (tiles + wp.waves - 1) / wp.waves;
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    if (wp.kend == 3) return bdx_launch_wave_end_aln(&a, wp, lds, blocks, stream);  // (bdx_wave_aln.hip)
    if ((wp.kend == 2) != (a.trim0 == 3 || a.trim1 == 3)) return BDX_BAD_PLAN();
    if (wp.kend == 2) return bdx_launch_wave_end_rev(&a, wp, lds, blocks, stream);  // (bdx_wave_rev.hip)
    const int tf = wp.track_from;
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 1>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

// The general form of the non-split kernel (dual configs, ref_search_range windows) for bdx_launch_wave.
hipError_t bdx_launch_wave_gen(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    const int tf = wp.track_from;
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 0, true>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

#else  // BDX_WAVE_TU_KREV: the known-trim instantiations with reversed sweeps (a trim_side = 3 pass; bdx_wave_rev.hip)

hipError_t bdx_launch_wave_end_rev(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    const int tf = wp.track_from;
#define BDX_WAVE_SP(RWV, TFV, NVV, QV) launch_wave<RWV, TFV, NVV, QV, false, 0, 0, false, 2>(a, lds, wp.waves, blocks, stream)
#define BDX_WAVE_NV(RWV, TFV, QV) (wp.span_cap <= 5 * 1024 ? BDX_WAVE_SP(RWV, TFV, 5, QV) : BDX_WAVE_SP(RWV, TFV, 10, QV))
#define BDX_WAVE_TF(RWV)                                                                             \
    return wp.q == 8 ? (tf >= 20 ? BDX_WAVE_NV(RWV, 20, 8) : tf >= 12 ? BDX_WAVE_NV(RWV, 12, 8) : BDX_WAVE_NV(RWV, 0, 8)) \
           : wp.q == 7 ? (tf >= 12 ? BDX_WAVE_NV(RWV, 12, 7) : BDX_WAVE_NV(RWV, 0, 7))                      \
                       : BDX_WAVE_NV(RWV, 0, 6)
    switch (wp.rw) {
        case 32:
            BDX_WAVE_TF(32);
        case 16:
            BDX_WAVE_TF(16);
        case 8:
            BDX_WAVE_TF(8);
        default:
            return BDX_BAD_PLAN();
    }
#undef BDX_WAVE_TF
#undef BDX_WAVE_NV
#undef BDX_WAVE_SP
}

hipError_t bdx_launch_pairs_rev(const void *wave_args, const BdxWavePlan &wp, size_t lds, long long blocks, hipStream_t stream) {
    const WaveArgs &a = *(const WaveArgs *)wave_args;
    if (wp.pairs_kb > 4 || wp.nw > 4 || wp.track_from < 12 || wp.groups > 1 || wp.split || wp.kend != 2) return BDX_BAD_PLAN();
#define BDX_PAIRS_SP(RWV, NVV, KBV, NWV) launch_wave<RWV, 12, NVV, 4, false, KBV, NWV, false, 2>(a, lds, wp.waves, blocks, stream)
#define BDX_PAIRS_NW(RWV, NVV, KBV) (wp.nw <= 2 ? BDX_PAIRS_SP(RWV, NVV, KBV, 2) : wp.nw == 3 ? BDX_PAIRS_SP(RWV, NVV, KBV, 3) : BDX_PAIRS_SP(RWV, NVV, KBV, 4))
#define BDX_PAIRS_KB(RWV, NVV) (wp.pairs_kb <= 3 ? BDX_PAIRS_NW(RWV, NVV, 3) : BDX_PAIRS_NW(RWV, NVV, 4))
    if (wp.rw == 16 && wp.span_cap <= 3 * 1024 + 16) return BDX_PAIRS_KB(16, 3);
    if (wp.rw == 16 && wp.span_cap <= 6 * 1024 + 16) return BDX_PAIRS_KB(16, 6);
    return BDX_BAD_PLAN();
#undef BDX_PAIRS_KB
#undef BDX_PAIRS_NW
#undef BDX_PAIRS_SP
}

#endif
