// bdx_bitpar.hip — fused "bit-parallel lower bound -> exact verify" kernel for gfx950.
//
// Stage 1 (filter, the dominant cost): for every (read, barcode) pair of the workgroup one lane
// sweeps the read's column window with Myers' bit-vector recurrence (unit costs, one 32-bit
// word per barcode, free start and free end in the read) and obtains
//      d*(read, barcode) = min over substrings of the unit-cost edit distance,
// with barcode 'N' treated as a wildcard whenever the reference does (NScoring / :hamming).
// The reference can only RECORD an alignment whose cost is <= allowed_error, every edit
// operation costs >= cmin = min(mismatch, indel[, nindel]) >= 1 and matches cost >= 0, so a
// recorded alignment has at most floor(allowed_error / cmin) operations; band seeds, the
// cut-off, the last-row rule and the start/end ranges only ever REMOVE alignments
// (classification.jl:238-445).  Hence
//      d* > kb := floor(floor(rate * norm) / cmin)   ==>   the reference returns Inf for this
// barcode at the initial threshold and at every tightened one (:661, :701, :706 only lower it),
// and an Inf result never changes the reducer state (:658, :696).  Dropping such barcodes is
// therefore lossless; the filter needs no parity argument of its own beyond this one.
//
// Stage 2 (verify): one lane per read runs the line-faithful evaluation of bdx_core.h over the
// surviving barcodes in file order — bit-exact by construction.
//
// MI355X mapping: a workgroup of BS lanes owns R consecutive reads.  Their bytes are copied
// HBM -> LDS once with 16-byte coalesced loads and transcoded to <= 8 symbol codes.  The Peq
// table lives in LDS as peq[code][barcode] with the barcode stride padded to a multiple of 32
// dwords: the 64 lanes of a wave hold consecutive barcodes (stride a power of two), so the table read is conflict-free
// for any code, and lanes of the same read fetch the same symbol byte (LDS broadcast).  Pairs
// are flattened (pair = read * B + barcode) so lanes stay busy for any B (96 = 1.5 waves).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "bdx_core.h"

namespace {

struct BitparArgs {
    BdxDevCfg cfg;
    const uint8_t *seq;
    const long long *off;
    long long n_reads;
    BdxDevOut out;
    unsigned long long *counts;
    int stage_bytes;     // capacity of EACH of the two staging areas (raw bytes, codes)
    int hist_entries;
    const uint8_t *lut;  // 256 bytes: byte -> symbol code
    const void *peq[2];     // [ncodes][bpad] sweep words: uint32_t, or uint64_t for barcodes of 33..64 nt (W64)
    const void *pvinit[2];  // [B]
    const int32_t *kb[2];
    int ncodes;
    int bpad[2];
    int bshift[2];
    // q-gram seeding (SEED variant only)
    int seed_q, seed_groups, seed_hash_log2, seed_bm_words, seed_bm_log2;
    const uint32_t *dmeta[2];  // DIAG variant: per barcode pieces | piece length << 8 (0: not seeded)
    const uint32_t *dkeys[2];  //   ... and the 8-bit keys of its pieces (2 words)
    int diag_kmax, diag_qcap;
    int seed_rcap;         // sweep records per read (power of two, sized to the expected seeded barcodes)
    int seed_qmul;         // sweep-record queue entries per read (the hit queue has twice as many)
    int seed_hash_in_lds;  // 0: the hash table is probed in L2 (large barcode sets)
    int seed_n_always[2];
    const uint32_t *seed_bitmap;
    const uint32_t *seed_hash;
    const uint8_t *seed_hash_ps;
    const uint16_t *seed_always[2];
    uint32_t *wins_out[2];  // split mode: [n_reads][BDX_WCAP][3] = {barcode, first column, last column} of the exact run
    uint8_t *wcnt_out[2];   // split mode: entries valid per read (255 = none: whole window)
    uint32_t *cand_out[2];  // candidate masks in HBM for the reads the exact kernel evaluates (split: all of them)
    int split;              // 1: this kernel only filters, every read's verdict comes from the exact kernel
    uint32_t *exc_list;     // known-score mode: reads handed to the exact kernel after all (see stage 2)
    unsigned int *exc_count;
    int *tile_counter;  // zeroed before every launch: dynamic tile queue
    int known_ok[2];  // config-level eligibility of the known-score class per pass
    int ncode;  // symbol code of 'N' (255 when no barcode contains it)
    int slot_bytes;  // > 0: per-read window slots instead of the flat span copy
    int slot_cap;    // survivors per read and pass the reducer replay can take (>= 4)
    int dense_d;     // plain-sweep kernels: byte table of every candidate's d instead of the slots
    int dense_w;     // plain-sweep kernels, split mode: dense window table wins_out[pass][read][barcode]
    int short_lb[2]; // per pass: the exact kernel only reports score (+ end) through the clean-class DP: its restricted
                     // run may start m + kb columns before the first end column instead of 2 (m + kb) + 1
    // tiered budgets (bdx_abi.cpp): tier 1 appends the reads it cannot settle to tier_list; tier 0 then runs in
    // LIST MODE over exactly those reads (in_list / *in_count; slot staging, as the reads are scattered)
    int tier;                      // 1: this launch is tier 1 (capped budgets)
    double tier_slo[2];            //   per pass: smallest score of a barcode beyond its capped budget
    uint32_t *tier_list;
    unsigned int *tier_count;
    const uint32_t *in_list;
    const unsigned int *in_count;
    int dbg;  // timing experiments (env BDX_DEBUG), compiled in ONLY with -DBDX_TUNING — results are wrong when a
              // skip bit is set: 1 skip stage 2, 2 skip sweeps, 4 skip hit resolve, 8 skip seed scan, 32 skip
              // transcode, 64 skip copy; 128 = sweep statistics (results stay correct)
};

// The product library has no phase-skip switches: BDX_DBG folds to 0 and the branches disappear.
// reads indexed at a time by the two-intact-pieces variant (5 KiB of index each).  4 instead of 8 gives a fourth
// workgroup per CU in list mode: measured +3 % on C2d, +6 % at 384 barcodes, -3.5 % at 96 barcodes on small batches
#ifndef BDX_DIAG_SB_NARROW
#define BDX_DIAG_SB_NARROW 8
#endif
#ifdef BDX_TUNING
#define BDX_DBG(bit) (a.dbg & (bit))
#else
#define BDX_DBG(bit) 0
#endif

// The kernel holds no DP state (the exact stage lives in bdx_generic_kernel), which keeps it at ~100
// VGPRs and ~37 KiB of LDS for a 64-read tile: 4 workgroups = 16 waves per CU.  The phases of a tile
// are short and barrier-separated, so throughput follows the number of resident waves closely.
// WL: log2 of the sweep word in 32-bit units — 1: 64-bit words (barcodes of 33..64 nt; every operation is two VALU
// instructions), 2: 128-bit words (65..128 nt; four) — the plain sweep and the single-seed variant; the diagonal variant
// stays 32-bit.
template <int BS, int R, bool SEED, bool DIAG, int NW = 5, int WL = 0>
__global__ __launch_bounds__(BS, 4) void bdx_bitpar_kernel(const BitparArgs a) {
    static_assert(!DIAG || SEED, "the diagonal variant is a seeded variant");
    static_assert(!DIAG || WL == 0, "the diagonal variant has 32-bit sweep words");
    using WT = typename std::conditional<WL == 2, unsigned __int128, typename std::conditional<WL == 1, unsigned long long, uint32_t>::type>::type;
    constexpr int WB = 4 << WL;  // bytes per sweep word
    const auto popw = [](const WT x) __attribute__((always_inline)) {
        if constexpr (WL == 2)
            return (int)__builtin_popcountll((unsigned long long)x) + (int)__builtin_popcountll((unsigned long long)(x >> 64));
        else if constexpr (WL == 1)
            return (int)__builtin_popcountll((unsigned long long)x);
        else
            return (int)__builtin_popcount((uint32_t)x);
    };
    constexpr bool HASH = SEED && !DIAG;  // single-piece seeds: bitmap + hash table + record tables
    // NW (DIAG): position words per 4-mer key: 5 for reads of <= 152 staged bases, 10 for <= 312
    constexpr int SBMAX = NW <= 5 ? BDX_DIAG_SB_NARROW : 4;  // index sub-batch: SBMAX x 5 KiB (NW = 5) or x 10 KiB
    constexpr int SB = DIAG ? (R < SBMAX ? R : SBMAX) : R;  // DIAG: reads indexed at a time (5 KiB of index each); larger tiles are walked in sub-batches
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LDS unsigned char *smem = (LDS unsigned char *)smem_raw;
    const BdxDevCfg &cfg = a.cfg;
    const int tid = threadIdx.x;
    const int RCAP = SEED ? a.seed_rcap : 8;
    const int RCAP_LOG2 = 31 - __builtin_clz(RCAP);
    // queue capacities per read follow the expected number of seeded (read, barcode) pairs (seed_qmul, >= 4: sized
    // on the host from the piece count; very many barcodes seed ~10 pairs per read by chance)
    const int SQCAP = SEED && !DIAG ? 2 * a.seed_qmul * R : 8 * R;  // capacity of the seed-hit queue
    const int PQCAP = DIAG ? a.diag_qcap * SB : (SEED ? a.seed_qmul * R : 4 * R);  // capacity of the sweep-record queue
    const int npass = cfg.is_dual ? 2 : 1;
    const int B0 = cfg.pass[0].n_barcodes;
    const int B1 = cfg.is_dual ? cfg.pass[1].n_barcodes : 0;
    const int cw0 = cfg.pass[0].cand_words, cw1 = cfg.is_dual ? cfg.pass[1].cand_words : 0;

    // ---- LDS carve-up (every region 16-byte aligned) ----
    size_t o = 0;
    auto take = [&](size_t bytes) -> LDS unsigned char * {
        LDS unsigned char *p = smem + o;
        o = (o + bytes + 15) & ~(size_t)15;
        return p;
    };
    // the seed bitmap sits at LDS offset 0: its byte probes then need no base address arithmetic
    LDS uint32_t *sbm = (LDS uint32_t *)take(HASH ? (size_t)a.seed_bm_words * 4 : 0);
    LDS int *hist = (LDS int *)take((size_t)a.hist_entries * 4);
    LDS unsigned char *lut = take(256);
    LDS WT *peq0 = (LDS WT *)take((size_t)a.ncodes * a.bpad[0] * WB);
    LDS WT *peq1 = (LDS WT *)take(cfg.is_dual ? (size_t)a.ncodes * a.bpad[1] * WB : 0);
    LDS WT *pv0 = (LDS WT *)take((size_t)B0 * WB);
    LDS WT *pv1 = (LDS WT *)take((size_t)B1 * WB);
    LDS int *kb0 = (LDS int *)take((size_t)B0 * 4);
    LDS int *kb1 = (LDS int *)take((size_t)B1 * 4);
    LDS uint32_t *cand = (LDS uint32_t *)take((size_t)R * (cw0 + cw1) * 4);
    LDS int *roff = (LDS int *)take((size_t)(R + 1) * 4);   // stage offset of base 1 of each read (may precede its slot)
    LDS int *rlen = (LDS int *)take((size_t)R * 4);         // read length n
    LDS uint32_t *rids = (LDS uint32_t *)take((size_t)R * 4);  // index of each read in the batch (list mode: from the list)
    LDS int *wlo = (LDS int *)take((size_t)R * 4);          // first staged base (0-based) of each read
    LDS int *wlen = (LDS int *)take((size_t)R * 4);         // staged bases of each read
    LDS int *win = (LDS int *)take((size_t)R * 4 * 4);      // [pass][first|last][R]
    const int SC = (!SEED && a.dense_d) ? 4 : a.slot_cap;  // survivors per read and pass the reducer replay can take (4; more for short barcodes)
    LDS uint32_t *slots = (LDS uint32_t *)take((size_t)2 * R * SC * 4);  // [pass][R][SC] (barcode << 8 | d)
    LDS int *scnt = (LDS int *)take((size_t)2 * R * 4);                 // [pass][R] entries pushed
    LDS unsigned char *full = take((size_t)2 * R);                      // [pass][R] read is in the known-score class
    // one staging area: the raw bytes are transcoded IN PLACE to symbol codes; the in-kernel exact
    // stage then compares codes (equal bytes <=> equal codes for every byte a barcode contains)
    LDS unsigned char *rstage = take((size_t)a.stage_bytes + 16);
    LDS unsigned char *codes = rstage;
    // seeding work areas (SEED variant only)
    LDS uint32_t *shash = (LDS uint32_t *)take(HASH && a.seed_hash_in_lds ? ((size_t)4 << a.seed_hash_log2) : 0);
    LDS unsigned char *spk = take(SEED ? (size_t)(a.stage_bytes >> 2) + 32 : 0);  // flat 2-bit image of the staging area
    LDS unsigned char *shps = take(HASH && a.seed_hash_in_lds ? ((size_t)1 << a.seed_hash_log2) : 0);  // piece start of each hash entry
    LDS uint32_t *shq = (LDS uint32_t *)take(HASH ? (size_t)SQCAP * 4 : (DIAG ? (size_t)2 * PQCAP * 4 : 0));      // seed hits: position << 16 | key
    LDS unsigned char *shr = take(HASH ? (size_t)SQCAP : 0);                       // ... and their read
    LDS uint32_t *srid = (LDS uint32_t *)take(HASH ? (size_t)R * RCAP * 4 : 0);   // per-read sweep records: id
    LDS int *srlo = (LDS int *)take(HASH ? (size_t)R * RCAP * 4 : 0);             //   window start (min)
    LDS int *srhi = (LDS int *)take(HASH ? (size_t)R * RCAP * 4 : 0);             //   window end (max)
    // the sweep-record queue reuses the hit queue (dead once the hits are resolved; SQCAP = 2 * PQCAP)
    LDS uint32_t *spq = shq;          // read << 16 | pass << 15 | barcode + 1
    LDS uint32_t *spw = shq + PQCAP;  // window lo << 16 | hi
    LDS int *sqn = (LDS int *)take(32);  // [0] hits, [1] pairs, [2] current tile, [3] slot overflow, [4] window-queue fill, [5] some read needs the whole-read fallback
    // split mode never enters the known-score class, so its work areas reuse that class's slots / counters
    LDS uint32_t *wq = slots;  // split mode: candidates to re-sweep with column tracking (4 R entries)
    LDS int *wcl = scnt;       // split mode: window entries written per read and pass
    LDS unsigned char *sall = take(SEED ? (size_t)R : 0);
    LDS unsigned char *act = take(!SEED && cfg.is_dual ? (size_t)R : 0);  // plain sweep, dual: reads whose first pass has a candidate
    // plain sweep, known-score class, few barcodes: d of every (read, barcode) candidate — each pair is swept exactly once
    const bool dense = !SEED && a.dense_d;
    LDS unsigned char *dtab = take(dense ? (size_t)R * (B0 + B1) : 0);  // [R][B0] then [R][B1]
    LDS uint32_t *slh = (LDS uint32_t *)take(SEED ? (size_t)R * 4 : 0);  // per read: first | (last + 1) << 16 seed start, relative to the first staged base
    LDS int *srw = (LDS int *)take(SEED ? (size_t)R * 4 : 0);            // per read: stage offset of its first staged base
    // DIAG variant: per-read inverted index of 4-mers (bit p of occ[r][key][.] <=> the 4-mer at staged position p is key)
    LDS uint32_t *occ = (LDS uint32_t *)take(DIAG ? (size_t)SB * 256 * NW * 4 : 0);
    LDS uint32_t *dm0 = (LDS uint32_t *)take(DIAG ? (size_t)B0 * 4 : 0);
    LDS uint32_t *dm1 = (LDS uint32_t *)take(DIAG ? (size_t)B1 * 4 : 0);
    LDS uint32_t *dk0 = (LDS uint32_t *)take(DIAG ? (size_t)B0 * 8 : 0);
    LDS uint32_t *dk1 = (LDS uint32_t *)take(DIAG ? (size_t)B1 * 8 : 0);

    // list mode with an empty list (the usual case behind the wave kernel): nothing to do — leave before the tables are loaded.
    // (Also letting the workgroups beyond a SHORT list's tiles leave here made C2's list launch 36 -> 24 us but C5's tier 0
    // 0.21 -> 0.27 ms: dropped.)
    if (a.in_list != nullptr && *a.in_count == 0u) return;
    // ---- tables -> LDS ----
    for (int i = tid; i < B0; i += BS) {
        pv0[i] = ((const WT *)a.pvinit[0])[i];
        kb0[i] = a.kb[0][i];
    }
    for (int i = tid; i < a.ncodes * a.bpad[0]; i += BS) peq0[i] = ((const WT *)a.peq[0])[i];
    if (cfg.is_dual) {
        for (int i = tid; i < B1; i += BS) {
            pv1[i] = ((const WT *)a.pvinit[1])[i];
            kb1[i] = a.kb[1][i];
        }
        for (int i = tid; i < a.ncodes * a.bpad[1]; i += BS) peq1[i] = ((const WT *)a.peq[1])[i];
    }
    for (int i = tid; i < 256; i += BS) lut[i] = a.lut[i];
    for (int i = tid; i < a.hist_entries; i += BS) hist[i] = 0;
    if (DIAG) {
        for (int i = tid; i < B0; i += BS) dm0[i] = a.dmeta[0][i];
        for (int i = tid; i < 2 * B0; i += BS) dk0[i] = a.dkeys[0][i];
        for (int i = tid; i < B1; i += BS) dm1[i] = a.dmeta[1][i];
        for (int i = tid; i < 2 * B1; i += BS) dk1[i] = a.dkeys[1][i];
    }
    if (HASH) {
        for (int i = tid; i < a.seed_bm_words; i += BS) sbm[i] = a.seed_bitmap[i];
        if (a.seed_hash_in_lds)
            for (int i = tid; i < (1 << a.seed_hash_log2); i += BS) {
                shash[i] = a.seed_hash[i];
                shps[i] = a.seed_hash_ps[i];
            }
    }
    __syncthreads();
    // kernel-argument arrays indexed by a per-lane pass number would be fetched with vector loads from
    // the argument segment: select between two scalars instead
    const int bsh0 = a.bshift[0], bsh1 = a.bshift[1];

    // ---- persistent workgroup: the tables above are loaded once, then the workgroup walks
    // tiles of R consecutive reads (tile = blockIdx.x, + gridDim.x, ...).  Tiles are independent;
    // nothing is exchanged between workgroups, so no placement or ordering is assumed. ----
    const bool lm = a.in_list != nullptr;  // list mode: the reads named by in_list[0 .. *in_count)
    long long n_eff = a.n_reads;
    if (lm) {
        const long long c = (long long)*a.in_count;
        n_eff = c < a.n_reads ? c : a.n_reads;
    }
    const long long ntiles = (n_eff + R - 1) / R;
    // the tile queue is read one tile ahead: the returning atomic's L2 round trip (1-3 us) overlaps
    // the current tile's work instead of heading every tile (each workgroup over-fetches one index)
    int next_tile = 0;
    if (tid == 0) next_tile = atomicAdd(a.tile_counter, 1);
    for (;;) {
    __syncthreads();  // the previous tile's stage 2 is done with the per-tile LDS state
    if (tid == 0) {
        sqn[2] = next_tile;  // dynamic tile queue (exit: queue drained)
        // per-tile flags that other lanes SET before the next barrier are cleared here, one barrier earlier
        sqn[3] = 0;
        sqn[4] = 0;
        sqn[5] = 0;
        next_tile = atomicAdd(a.tile_counter, 1);
    }
    __syncthreads();
    const long long tile = sqn[2];
    if (tile >= ntiles) break;
    // Phases with fewer items than lanes (per-read setup, hit resolve, sweeps, reducer replay) would always land
    // on the first waves — and with them on the same SIMDs of the CU.  The lane numbering of those phases is
    // rotated by one wave per tile, so that over a workgroup's tiles all four SIMDs carry them.
    const int ltid = (tid + (((int)tile & (BS / 64 - 1)) << 6)) & (BS - 1);
    for (int i = tid; i < R * (cw0 + cw1); i += BS) cand[i] = 0;
    for (int i = tid; i < 2 * R; i += BS) scnt[i] = 0;
    if (SEED) {
        for (int i = ltid; i < R; i += BS) sall[i] = 0;  // (same lane as the per-read setup below, which may set it)
        if (HASH)
            for (int i = tid; i < R * RCAP; i += BS) {
                srid[i] = 0u;
                srlo[i] = 0x7FFFFFFF;
                srhi[i] = 0;
            }
        if (DIAG) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = tid; i < SB * 256 * NW / 4; i += BS) ((LDS u32x4 *)occ)[i] = z;
        }
        if (tid < 2) sqn[tid] = 0;
    }
    // ---- this tile's reads [r0, r1): one contiguous span of the packed batch ----
    const long long r0 = tile * R;  // (list mode: a position in the list)
    long long r1 = r0 + R;
    if (r1 > n_eff) r1 = n_eff;
    const int nr = (int)(r1 - r0);
    // Two staging modes (chosen on the host from the config's ranges and the read-length hint):
    //  * flat: the tile's reads are one contiguous span of the batch -> one coalesced copy;
    //  * slot (a.slot_bytes > 0): only each read's column window is needed (e.g. 10 kbp reads
    //    with ref_search_range "1:200"): every read gets a fixed-size slot holding just the
    //    union of its pass windows, so HBM traffic and LDS follow the window, not the read.
    const int slot = a.slot_bytes;
    const bool sgm = cfg.algorithm == BDX_ALG_SEMIGLOBAL;
    const bool virt = cfg.vlen != nullptr;  // window upload: per-read slots as well (the host plans slot staging)
    const long long span0 = (lm || virt) ? 0 : a.off[r0];  // (list mode always stages per-read slots)
    const long long span1 = (lm || virt) ? 0 : a.off[r1];
    const uintptr_t g0 = (uintptr_t)(a.seq + span0);
    const uintptr_t g0a = g0 & ~(uintptr_t)15;
    const int head = (int)(g0 - g0a);
    const long long need = slot ? (long long)nr * slot : (span1 - span0) + head;
    bool staged = slot ? true : (need + 16 <= (long long)a.stage_bytes);  // wave-uniform (whole workgroup)
    if (!slot && staged && !BDX_DBG(64)) {
        const int nvec = (int)((need + 15) >> 4);
        const GlobalVec16 src = (GlobalVec16)g0a;
        LDS u32x4 *dst = (LDS u32x4 *)rstage;
        for (int k = tid; k < nvec; k += BS) dst[k] = __builtin_nontemporal_load(src + k);
    }
    // per-read lengths, stage offsets and column windows of both passes
    for (int t = ltid; t < nr; t += BS) {
        const long long rid = lm ? (long long)a.in_list[r0 + t] : r0 + t;
        rids[t] = (uint32_t)rid;
        const long long ro = a.off[rid] - (virt ? (long long)cfg.vlo[rid] : 0);  // where position 0 of the read would be
        const long long rn = virt ? (long long)cfg.vlen[rid] : a.off[rid + 1] - a.off[rid];
        const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
        rlen[t] = n;
        int ulo = 0x7FFFFFFF, uhi = 0;
        int f_p0 = 1, l_p0 = 0;
        for (int p = 0; p < npass; ++p) {
            PassWindow w;
            const bool ok = pass_window(cfg.pass[p], n, w);
            int f = ok ? (w.first > 1 ? w.first : 1) : 1;
            int l = ok ? (w.last < n ? w.last : n) : 0;
            if (p == 0) {
                f_p0 = f;
                l_p0 = l;
            }
            win[(p * 2 + 0) * R + t] = f;
            win[(p * 2 + 1) * R + t] = l;
            if (l >= f) {
                // :hamming / :exact occurrences reach max_m - 1 past the last start position
                int h = sgm ? l : l + cfg.max_m - 1;
                if (h > n) h = n;
                ulo = f - 1 < ulo ? f - 1 : ulo;
                uhi = h > uhi ? h : uhi;
            }
            // known-score class, per read: neither the start nor the end range binds (band_offset =
            // m-n-steps, min_valid_start <= 1, j >= min_end always); any column window first:last
            full[p * R + t] = (unsigned char)(a.known_ok[p] && ok && n > 0 && w.max_start >= n && w.min_end <= 1);
        }
        if (uhi <= ulo) {
            ulo = 0;
            uhi = 0;
        }
        if (slot) {
            const int hd = (int)((uintptr_t)(a.seq + ro + ulo) & 15);
            if (uhi - ulo + hd + 16 > slot) sqn[3] = 1;  // window larger than planned: tile not staged
            roff[t] = t * slot + hd - ulo;
            wlo[t] = ulo;
            wlen[t] = uhi - ulo;
        } else {
            roff[t] = head + (int)(ro - span0);
            wlo[t] = 0;
            wlen[t] = n;
        }
        if (SEED) {
            // seed scan parameters of this read: the 0-based start positions [lo, hi] that may begin a
            // seed, relative to the first staged base
            const int q = a.seed_q;
            const int base = slot ? ulo : 0, wl = slot ? uhi - ulo : n;
            int lo, hi;
            if (npass == 1) {
                lo = f_p0 - 1;
                hi = (sgm ? l_p0 : n) - q;
                if (l_p0 < f_p0) hi = -1;
            } else {
                lo = base;
                hi = base + wl - q;
            }
            if (lo < base) lo = base;
            if (hi > base + wl - q) hi = base + wl - q;  // never beyond the staged bases
            int lor = lo - base, hir = hi - base;
            if (hir < lor || hir < 0 || lor > 0xFFFF) {
                lor = 1;
                hir = 0;
            }
            if (hir > 0xFFFE) hir = 0xFFFE;  // (positions are 16-bit throughout the seeded path)
            slh[t] = (uint32_t)lor | ((uint32_t)(hir + 1) << 16);
            srw[t] = roff[t] + base;
            // a read longer than the planned group count would leave its tail unscanned:
            // sweep every barcode of it instead (lossless fallback)
            if (wl + 15 > 16 * a.seed_groups || wl > 0xFFF0 || (DIAG && wl > 32 * NW - 8)) sall[t] = 1, sqn[5] = 1;
        }
    }
    __syncthreads();
    if (slot) {
        staged = sqn[3] == 0;
        if (staged) {
            const int cpr = slot >> 4;  // 16-byte chunks per slot
            for (int idx = tid; idx < nr * cpr; idx += BS) {
                const int r = idx / cpr, k = idx - r * cpr;
                const int hd = roff[r] - r * slot + wlo[r];
                if (16 * k < wlen[r] + hd) {
                    const long long rb = a.off[rids[r]] - (virt ? (long long)cfg.vlo[rids[r]] : 0);
                    const uintptr_t src = ((uintptr_t)(a.seq + rb + wlo[r]) & ~(uintptr_t)15) + 16u * (unsigned)k;
                    *(LDS u32x4 *)(rstage + r * slot + 16 * k) = __builtin_nontemporal_load((GlobalVec16)src);
                }
            }
        }
        __syncthreads();
    }

    // ---- stage 1: Myers bit-vector sweep, one lane per (read, barcode) pair ----
    // Two independent pairs per lane per trip: the recurrence is a serial chain of ~14
    // dependent VALU ops per column, a second chain fills the issue slots the first leaves.
    const bool sg = cfg.algorithm == BDX_ALG_SEMIGLOBAL;
    struct Sweep {
        const LDS unsigned char *c;
        const LDS unsigned char *pq;
        WT Pv, Mv;
        int score, best, ncol, r, b, p;
        int e_lo, e_hi, kbv;  // tracked sweeps (split mode of the seeded variants): first / last column j with score <= kbv
    };
    // wlo_rel / whi_rel: optional column sub-window [lo, hi) relative to the read's first staged
    // base (seeded path); (0, 0xFFFF) = the whole pass window
    auto setup = [&](const bool valid, const int p, const int r, const int b, Sweep &w, const int wlo_rel = 0,
                     const int whi_rel = 0xFFFF) __attribute__((always_inline)) {
        w.ncol = 0;
        w.r = 0;
        w.b = 0;
        w.p = p;
        w.c = codes;
        w.pq = (const LDS unsigned char *)peq0;
        w.Pv = 0;
        w.Mv = 0;
        w.score = 0;
        w.best = 0x7FFFFFFF;
        w.e_lo = 0;
        w.e_hi = -1;
        w.kbv = -1;
        if (!valid) return;
        int jf = win[(p * 2 + 0) * R + r];
        int jl = win[(p * 2 + 1) * R + r];
        if (jl < jf) return;
        const bool sub = whi_rel != 0xFFFF;
        w.r = r;
        w.b = b;
        w.Pv = (p ? pv1 : pv0)[b];
        w.score = popw(w.Pv);  // = barcode length m
        w.kbv = (p ? kb1 : kb0)[b];
        w.best = w.score;
        if (!sg) {
            // :hamming / :exact bound the START positions by the window (SURVEY Q11,
            // classification.jl:490-491, :570-571); the occurrence itself reaches m-1 further
            const int nread = rlen[r];
            jl = jl + w.score - 1 < nread ? jl + w.score - 1 : nread;
        }
        if (sub) {  // intersect with the seed window (1-based inclusive columns)
            const int a1 = wlo[r] + wlo_rel + 1, b1 = wlo[r] + whi_rel;
            jf = a1 > jf ? a1 : jf;
            jl = b1 < jl ? b1 : jl;
            if (jl < jf) {
                w.ncol = 0;
                return;
            }
        }
        w.c = codes + roff[r] + (jf - 1);
        w.ncol = jl - jf + 1;
        w.pq = (const LDS unsigned char *)((p ? peq1 : peq0) + b);
    };
    auto step = [&](Sweep &w, const int j, const int sh) __attribute__((always_inline)) {
        const WT Eq = *(const LDS WT *)(w.pq + ((uint32_t)w.c[j] << sh));
        const WT Xv = Eq | w.Mv;
        const WT Xh = (((Eq & w.Pv) + w.Pv) ^ w.Pv) | Eq;
        WT Ph = w.Mv | ~(Xh | w.Pv);
        WT Mh = w.Pv & Xh;
        // << 1 as an add whose carry-out is the bit of the barcode's last row (the patterns are top-aligned):
        // v_add_co + v_addc instead of two shifts and a three-operand add (shifts and v_add3 issue at 2/3 rate)
        WT cp, cm;
        if constexpr (WL == 2) {
            cp = Ph >> 127;
            cm = Mh >> 127;
            Ph = Ph + Ph;
            Mh = Mh + Mh;
        } else if constexpr (WL == 1) {
            Ph = __builtin_addcll(Ph, Ph, 0ull, &cp);
            Mh = __builtin_addcll(Mh, Mh, 0ull, &cm);
        } else {
            Ph = __builtin_addc(Ph, Ph, 0u, &cp);
            Mh = __builtin_addc(Mh, Mh, 0u, &cm);
        }
        w.score += (int)cp;
        w.score -= (int)cm;
        w.Pv = Mh | ~(Xv | Ph);
        w.Mv = Ph & Xv;
        w.best = w.score < w.best ? w.score : w.best;
    };
    // Split mode of the seeded variants: the sweeps themselves record the first / last column whose unit
    // distance is <= kb instead of a second, sparsely populated pass over the survivors.
    // dense window table (plain-sweep kernels, few barcodes, many genuine candidates per read): every sweep is tracked
    // and writes its pair's window to wins_out[pass][read][barcode] (lo + 1024 | hi << 16) — no per-read entry cap
    const bool dense_w = !SEED && a.dense_w && a.split && a.wins_out[0] != nullptr;
    const bool track = (SEED || dense_w) && a.split && a.wins_out[0] != nullptr;
    auto step_tracked = [&](Sweep &w, const int j, const int sh) __attribute__((always_inline)) {
        step(w, j, sh);
        const bool in = w.score <= w.kbv;
        w.e_lo = (in && w.e_hi < 0) ? j : w.e_lo;
        w.e_hi = in ? j : w.e_hi;
    };
    auto finish = [&](const Sweep &w) __attribute__((always_inline)) {
        if (dense_w && w.ncol > 0 && w.e_hi >= 0) {
            const int jf_abs = (int)(w.c - (codes + roff[w.r])) + 1;  // 1-based column of sweep column 0
            const int mm = popw((w.p ? pv1 : pv0)[w.b]);
            const int lo = jf_abs + w.e_lo - (sg ? (a.short_lb[w.p] ? mm + w.kbv : 2 * (mm + w.kbv) + 1) : mm - 1);
            (w.p ? a.wins_out[1] : a.wins_out[0])[(long long)rids[w.r] * (w.p ? B1 : B0) + w.b] =
                (uint32_t)(lo + 1024) | ((uint32_t)(jf_abs + w.e_hi) << 16);
        } else if (track && w.ncol > 0 && w.e_hi >= 0) {
            const int kk = __hip_atomic_fetch_add(&wcl[w.p * R + w.r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (kk < BDX_WCAP) {
                const int jf_abs = (int)(w.c - (codes + roff[w.r])) + 1;  // 1-based column of sweep column 0
                const int mm = popw((w.p ? pv1 : pv0)[w.b]);
                uint32_t *dst = (w.p ? a.wins_out[1] : a.wins_out[0]) + ((long long)rids[w.r] * BDX_WCAP + kk) * 3;
                dst[0] = (uint32_t)w.b;
                // :semiglobal: first column of the restricted run; :hamming: first START position
                dst[1] = (uint32_t)(jf_abs + w.e_lo - (sg ? (a.short_lb[w.p] ? mm + w.kbv : 2 * (mm + w.kbv) + 1) : mm - 1));
                dst[2] = (uint32_t)(jf_abs + w.e_hi);
            }
        }
        if (w.ncol > 0 && w.best <= (w.p ? kb1 : kb0)[w.b]) {
            const int cw = w.p ? cw1 : cw0;
            LDS uint32_t *cnd = cand + (w.p ? R * cw0 : 0);
            __hip_atomic_fetch_or(&cnd[w.r * cw + (w.b >> 5)], 1u << (w.b & 31), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
            // (split mode reuses scnt / slots as its window counters / re-sweep queue: a dual config with ONE pass in the
            // known-score class — e.g. trim_side 5 + no trim_side2 — must not push replay slots there, or the window
            // counts handed to the exact kernel cover entries nobody wrote)
#ifdef BDX_REVERT_5CB01F9  // evidence builds only (tools/red_green_53109.sh): the defect fixed in 5cb01f9, back in
            if (full[w.p * R + w.r]) {
#else
            if (!a.split && full[w.p * R + w.r]) {
#endif
                const int k = __hip_atomic_fetch_add(&scnt[w.p * R + w.r], 1, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                if (dense)
                    dtab[(w.p ? R * B0 : 0) + w.r * (w.p ? B1 : B0) + w.b] = (unsigned char)w.best;
                else if (k < SC)
                    slots[(w.p * R + w.r) * SC + k] = ((uint32_t)w.b << 8) | (uint32_t)w.best;
            }
        }
    };
    auto sweep2 = [&](Sweep &A, Sweep &Bw) __attribute__((always_inline)) {
        // both chains use the same symbol shift only when they belong to the same pass;
        // the shifts are per chain (uniform in the non-seeded path, per lane otherwise)
        const int shA = A.p ? bsh1 : bsh0, shB = Bw.p ? bsh1 : bsh0;
        const int common = A.ncol < Bw.ncol ? A.ncol : Bw.ncol;
        int j = 0;
        if (track) {  // workgroup-uniform
#pragma unroll 4
            for (; j < common; ++j) {
                step_tracked(A, j, shA);
                step_tracked(Bw, j, shB);
            }
#pragma unroll 4
            for (int ja = j; ja < A.ncol; ++ja) step_tracked(A, ja, shA);
#pragma unroll 4
            for (int jb = j; jb < Bw.ncol; ++jb) step_tracked(Bw, jb, shB);
        } else {
#pragma unroll 4
            for (; j < common; ++j) {
                step(A, j, shA);
                step(Bw, j, shB);
            }
#pragma unroll 4
            for (int ja = j; ja < A.ncol; ++ja) step(A, ja, shA);
#pragma unroll 4
            for (int jb = j; jb < Bw.ncol; ++jb) step(Bw, jb, shB);
        }
        finish(A);
        finish(Bw);
    };

    if (staged) {
        // ---- transcode bytes -> symbol codes (4 per lane per step); the seeded variant also packs the
        // same four symbols to 2 bits each: one byte of the flat 2-bit image of the whole staging area
        // (packed byte k holds staged bytes 4k .. 4k+3, whatever read they belong to) ----
        const int nvec4 = BDX_DBG(32) ? 0 : (int)((need + 3) >> 2);
        for (int k = tid; k < nvec4; k += BS) {
            const uint32_t w = ((LDS uint32_t *)rstage)[k];
            const uint32_t c = (uint32_t)lut[w & 255] | ((uint32_t)lut[(w >> 8) & 255] << 8) |
                               ((uint32_t)lut[(w >> 16) & 255] << 16) | ((uint32_t)lut[w >> 24] << 24);
            ((LDS uint32_t *)codes)[k] = c;
            if (SEED) spk[k] = (unsigned char)((c & 3u) | ((c >> 6) & 0xCu) | ((c >> 12) & 0x30u) | ((c >> 18) & 0xC0u));
        }
        __syncthreads();

        if (!SEED) {
            for (int p = 0; p < npass; ++p) {
                const int B = p ? B1 : B0;
                int nrp = nr;
                if (p == 1) {
                    // the second pass only runs on reads whose first pass matched (classification.jl:884-891): a read
                    // without any first-pass candidate is :unknown whatever its second pass holds — not swept
                    __syncthreads();
                    if (tid == 0) sqn[6] = 0;
                    __syncthreads();
                    for (int t = tid; t < nr; t += BS) {
                        uint32_t any = 0;
                        for (int w = 0; w < cw0; ++w) any |= cand[t * cw0 + w];
                        if (any) act[__hip_atomic_fetch_add(&sqn[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = (unsigned char)t;
                    }
                    __syncthreads();
                    nrp = sqn[6];
                }
                const int total = BDX_DBG(2) ? 0 : nrp * B;
                for (int pair = tid; pair < total; pair += 2 * BS) {
                    Sweep A, Bw;
                    const int pb = pair + BS;
                    const int iA = pair / B, iB = pb < total ? pb / B : 0;
                    const int rA = p ? (int)act[iA] : iA, rB = p ? (int)act[iB] : iB;
                    setup(true, p, rA, pair - iA * B, A);
                    setup(pb < total, p, rB, pb - iB * B, Bw);
                    sweep2(A, Bw);
                }
            }
        } else {
            // ---- stage 0: q-gram seeds (pigeonhole) decide which pairs are swept, and WHERE ----
            // A recordable alignment of barcode b has <= kb[b] operations, so one of its kb+1 pieces
            // sits unchanged in the read; with the piece starting at barcode offset ps and found at
            // read position pos, the alignment starts within kb of  diag = pos - ps  and ends no
            // later than diag + m + kb.  Only that column window is swept (all windows of one
            // (read, barcode) pair are merged into one).  The windowed minimum equals the
            // whole-window minimum whenever the latter is <= kb — the only case anything
            // downstream looks at.
            const int q = a.seed_q;
            // DIAG walks the tile in sub-batches of SB reads (index, pair test, sweeps); HASH: one pass over the tile
            const int nbatch = DIAG ? (nr + SB - 1) / SB : 1;
            for (int bi = 0; bi < nbatch; ++bi) {
            const int rb0 = DIAG ? bi * SB : 0;
            const int rbn = DIAG ? (nr - rb0 < SB ? nr - rb0 : SB) : nr;  // reads of this sub-batch
            if constexpr (HASH) {
            // scan: lane = (read, group of 16 consecutive bases of the flat 2-bit image); the group's 16
            // start positions share one 64-bit window (16 + 7 bases), every key is probed in the bitmap.  The
            // hit counts of a wave are prefix-summed with four ballots (counts are <= 16), one lane
            // reserves the wave's range of the hit queue, every lane then writes its own hits.
            {
                const int G = a.seed_groups;  // 16-base groups per read (uniform upper bound)
                const int lane = tid & 63;
                const int dr = BS / G, dg = BS - dr * G;
                int r = tid / G, g = tid - r * G;
                const int total_items = BDX_DBG(8) ? 0 : nr * G;
                const int bml = a.seed_bm_log2;
                const uint32_t bmmask = (1u << bml) - 1u;
                const bool bm_direct = bml >= 2 * q;
                const LDS uint32_t *spk32 = (const LDS uint32_t *)spk;
                // the bitmap starts at LDS address 0 (first region of the carve-up, the kernel has no static
                // LDS): a byte's address is its index, no base register and no add per probe
                const auto probe = [](const uint32_t hb) __attribute__((always_inline)) {
                    return (uint32_t) * (const LDS unsigned char *)(uintptr_t)(hb >> 3);
                };
                for (int idx = tid; idx < total_items; idx += BS, r += dr, g += dg, r += (g >= G), g -= (g >= G) ? G : 0) {
                    const int rw = srw[r];
                    const uint32_t lh = slh[r];
                    const int F = (rw >> 4) + g;  // flat group: staged bytes 16F .. 16F+15
                    const int p0 = 16 * F - rw;   // position of the group's first base relative to the first staged base
                    int i0 = (int)(lh & 0xFFFFu) - p0, i1 = (int)(lh >> 16) - 1 - p0;  // valid starts: i0 <= i <= i1
                    i0 = i0 < 0 ? 0 : i0;
                    i1 = i1 > 15 ? 15 : i1;
                    const uint32_t w0 = spk32[F], w1 = spk32[F + 1];  // bases 0..15 and 16..31 of the group's window
                    uint32_t hits = 0;
                    if (bm_direct) {  // workgroup-uniform: the bitmap spans the key space
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const uint32_t hb = (i <= 8 ? __builtin_amdgcn_ubfe(w0, 2 * i, 2 * q)  /* q <= 8: the key lies inside w0 */
                                                             : __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, 2 * i), 0, 2 * q));
                            hits |= __builtin_amdgcn_ubfe(probe(hb), hb & 7u, 1) << i;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const uint32_t key = (i <= 8 ? __builtin_amdgcn_ubfe(w0, 2 * i, 2 * q)  /* q <= 8: the key lies inside w0 */
                                                             : __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, 2 * i), 0, 2 * q));
                            const uint32_t hb = (key ^ (key >> bml)) & bmmask;
                            hits |= __builtin_amdgcn_ubfe(probe(hb), hb & 7u, 1) << i;
                        }
                    }
                    hits = i0 <= i1 ? (hits & ((2u << i1) - (1u << i0))) : 0u;
                    // exclusive prefix sum of the per-lane hit counts over the wave, bit plane by bit plane
                    const uint32_t cnt = (uint32_t)__builtin_popcount(hits);
                    int pre = 0, tot = 0;
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const unsigned long long m = __builtin_amdgcn_ballot_w64((cnt >> k) & 1u);
                        pre += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << k;
                        tot += (int)__builtin_popcountll(m) << k;
                    }
                    if (tot) {  // wave-uniform
                        int basek = 0;
                        if (lane == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                            basek = __hip_atomic_fetch_add(&sqn[0], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        basek = __builtin_amdgcn_readfirstlane(basek);
                        int k = basek + pre;
                        while (hits) {
                            const int i = __builtin_ctz(hits);
                            hits &= hits - 1u;
                            if (k < SQCAP) {
                                shq[k] = ((uint32_t)(p0 + i) << 16) | __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, 2 * i), 0, 2 * q);
                                shr[k] = (unsigned char)r;
                            } else {
                                sall[r] = 1, sqn[5] = 1;  // hit queue full: sweep every barcode of this read instead
                            }
                            ++k;
                        }
                    }
                }
            }
            __syncthreads();
            // resolve: one lane per hit probes the hash table; every (pass, barcode) it finds is merged
            // into the read's small record table (CAS on the id, atomic min/max on the window)
            {
                const int nh = BDX_DBG(4) ? 0 : (sqn[0] < SQCAP ? sqn[0] : SQCAP);
                const uint32_t hmask = (1u << a.seed_hash_log2) - 1u;
                for (int k = ltid; k < nh; k += BS) {
                    const uint32_t h = shq[k];
                    const int t = shr[k];
                    const int prel = (int)(h >> 16);
                    const uint32_t key = h & 0xFFFFu;
                    const int wl_r = wlen[t];
                    uint32_t slot = (key * 0x9E3779B1u) >> (32 - a.seed_hash_log2);
                    for (;;) {
                        const uint32_t e = a.seed_hash_in_lds ? shash[slot] : a.seed_hash[slot];
                        if (e == 0u) break;
                        if ((e >> 16) == key) {
                            const int p = (int)((e >> 15) & 1u);
                            const int b = (int)(e & 0x7FFFu) - 1;
                            const int kk = (p ? kb1 : kb0)[b];
                            const int mm = popw((p ? pv1 : pv0)[b]);
                            const int diag = prel - (int)(a.seed_hash_in_lds ? shps[slot] : a.seed_hash_ps[slot]);
                            int lo = diag - kk - 1, hi = diag + mm + kk + 1;  // [lo, hi) relative to the staged base
                            if (lo < 0) lo = 0;
                            if (hi > wl_r) hi = wl_r;
                            const uint32_t pb = e & 0xFFFFu;  // pass << 15 | barcode + 1  (never 0)
                            int rs = (int)(pb & (RCAP - 1));
                            bool placed = false;
                            for (int tries = 0; tries < RCAP && !placed; ++tries) {
                                LDS uint32_t *id = srid + t * RCAP + rs;
                                uint32_t old = *id;
                                if (old == 0u) {
                                    uint32_t expect = 0u;
                                    __hip_atomic_compare_exchange_strong(id, &expect, pb, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                                    old = expect == 0u ? pb : expect;
                                }
                                if (old == pb) {
                                    __hip_atomic_fetch_min(&srlo[t * RCAP + rs], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_max(&srhi[t * RCAP + rs], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    placed = true;
                                }
                                rs = (rs + 1) & (RCAP - 1);
                            }
                            if (!placed) sall[t] = 1, sqn[5] = 1;  // more than RCAP distinct barcodes seeded in this read
                        }
                        slot = (slot + 1) & hmask;
                    }
                }
            }
            __syncthreads();
            // emit: every occupied record becomes one sweep
            for (int idx0 = 0; idx0 < nr * RCAP; idx0 += BS) {  // uniform trip count: wave-aggregated append
                const int idx = idx0 + tid;
                const uint32_t pb = idx < nr * RCAP ? srid[idx] : 0u;
                const bool has = pb != 0u;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(has);
                if (mk) {
                    const int lane = tid & 63;
                    const int leader = __builtin_ctzll(mk);
                    int basek = 0;
                    if (lane == leader)
                        basek = __hip_atomic_fetch_add(&sqn[1], (int)__builtin_popcountll(mk), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    basek = __shfl(basek, leader, 64);
                    if (has) {
                        const int t = idx >> RCAP_LOG2;
                        const int kq = basek + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
                        if (kq < PQCAP) {
                            spq[kq] = ((uint32_t)t << 16) | pb;
                            spw[kq] = ((uint32_t)srlo[idx] << 16) | (uint32_t)srhi[idx];
                        } else {
                            sall[t] = 1, sqn[5] = 1;
                        }
                    }
                }
            }
            } else {
            // ---- two-intact-pieces ("diagonal") variant (see build_diag_tables): budgets too large for a
            // selective single piece.
            //  1) inverted index of the read's 4-mers, every occurrence smeared over h = ceil(kmax / 2)
            //     positions to either side (index bit p + h + s, s = -h..h, for a 4-mer at position p);
            //  2) lane = (read, barcode): the bits of each of the barcode's kb+2 piece keys are shifted onto
            //     diagonals (bit 32 + h + position - piece offset); two DIFFERENT pieces on diagonals at most
            //     kb <= 2h apart have smeared bits in common, so a pair is a candidate iff the AND of one piece's
            //     bits with the union of the earlier pieces' bits is non-zero (exact for even kb = 2h, a
            //     superset otherwise): two loads, one funnel shift and two logic ops per piece and word;
            //  3) a common bit b is within h of both diagonals and the alignment starts within kb of either:
            //     candidates are swept over [b_min - h - kb - 1, b_max + h + m + kb + 1).
            const int hsm = (a.diag_kmax + 1) >> 1;
            {
                if (bi > 0) {  // the previous sub-batch is done with the index and the sweep queue
                    __syncthreads();
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    for (int i = tid; i < SB * 256 * NW / 4; i += BS) ((LDS u32x4 *)occ)[i] = z;
                    if (tid == 0) sqn[1] = 0;
                    __syncthreads();
                }
                const int G = a.seed_groups;
                const int dr = BS / G, dg = BS - dr * G;
                int rl = tid / G, g = tid - rl * G;  // rl: read within the sub-batch
                const int total_items = BDX_DBG(8) ? 0 : rbn * G;
                const LDS uint32_t *spk32 = (const LDS uint32_t *)spk;
                for (int idx = tid; idx < total_items; idx += BS, rl += dr, g += dg, rl += (g >= G), g -= (g >= G) ? G : 0) {
                    const int r = rb0 + rl;
                    const int rw = srw[r];
                    const uint32_t lh = slh[r];
                    const int F = (rw >> 4) + g;
                    const int p0 = 16 * F - rw;
                    int i0 = (int)(lh & 0xFFFFu) - p0, i1 = (int)(lh >> 16) - 1 - p0;
                    i0 = i0 < 0 ? 0 : i0;
                    i1 = i1 > 15 ? 15 : i1;
                    if (sall[r] || i0 > i1) continue;
                    const uint32_t w0 = spk32[F], w1 = spk32[F + 1];
                    LDS uint32_t *oc = occ + (size_t)rl * 256 * NW;
                    for (int i = i0; i <= i1; ++i) {
                        const uint32_t key = __builtin_amdgcn_alignbit(w1, w0, 2 * i) & 255u;
                        const int lo = p0 + i, hi = lo + 2 * hsm;  // bits [pos, pos + 2h]: <= 32 NW - 1 (reads <= 32 NW - 8 bases)
                        const int wlo_ = lo >> 5, whi_ = hi >> 5;
                        const uint32_t mlo = 0xFFFFFFFFu << (lo & 31), mhi = 0xFFFFFFFFu >> (31 - (hi & 31));
                        if (wlo_ == whi_) {
                            __hip_atomic_fetch_or(&oc[key * NW + wlo_], mlo & mhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            __hip_atomic_fetch_or(&oc[key * NW + wlo_], mlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&oc[key * NW + whi_], mhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            __syncthreads();
            for (int p = 0; p < npass; ++p) {
                const int B = p ? B1 : B0;
                const LDS uint32_t *dm = p ? dm1 : dm0;
                const LDS uint32_t *dk = p ? dk1 : dk0;
                const int total = BDX_DBG(4) ? 0 : rbn * B;
                for (int pair0 = 0; pair0 < total; pair0 += BS) {
                    const int pair = pair0 + tid;
                    const bool in = pair < total;
                    const int rl = in ? pair / B : 0, b = in ? pair - rl * B : 0;
                    const int r = rb0 + rl;
                    const uint32_t meta = in ? dm[b] : 0u;
                    uint32_t Cm[NW + 1];
                    bool cand_pair = false;
                    int kk = 0, mm = 0;
                    if (meta != 0u && !sall[r]) {
                        const int P = (int)(meta & 255u), L = (int)(meta >> 8);
                        const uint32_t k0 = dk[2 * b], k1 = dk[2 * b + 1];
                        kk = (p ? kb1 : kb0)[b];
                        mm = popw((p ? pv1 : pv0)[b]);
                        const LDS uint32_t *oc = occ + (size_t)rl * 256 * NW;
                        uint32_t SU[NW + 1];
#pragma unroll
                        for (int w = 0; w <= NW; ++w) SU[w] = Cm[w] = 0u;
                        for (int t = 0; t < P; ++t) {
                            const uint32_t key = (t < 4 ? k0 >> (8 * t) : k1 >> (8 * (t - 4))) & 255u;
                            const int o = t * L;  // piece offset, <= 28
                            uint32_t in_[NW];
#pragma unroll
                            for (int w = 0; w < NW; ++w) in_[w] = oc[key * NW + w];
                            // S: bit 32 + index bit - o  (({in[w], in[w-1]} >> o) as one 64-bit funnel per word)
#pragma unroll
                            for (int w = 0; w <= NW; ++w) {
                                const uint32_t S = __builtin_amdgcn_alignbit(w < NW ? in_[w] : 0u, w > 0 ? in_[w - 1] : 0u, o);
                                Cm[w] |= SU[w] & S;
                                SU[w] |= S;
                            }
                        }
                        uint32_t any = 0;
#pragma unroll
                        for (int w = 0; w <= NW; ++w) any |= Cm[w];
                        cand_pair = any != 0u;
                    }
                    if (cand_pair) {
                        // one sweep per CLUSTER of candidate diagonals (bits closer than 2 kk + 2 share a window):
                        // far-apart candidates of one pair must not be merged into one long window — the
                        // longest window of a wave sets that wave's sweep time
                        const int wl_r = wlen[r];
                        const int gap = 2 * kk + 2;
                        int c_lo = -1, c_hi = -1;
                        auto flush = [&]() __attribute__((always_inline)) {
                            // common bits carry the index bias h and lie within h of both diagonals
                            int lo = c_lo - 32 - 2 * hsm - kk - 1, hi = c_hi - 32 + mm + kk + 1;  // [lo, hi) relative to the staged base
                            if (lo < 0) lo = 0;
                            if (hi > wl_r) hi = wl_r;
                            const int kq = __hip_atomic_fetch_add(&sqn[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (kq < PQCAP) {
                                spq[kq] = ((uint32_t)r << 16) | ((uint32_t)p << 15) | (uint32_t)(b + 1);
                                spw[kq] = ((uint32_t)lo << 16) | (uint32_t)hi;
                            } else {
                                sall[r] = 1, sqn[5] = 1;
                            }
                        };
                        // first / last common bit; nearly always they form ONE cluster (a single region of a few
                        // consecutive bits), which needs no walk over the bits
                        int f_i = 0, l_i = 0;
                        bool got = false;
#pragma unroll
                        for (int w = 0; w <= NW; ++w) {
                            if (!got && Cm[w]) {
                                f_i = 32 * w + __builtin_ctz(Cm[w]);
                                got = true;
                            }
                            if (Cm[w]) l_i = 32 * w + 31 - __builtin_clz(Cm[w]);
                        }
                        if (l_i - f_i <= gap) {
                            c_lo = f_i;
                            c_hi = l_i;
                        } else {
#pragma unroll
                            for (int w = 0; w <= NW; ++w) {
                                uint32_t bits = Cm[w];
                                while (bits) {
                                    const int g = 32 * w + __builtin_ctz(bits);
                                    bits &= bits - 1u;
                                    if (c_lo >= 0 && g - c_hi > gap) {
                                        flush();
                                        c_lo = g;
                                    }
                                    if (c_lo < 0) c_lo = g;
                                    c_hi = g;
                                }
                            }
                        }
                        flush();
                    }
                }
            }
            }
            // barcodes that are swept unconditionally (wildcards, pieces shorter than 5): whole window
            for (int p = 0; p < npass; ++p) {
                const int na = a.seed_n_always[p];
                for (int idx = tid; idx < rbn * na; idx += BS) {
                    const int rl = idx / na;
                    const int r = rb0 + rl;
                    const int b = a.seed_always[p][idx - rl * na];
                    const int kq = __hip_atomic_fetch_add(&sqn[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (kq < PQCAP) {
                        spq[kq] = ((uint32_t)r << 16) | ((uint32_t)p << 15) | (uint32_t)(b + 1);
                        spw[kq] = 0x0000FFFFu;
                    } else {
                        sall[r] = 1, sqn[5] = 1;
                    }
                }
            }
            __syncthreads();
            {
                const int np = BDX_DBG(2) ? 0 : (sqn[1] < PQCAP ? sqn[1] : PQCAP);
                for (int k = ltid; k < np; k += 2 * BS) {
                    Sweep A, Bw;
                    const uint32_t ea = spq[k], wa = spw[k];
                    const bool hb = k + BS < np;
                    const uint32_t eb = hb ? spq[k + BS] : 0u, wb = hb ? spw[k + BS] : 0u;
                    const int ra = (int)(ea >> 16), rb = (int)(eb >> 16);
                    setup(!sall[ra], (int)((ea >> 15) & 1u), ra, (int)(ea & 0x7FFFu) - 1, A, (int)(wa >> 16), (int)(wa & 0xFFFFu));
                    setup(hb && !sall[rb], (int)((eb >> 15) & 1u), rb, (int)(eb & 0x7FFFu) - 1, Bw, (int)(wb >> 16), (int)(wb & 0xFFFFu));
                    if (BDX_DBG(128)) {  // statistics for tuning (results stay correct): sweeps and swept columns
                        atomicAdd(a.exc_count + 1, 1u + (hb ? 1u : 0u));
                        atomicAdd(a.exc_count + 2, (unsigned)(A.ncol + Bw.ncol));
                    }
                    sweep2(A, Bw);
                }
            }
            }  // sub-batches
            if (DIAG) __syncthreads();  // orders the last sub-batch's flag writes before the read below
            const int any_sall = sqn[5];  // does any read of this tile need the whole-read fallback below?
            if (BDX_DBG(128) && tid == 0 && any_sall) atomicAdd(a.exc_count + 3, 1u);
            {
                // reads whose lists overflowed: every barcode over the whole window, exactly once
                for (int p = 0; p < npass; ++p) {
                    const int B = p ? B1 : B0;
                    const int total = (BDX_DBG(2) || !any_sall) ? 0 : nr * B;
                    for (int pair = tid; pair < total; pair += 2 * BS) {
                        Sweep A, Bw;
                        const int pb = pair + BS;
                        const int rA = pair / B, rB = pb / B;
                        const bool va = sall[rA] != 0;
                        const bool vb = pb < total && sall[rB < nr ? rB : 0] != 0;
                        if (!__builtin_amdgcn_ballot_w64(va || vb)) continue;
                        setup(va, p, rA, pair - rA * B, A, 0, 0xFFFF);
                        setup(vb, p, rB, pb - rB * B, Bw, 0, 0xFFFF);
                        sweep2(A, Bw);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (a.split) {
        // split mode: hand the candidate masks to the full-width exact kernel (bdx_generic_kernel,
        // 256 reads per workgroup).  Tiles that could not be staged pass every barcode.
        for (int p = 0; p < npass; ++p) {
            const int cw = p ? cw1 : cw0;
            const LDS uint32_t *cnd = cand + (p ? R * cw0 : 0);
            uint32_t *dst = a.cand_out[p];
            for (int i = tid; i < nr * cw; i += BS) {
                const int r = i / cw;
                dst[(long long)rids[r] * cw + (i - r * cw)] = staged ? cnd[i] : 0xFFFFFFFFu;
            }
        }
        if (a.wins_out[0]) {
            // Column windows for the exact kernel: every surviving candidate is swept
            // once more over its whole pass window while tracking the first / last column whose unit
            // distance is <= kb; recordable alignments end inside [e_lo, e_hi] and start at most m + kb
            // earlier, and another m + kb + 1 columns of warm-up make every cell <= allowed_error
            // independent of the fresh start -> the exact DP runs over e_lo - 2(m+kb) - 1 .. e_hi only.
            constexpr int WQCAP = 4 * R;
            if (staged && !track) {
                for (int idx = tid; idx < nr * (cw0 + cw1); idx += BS) {
                    const int p = idx >= nr * cw0 ? 1 : 0;
                    const int loc = p ? idx - nr * cw0 : idx;
                    const int cw = p ? cw1 : cw0;
                    const int r = loc / cw, w = loc - r * cw;
                    uint32_t bits = cand[(p ? R * cw0 : 0) + r * cw + w];
                    while (bits) {
                        const int b = w * 32 + __builtin_ctz(bits);
                        bits &= bits - 1u;
                        const int k = __hip_atomic_fetch_add(&sqn[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (k < WQCAP) wq[k] = ((uint32_t)r << 16) | ((uint32_t)p << 15) | (uint32_t)b;
                    }
                }
            }
            __syncthreads();
            const int nq = sqn[4];
            const bool usable = staged && nq <= WQCAP;  // overflow / unstaged tile: no windows, whole-window DP
            if (usable && !track) {
                for (int k = tid; k < nq; k += BS) {
                    const uint32_t e = wq[k];
                    const int r = (int)(e >> 16), p = (int)((e >> 15) & 1u), b = (int)(e & 0x7FFFu);
                    Sweep A;
                    setup(true, p, r, b, A);
                    const int kbv = (p ? kb1 : kb0)[b];
                    const int mm = A.score;  // barcode length before the first column
                    const int sh = p ? bsh1 : bsh0;
                    const int jf_abs = (int)(A.c - (codes + roff[r])) + 1;
                    int e_lo = 0, e_hi = -1;
                    for (int j = 0; j < A.ncol; ++j) {
                        step(A, j, sh);
                        if (A.score <= kbv) {
                            if (e_hi < 0) e_lo = jf_abs + j;
                            e_hi = jf_abs + j;
                        }
                    }
                    if (e_hi >= 0) {
                        const int kk = __hip_atomic_fetch_add(&wcl[p * R + r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (kk < BDX_WCAP) {
                            uint32_t *dst = (p ? a.wins_out[1] : a.wins_out[0]) + ((long long)rids[r] * BDX_WCAP + kk) * 3;
                            dst[0] = (uint32_t)b;
                            dst[1] = (uint32_t)(e_lo - (sg ? (a.short_lb[p] ? mm + kbv : 2 * (mm + kbv) + 1) : mm - 1));
                            dst[2] = (uint32_t)e_hi;
                        }
                    } else {
                        wcl[p * R + r] = 1000;  // cannot happen for a candidate; if it did: no restriction
                    }
                }
            }
            __syncthreads();
            for (int p = 0; p < npass; ++p)
                for (int t = tid; t < nr; t += BS) {
                    const int c = wcl[p * R + t];
                    a.wcnt_out[p][rids[t]] = (unsigned char)(dense_w ? ((staged && rlen[t] <= 60000) ? 254 : 255 /* columns beyond 16 bits (a wrong length hint): whole window */) : ((usable && c <= BDX_WCAP) ? c : 255));
                }
        }
        continue;
    }

    // ---- stage 2 (configs whose passes all sit in the known-score class): the verdict of a read
    // with at most four survivors per pass is a replay of the reducer on their unit distances
    // (DESIGN.md §3.1).  The other reads — more survivors, a range that binds, or a tile that did not
    // fit the staging area — are handed to the exact kernel through a list in HBM together with
    // their candidate masks; this kernel holds no DP state at all. ----
    const bool active = ltid < nr;
    const long long ridx = active ? (long long)rids[ltid] : 0;
    Verdict v{0, 0, -1, -1};
    PassOut p1{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()}, p2{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    bool done = false;
    if (active && !BDX_DBG(1)) {
        const int cnt0 = scnt[0 * R + ltid], cnt1 = scnt[1 * R + ltid];
        bool known = staged && full[0 * R + ltid] && (dense || cnt0 <= SC);
        if (npass > 1) known = known && full[1 * R + ltid] && (dense || cnt1 <= SC);
        if (known) {
            const LDS uint32_t *e0 = slots + (0 * R + ltid) * SC;
            const LDS uint32_t *e1 = slots + (1 * R + ltid) * SC;
            const bool many = cnt0 > 4 || cnt1 > 4;  // (then the replay scans its entries in LDS)
            const KnownPass kn0{true, e0[0], e0[1], e0[2], e0[3], cnt0, many ? e0 : nullptr,
                                dense ? dtab + ltid * B0 : nullptr, cand + ltid * cw0, cw0};
            const KnownPass kn1{true, e1[0], e1[1], e1[2], e1[3], cnt1, many ? e1 : nullptr,
                                dense ? dtab + R * B0 + ltid * B1 : nullptr, cand + R * cw0 + ltid * cw1, cw1};
            const auto m0 = [&](const int b) { return popw(pv0[b]); };
            const auto m1 = [&](const int b) { return popw(pv1[b]); };
            classify_known(cfg, m0, m1, rlen[ltid], kn0, kn1, v, p1, p2);
            done = true;
            if (a.tier) {
                // Tier settle rule.  The replay saw every barcode b with d*(b) <= its CAPPED budget; a barcode it
                // did not see scores >= slo (tier_slo of its pass).  Both reducers (:632-713) return the smallest
                // score — first in file order among ties — and the with_delta one the second smallest as sub_min:
                //   * min_score < slo (strictly): no unseen barcode can win or tie -> bc, score, raw are final;
                //   * no_delta: delta = Inf, nothing else to know;
                //   * with_delta: sub_min is final if the replay's own sub_min <= slo (an unseen barcode cannot be
                //     smaller); else the true sub_min lies in [slo, replay's], so delta >= fl(slo - min): the pass is a
                //     match — not ambiguous — if that is >= min_delta (IEEE subtraction is monotone), which settles
                //     the verdict when nobody asked for the delta value itself.
                // Anything else (no barcode within the capped budgets at all, a possible tie, an undecided
                // ambiguity; also reads outside the known-score class or with more than four survivors) goes to
                // tier 0, which filters the read at the full budgets.
                const bool nd = cfg.min_delta == 0.0;
                const auto settled = [&](const PassOut &po, const int cnt, const double slo) {
                    if (cnt < 1 || !(po.score < slo)) return false;
                    if (nd) return true;
                    if (cnt >= 2 && po.sub <= slo) return true;
                    return a.out.pass_delta == nullptr && (slo - po.score) >= cfg.min_delta && po.status == 1;
                };
                bool ok = settled(p1, cnt0, a.tier_slo[0]);
                if (ok && npass > 1 && p1.status == 1) ok = settled(p2, cnt1, a.tier_slo[1]);
                done = ok;
            }
        } else if (!a.tier) {  // (tier 1 hands every read it cannot settle to tier 0, below)
            for (int p = 0; p < npass; ++p) {
                const int cw = p ? cw1 : cw0;
                const LDS uint32_t *cnd = cand + (p ? R * cw0 : 0) + ltid * cw;
                uint32_t *dst = (p ? a.cand_out[1] : a.cand_out[0]) + ridx * cw;
                for (int w = 0; w < cw; ++w) dst[w] = staged ? cnd[w] : 0xFFFFFFFFu;
            }
            a.exc_list[atomicAdd(a.exc_count, 1u)] = (uint32_t)ridx;
        }
    }
    if (a.tier) {
        // unsettled reads -> tier 0's list: one queue reservation per wave (a tile's stage 2 is one wave)
        const bool hand = active && !done && !BDX_DBG(1);
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(hand);
        if (mk) {
            const int lane = tid & 63;
            const int leader = __builtin_ctzll(mk);
            unsigned int basek = 0;
            if (lane == leader) basek = atomicAdd(a.tier_count, (unsigned int)__builtin_popcountll(mk));
            basek = (unsigned int)__shfl((int)basek, leader, 64);
            if (hand) a.tier_list[basek + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (uint32_t)ridx;
        }
    }
    if (done) {
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

    // ---- DemuxStats scalar counters (accumulated in LDS across this workgroup's tiles) ----
    if (a.counts && done) {
        int slot = -1;
        if (v.bc1 > 0) slot = 4 + (v.bc1 - 1) * cfg.counts_stride2 + (v.bc2 > 0 ? v.bc2 - 1 : 0);
        const int cls = v.bc1 > 0 ? 1 : (v.bc1 == 0 ? 2 : 3);
        // the four scalar counters (and as many per-barcode slots as the LDS histogram holds) are accumulated in LDS;
        // slots beyond it (very many barcodes) go to HBM directly — different addresses, little contention
        __hip_atomic_fetch_add(&hist[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&hist[cls], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (slot >= 0 && slot < a.hist_entries)
            __hip_atomic_fetch_add(&hist[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (slot >= 0)
            atomicAdd(&a.counts[slot], 1ULL);
    }
    }  // tile loop

    if (a.counts) {
        __syncthreads();
        for (int i = tid; i < a.hist_entries; i += BS) {
            const int h = hist[i];
            if (h) atomicAdd(&a.counts[i], (unsigned long long)h);
        }
    }
}

template <int BS, int R, bool SEED, bool DIAG = false, int NW = 5, int WL = 0>
hipError_t launch_one(const BitparArgs &a, size_t lds, long long n_reads, hipStream_t stream, long long grid_override, int n_cu) {
    // the attribute is per device: one flag per device of this process.  Contexts of several OS threads may
    // launch concurrently: setting the attribute twice is harmless, the flag itself must not be a data race.
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)bdx_bitpar_kernel<BS, R, SEED, DIAG, NW, WL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
    }
    // persistent grid: enough workgroups to fill every CU at the LDS-limited residency,
    // never more than there are tiles
    const long long tiles = (n_reads + R - 1) / R;
    long long per_cu = (long long)((160 * 1024) / (lds ? ((lds + 1279) / 1280) * 1280 : 1));  // 1280-byte LDS granules
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    long long blocks = (long long)(n_cu > 0 ? n_cu : 256) * per_cu;  // exactly the resident set; the tile queue balances it
    if (grid_override > 0) blocks = grid_override;
    if (blocks > tiles) blocks = tiles;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((bdx_bitpar_kernel<BS, R, SEED, DIAG, NW, WL>), dim3((unsigned)blocks), dim3(BS), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

size_t bdx_bitpar_lds_bytes(const BdxDevCfg &cfg, const BdxBitparPlan &bp, const BdxGenericPlan &gp,
                            const BdxSeedPlan *sp) {
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const int R = bp.reads_per_block;
    const int B0 = cfg.pass[0].n_barcodes, B1 = cfg.is_dual ? cfg.pass[1].n_barcodes : 0;
    const int cw0 = cfg.pass[0].cand_words, cw1 = cfg.is_dual ? cfg.pass[1].cand_words : 0;
    size_t o = 0;
    o += al((size_t)gp.hist_entries * 4) + al(256);
    const size_t wb = bp.word_bytes >= 8 ? (size_t)bp.word_bytes : 4;
    o += al((size_t)bp.ncodes * bp.bpad[0] * wb) + al(cfg.is_dual ? (size_t)bp.ncodes * bp.bpad[1] * wb : 0);
    o += al((size_t)B0 * wb) + al((size_t)B1 * wb) + al((size_t)B0 * 4) + al((size_t)B1 * 4);
    o += al((size_t)R * (cw0 + cw1) * 4) + al((size_t)(R + 1) * 4) + 4 * al((size_t)R * 4) + al((size_t)R * 16);
    const bool seeded = sp && sp->enabled;
    const int sc = (!seeded && bp.dense_d) ? 4 : (bp.slot_cap > 4 ? bp.slot_cap : 4);
    o += al((size_t)2 * R * sc * 4) + al((size_t)2 * R * 4) + al((size_t)2 * R);
    o += al((size_t)bp.stage_bytes + 16);
    if (sp && sp->enabled && sp->diag) {
        const int nw = bp.diag_nw > 0 ? bp.diag_nw : 5;
        const int sbmax = nw <= 5 ? BDX_DIAG_SB_NARROW : 4;
        const int SBh = R < sbmax ? R : sbmax;  // index sub-batch (see the kernel)
        o += al((size_t)(bp.stage_bytes >> 2) + 32) + al((size_t)2 * (bp.diag_qcap > 0 ? bp.diag_qcap : 64) * SBh * 4);
        o += al((size_t)R) + 2 * al((size_t)R * 4);
        o += al((size_t)SBh * 256 * nw * 4) + al((size_t)B0 * 4) + al((size_t)B1 * 4) + al((size_t)B0 * 8) + al((size_t)B1 * 8);
    } else if (sp && sp->enabled) {
        o += al((size_t)sp->bm_words * 4) + al((size_t)(bp.stage_bytes >> 2) + 32);
        if (sp->hash_in_lds) o += al((size_t)4 << sp->hash_log2) + al((size_t)1 << sp->hash_log2);
        const size_t sq = (size_t)2 * (sp->qmul >= 4 ? sp->qmul : 4) * R;  // hit-queue entries
        o += al(sq * 4) + al(sq) + 3 * al((size_t)R * sp->rcap * 4);
        o += al((size_t)R) + 2 * al((size_t)R * 4);
    }
    o += al(32);
    if (!(sp && sp->enabled) && cfg.is_dual) o += al((size_t)R);  // act[]
    if (!(sp && sp->enabled) && bp.dense_d) o += al((size_t)R * (B0 + B1));  // dtab[]
    return o;
}

hipError_t bdx_launch_bitpar(const BdxDevCfg &cfg, const BdxGenericPlan &gp, const BdxBitparPlan &bp,
                             const BdxSeedPlan &sp, const uint8_t *d_seq, const long long *d_off, long long n_reads, const BdxDevOut &out,
                             unsigned long long *d_counts, uint32_t *cand_out0, uint32_t *cand_out1,
                             hipStream_t stream, uint32_t *wins_out0, uint32_t *wins_out1, uint8_t *wcnt_out0,
                             uint8_t *wcnt_out1, int split, uint32_t *exc_list, unsigned int *exc_count,
                             const BdxTierArgs *tier) {
    if (n_reads <= 0) return hipSuccess;
    BitparArgs a;
    a.tier = 0;
    a.tier_slo[0] = a.tier_slo[1] = 0.0;
    a.tier_list = nullptr;
    a.tier_count = nullptr;
    a.in_list = nullptr;
    a.in_count = nullptr;
    if (tier) {
        a.tier = tier->tier1;
        a.tier_slo[0] = bp.tier_slo[0];
        a.tier_slo[1] = bp.tier_slo[1];
        a.tier_list = tier->out_list;
        a.tier_count = tier->out_count;
        a.in_list = tier->in_list;
        a.in_count = tier->in_count;
    }
    a.cfg = cfg;
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.out = out;
    a.counts = d_counts;
    a.stage_bytes = bp.stage_bytes;
    a.hist_entries = gp.hist_entries;
    a.lut = bp.d_lut;
    for (int k = 0; k < 2; ++k) {
        a.peq[k] = bp.d_peq[k];
        a.pvinit[k] = bp.d_pvinit[k];
        a.kb[k] = bp.d_kb[k];
        a.bpad[k] = bp.bpad[k];
        a.bshift[k] = 2;  // log2 of a peq row in bytes: bpad (a power of two) sweep words
        while ((1 << a.bshift[k]) < bp.bpad[k] * (bp.word_bytes >= 8 ? bp.word_bytes : 4)) a.bshift[k]++;
    }
    a.ncodes = bp.ncodes;
    a.dbg = bp.dbg;
    a.slot_bytes = bp.slot_bytes;
    a.slot_cap = bp.slot_cap > 4 ? bp.slot_cap : 4;
    a.dense_d = bp.dense_d;
    a.dense_w = bp.dense_w;
    a.short_lb[0] = bp.short_lb[0];
    a.short_lb[1] = bp.short_lb[1];
    a.ncode = bp.ncode_N;
    a.split = split;
    a.exc_list = exc_list;
    a.exc_count = exc_count;
    a.cand_out[0] = cand_out0;
    a.cand_out[1] = cand_out1;
    a.wins_out[0] = wins_out0;
    a.wins_out[1] = wins_out1;
    a.wcnt_out[0] = wcnt_out0;
    a.wcnt_out[1] = wcnt_out1;
    a.tile_counter = bp.d_tile_counter;
    a.known_ok[0] = bp.known_ok[0];
    a.known_ok[1] = bp.known_ok[1];
    a.seed_q = sp.q;
    a.seed_groups = (bp.seed_span + 15 + 15) / 16;  // 16-base groups of the flat image that can overlap one read
    a.seed_hash_log2 = sp.hash_log2;
    a.seed_bm_words = sp.bm_words;
    a.seed_bm_log2 = sp.bm_log2;
    a.seed_rcap = sp.rcap > 0 ? sp.rcap : 8;
    a.seed_qmul = sp.qmul >= 4 ? sp.qmul : 4;
    a.diag_kmax = sp.diag_kmax;
    a.diag_qcap = bp.diag_qcap > 0 ? bp.diag_qcap : 64;
    for (int k = 0; k < 2; ++k) {
        a.dmeta[k] = sp.d_dmeta[k];
        a.dkeys[k] = sp.d_dkeys[k];
    }
    a.seed_hash_in_lds = sp.hash_in_lds;
    a.seed_bitmap = sp.d_bitmap;
    a.seed_hash = sp.d_hash;
    a.seed_hash_ps = sp.d_hash_ps;
    for (int k = 0; k < 2; ++k) {
        a.seed_n_always[k] = sp.n_always[k];
        a.seed_always[k] = sp.d_always[k];
    }
    const size_t lds = bdx_bitpar_lds_bytes(cfg, bp, gp, &sp);
    const bool seed = sp.enabled != 0;
    if (seed && sp.diag) {
#define BDX_LAUNCH_D(RR) return bp.diag_nw > 5 ? launch_one<256, RR, true, true, 10>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu) : launch_one<256, RR, true, true, 5>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)
        switch (bp.reads_per_block) {
            case 32:
                BDX_LAUNCH_D(32);
            case 16:
                BDX_LAUNCH_D(16);
            case 8:
                BDX_LAUNCH_D(8);
            case 4:
                BDX_LAUNCH_D(4);
            default:
                return hipErrorInvalidValue;
        }
#undef BDX_LAUNCH_D
    }
#define BDX_LAUNCH_R(RR)                                                                                                  \
    return bp.word_bytes == 8 ? (seed ? launch_one<256, RR, true, false, 5, 1>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)    \
                                      : launch_one<256, RR, false, false, 5, 1>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)) \
                              : (seed ? launch_one<256, RR, true>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)                    \
                                      : launch_one<256, RR, false>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu))
#define BDX_LAUNCH_Q(RR)                                                                                                  \
    return seed ? launch_one<256, RR, true, false, 5, 2>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)                \
                : launch_one<256, RR, false, false, 5, 2>(a, lds, n_reads, stream, bp.grid_override, bp.n_cu)
    if (bp.word_bytes == 16) {  // barcodes of 65..128 nt: tiles of 64 / 32 / 16 reads (size_bitpar)
        switch (bp.reads_per_block) {
            case 64:
                BDX_LAUNCH_Q(64);
            case 32:
                BDX_LAUNCH_Q(32);
            case 16:
                BDX_LAUNCH_Q(16);
            default:
                return hipErrorInvalidValue;
        }
    }
#undef BDX_LAUNCH_Q
    switch (bp.reads_per_block) {
        case 256:
            BDX_LAUNCH_R(256);
        case 128:
            BDX_LAUNCH_R(128);
        case 64:
            BDX_LAUNCH_R(64);
        case 32:
            BDX_LAUNCH_R(32);
        case 16:
            BDX_LAUNCH_R(16);
        default:
            return hipErrorInvalidValue;
    }
#undef BDX_LAUNCH_R
}
