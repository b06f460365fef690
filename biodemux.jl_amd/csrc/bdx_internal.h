// bdx_internal.h — structures shared between the C-ABI translation unit (bdx_abi.cpp) and the
// gfx950 kernels (bdx_device.hip).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/biodemux_hip.h"

// Integer domain of the device DP.  The reference computes in Int64 with
// INF_INT = typemax(Int) ÷ 4 (classification.jl:7); every DP value is bounded by
// (max_m + 2) * max|cost| + allowed_error, so with the limits enforced in bdx_create
// (|cost| <= BDX_MAX_COST, max_m <= BDX_MAX_M, |allowed_error| < 2^27) int32 arithmetic
// is exact and BDX_INF32 + value never overflows.
#define BDX_INF32 0x3FFFFFFF
#define BDX_MAX_COST 32767
#define BDX_MAX_M 8192
#define BDX_MAX_RATE 1.0e4
#define BDX_WCAP 4       // column-window entries per read and pass handed from the filter to the exact kernel
#define BDX_REG_ROWS 32  // barcodes up to this length run the register-resident exact DP

struct BdxDevRange {
    long long start_offset;
    long long end_offset;
    int start_from_end;
    int end_from_end;
};

struct BdxDevPass {
    BdxDevRange ref_search, bc_start, bc_end;
    int trim_side;
    int n_barcodes;
    int explicit_window;
    int cand_words;  // ceil(n_barcodes / 32)
    long long win_first, win_last, win_max_start, win_min_end;
    const uint8_t *bc_bytes;    // device
    const uint32_t *bc_off;     // device, n_barcodes + 1
    const int32_t *bc_len_no_N; // device
};

struct BdxDevCfg {
    int algorithm;
    int is_dual;
    double max_error_rate;
    double min_delta;
    int match, mismatch, indel;
    int has_nindel, nindel;
    int need_traceback;
    int end_only_ok;   // per launch: the caller did not ask for pass_start (an end-only DP may serve trim_side 5)
    int max_m;
    int force_lds_dp;   // testing: use the LDS-resident DP even for short barcodes (env BDX_LDS_DP)
    int any_traceback;  // origin array needed (trim or summary in any pass)
    int counts_stride2; // max(1, B2 when dual)
    int n_counts;
    // "window upload" of the host entry point (long reads with short column windows): seq / off hold only each
    // read's window bytes; vlen[i] is the read's true length, vlo[i] the 0-based position its first uploaded byte
    // stands for.  Position p of read i lives at seq[off[i] + p - vlo[i]].  NULL: ordinary batches.
    const int32_t *vlen;
    const int32_t *vlo;
    // per launch of the exact kernel, split mode with column windows: the budget (unit operations) the fused kernel's
    // tracked sweeps used for every barcode of the pass and the lookback it subtracted from the first end column;
    // with both the exact kernel rebuilds [e_lo, e_hi] and runs the diagonal-band DP (sg_core_band).  -1: off.
    int band_kb[2];
    int band_lb[2];
    int dense_w;       // per launch: the column windows are a dense table wins[pass][read][barcode] (lo + 1024 | hi << 16), wcnt = 254
    int band_m;        // the common barcode length the band bodies run with (8, 10, 12, 16, 20, 24 or 32)
    int band_hcap;     // rolling band (sg_band_roll, barcodes beyond 32 rows in the clean class): diagonals a lane's LDS cells hold; 0: off
    unsigned int *dbg_rejected;  // device counter: hand-over windows the exact kernel refused as "not a window" (must stay 0)
    BdxDevPass pass[2];
};

struct BdxDevOut {
    int32_t *bc1, *bc2, *keep_start, *keep_end;
    int32_t *pass_start, *pass_end, *pass_raw, *pass_bc;
    double *pass_score, *pass_delta;
};

// DemuxStats histograms (classification.jl:827-865), accumulated by the exact kernel when the config asks for
// statistics (summary = true): per pass three int64 tables [rows][n_barcodes] — start position (row = start - 1
// + pos_bias: origins may lie before the read), length end - start + 1, and the integer numerator of
// the score (the host maps it to round(raw / norm, digits = 2), :835).  Row-major by KEY so that a longer read
// in a later batch only appends rows.
struct BdxDevStats {
    unsigned long long *pos[2], *len[2], *raw[2];
    long long rows;       // rows of pos (0: no statistics)
    int raw_rows;
    // len and raw live TRANSPOSED on the device, [barcode][key] with the key stride a multiple of 16 counters: nearly
    // every match of a barcode has the same length and one of two or three scores, so in [key][barcode] order the
    // whole batch would hammer the half dozen cache lines of those rows (measured: 1.0 of 3.2 ms per 2 M reads);
    // per barcode the hot keys share one line of their own.  bdx_get_stats hands them out as [key][barcode].
    int len_rows, len_stride, raw_stride;
    int pos_bias;         // = max barcode length
    unsigned int *overflow;  // set when a key does not fit (cannot happen with rows sized from the batch)
};

// Launch geometry chosen on the host for the generic (unfiltered / verify) kernel.
struct BdxGenericPlan {
    int threads;         // 64 / 128 / 256
    int reg_rows;        // 24 / 32: register-resident exact DP (no DP columns in LDS); 0: LDS columns
    int clean;           // register DP in its clean-class form (sg_core_clean): in-domain costs, start / end ranges "1:end"
    int uniform_m;       //   ... and every barcode has exactly reg_rows rows
    int uniform_len;     //   ... every barcode of the config has this length and band bodies exist for it (else 0)
    int band_roll;       // barcodes beyond 32 rows inside the clean class: the exact DP is the rolling diagonal band (sg_band_roll), dp_rows = its H + 1
    int same_len;        // every barcode of the config has the same length (the filter's first end column can be rebuilt from a hand-over row)
    int dp_rows;         // generic kernel: max_m + 1, or 1 in register mode
    int dp_rows_fused;   // fused kernel's in-kernel exact stage always keeps LDS columns: max_m + 1
    int stage_bytes;     // LDS bytes reserved for staged read bytes (0 = read from HBM/L2 directly)
    int bc_stage_bytes;  // LDS bytes for the staged barcode bytes of both passes (0 = not staged)
    int hist_entries;    // LDS histogram entries (0 = global atomics)
    size_t lds_bytes;
    int n_cu;            // compute units of the device (list-mode grid: 4 workgroups per unit)
};

// Bit-parallel (Myers) pre-filter: tables built on the host in bdx_abi.cpp, used by
// bdx_bitpar.hip.  enabled == 0 -> the config is outside the filter's domain.
struct BdxBitparPlan {
    int enabled;
    int reads_per_block;   // R: 256 / 128 / 64 / 32 / 16
    int stage_bytes;       // capacity of each staging area (raw bytes, symbol codes)
    int read_len_hint;     // the read length the geometry was planned for
    int r_cap;             // ... and the tile-size cap the batch size implied
    int diag_nw;           // diagonal variant: index words per key for this read length (5: <= 152 bases, 10: <= 312)
    int diag_qcap;         //   ... and sweep-queue entries to provide per read
    int read_len_hint_for_lds;  // same value, set before sizing (used for the seed work areas)
    int slot_bytes;        // > 0: window-slot staging (long reads with a short column window)
    int dense_w;           // per launch: split mode of the plain-sweep kernel hands the column windows over as a dense [read][barcode] table
    int dense_d;           // known-score class, plain-sweep kernels: byte table of every candidate's d (few barcodes, many genuine candidates)
    int slot_cap;          // known-score class: survivors per read and pass the replay takes (4 .. 32, from the expected number of genuine candidates)
    int seed_span;         // bases per read the seed scan covers (read length, or the window in slot mode)
    int ncode_N;           // symbol code of 'N' (255 if no barcode contains it)
    int ncodes;            // symbol codes incl. the trailing "other" code (<= 16)
    long long grid_override;  // > 0: forced persistent grid (tuning, BdxTuning::grid)
    int n_cu;              // compute units of the device: the persistent grid is n_cu x the LDS-limited residency
    int short_lb[2];       // per launch and pass: short lookback of the restricted runs (score / end-only clean-class passes)
    int dbg;               // BdxTuning::debug (only builds with -DBDX_TUNING look at it)
    int known_ok[2];       // pass qualifies for the known-score class (see bdx_bitpar.hip)
    int kb_uniform[2];  // the budget every barcode of the pass has in this filter set, or -1 (mixed)
    int tier_capped;       // tier 1: some barcode's budget was capped below its full budget
    double tier_slo[2];    //   ... per pass the smallest score a barcode beyond its capped budget can have (else +Inf)
    int bpad[2];           // barcode stride of peq[code][barcode], multiple of 32
    int *d_tile_counter;           // device int, zeroed before each launch
    const uint8_t *d_lut;          // device, 256 bytes
    int word_bytes;                // 4: 32-bit sweep words (barcodes <= 32 nt); 8: 64-bit (33..64 nt); 16: 128-bit (65..128 nt)
    const void *d_peq[2];          // device, [ncodes][bpad] sweep words
    const void *d_pvinit[2];       // device, [B]: top-aligned mask of the barcode's rows
    const int32_t *d_kb[2];        // device, [B]: max unit edit operations of a recordable alignment
};

// q-gram seeding in front of the sweep (pigeonhole): tables built in bdx_abi.cpp.
struct BdxSeedPlan {
    int enabled;
    int q;                 // seed length in bases (5..8); key = 2 bits per base
    int bm_words;          // bitmap words (hashed: bit index = (key * 0x9E3779B1) >> (32 - bm_log2))
    int bm_log2;
    int rcap;              // sweep records per read (power of two)
    int qmul;              // sweep-record queue entries per read (the hit queue has twice as many)
    int hash_in_lds;       // hash table small enough to live in LDS
    int hash_log2;         // hash slots = 1 << hash_log2; entry = key << 16 | pass << 15 | (barcode + 1)
    int n_always[2];       // barcodes swept unconditionally (wildcards / too-short pieces)
    const uint32_t *d_bitmap;
    const uint32_t *d_hash;
    const uint8_t *d_hash_ps;      // piece start offset (bases) of every hash entry
    const uint16_t *d_always[2];
    // two-intact-pieces ("diagonal") variant for budgets too large for single seeds (see bdx_bitpar.hip)
    int diag;                      // 1: q = 4 inverted index per read + per-pair diagonal test instead of bitmap / hash
    int diag_kmax;                 // largest operation budget among the seeded barcodes
    double diag_flag_coef;         // expected flagged pairs per read = coef * ((L - 3) / 256)^2 / (L + 24) for reads of L bases
    const uint32_t *d_dmeta[2];    // per barcode: pieces | piece length << 8 (0 = swept unconditionally)
    const uint32_t *d_dkeys[2];    // per barcode: 2 words, 8 bits per piece key (first 4 bases of the piece)
};

// Wave-autonomous seeded kernel (bdx_wave.hip): the single-seed filter + reducer replay of known-score configs whose
// barcodes are plain A/C/G/T, every wave on a tile of its own (no workgroup barriers), bytes transcoded arithmetically.
// Tables are built next to the seed tables of a filter set (bdx_abi.cpp, build_wave_tables); the geometry per batch.
struct BdxWavePlan {
    int enabled;           // config-level eligibility of this filter set
    uint32_t *d_carry = nullptr;  // per launch (dual tiered known-class configs, min_delta = 0): tier 1 leaves the winning survivor of the ONE pass it settled
                                  // for a read it lists here (indexed by read; two state bits ride on the list entry) and the pairs mode takes it over
    int q;                 // seed length (6..8)
    int n_ent;             // seed table entries (one per (key, barcode, piece start))
    int n_barcodes;        // of all passes together (the barcodes of pass 1 are numbered behind those of pass 0)
    int b0;                // barcodes of pass 0
    int split;             // the set's config is outside the known-score class: the kernel only filters (candidate masks + column windows for the exact kernel)
    int bm_bytes;          // direct bitmap over the 4^q keys
    int track_from;        // columns [0, track_from) of a sweep cannot end an alignment within any barcode's budget
    const uint8_t *d_bitmap;
    const uint16_t *d_rank;      // [bm_bytes / 4]: keys present below each 32-bit word of the bitmap
    const uint32_t *d_ent;       // barcode + 1 | piece start << 11 | next entry of the same key << 16
    const uint32_t *d_peq8;      // [B][9] (stride 9 dwords): rows A, C, T, G ((byte >> 1) & 3), 4..7 = symbols no barcode contains
    const uint32_t *d_peq8r;     // the same for the reversed barcodes (known-trim class: trim_side = 3 passes are swept right to left)
    const uint32_t *d_meta;      // [B]: m | kb << 8 | lone-survivor accept threshold << 16
    const uint32_t *d_settle;    // [B]: tier 1 settle bits of a lone survivor per distance (no_delta | with_delta << 16)
    // per batch (size_wave)
    int rw;                // reads per wave tile (32 / 16 / 8)
    int waves;             // waves per workgroup
    int blocks;            // persistent grid
    int span_cap;          // bytes of a tile's span the images hold
    int read_len_hint;     // the read length the geometry was planned for
    int hq_cap, sq_cap;    // entries of a tile's hit queue / sweep list (from the expected chance hits per read)
    int cand_words;        // split mode: candidate mask words per read (both passes)
    int scan_gpr;          // ranged single-pass configs (per batch): groups of sixteen positions scanned per read, 0: the whole flat image
    int ranged;            // some pass has a ref_search_range other than the whole read: per-read column windows in the kernel
    int winm;              // per batch: window mode (bdx_wave_win.hip): scattered tiles that hold only each read's ref_search_range window (`slot` positions per read)
    int kend;              // known-trim class (any trim side per pass): the non-split kernel with position keys — 1: trim sides 5 / none (bdx_wave_end.hip), 2: with reversed
                           // sweeps for trim_side = 3 (bdx_wave_rev.hip), 3: the known-ALIGNMENT class — both positions of every winner by anchored sweeps, statistics (bdx_wave_aln.hip)
    double chance;         // expected chance seed hits per 150-base read (config)
    // pairs mode (two-intact-pieces filter over a gathered list of reads; bdx_pairs.hip): d_bitmap holds the piece
    // tables [kb + 2][256] of barcode masks, there is no hash
    int pairs_kb;          // 0: single seeds; 3 / 4: two intact 4-base pieces within that many diagonals (the largest budget); 8 / 9: two intact
                           // pieces on the SAME diagonal, of six 4-base / eight 3-base pieces (configs whose indels cost more than their mismatches)
    int pairs_spread;      // pairs mode: columns a flagged alignment can lie off its diagonal (classic: the budget; same-diagonal variants: the largest number of indels)
    int nw;                // words of a barcode mask
    int groups;            // groups of 128 barcodes (more than 128 barcodes: one set of piece tables per group, nw = 4)
    int slot;              // pairs mode: flat positions per read of a (scattered) tile: the read length + 15 (its address mod 16), rounded up to 16
    int cpr;               // 16-diagonal chunks scanned per read
};

// Tiered budgets: tier 1 (tier1 = 1) appends the reads it cannot settle to out_list / *out_count; tier 0 then
// runs over in_list[0 .. *in_count) only (list mode).
struct BdxTierArgs {
    int tier1;
    uint32_t *out_list;
    unsigned int *out_count;
    const uint32_t *in_list;
    const unsigned int *in_count;
};

// Implemented in bdx_bitpar.hip.
size_t bdx_bitpar_lds_bytes(const BdxDevCfg &cfg, const BdxBitparPlan &bp, const BdxGenericPlan &gp,
                            const BdxSeedPlan *sp = nullptr);
hipError_t bdx_launch_bitpar(const BdxDevCfg &cfg, const BdxGenericPlan &gp, const BdxBitparPlan &bp,
                             const BdxSeedPlan &sp, const uint8_t *d_seq, const long long *d_off, long long n_reads, const BdxDevOut &out,
                             unsigned long long *d_counts, uint32_t *cand_out0, uint32_t *cand_out1, hipStream_t stream,
                             uint32_t *wins_out0, uint32_t *wins_out1, uint8_t *wcnt_out0, uint8_t *wcnt_out1, int split,
                             uint32_t *exc_list, unsigned int *exc_count, const BdxTierArgs *tier = nullptr);
// max read length of a device-resident batch (one tiny kernel; result written to *d_out)
hipError_t bdx_launch_maxlen(const long long *d_off, long long n_reads, int *d_out, hipStream_t stream);
hipError_t bdx_launch_copy(void *d_dst, const void *src_mapped, size_t bytes, hipStream_t stream, void *d_zero = nullptr, int zero_bytes = 0);

// split-mode outputs of the wave kernel (same buffers and formats as bdx_bitpar.hip's split mode)
struct BdxWaveSplit {
    int cw[2];
    uint32_t *cand_out[2];
    uint32_t *wins_out[2];
    uint8_t *wcnt_out[2];
    int short_lb[2];
};

// Implemented in bdx_wave.hip.
size_t bdx_wave_table_bytes(const BdxWavePlan &wp, int hist_entries);
size_t bdx_wave_area_bytes(int rw, int span_cap, bool pairs, int hq_cap, int sq_cap, int cand_words, bool winm = false);
// Implemented in bdx_wave_end.hip (the known-end instantiations of the same kernel).
hipError_t bdx_launch_wave_end(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                               long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int tier1, double tier_slo, uint32_t *list,
                               unsigned int *list_count, hipStream_t stream, int dbg = 0, double tier_slo1 = 0.0,
                               const BdxDevStats *stats = nullptr);  // (stats / pass_start: the known-alignment class only, kend = 3)
// Implemented in bdx_wave_win.hip (the window-mode instantiations: single-pass known-score configs whose column window is much
// shorter than their reads — only the windows are fetched).
hipError_t bdx_launch_wave_win(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                               long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int tier1, double tier_slo, uint32_t *list,
                               unsigned int *list_count, hipStream_t stream, int dbg = 0);
// Implemented in bdx_pairs.hip (the pairs-mode instantiations of the same kernel).
// (the listed reads d_idmap[0 .. *d_count) are fetched straight from the batch; d_idmap == NULL: every read of the batch)
hipError_t bdx_launch_pairs(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                            long long n_reads, const uint32_t *d_idmap, const unsigned int *d_count, const BdxDevOut &out,
                            unsigned long long *d_counts, uint32_t *list, unsigned int *list_count, hipStream_t stream, int dbg = 0,
                            const BdxWaveSplit *sp = nullptr, const BdxDevStats *stats = nullptr);
hipError_t bdx_launch_wave(const BdxDevCfg &cfg, const BdxWavePlan &wp, int hist_entries, const uint8_t *d_seq, const long long *d_off,
                           long long n_reads, const BdxDevOut &out, unsigned long long *d_counts, int *d_tile_counter, int tier1,
                           double tier_slo, uint32_t *list, unsigned int *list_count, hipStream_t stream, int dbg = 0, const BdxWaveSplit *sp = nullptr,
                           double tier_slo1 = 0.0);

// Implemented in bdx_device.hip.
hipError_t bdx_launch_generic(const BdxDevCfg &cfg, const BdxGenericPlan &plan, const uint8_t *d_seq,
                              const long long *d_off, long long n_reads, const BdxDevOut &out,
                              unsigned long long *d_counts, const uint32_t *d_cand0,
                              const uint32_t *d_cand1, hipStream_t stream, const uint32_t *d_wins0 = nullptr,
                              const uint32_t *d_wins1 = nullptr, const uint8_t *d_wcnt0 = nullptr,
                              const uint8_t *d_wcnt1 = nullptr, const uint32_t *d_list = nullptr,
                              const unsigned int *d_list_count = nullptr, const BdxDevStats *stats = nullptr,
                              const BdxTierArgs *tier = nullptr, const double *tier_slo = nullptr, uint32_t *zero_words = nullptr);
// scratch words of a classify call (tile queues, hand-over / tier list lengths): bytes [64, 512) of one half of the context's
// 1 KiB scratch block; the halves alternate between calls and a call's last launch clears the other one (bdx_abi.cpp)
#define BDX_SCRATCH_WORDS 112
hipError_t bdx_generic_set_lds_limit(size_t bytes);
// test switch BDX_POISON: checks (and sanitises) one hand-over between a producer and its consumer (bdx_device.hip)
hipError_t bdx_launch_poison_check(uint32_t *list, const unsigned int *list_count, long long n_reads, const uint32_t *wins, uint8_t *wcnt,
                                   const uint32_t *cand, int cand_words, int n_barcodes, int check_list, unsigned int *dbg, hipStream_t stream);
