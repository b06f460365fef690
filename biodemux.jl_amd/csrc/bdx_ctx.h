// bdx_ctx.h — the context object behind the C-ABI (private; shared by bdx_abi.cpp and bdx_comm.cpp).
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "bdx_internal.h"

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 256 ? 256 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Developer switches.  Read from the environment ONCE, in bdx_create; none of them
// changes a result — they only select between kernel paths that must agree (the parity tests run them).
struct BdxTuning {
    int no_known = 0;     // BDX_NO_KNOWN: no reducer replay, every config runs split (filter -> exact kernel)
    int no_seed = 0;      // BDX_NO_SEED: no q-gram seeds at all
    int no_diag = 0;      // BDX_NO_DIAG: no two-intact-pieces variant
    int no_windows = 0;   // BDX_NO_WINDOWS: split mode without column windows
    int no_slot = 0;      // BDX_NO_SLOT: long reads use flat staging instead of window slots
    int lds_dp = 0;       // BDX_LDS_DP: exact kernel with LDS columns instead of the register DP
    int bitpar_r = 0;     // BDX_BITPAR_R: forced tile size of the fused kernel
    long long grid = 0;   // BDX_GRID: forced persistent grid
    int diag_min_b = 48;  // BDX_DIAG_MIN_B: barcode threshold of the diagonal filter
    int no_window_upload = 0;  // BDX_NO_WINDOW_UPLOAD: the host entry point always uploads whole reads
    int seed_hash_l2 = 0;  // BDX_SEED_HASH_L2: the piece hash table stays in global memory
    int seed_bm_log2 = 0;  // BDX_SEED_BM_LOG2: size of the seed bitmap (log2 of its bits)
    int no_clean = 0;     // BDX_NO_CLEAN: exact kernel's register DP always in its predicated by-construction form
    int tier_q = 0;       // BDX_TIER_Q: piece length (5..8) the capped budgets of tier 1 are derived from (default: chosen per config)
    int no_pipeline = 0;  // BDX_NO_PIPELINE: the host entry point uploads large batches in one piece
    int no_dense = 0;     // BDX_NO_DENSE: plain-sweep kernels keep the 4-entry slots / window entries also for short barcodes
    int no_band = 0;      // BDX_NO_BAND: the exact kernel never takes the diagonal-band DP
    int poison = 0;       // BDX_POISON: every hand-over buffer is filled with 0xA5 before each classify call (tests: a consumer that reads what no producer wrote gets garbage on every run, not only when the allocator happens to hand back dirty memory)
    int tier0_div = 0;    // BDX_TIER0_DIV: tier 0's list is planned for n_reads / this many reads (default 16; 1: the whole batch)
    int no_kend = 0;      // BDX_NO_KEND: trim_side = 5 configs never take the known-end form of the wave kernel (filter + exact kernel instead)
    int no_pairs = 0;     // BDX_NO_PAIRS: never the pairs-mode kernel (bdx_pairs.hip) between tier 1 and the general kernel
    int no_win = 0;       // BDX_NO_WIN: never the window mode of the wave kernel (bdx_wave_win.hip): reads with a short column window stage whole tiles or stay on the general kernel
    int no_wave = 0;      // BDX_NO_WAVE: never the wave-autonomous kernel (bdx_wave.hip): the general fused kernel answers every read
    int wave_rw = 0;      // BDX_WAVE_RW / BDX_WAVE_WAVES: forced tile size / waves per workgroup of the wave kernel (tuning)
    int wave_waves = 0;
    int no_carry = 0;     // BDX_NO_CARRY: tier 1 of a dual config hands a listed read on without the pass it settled (the pairs mode evaluates both passes again)
    int no_staged_download = 0;  // BDX_NO_STAGED_DOWNLOAD: large result vectors go back with the runtime's own pageable copies
    int wave_maxres = 0;  // BDX_WAVE_MAXRES: resident waves per compute unit the wave kernel's geometry may plan for (default 16 = four per SIMD: the kernels need 114-128 VGPRs; tuning: the occupancy experiment of DESIGN §4)
    int cu_count = 0;     // BDX_CU_COUNT: pretend the device has this many compute units (tests of the grid sizing)
    int no_tier = 0;      // BDX_NO_TIER: no tiered budgets (every read filtered at the full budget)
    int debug = 0;        // BDX_DEBUG: honoured only by builds with -DBDX_TUNING (phase skips: results are wrong)
};

struct bdx_comm_state;  // bdx_comm.cpp

// One complete filter configuration of the fused kernel: sweep tables + seed tables + launch geometry.
// A context holds two: fs[0] filters at the config's full operation budgets; fs[1] — "tier 1" — at budgets
// capped so that single 8-base seeds stay selective (see bdx_abi.cpp, tiered budgets).
struct BdxFilterSet {
    BdxBitparPlan bplan{};
    BdxSeedPlan splan{};
    // weak single seeds kept beside a two-intact-pieces plan: taken when the latter's index does not fit the
    // batch at hand (very many barcodes, reads beyond 312 bases); built at create, while the barcodes are there
    BdxSeedPlan splan_alt{};
    DevBuf bp_tables, seed_tables, seed_tables_alt;
    BdxWavePlan wplan{};   // wave-autonomous kernel (bdx_wave.hip) for this set, when the config qualifies
    DevBuf wave_tables;
    BdxWavePlan wplan_k{};  // known-trim class (ScoreOnly conditions + trim sides): the same tables, the non-split kernel with position keys
    BdxWavePlan wplan_a{};  // known-alignment class (... + summary statistics / per-pass positions wanted): kend = 3
    BdxWavePlan pplan{};   // the same kernel in pairs mode (bdx_pairs.hip) at this set's full budgets, over listed reads
    DevBuf pair_tables;
    BdxWavePlan pplan_k{};  // ... in its known-end form (trim_side = 5 configs)
    BdxWavePlan pplan_a{};  // ... in its known-alignment form (kend = 3)
};

struct bdx_ctx {
    bdx_config_t cfg{};
    BdxDevCfg dev{};
    BdxGenericPlan plan{};
    BdxFilterSet fs[2];
    int cur = 0;        // the set the table builders / planners below work on
    int tiered = 0;     // fs[1] is usable: classify runs tier 1 first, tier 0 on the reads it cannot settle
    BdxFilterSet &F() { return fs[cur]; }
    const BdxFilterSet &F() const { return fs[cur]; }
    BdxTuning tune{};
    DevBuf d_maxlen;
    DevBuf d_tier;      // tiered budgets: reads handed from tier 1 to tier 0
    DevBuf d_carry;     // dual tiered known-class configs: the winning survivor of the pass tier 1 settled, per listed read (BdxWavePlan::d_carry)
    int user_len_hint = 0;  // 0 = measure every device batch
    int filter_used = BDX_FILTER_OFF;
    int device = 0;
    int n_cu = 256;          // compute units of the device (hipDeviceAttributeMultiprocessorCount, read in bdx_create)
    int64_t wave_launches = 0;  // launches of the wave-autonomous kernel
    int64_t pair_launches = 0;  // launches of its pairs mode
    int pair_mmin = 0;          // shortest barcode of the pairs plan
    DevBuf d_dbg;            // [0] hand-over windows the exact kernel refused (not a window: defence in depth; must stay 0)
    DevBuf d_wlist;          // reads the wave kernel hands to the general kernel (plain configs; tiered ones use d_tier)
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;   // host entry point, large batches: chunk uploads beside the previous chunk's kernels
    hipEvent_t copy_events[8] = {};
    int64_t pipelined_calls = 0;
    int64_t staged_downloads = 0;  // large result downloads that went through the page-locked staging buffer (download_items)
    hipStream_t stream = nullptr;
    // device tables
    DevBuf bc_bytes[2], bc_off[2], bc_nn[2];
    DevBuf counts_own;
    unsigned long long *counts = nullptr;
    // staging for the host entry point
    DevBuf d_seq, d_off, d_out_i32, d_out_f64;
    // window upload (host entry point, long reads with short column windows): per-read true lengths / window starts
    DevBuf d_vlen, d_vlo;
    int virt_maxlen = 0;     // > 0 while a window-upload batch is being classified: its longest read
    int tier_q = 8;          // piece length behind tier 1's capped budgets: cap(m) = m / tier_q - 1
    int tier_cap_fixed = -1; // >= 0: the pairs tier — tier 1's budgets are capped at this many operations for every barcode and its
                             // filter is the same-diagonal pairs mode over the whole batch (configs whose min_delta the seed tier cannot prove)
    int pairs_tier = 0;
    bool band_roll_off = false;      // bdx_create: the config has no filter (no hand-over windows) — the exact kernel keeps its by-construction LDS form
    int scratch_par = 0;             // which half of the scratch block the next classify call uses
    bool scratch_clean[2] = {false, false};  // that half is known to hold zeros (cleared by the previous call's last launch)
    bool scratch_zeroed = false;  // the small-batch copy kernel has already cleared the filter kernels' scratch words
    int host_maxlen = 0;     // > 0 while bdx_classify_host runs an ordinary batch: its longest read (seen on the host)
    void *h_stage = nullptr;  // page-locked staging for the verdict vectors of small batches
    size_t h_stage_bytes = 0;
    void *h_in = nullptr;     // page-locked staging for the bytes + offsets of small batches
    void *h_back = nullptr;   // page-locked staging for the result vectors of large batches handed over in pageable memory (download_items)
    size_t h_back_bytes = 0;
    hipEvent_t back_events[10] = {};
    bool back_events_made = false;
    size_t h_in_bytes = 0;
    int64_t window_uploads = 0;
    int64_t band_launches = 0;  // (pass, exact-kernel launch) pairs that ran with the diagonal-band DP enabled
    std::vector<uint8_t> h_win;    // host staging of the compacted windows
    std::vector<int64_t> h_coff;
    std::vector<int32_t> h_vlen, h_vlo;
    // candidate masks (filtered paths)
    DevBuf d_cand[2];
    DevBuf d_wins[2], d_wcnt[2];  // split mode: column windows for the exact kernel
    DevBuf d_exc;                 // known-score mode: reads handed over to the exact kernel
    // DemuxStats histograms (summary = true): [pass][pos | len | raw] int64 tables of st_rows (raw: st_raw_rows)
    // rows x n_barcodes, their all-reduced twins, and the overflow flag
    DevBuf st_tab[2][3], st_sum[2][3], st_flag;
    long long st_rows = 0, st_sum_rows = 0;
    int st_sum_len_rows = 0;  // height of the length table when the summed twins were last filled (its pitch follows from it)
    int st_raw_rows = 0, st_len_rows = 0;
    bool st_len_fixed = false;  // the length table has a height known from the config (else it grows with the reads)
    // multi-GPU: communicator + the reduced counter vector (bdx_comm.cpp)
    bdx_comm_state *comm = nullptr;
    DevBuf counts_sum;
    std::string err;
    std::string path;
    int64_t launches = 0;
    int64_t last_blocks = 0;
};

// records the message on ctx (or as the create error when ctx is NULL) and returns `code`
int bdx_fail(bdx_ctx *ctx, int code, const char *fmt, ...);
void bdx_comm_release(bdx_ctx *ctx);  // bdx_comm.cpp: frees ctx->comm (called by bdx_destroy)
// histogram tables: grow (append zero rows) to at least `rows` rows; synchronises the stream when it grows
int bdx_stats_reserve(bdx_ctx *ctx, long long rows, bool exact = false);  // exact: the height the ranks agreed on
// logical size of a table as bdx_get_stats hands it out: [keys][barcodes]
inline size_t bdx_stats_words(const bdx_ctx *ctx, int pass, int which, long long rows) {
    return (size_t)(which == 2 ? ctx->st_raw_rows : which == 1 ? ctx->st_len_rows : rows) * (size_t)ctx->dev.pass[pass].n_barcodes;
}
inline int bdx_stats_stride(const bdx_ctx *ctx, int which) {  // key stride of the transposed tables (len, raw)
    return ((which == 2 ? ctx->st_raw_rows : ctx->st_len_rows) + 15) & ~15;
}
// words a table occupies on the device (pos: [rows][barcodes]; len, raw: [barcode][key stride])
inline size_t bdx_stats_phys_words(const bdx_ctx *ctx, int pass, int which, long long rows) {
    if (which == 0) return (size_t)rows * (size_t)ctx->dev.pass[pass].n_barcodes;
    return (size_t)bdx_stats_stride(ctx, which) * (size_t)ctx->dev.pass[pass].n_barcodes;
}

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return bdx_fail(ctx, BDX_E_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)
