// bdx_abi.cpp — C-ABI of libbiodemux_hip.so (see include/biodemux_hip.h for the contract and
// the reference lines each entry point replaces).  Host-side only: validation, table upload,
// launch planning, staging buffers.  All arithmetic of the hot path runs in the gfx950
// kernels of bdx_device.hip / bdx_filter.hip; there is NO CPU fallback — without a usable
// HIP device every entry point fails with BDX_E_DEVICE.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bdx_ctx.h"

thread_local std::string g_create_error;
static std::atomic<long long> g_rejected_windows{0};  // summed over the contexts destroyed so far (see bdx_debug_rejected_windows_total)

int bdx_fail(bdx_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

namespace {

#define fail bdx_fail

BdxTuning read_tuning() {
    BdxTuning t;
    t.no_known = getenv("BDX_NO_KNOWN") != nullptr;
    t.no_seed = getenv("BDX_NO_SEED") != nullptr;
    t.no_diag = getenv("BDX_NO_DIAG") != nullptr;
    t.no_windows = getenv("BDX_NO_WINDOWS") != nullptr;
    t.no_slot = getenv("BDX_NO_SLOT") != nullptr;
    t.lds_dp = getenv("BDX_LDS_DP") != nullptr;
    t.no_tier = getenv("BDX_NO_TIER") != nullptr;
    t.no_wave = getenv("BDX_NO_WAVE") != nullptr;
    t.no_win = getenv("BDX_NO_WIN") != nullptr;
    t.no_pairs = getenv("BDX_NO_PAIRS") != nullptr;
    t.no_kend = getenv("BDX_NO_KEND") != nullptr;
    if (const char *e = getenv("BDX_TIER0_DIV")) t.tier0_div = atoi(e);
    t.poison = getenv("BDX_POISON") != nullptr;
    if (const char *e = getenv("BDX_WAVE_RW")) t.wave_rw = atoi(e);
    if (const char *e = getenv("BDX_WAVE_WAVES")) t.wave_waves = atoi(e);
    if (const char *e = getenv("BDX_WAVE_MAXRES")) t.wave_maxres = atoi(e);
    t.no_staged_download = getenv("BDX_NO_STAGED_DOWNLOAD") != nullptr;
    t.no_carry = getenv("BDX_NO_CARRY") != nullptr;
    if (const char *e = getenv("BDX_CU_COUNT")) t.cu_count = atoi(e);
    t.no_clean = getenv("BDX_NO_CLEAN") != nullptr;
    t.no_band = getenv("BDX_NO_BAND") != nullptr;
    t.no_dense = getenv("BDX_NO_DENSE") != nullptr;
    t.no_pipeline = getenv("BDX_NO_PIPELINE") != nullptr;
    if (const char *e = getenv("BDX_TIER_Q")) t.tier_q = atoi(e);
    t.no_window_upload = getenv("BDX_NO_WINDOW_UPLOAD") != nullptr;
    t.seed_hash_l2 = getenv("BDX_SEED_HASH_L2") != nullptr;
    if (const char *e = getenv("BDX_SEED_BM_LOG2")) t.seed_bm_log2 = atoi(e);
    if (const char *e = getenv("BDX_BITPAR_R")) t.bitpar_r = atoi(e);
    if (const char *e = getenv("BDX_GRID")) t.grid = atoll(e);
    if (const char *e = getenv("BDX_DIAG_MIN_B")) t.diag_min_b = atoi(e);
#ifdef BDX_TUNING
    if (const char *e = getenv("BDX_DEBUG")) t.debug = atoi(e);
#endif
    // BDX_NO_WAVE_FALLBACK: known-score forms of the wave kernel list a read whose record tables overflow instead of sweeping
    // it over every barcode themselves (results identical; bdx_last_list_reads shows the difference) — bit 30 of the kernels' dbg word
    if (getenv("BDX_NO_WAVE_FALLBACK")) t.debug |= 1 << 30;
    return t;
}

BdxDevRange cvt_range(const bdx_range_t &r) {
    BdxDevRange d;
    d.start_offset = r.start_offset;
    d.end_offset = r.end_offset;
    d.start_from_end = r.start_from_end != 0;
    d.end_from_end = r.end_from_end != 0;
    return d;
}

const size_t LDS_MAX = 160 * 1024;

// Launch planning for the exact-evaluation kernel: per-lane DP (+origin) columns, barcode
// tables, count histogram and the read staging area must fit the CU's 160 KiB of LDS.
int plan_generic(bdx_ctx *ctx) {
    const BdxDevCfg &d = ctx->dev;
    BdxGenericPlan &p = ctx->plan;
    // SimpleScoring barcodes of <= 32 rows run the register-resident DP: no LDS columns at all
    p.reg_rows = (!d.has_nindel && d.algorithm == BDX_ALG_SEMIGLOBAL && !d.force_lds_dp)
                     ? (d.max_m <= 24 ? 24 : (d.max_m <= 32 ? 32 : 0)) : 0;
    // clean class (bdx_core.h sg_core_clean): costs match >= 0, mismatch / indel >= 1, and barcode_start_range /
    // barcode_end_range that resolve to 1:n for every read (no offset from either end) — then neither binds
    p.clean = 0;
    p.uniform_m = 0;
    p.uniform_len = 0;
    if (p.reg_rows && !ctx->tune.no_clean && d.match >= 0 && d.mismatch >= 1 && d.indel >= 1) {
        bool free_ranges = true, uniform = true;
        int len0 = -1;
        bool same_len = true;
        for (int k = 0; k < (d.is_dual ? 2 : 1); ++k) {
            const bdx_pass_t &ps = ctx->cfg.pass[k];
            const auto whole = [](const bdx_range_t &r) { return !r.start_from_end && r.start_offset <= 1 && r.end_from_end && r.end_offset >= 0; };
            free_ranges = free_ranges && ps.explicit_window == 0 && whole(ps.barcode_start_range) && whole(ps.barcode_end_range);
            for (int b = 0; b < ps.n_barcodes; ++b) {
                const int m = (int)(ps.bc_off[b + 1] - ps.bc_off[b]);
                uniform = uniform && m == p.reg_rows;
                if (len0 < 0) len0 = m;
                same_len = same_len && m == len0;
            }
        }
        p.clean = free_ranges;
        p.uniform_m = free_ranges && uniform;
        // the diagonal-band bodies exist for these barcode lengths (every barcode of the config alike)
        const bool band_len = len0 == 8 || len0 == 10 || len0 == 12 || len0 == 16 || len0 == 20 || len0 == 24 || len0 == 32;
        p.uniform_len = (free_ranges && same_len && band_len) ? len0 : 0;
    }
    // Barcodes beyond the register DP's 32 rows inside the clean class: the rolling diagonal band (bdx_core.h sg_band_roll) —
    // H = two operation budgets at the configured rate + 9 end columns per chunk — instead of max_m + 1 LDS rows per lane
    // (80-nt barcodes with trimming: 648 B per lane = ONE 128-lane workgroup per CU; 26 cells: two 256-lane workgroups).
    p.band_roll = 0;
    p.same_len = 0;
    ctx->dev.band_hcap = 0;
    // (only behind a filter: without hand-over windows — filter off, barcodes beyond the sweep's 128 rows — every candidate
    // would be walked over its whole window in chunks, three times the full matrix, where sg_core's cut-off visits a few rows
    // per column: 160-nt barcodes unfiltered 0.3 -> 0.03 M reads/s, measured; bdx_create plans again once the filter is known)
    if (!p.reg_rows && d.max_m > 32 && !d.has_nindel && d.algorithm == BDX_ALG_SEMIGLOBAL && !d.force_lds_dp && !ctx->tune.no_clean &&
        !ctx->band_roll_off && !getenv("BDX_NO_BAND_ROLL") && d.match >= 0 && d.mismatch >= 1 && d.indel >= 1) {
        bool free_ranges = true, same_len = true;
        int len0 = -1, kb_max = 0;
        const int cmin = d.mismatch < d.indel ? d.mismatch : d.indel;
        for (int k = 0; k < (d.is_dual ? 2 : 1); ++k) {
            const bdx_pass_t &ps = ctx->cfg.pass[k];
            const auto whole = [](const bdx_range_t &r) { return !r.start_from_end && r.start_offset <= 1 && r.end_from_end && r.end_offset >= 0; };
            free_ranges = free_ranges && ps.explicit_window == 0 && whole(ps.barcode_start_range) && whole(ps.barcode_end_range);
            for (int b = 0; b < ps.n_barcodes; ++b) {
                const int m = (int)(ps.bc_off[b + 1] - ps.bc_off[b]);
                if (len0 < 0) len0 = m;
                same_len = same_len && m == len0;
                const int kb = (int)std::floor(d.max_error_rate * (double)m) / cmin;  // (ae as the device computes it, :254; the threshold only tightens)
                kb_max = kb > kb_max ? kb : kb_max;
            }
        }
        // H: at least two budgets + 9 end columns per chunk; up to four budgets + 9 (a clean occurrence has end columns within
        // the budget on either side: one chunk) while two 256-lane workgroups still fit a CU (8 bytes per cell and lane)
        int hcap = 2 * kb_max + 9;
        {
            size_t bc_bytes = 0;
            int nb = 0;
            for (int k = 0; k < (d.is_dual ? 2 : 1); ++k) {
                bc_bytes += ctx->cfg.pass[k].bc_off[ctx->cfg.pass[k].n_barcodes];
                nb += ctx->cfg.pass[k].n_barcodes;
            }
            // (what the workgroup keeps in LDS besides the cells — barcode bytes and tables, the counter histogram — as below)
            const size_t other = (bc_bytes <= 32 * 1024 ? bc_bytes : 0) + (size_t)nb * 8 + (size_t)(d.n_counts <= 2048 ? d.n_counts : 2048) * 4 + 256;
            const size_t room = other + 2048 < (size_t)76 * 1024 ? (size_t)76 * 1024 - other - 2048 : 0;
            const int fit2 = (int)(room / (256 * (d.any_traceback ? 8 : 4))) - 1;  // cells per lane of a workgroup that shares the CU with another one
            const int want = 4 * kb_max + 9;
            const int roomy = want < fit2 ? want : fit2;
            if (roomy > hcap) hcap = roomy;
        }
        if (free_ranges && hcap + 1 < d.max_m + 1 && d.max_error_rate >= 0.0 && d.max_error_rate <= 1.0) {
            p.band_roll = 1;
            p.same_len = same_len ? 1 : 0;
            ctx->dev.band_hcap = hcap;
        }
    }
    p.dp_rows = p.reg_rows ? 1 : (p.band_roll ? ctx->dev.band_hcap + 1 : d.max_m + 1);
    p.dp_rows_fused = d.max_m + 1;
    const size_t per_thread = (size_t)p.dp_rows * 4 * (d.any_traceback ? 2 : 1);
    const int B0 = d.pass[0].n_barcodes, B1 = d.is_dual ? d.pass[1].n_barcodes : 0;
    size_t bc_total = 0;
    for (int k = 0; k < (d.is_dual ? 2 : 1); ++k) bc_total += ctx->cfg.pass[k].bc_off[ctx->cfg.pass[k].n_barcodes];
    p.bc_stage_bytes = bc_total <= 32 * 1024 ? (int)((bc_total + 15) & ~(size_t)15) : 0;
    p.hist_entries = d.n_counts <= 2048 ? d.n_counts : 2048;  // LDS histogram: the scalars + the first per-barcode slots (>= 4)
    const size_t fixed = (size_t)(B0 + 1 + B1 + 1 + B0 + B1) * 4 + 16 + (size_t)p.bc_stage_bytes + 16 +
                         (size_t)p.hist_entries * 4 + 16;
    const int tries[3] = {256, 128, 64};
    for (int t : tries) {
        const size_t need = fixed + per_thread * (size_t)t;
        if (need + 4096 > LDS_MAX && !(t == 64 && need <= LDS_MAX)) continue;
        p.threads = t;
        // Read staging: aim for two resident workgroups per CU (<= 80 KiB each) when that
        // still leaves room for ~192 B per read; otherwise take what is left of the CU.
        // two workgroups per CU (the exact kernels are compiled for two waves per SIMD); the rolling band keeps clear of the last
        // granules (measured on the fused kernel: three workgroups of 54 128 B do not share a CU, three of 51 872 B do)
        const size_t share = p.band_roll ? 77 * 1024 : 80 * 1024;
        size_t budget = need < share ? share - need : 0;
        // (the rolling band is bound by the latency of its LDS chain: resident waves first — two workgroups per CU with whatever
        // staging still fits, reads that do not fit come straight from L2)
        if (budget < (size_t)t * 192 && !(p.band_roll && need <= share)) budget = LDS_MAX - need;
        size_t stage = budget > 64 * 1024 ? 64 * 1024 : budget;
        stage &= ~(size_t)15;
        if (stage < 1024 || p.bc_stage_bytes == 0) stage = 0;
        p.stage_bytes = (int)stage;
        p.lds_bytes = need + stage;
        return BDX_OK;
    }
    return fail(ctx, BDX_E_INVALID,
                "barcodes too long for the on-chip DP columns: max length %d needs %zu B of LDS per lane "
                "(limit: 64 lanes within 160 KiB)",
                d.max_m, per_thread);
}

// ---- tiered budgets --------------------------------------------------------------------------
// Single q-gram seeds are only selective when a barcode's kb + 1 pieces keep >= 8 bases (C2: kb = 2 on 24 nt).
// The reference's default rate 0.2 allows kb = 4 there, which needs the much costlier two-intact-pieces
// filter — although nearly every read that carries a barcode carries it with 0..2 errors.  Tier 1 therefore
// filters with budgets CAPPED at kb1 = m / 8 - 1: it finds, losslessly, every barcode within kb1 operations
// (exact unit distances).  Both reducers of the reference only ever look at the smallest (and second
// smallest) score, so whenever tier 1 finds a barcode and no barcode it cannot see could tie or beat it
// (bdx_bitpar.hip, "tier settle rule"), the read's verdict is final; only the other reads — those without a
// barcode, or with one beyond kb1 — are filtered again at the full budget (tier 0, in list mode).
long long tier_cap(const bdx_ctx *ctx, int m) {
    if (ctx->cur == 0) return (1LL << 40);
    if (ctx->tier_cap_fixed >= 0) return ctx->tier_cap_fixed;  // (the pairs tier)
    const int q = ctx->tier_q >= 5 && ctx->tier_q <= 8 ? ctx->tier_q : 8;
    const int c = m / q - 1;
    return c > 0 ? c : 0;
}

// ---- bit-parallel pre-filter: eligibility and tables (see bdx_bitpar.hip for the argument) ----
int build_bitpar_tables(bdx_ctx *ctx) {
    const bdx_config_t &c = ctx->cfg;
    BdxBitparPlan &bp = ctx->F().bplan;
    bp = BdxBitparPlan{};
    bp.tier_slo[0] = bp.tier_slo[1] = HUGE_VAL;
    if (c.filter == BDX_FILTER_OFF) return BDX_OK;
    const int npass = c.is_dual ? 2 : 1;
    // cost domain: every edit operation must cost >= 1 and a match >= 0
    int cmin = 1;
    if (c.algorithm == BDX_ALG_SEMIGLOBAL) {
        cmin = c.mismatch < c.indel ? c.mismatch : c.indel;
        if (c.has_nindel && c.nindel < cmin) cmin = c.nindel;
        if (c.match < 0 || cmin < 1) return BDX_OK;
    }
    const bool n_wild = (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel) || c.algorithm == BDX_ALG_HAMMING;
    // alphabet = distinct barcode bytes (<= 15: the IUPAC letters), everything else shares the "other" code
    int code_of[256];
    for (int i = 0; i < 256; ++i) code_of[i] = -1;
    int K = 0;
    size_t cand_words = 0;
    size_t wb = 4;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        cand_words += (size_t)(p.n_barcodes + 31) / 32;
        for (int b = 0; b < p.n_barcodes; ++b) {
            const uint32_t m = p.bc_off[b + 1] - p.bc_off[b];
            if (m > 128) return BDX_OK;  // one sweep word per barcode: 32 bits, 64 for barcodes of 33..64 nt, 128 for 65..128 nt
            if (m > 64) wb = 16;
            else if (m > 32 && wb < 8) wb = 8;
            for (uint32_t i = 0; i < m; ++i) {
                const uint8_t ch = p.bc_bytes[p.bc_off[b] + i];
                if (code_of[ch] < 0) {
                    if (K == 15) return BDX_OK;
                    code_of[ch] = K++;
                }
            }
        }
    }
    if (cand_words > 128) return BDX_OK;  // <= 4096 barcodes per config (both passes together)
    bp.ncodes = K + 1;
    bp.ncode_N = code_of['N'] >= 0 ? code_of['N'] : 255;
    std::vector<uint8_t> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)(code_of[i] < 0 ? K : code_of[i]);
    size_t bytes = 256;
    bp.word_bytes = (int)wb;
    size_t o_peq[2] = {0, 0}, o_pv[2] = {0, 0}, o_kb[2] = {0, 0};
    for (int k = 0; k < npass; ++k) {
        const int B = c.pass[k].n_barcodes;
        bp.bpad[k] = 32;  // power of two >= B: peq row address = code << log2(4*bpad)
        while (bp.bpad[k] < B) bp.bpad[k] <<= 1;
        if ((size_t)bp.ncodes * bp.bpad[k] * wb > 96 * 1024) return BDX_OK;  // the table lives in LDS
        bytes = (bytes + 15) & ~(size_t)15;
        o_peq[k] = bytes;
        bytes += (size_t)bp.ncodes * bp.bpad[k] * wb;
        o_pv[k] = bytes;
        bytes += (size_t)B * wb;
        o_kb[k] = bytes;
        bytes += (size_t)B * 4;
    }
    std::vector<uint8_t> blob(bytes, 0);
    memcpy(blob.data(), lut.data(), 256);
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        uint8_t *peq = blob.data() + o_peq[k];
        uint8_t *pv = blob.data() + o_pv[k];
        int32_t *kb = (int32_t *)(blob.data() + o_kb[k]);
        const int bits = (int)wb * 8;
        bp.kb_uniform[k] = -2;  // (unset)
        typedef unsigned __int128 u128;
        const auto put = [&](uint8_t *dst, size_t idx, u128 v) {
            if (wb == 16)
                memcpy(dst + idx * 16, &v, 16);
            else if (wb == 8)
                ((uint64_t *)dst)[idx] = (uint64_t)v;
            else
                ((uint32_t *)dst)[idx] = (uint32_t)v;
        };
        for (int b = 0; b < p.n_barcodes; ++b) {
            const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
            const int shift = bits - m;
            const u128 all = bits == 128 ? ~(u128)0 : (((u128)1 << bits) - 1);
            const u128 rows = m == bits ? all : ((((u128)1 << m) - 1) << shift);
            const u128 pad = ~rows & all;  // virtual rows below the barcode: match everything, D stays 0
            put(pv, (size_t)b, rows);
            for (int code = 0; code < bp.ncodes; ++code) {
                u128 mask = pad;
                for (int i = 0; i < m; ++i) {
                    const uint8_t ch = p.bc_bytes[p.bc_off[b] + i];
                    const bool wild = n_wild && ch == 'N';
                    if (wild || (code < K && code_of[ch] == code)) mask |= (u128)1 << (shift + i);
                }
                put(peq, (size_t)code * bp.bpad[k] + b, mask);
            }
            // allowed_error at the initial threshold, exactly as the device computes it
            long long ae;
            if (c.algorithm == BDX_ALG_EXACT)
                ae = 0;
            else if (c.algorithm == BDX_ALG_HAMMING)
                ae = (long long)std::floor(c.max_error_rate * (double)m);
            else
                ae = (long long)std::floor(c.max_error_rate * (double)(c.has_nindel ? p.bc_len_no_N[b] : m));
            long long kfull = ae < 0 ? -1 : ae / cmin;
            long long kcap = kfull;
            if (kcap > tier_cap(ctx, m)) kcap = tier_cap(ctx, m);
            kb[b] = (int32_t)kcap;
            bp.kb_uniform[k] = bp.kb_uniform[k] == -2 ? (int)kcap : (bp.kb_uniform[k] == (int)kcap ? (int)kcap : -1);
            if (kcap < kfull) {
                // the smallest score a barcode tier 1 cannot see may have: (kb1 + 1) operations of cost >= cmin
                // each, over this barcode's normalisation (computed as the device computes a score)
                const double norm = (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel) ? (double)p.bc_len_no_N[b] : (double)m;
                const double lo = (double)((kcap + 1) * cmin) / norm;
                if (lo < bp.tier_slo[k]) bp.tier_slo[k] = lo;
                bp.tier_capped = 1;
            }
        }
    }
    HIP_TRY(ctx, ctx->F().bp_tables.ensure(bytes));
    HIP_TRY(ctx, hipMemcpy(ctx->F().bp_tables.p, blob.data(), bytes, hipMemcpyHostToDevice));
    const uint8_t *base = (const uint8_t *)ctx->F().bp_tables.p;
    bp.d_lut = base;
    for (int k = 0; k < npass; ++k) {
        bp.d_peq[k] = base + o_peq[k];
        bp.d_pvinit[k] = base + o_pv[k];
        bp.d_kb[k] = (const int32_t *)(base + o_kb[k]);
    }
    // Reducer replay capacity: short barcodes at high rates have many GENUINE candidates per read (a 10-mer within two
    // edits of a random 150-base read is common: ~25 of 96 barcodes), and a read with more survivors than the replay
    // holds costs a full exact DP per candidate.  Expected candidates per read ~ sum over barcodes of
    // 150 * V(m, kb) / 4^m with V = sum_{e <= kb} C(m, e) 8^e (3 substitutions, 4 insertions, 1 deletion per site).
    {
        double expected = 0.0;
        for (int k = 0; k < npass; ++k) {
            const bdx_pass_t &p = c.pass[k];
            const int32_t *kbh = (const int32_t *)(blob.data() + o_kb[k]);
            for (int b = 0; b < p.n_barcodes; ++b) {
                const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
                if (kbh[b] < 0 || m > 20) continue;
                double v = 0.0, term = 1.0;
                for (int e = 0; e <= kbh[b] && e <= m; ++e) {
                    v += term;
                    term *= 8.0 * (double)(m - e) / (double)(e + 1);
                }
                expected += 150.0 * v / std::pow(4.0, (double)m);
            }
        }
        bp.slot_cap = expected < 1.0 ? 4 : expected < 2.5 ? 8 : expected < 8.0 ? 16 : 32;
        int total_b = 0;
        for (int k = 0; k < npass; ++k) total_b += c.pass[k].n_barcodes;
        bp.dense_d = expected >= 1.0 && total_b <= 256 && !ctx->tune.no_dense;  // (used by the kernels without seeds only; they then keep four slots)
    }
    // known-score class (config level): SimpleScoring with unit costs, ScoreOnly output.
    // (:exact with whole ranges IS the class at a budget of 0: exact_align, classification.jl:485-548, returns (0.0, s, s + m - 1)
    // for an occurrence — the leftmost, or the rightmost with trim_side = 3 — else Inf: the value, the end of the first column
    // at distance 0 and the largest origin of a distance-0 alignment; raw bytes are compared, N is a literal: SimpleScoring.
    // With a ref_search_range its meaning differs — allowed START positions, SURVEY Q11 — so only whole ranges qualify.)
    const auto whole_rng = [](const bdx_range_t &r) { return !r.start_from_end && r.start_offset <= 1 && r.end_from_end && r.end_offset >= 0; };
    for (int k = 0; k < npass; ++k) {
        const bool score_only = c.pass[k].trim_side == 0 && !c.need_traceback;
        const bool unit_sg = c.algorithm == BDX_ALG_SEMIGLOBAL && !c.has_nindel && c.match == 0 && c.mismatch == 1 && c.indel == 1;
        const bool exact_whole = c.algorithm == BDX_ALG_EXACT && c.pass[k].explicit_window == 0 && whole_rng(c.pass[k].ref_search_range) &&
                                 whole_rng(c.pass[k].barcode_start_range) && whole_rng(c.pass[k].barcode_end_range) && !getenv("BDX_NO_KNOWN_EXACT");
        bp.known_ok[k] = (unit_sg || exact_whole) && score_only && c.pass[k].explicit_window != BDX_WINDOW_ALIGN_ONE && !ctx->tune.no_known;
    }
    bp.enabled = 1;
    return BDX_OK;
}


// ---- q-gram seeding (pigeonhole) in front of the sweep -------------------------------------
// A recordable alignment of barcode b has at most kb[b] edit operations (see the sweep), so of
// kb[b]+1 disjoint pieces of the barcode at least one occurs in the read unchanged; a fortiori
// the first q bases of that piece do.  Pairs without any such seed hit cannot be candidates and
// are not swept.  Keys use 2 bits per base (symbol code & 3): equal bytes give equal keys, other
// bytes may alias — that only adds sweeps, never removes one.
int build_seed_tables(bdx_ctx *ctx, bool strict, bool alt = false) {
    const bdx_config_t &c = ctx->cfg;
    BdxSeedPlan &sp = alt ? ctx->F().splan_alt : ctx->F().splan;
    DevBuf &tables = alt ? ctx->F().seed_tables_alt : ctx->F().seed_tables;
    sp = BdxSeedPlan{};
    if (!ctx->F().bplan.enabled || c.filter == BDX_FILTER_BITPAR || ctx->tune.no_seed) return BDX_OK;
    const int npass = c.is_dual ? 2 : 1;
    int cmin = 1;
    if (c.algorithm == BDX_ALG_SEMIGLOBAL) {
        cmin = c.mismatch < c.indel ? c.mismatch : c.indel;
        if (c.has_nindel && c.nindel < cmin) cmin = c.nindel;
    }
    const bool n_wild = (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel) || c.algorithm == BDX_ALG_HAMMING;
    // same symbol coding as the sweep
    int code_of[256];
    for (int i = 0; i < 256; ++i) code_of[i] = -1;
    int K = 0;
    for (int k = 0; k < npass; ++k)
        for (uint32_t i = 0; i < c.pass[k].bc_off[c.pass[k].n_barcodes]; ++i) {
            const uint8_t ch = c.pass[k].bc_bytes[i];
            if (code_of[ch] < 0) code_of[ch] = K++;
        }
    struct Piece { int pass, b, start; };
    std::vector<Piece> pieces;
    std::vector<uint16_t> always[2];
    int q = 8;
    int total_bc = 0;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        if (p.n_barcodes > 32767) return BDX_OK;
        total_bc += p.n_barcodes;
        for (int b = 0; b < p.n_barcodes; ++b) {
            const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
            long long ae;
            if (c.algorithm == BDX_ALG_EXACT) ae = 0;
            else if (c.algorithm == BDX_ALG_HAMMING) ae = (long long)std::floor(c.max_error_rate * (double)m);
            else ae = (long long)std::floor(c.max_error_rate * (double)(c.has_nindel ? p.bc_len_no_N[b] : m));
            if (ae < 0) continue;  // can never be recorded: neither seeded nor swept
            long long kb = ae / cmin;
            if (kb > tier_cap(ctx, m)) kb = tier_cap(ctx, m);
            bool wild = false;
            for (int i = 0; i < m; ++i) wild |= n_wild && p.bc_bytes[p.bc_off[b] + i] == 'N';
            const long long L = m / (kb + 1);
            if (wild || L < 5) {
                always[k].push_back((uint16_t)b);
                continue;
            }
            if (L < q) q = (int)L;
            for (long long t = 0; t <= kb; ++t) pieces.push_back(Piece{k, b, (int)(t * L)});
        }
    }
    if (pieces.empty()) return BDX_OK;
    if ((int)(always[0].size() + always[1].size()) * 4 > total_bc) return BDX_OK;  // seeding would not pay
    if (pieces.size() > 16384) return BDX_OK;
    // selectivity: expected seed-hit pairs per read of ~150 bases must be well below B
    const double space = std::pow(4.0, q);
    const double expected = 150.0 * (double)pieces.size() / space + 1.0 + (double)(always[0].size() + always[1].size());
    if (expected * 3.0 > (double)total_bc) return BDX_OK;
    // strict: single seeds only when they are really selective (q = 7, 8 in practice).  With ~14 falsely
    // seeded barcodes per read (q = 6 at B = 96) the hit queue / record tables cost more than the
    // two-intact-pieces variant, which is tried next (measured at kb = 3: 7.5 ms vs 5.7 ms per 2 M reads).
    if (strict && expected > 7.0) return BDX_OK;
    sp.q = q;
    // hashed bitmap with >= 96 bits per key (<= ~1 % false hits per position), at most the key space
    // itself (then it is exact).  Too many false hits overflow the hit queue, and an overflow costs a
    // whole-read sweep of every barcode.
    sp.bm_log2 = 5;
    while ((1u << sp.bm_log2) < pieces.size() * 96 && sp.bm_log2 < 2 * q) sp.bm_log2++;
    // sweep records per read: the true barcode(s) plus the expected falsely seeded ones, generously
    {
        const double false_pairs = 150.0 * (double)pieces.size() / space;
        sp.rcap = 8;
        while (sp.rcap < 64 && (double)sp.rcap < 4.0 + 4.0 * false_pairs) sp.rcap *= 2;
        // queues: the planted pair plus the chance pairs, with slack for the spread between the reads of a tile
        sp.qmul = sp.rcap >= 16 ? 8 : 4;
        const int want = (int)std::ceil((1.0 + false_pairs) * 1.5 + 1.0);  // (C2: 4 — one more entry per read would cost the fourth workgroup per CU)
        if (want > sp.qmul) sp.qmul = want > 48 ? 48 : want;
    }
    if (ctx->tune.seed_bm_log2 > 0) sp.bm_log2 = ctx->tune.seed_bm_log2 < 2 * q ? ctx->tune.seed_bm_log2 : 2 * q;
    sp.bm_words = (1 << sp.bm_log2) / 32;
    sp.hash_log2 = 8;
    while ((1u << sp.hash_log2) < pieces.size() * 2) sp.hash_log2++;
    sp.hash_in_lds = ((size_t)5 << sp.hash_log2) <= 8 * 1024;  // larger tables are probed in L2 (a few probes per read)
    if (ctx->tune.seed_hash_l2) sp.hash_in_lds = 0;
    std::vector<uint32_t> bitmap(sp.bm_words, 0), hash((size_t)1 << sp.hash_log2, 0);
    std::vector<uint8_t> hash_ps((size_t)1 << sp.hash_log2, 0);
    const uint32_t hmask = (1u << sp.hash_log2) - 1;
    for (const Piece &pc : pieces) {
        const bdx_pass_t &p = c.pass[pc.pass];
        uint32_t key = 0;
        for (int i = 0; i < q; ++i) key |= (uint32_t)(code_of[p.bc_bytes[p.bc_off[pc.b] + pc.start + i]] & 3) << (2 * i);
        // same cheap fold as the kernel's scan (direct index when the bitmap spans the key space)
        const uint32_t hb = sp.bm_log2 >= 2 * q ? key : ((key ^ (key >> sp.bm_log2)) & ((1u << sp.bm_log2) - 1u));
        bitmap[hb >> 5] |= 1u << (hb & 31);
        const uint32_t entry = (key << 16) | ((uint32_t)pc.pass << 15) | (uint32_t)(pc.b + 1);
        uint32_t slot = (key * 0x9E3779B1u) >> (32 - sp.hash_log2);
        // one entry per (key, barcode, piece start): two pieces of one barcode may share a key
        bool dup = false;
        while (hash[slot] != 0) {
            if (hash[slot] == entry && hash_ps[slot] == (uint8_t)pc.start) { dup = true; break; }
            slot = (slot + 1) & hmask;
        }
        if (!dup) {
            hash[slot] = entry;
            hash_ps[slot] = (uint8_t)pc.start;
        }
    }
    size_t bytes = bitmap.size() * 4 + hash.size() * 4 + ((hash_ps.size() + 15) & ~(size_t)15);
    const size_t o_always[2] = {bytes, bytes + ((always[0].size() * 2 + 15) & ~(size_t)15)};
    bytes = o_always[1] + ((always[1].size() * 2 + 15) & ~(size_t)15) + 16;
    std::vector<uint8_t> blob(bytes, 0);
    memcpy(blob.data(), bitmap.data(), bitmap.size() * 4);
    memcpy(blob.data() + bitmap.size() * 4, hash.data(), hash.size() * 4);
    memcpy(blob.data() + bitmap.size() * 4 + hash.size() * 4, hash_ps.data(), hash_ps.size());
    for (int k = 0; k < 2; ++k)
        if (!always[k].empty()) memcpy(blob.data() + o_always[k], always[k].data(), always[k].size() * 2);
    HIP_TRY(ctx, tables.ensure(bytes));
    HIP_TRY(ctx, hipMemcpy(tables.p, blob.data(), bytes, hipMemcpyHostToDevice));
    const uint8_t *base = (const uint8_t *)tables.p;
    sp.d_bitmap = (const uint32_t *)base;
    sp.d_hash = (const uint32_t *)(base + bitmap.size() * 4);
    sp.d_hash_ps = base + bitmap.size() * 4 + hash.size() * 4;
    for (int k = 0; k < 2; ++k) {
        sp.n_always[k] = (int)always[k].size();
        sp.d_always[k] = (const uint16_t *)(base + o_always[k]);
    }
    sp.enabled = 1;
    return BDX_OK;
}

// ---- two-intact-pieces ("diagonal") seeding for budgets where single pieces are too short -----
// With kb operations allowed, kb+2 disjoint pieces of the barcode leave at least TWO untouched; they
// occur in the read on diagonals (read position - barcode offset) that differ by at most kb (the
// indels between them), and the alignment starts within kb of either diagonal.  The kernel keeps,
// per read, an inverted index of its 4-mers (256 keys x position bits) and tests every (read,
// barcode) pair with a handful of word operations per piece; only pairs with two such pieces are
// swept, over the columns [d_min - kb - 1, d_max + m + kb + 1).  Lossless for the same reason as the
// single-piece seeds: it only skips pairs whose unit distance exceeds kb.
int build_diag_tables(bdx_ctx *ctx) {
    const bdx_config_t &c = ctx->cfg;
    BdxSeedPlan &sp = ctx->F().splan;
    if (sp.enabled || !ctx->F().bplan.enabled || c.filter == BDX_FILTER_BITPAR || ctx->tune.no_seed || ctx->tune.no_diag ||
        ctx->F().bplan.word_bytes != 4)  // (the diagonal variant has 32-bit sweep words)
        return BDX_OK;
    const int npass = c.is_dual ? 2 : 1;
    int cmin = 1;
    if (c.algorithm == BDX_ALG_SEMIGLOBAL) {
        cmin = c.mismatch < c.indel ? c.mismatch : c.indel;
        if (c.has_nindel && c.nindel < cmin) cmin = c.nindel;
    }
    const bool n_wild = (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel) || c.algorithm == BDX_ALG_HAMMING;
    int code_of[256];
    for (int i = 0; i < 256; ++i) code_of[i] = -1;
    int K = 0;
    for (int k = 0; k < npass; ++k)
        for (uint32_t i = 0; i < c.pass[k].bc_off[c.pass[k].n_barcodes]; ++i) {
            const uint8_t ch = c.pass[k].bc_bytes[i];
            if (code_of[ch] < 0) code_of[ch] = K++;
        }
    std::vector<uint32_t> meta[2], keys[2];
    std::vector<uint16_t> always[2];
    int total_bc = 0, kmax = 0;
    double flagged = 0.0;  // expected falsely flagged pairs per read of ~150 bases
    double flag_coef = 0.0;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        if (p.n_barcodes > 32767) return BDX_OK;
        total_bc += p.n_barcodes;
        meta[k].assign((size_t)p.n_barcodes, 0u);
        keys[k].assign((size_t)p.n_barcodes * 2, 0u);
        for (int b = 0; b < p.n_barcodes; ++b) {
            const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
            long long ae;
            if (c.algorithm == BDX_ALG_EXACT) ae = 0;
            else if (c.algorithm == BDX_ALG_HAMMING) ae = (long long)std::floor(c.max_error_rate * (double)m);
            else ae = (long long)std::floor(c.max_error_rate * (double)(c.has_nindel ? p.bc_len_no_N[b] : m));
            if (ae < 0) continue;  // can never be recorded: neither seeded nor swept (meta 0 and not in `always`)
            const long long kb = ae / cmin;
            bool wild = false;
            for (int i = 0; i < m; ++i) wild |= n_wild && p.bc_bytes[p.bc_off[b] + i] == 'N';
            const long long P = kb + 2;
            const long long L = m / P;
            if (wild || L < 4 || P > 8 || kb > 6 || (P - 1) * L > 28) {
                always[k].push_back((uint16_t)b);
                continue;
            }
            if (kb > kmax) kmax = (int)kb;
            meta[k][b] = (uint32_t)P | ((uint32_t)L << 8);
            uint64_t kk = 0;
            for (long long t = 0; t < P; ++t) {
                uint32_t key = 0;
                for (int i = 0; i < 4; ++i) key |= (uint32_t)(code_of[p.bc_bytes[p.bc_off[b] + t * L + i]] & 3) << (2 * i);
                kk |= (uint64_t)key << (8 * t);
            }
            keys[k][2 * b] = (uint32_t)kk;
            keys[k][2 * b + 1] = (uint32_t)(kk >> 32);
            const double hits = 147.0 / 256.0;  // occurrences of one 4-mer in the read
            flagged += (double)(P * (P - 1) / 2) * hits * hits * (double)(2 * kb + 1) / (150.0 + m);
            flag_coef += (double)(P * (P - 1) / 2) * (double)(2 * kb + 1);
        }
    }
    const size_t n_always = always[0].size() + always[1].size();
    if (n_always == (size_t)total_bc) return BDX_OK;
    if (n_always * 4 > (size_t)total_bc) return BDX_OK;
    // worth it only if clearly fewer pairs are swept (a flagged pair costs ~ a quarter of a whole-read sweep)
    if ((flagged + (double)n_always) * 2.0 > (double)total_bc) return BDX_OK;
    // ... and only for enough barcodes: the index forces 4..8-read tiles, whose per-tile latency costs about
    // as much as sweeping ~40 barcodes over a whole 150-base read (measured: 1.44 us/read + 0.017 us/pair
    // against 0.054 us/pair of the plain sweep)
    {
        if (total_bc < ctx->tune.diag_min_b) return BDX_OK;  // 48 unless overridden for tuning experiments
    }
    const double coef_keep = flag_coef;
    size_t bytes = 0;
    size_t o_meta[2], o_keys[2], o_always[2];
    for (int k = 0; k < 2; ++k) {
        o_meta[k] = bytes;
        bytes += (meta[k].size() * 4 + 15) & ~(size_t)15;
        o_keys[k] = bytes;
        bytes += (keys[k].size() * 4 + 15) & ~(size_t)15;
        o_always[k] = bytes;
        bytes += (always[k].size() * 2 + 15) & ~(size_t)15;
    }
    bytes += 16;
    std::vector<uint8_t> blob(bytes, 0);
    for (int k = 0; k < 2; ++k) {
        if (!meta[k].empty()) memcpy(blob.data() + o_meta[k], meta[k].data(), meta[k].size() * 4);
        if (!keys[k].empty()) memcpy(blob.data() + o_keys[k], keys[k].data(), keys[k].size() * 4);
        if (!always[k].empty()) memcpy(blob.data() + o_always[k], always[k].data(), always[k].size() * 2);
    }
    HIP_TRY(ctx, ctx->F().seed_tables.ensure(bytes));
    HIP_TRY(ctx, hipMemcpy(ctx->F().seed_tables.p, blob.data(), bytes, hipMemcpyHostToDevice));
    const uint8_t *base = (const uint8_t *)ctx->F().seed_tables.p;
    sp = BdxSeedPlan{};
    sp.diag_flag_coef = coef_keep;
    for (int k = 0; k < 2; ++k) {
        sp.d_dmeta[k] = (const uint32_t *)(base + o_meta[k]);
        sp.d_dkeys[k] = (const uint32_t *)(base + o_keys[k]);
        sp.n_always[k] = (int)always[k].size();
        sp.d_always[k] = (const uint16_t *)(base + o_always[k]);
    }
    sp.q = 4;
    sp.diag = 1;
    sp.diag_kmax = kmax;
    sp.rcap = 8;
    sp.enabled = 1;
    return BDX_OK;
}

// ---- wave-autonomous kernel (bdx_wave.hip): tables of one filter set ------------------------------
// Eligible: single pass in the known-score class (ScoreOnly, unit costs), strict single seeds for every barcode
// (no barcode swept unconditionally), barcodes of plain A / C / G / T up to 32 nt, and ranges that resolve to 1:n
// for every read (then final_search_range = 1:n, max_start_pos = n, min_end_pos = 1: neither binds, DESIGN.md
// §3.1).  Same pieces, keys and budgets as build_seed_tables / build_bitpar_tables of the set — only the symbol
// coding differs: the kernel transcodes arithmetically, code = (byte >> 1) & 3 (A 0, C 1, T 2, G 3).
// (developer aid: with BDX_TRACE_LAUNCH set, build_wave_tables says where it turned a filter set away)
#define WAVE_NO() (getenv("BDX_TRACE_LAUNCH") ? (void)fprintf(stderr, "[bdx] no wave tables for set %d: bdx_abi.cpp:%d\n", ctx->cur, __LINE__) : (void)0, BDX_OK)
int build_wave_tables(bdx_ctx *ctx) {
    const bdx_config_t &c = ctx->cfg;
    BdxFilterSet &F = ctx->F();
    BdxWavePlan &wp = F.wplan;
    wp = BdxWavePlan{};
    const BdxBitparPlan &bp = F.bplan;
    const BdxSeedPlan &sp = F.splan;
    const int npass = c.is_dual ? 2 : 1;
    if (ctx->tune.no_wave || !bp.enabled || !sp.enabled || sp.diag || bp.word_bytes != 4 || sp.n_always[0] != 0 || sp.n_always[1] != 0 ||
        sp.q < 6 || sp.q > 8 || (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel))
        return WAVE_NO();
    // known-score configs: the kernel replays the reducer itself (single pass); everything else in the filters' domain:
    // "split" — it only filters, candidate masks and column windows go to the exact kernel (either pass count)
    bool split = false;
    for (int k = 0; k < npass; ++k) split |= !bp.known_ok[k];
    // :hamming / :exact (always split: their scans run in the exact kernel, restricted to the hand-over windows): the
    // budget is floor(rate * m) substitutions / 0, one operation costs 1
    const bool sgm = c.algorithm == BDX_ALG_SEMIGLOBAL;
    // the known classes' config condition: unit-cost SimpleScoring, or :exact (whole ranges: checked below / in build_bitpar_tables)
    const bool kclass = (sgm && !c.has_nindel && c.match == 0 && c.mismatch == 1 && c.indel == 1) || (c.algorithm == BDX_ALG_EXACT && !getenv("BDX_NO_KNOWN_EXACT"));
    int cmin = sgm ? (c.mismatch < c.indel ? c.mismatch : c.indel) : 1;
    if (cmin < 1 || (sgm && c.match < 0)) return WAVE_NO();
    const auto whole = [](const bdx_range_t &r) { return !r.start_from_end && r.start_offset <= 1 && r.end_from_end && r.end_offset >= 0; };
    int Btot = 0, cwt = 0;
    bool ranged = false;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        // (a ref_search_range is allowed for :semiglobal: the kernel resolves every read's column window itself, classification.jl:795-807;
        // start / end ranges that could bind stay on the general kernel)
        if (p.explicit_window != 0 || !whole(p.barcode_start_range) || !whole(p.barcode_end_range)) return WAVE_NO();
        if (!whole(p.ref_search_range)) {
            if (c.algorithm != BDX_ALG_SEMIGLOBAL) return WAVE_NO();
            ranged = true;
        }
        if (p.n_barcodes < 1) return WAVE_NO();
        for (uint32_t i = 0; i < p.bc_off[p.n_barcodes]; ++i) {
            const uint8_t ch = p.bc_bytes[i];
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') return WAVE_NO();
        }
        Btot += p.n_barcodes;
        cwt += (p.n_barcodes + 31) / 32;
    }
    if (Btot > 1024 || (split && cwt > 16)) return WAVE_NO();  // (split mode keeps the candidate words of a read in LDS: up to 512 barcodes)
    const int q = sp.q;
    struct Piece { int g, start; const uint8_t *bc; };
    std::vector<Piece> pieces;
    std::vector<uint32_t> meta((size_t)Btot, 0u), peq8((size_t)Btot * 9, 0u), settle((size_t)Btot, 0u);  // (stride 9: bank spread, see the kernel)
    std::vector<uint32_t> peq8r((size_t)Btot * 9, 0u);  // the reversed barcodes (known-trim class: trim_side = 3 passes are swept right to left)
    int track = 1 << 20, g = 0;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        for (int b = 0; b < p.n_barcodes; ++b, ++g) {
            const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
            if (m < 1 || m > 32) return WAVE_NO();
            const uint8_t *bc = p.bc_bytes + p.bc_off[b];
            const int shift = 32 - m;
            const uint32_t rows = m == 32 ? 0xFFFFFFFFu : (((1u << m) - 1u) << shift);
            const uint32_t pad = ~rows;  // virtual rows below the barcode: match everything, D stays 0
            for (int code = 0; code < 8; ++code) {
                uint32_t mask = pad;
                if (code < 4)
                    for (int i = 0; i < m; ++i)
                        if (((bc[i] >> 1) & 3) == code) mask |= 1u << (shift + i);
                peq8[(size_t)g * 9 + code] = mask;
                uint32_t maskr = pad;
                if (code < 4)
                    for (int i = 0; i < m; ++i)
                        if (((bc[m - 1 - i] >> 1) & 3) == code) maskr |= 1u << (shift + i);
                peq8r[(size_t)g * 9 + code] = maskr;
            }
            const long long ae = c.algorithm == BDX_ALG_EXACT ? 0 : (long long)std::floor(c.max_error_rate * (double)m);  // (normalisation = m)
            if (ae < 0) {  // can never be recorded: neither seeded nor swept (budget field 255)
                meta[(size_t)g] = (uint32_t)m | (255u << 8) | (255u << 16);
                continue;
            }
            long long kb = ae / cmin;
            if (kb > tier_cap(ctx, m)) kb = tier_cap(ctx, m);
            if (kb > 15) return WAVE_NO();  // (a record keeps the diagonals of its hits as 2 kb + 1 bits)
            const long long L = m / (kb + 1);
            if (L < q) return WAVE_NO();  // (cannot happen: the set's q is the shortest piece)
            // lone-survivor tables of the replay (bdx_wave.hip): the reference accepts a survivor with distance d iff
            // d <= floor(max_error_rate * m) (:254) and score = d / m <= max_error_rate (:658 / :696) — both Float64, both
            // evaluated here exactly as the device would; tier 1 settles it iff score < slo (and, with_delta, the bound
            // slo - score >= min_delta proves "not ambiguous")
            int dmax = 255;
            uint32_t sbits = 0;
            for (long long d = 0; d <= kb && d <= 15; ++d) {
                const double score = (double)d / (double)m;
                if (d <= ae && score <= c.max_error_rate) dmax = (int)d;
                const double slo = bp.tier_slo[k];
                if (score < slo) {
                    sbits |= 1u << d;
                    if ((slo - score) >= c.min_delta) sbits |= 1u << (16 + d);
                }
            }
            settle[(size_t)g] = sbits;
            meta[(size_t)g] = (uint32_t)m | ((uint32_t)kb << 8) | ((uint32_t)dmax << 16);
            if (m - (int)kb - 1 < track) track = m - (int)kb - 1;
            for (long long t = 0; t <= kb; ++t) pieces.push_back(Piece{g, (int)(t * L), bc});
        }
    }
    if (pieces.empty() || pieces.size() > 8192) return WAVE_NO();
    // the per-read record table holds eight (barcode, diagonal cluster) records: the planted one(s) plus the chance pairs must nearly always fit
    {
        // chance seed hits per 150-base read: the hit queue and the sweep list of a tile are sized from it (size_wave)
        // (measured, 24-nt barcodes, 2 M reads: B = 192 / 384 / 768 at rate 0.1 — chance 1.3 / 2.6 / 5.3 — 1.99 -> 4.85, 1.49 -> 3.52,
        // 0.94 -> 1.62 G reads/s against the general kernel; as tier 1 of rate 0.2: 0.92 -> 1.22, 0.50 -> 0.64, 0.25 -> 0.17: whatever
        // overflows there costs a full-budget evaluation)
        double limit = ctx->cur == 1 ? 3.0 : 6.0;
        if (const char *e = getenv("BDX_WAVE_CHANCE")) limit = atof(e);  // (tuning experiment)
        wp.chance = 150.0 * (double)pieces.size() / std::pow(4.0, (double)q);
        if (wp.chance > limit) return WAVE_NO();
    }
    wp.q = q;
    wp.n_barcodes = Btot;
    wp.b0 = c.pass[0].n_barcodes;
    wp.split = split ? 1 : 0;
    wp.ranged = ranged ? 1 : 0;
    wp.cand_words = split ? cwt : (c.is_dual ? 4 : 0);  // (known-score dual configs: four survivor slots of pass 1 per read in that area)
    wp.bm_bytes = (1 << (2 * q)) / 8;
    wp.track_from = track < 0 ? 0 : (track > 28 ? 28 : track);
    // seed table: the bitmap is exact (one bit per key of the 4^q key space), so a hit's entry is found by the RANK of its
    // key among the keys present (prefix count per bitmap word + a popcount); pieces that share a key are chained
    std::vector<uint8_t> bitmap((size_t)wp.bm_bytes, 0);
    struct Ent { uint32_t key; int g, start; };
    std::vector<Ent> ents;
    for (const Piece &pc : pieces) {
        uint32_t key = 0;
        for (int i = 0; i < q; ++i) key |= (uint32_t)((pc.bc[pc.start + i] >> 1) & 3) << (2 * i);
        bool dup = false;  // one entry per (key, barcode, piece start)
        for (const Ent &e : ents) dup |= e.key == key && e.g == pc.g && e.start == pc.start;
        if (dup) continue;
        bitmap[key >> 3] |= (uint8_t)(1u << (key & 7));
        ents.push_back(Ent{key, pc.g, pc.start});
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.key < y.key; });
    std::vector<uint32_t> ent;  // heads (one per key, in key order) first, chained entries behind them
    std::vector<uint32_t> chain;
    {
        std::vector<size_t> head_of;  // index into ents of every head
        for (size_t i = 0; i < ents.size(); ++i)
            if (i == 0 || ents[i].key != ents[i - 1].key) head_of.push_back(i);
        const size_t D = head_of.size();
        if (ents.size() >= 65536) return WAVE_NO();
        ent.assign(ents.size(), 0u);
        size_t next_free = D;
        for (size_t h = 0; h < D; ++h) {
            const size_t first = head_of[h], last = h + 1 < D ? head_of[h + 1] : ents.size();
            size_t at = h;
            for (size_t i = first; i < last; ++i) {
                const size_t nxt = i + 1 < last ? next_free++ : 0;
                ent[at] = (uint32_t)(ents[i].g + 1) | ((uint32_t)ents[i].start << 11) | ((uint32_t)nxt << 16);
                at = nxt;
            }
        }
    }
    std::vector<uint16_t> rank((size_t)wp.bm_bytes / 4, 0);
    {
        uint32_t run = 0;
        for (size_t w = 0; w < rank.size(); ++w) {
            rank[w] = (uint16_t)run;
            uint32_t word;
            memcpy(&word, bitmap.data() + 4 * w, 4);
            run += (uint32_t)__builtin_popcount(word);
        }
    }
    wp.n_ent = (int)ent.size();
    // the tables must leave room for at least eight waves' work areas at the smallest tile
    if (bdx_wave_table_bytes(wp, ctx->plan.hist_entries) > 64 * 1024) return WAVE_NO();
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_bm = 0, o_rank = al(bitmap.size()), o_ent = o_rank + al(rank.size() * 2), o_peq = o_ent + al(ent.size() * 4),
                 o_meta = o_peq + al(peq8.size() * 4), o_settle = o_meta + al(meta.size() * 4), o_peqr = o_settle + al(settle.size() * 4),
                 bytes = o_peqr + al(peq8r.size() * 4);
    std::vector<uint8_t> blob(bytes, 0);
    memcpy(blob.data() + o_bm, bitmap.data(), bitmap.size());
    memcpy(blob.data() + o_rank, rank.data(), rank.size() * 2);
    memcpy(blob.data() + o_ent, ent.data(), ent.size() * 4);
    memcpy(blob.data() + o_peq, peq8.data(), peq8.size() * 4);
    memcpy(blob.data() + o_meta, meta.data(), meta.size() * 4);
    memcpy(blob.data() + o_settle, settle.data(), settle.size() * 4);
    memcpy(blob.data() + o_peqr, peq8r.data(), peq8r.size() * 4);
    HIP_TRY(ctx, F.wave_tables.ensure(bytes));
    HIP_TRY(ctx, hipMemcpy(F.wave_tables.p, blob.data(), bytes, hipMemcpyHostToDevice));
    const uint8_t *base = (const uint8_t *)F.wave_tables.p;
    wp.d_bitmap = base + o_bm;
    wp.d_rank = (const uint16_t *)(base + o_rank);
    wp.d_ent = (const uint32_t *)(base + o_ent);
    wp.d_peq8 = (const uint32_t *)(base + o_peq);
    wp.d_meta = (const uint32_t *)(base + o_meta);
    wp.d_settle = (const uint32_t *)(base + o_settle);
    wp.d_peq8r = (const uint32_t *)(base + o_peqr);
    wp.enabled = 1;
    // Known-trim class: the known-score conditions with a trim side in some pass (either pass count, no summary).  What a trim
    // side makes observable is one position per pass — trim_side = 5: the alignment's END (keep_start = end + 1,
    // classification.jl:912-914; the reference keeps the leftmost end of the best score, :142-153: strict `<`); trim_side = 3: its
    // START (keep_end = max(1, start) - 1, :910-911; the largest start among the alignments of the best score, :142-153 tie rule
    // + :310-321 origin order) — and the sweep delivers both (bdx_wave.hip, KEND: the lowering mask of a left-to-right sweep /
    // of a right-to-left sweep with the reversed barcode).  Such a config gets its verdicts from the non-split kernel whenever
    // the caller does not ask for per-pass start positions (nor for end positions of a trim_side = 3 pass).
    F.wplan_k = BdxWavePlan{};
    bool trims_ok = true;
    for (int k = 0; k < npass; ++k) trims_ok = trims_ok && c.pass[k].explicit_window != BDX_WINDOW_ALIGN_ONE;
    if (split && kclass && trims_ok && !c.need_traceback &&
        !ctx->tune.no_known && !ctx->tune.no_kend) {
        bool fits = true;  // entry = barcode << 22 | d << 16 | position key
        for (uint32_t x : meta) fits = fits && (((x >> 8) & 255u) == 255u || ((x >> 8) & 255u) < 64u);
        if (fits && c.pass[0].n_barcodes <= 1023 && (!c.is_dual || c.pass[1].n_barcodes <= 1023)) {
            F.wplan_k = wp;
            F.wplan_k.split = 0;
            F.wplan_k.cand_words = c.is_dual ? 4 : 0;  // (the four survivor slots of pass 1)
            F.wplan_k.kend = 1;
            for (int k = 0; k < npass; ++k)
                if (c.pass[k].trim_side == 3) F.wplan_k.kend = 2;  // (reversed sweeps: bdx_wave_rev.hip)
        }
    }
    // Known-alignment class: the same conditions with `summary` allowed — start AND end of every pass's winner come out of one
    // more (anchored) sweep per pass and read, so per-pass positions and the DemuxStats histograms need no exact kernel either
    // (bdx_wave_aln.hip); taken per launch when the caller wants positions the known-trim class does not know, or statistics.
    F.wplan_a = BdxWavePlan{};
    // (:exact reports the occurrence's positions whatever the output policy, classification.jl:485-548: its score-only form only
    // serves callers that do not ask for them — the others take this class per launch, bdx_classify_device)
    if ((split || c.algorithm == BDX_ALG_EXACT) && kclass && trims_ok && !ctx->tune.no_known &&
        !ctx->tune.no_kend && !getenv("BDX_NO_KALN")) {
        bool fits = true;
        for (uint32_t x : meta) fits = fits && (((x >> 8) & 255u) == 255u || ((x >> 8) & 255u) < 64u);
        if (fits && c.pass[0].n_barcodes <= 1023 && (!c.is_dual || c.pass[1].n_barcodes <= 1023)) {
            F.wplan_a = wp;
            F.wplan_a.split = 0;
            F.wplan_a.cand_words = c.is_dual ? 4 : 0;
            F.wplan_a.kend = 3;
        }
    }
    return BDX_OK;
}

#undef WAVE_NO
// Geometry of the wave kernel for a batch: the tile size and workgroup shape that keep the most waves resident
// per compute unit (tables once per workgroup + one work area per wave within 160 KiB, at most 16 waves: the
// kernel is compiled for four waves per SIMD).  false: this batch runs the general kernel.
bool size_wave(bdx_ctx *ctx, BdxWavePlan &wp, int read_len, long long n_reads);
bool size_wave(bdx_ctx *ctx, int set, int read_len, long long n_reads) { return size_wave(ctx, ctx->fs[set].wplan, read_len, n_reads); }
bool size_wave(bdx_ctx *ctx, BdxWavePlan &wp, int read_len, long long n_reads) {
    wp.winm = 0;
    if (!wp.enabled || ctx->dev.vlen) return false;  // (window uploads stage per-read slots: general kernel)
    if (read_len < 1) read_len = 1;
    const size_t tables = bdx_wave_table_bytes(wp, ctx->plan.hist_entries);
    const int rws[3] = {32, 16, 8};
    int best_waves = 0;
    for (int rw : rws) {
        if (ctx->tune.wave_rw && rw != ctx->tune.wave_rw) continue;
        // small batches: at least one tile per resident wave before the tile grows
        if (!ctx->tune.wave_rw && rw > 8 && n_reads / rw < (long long)ctx->n_cu * 16) continue;
        const long long span = (((long long)rw * read_len + 64 + 15) & ~15LL);
        if (span > 10 * 1024) continue;  // a tile's bytes wait in registers: at most ten 16-byte vectors per lane
        // queues: the planted barcode's pieces (up to kb + 1 = 3 hits, one or two records) + the chance hits, with slack
        const int hq_cap = rw * (int)std::ceil(std::max(6.0, 4.0 + 2.5 * wp.chance));
        const int sq_cap = rw * (int)std::ceil(std::max(3.0, 1.8 + 1.6 * wp.chance));
        const size_t area = bdx_wave_area_bytes(rw, (int)span, false, hq_cap, sq_cap, wp.cand_words + (wp.ranged ? 4 : 0));
        const int maxres = ctx->tune.wave_maxres > 0 ? ctx->tune.wave_maxres : 16;
        const int shapes[4] = {8, 16, 4, ctx->tune.wave_waves};  // (a forced shape may be any wave count up to 16)
        for (int w : shapes) {
            if (w < 1 || w > 16 || (ctx->tune.wave_waves && w != ctx->tune.wave_waves)) continue;
            const size_t lds = tables + (size_t)w * area;
            if (lds > LDS_MAX) continue;
            int per_cu = (int)(LDS_MAX / (((lds + 1279) / 1280) * 1280));  // 1280-byte LDS granules
            if (per_cu * w > maxres) per_cu = maxres / w;
            if (per_cu < 1) continue;
            const int resident = per_cu * w;
            if (resident > best_waves) {
                best_waves = resident;
                wp.rw = rw;
                wp.waves = w;
                wp.blocks = per_cu * ctx->n_cu;
                wp.span_cap = (int)span;
                wp.hq_cap = hq_cap;
                wp.sq_cap = sq_cap;
            }
        }
        if (best_waves >= 12) break;  // a larger tile at (nearly) full residency beats a smaller one
    }
    if (best_waves < 4) return false;
    wp.read_len_hint = read_len;
    // ranged single-pass configs: the seed scan only walks the groups of sixteen positions that overlap a read's window when
    // the window (resolved at the planned read length, classification.jl:795-800) is much shorter than the read
    wp.scan_gpr = 0;
    if (wp.ranged) {
        const int npw = ctx->dev.is_dual ? 2 : 1;
        long long gpr = 0;
        for (int k = 0; k < npw; ++k) {
            const BdxDevRange &dr = ctx->dev.pass[k].ref_search;
            long long f = dr.start_from_end ? read_len + dr.start_offset : dr.start_offset;
            long long l = dr.end_from_end ? read_len + dr.end_offset : dr.end_offset;
            if (f < 1) f = 1;
            if (l > read_len) l = read_len;
            const long long wlen = l >= f ? l - f + 1 : 0;
            const long long g = (wlen + 15) / 16 + 1;
            if (g > gpr) gpr = g;
        }
        if (gpr * npw * 16 * 10 <= (long long)read_len * 7 && (long long)wp.rw * npw * gpr < 2048) wp.scan_gpr = (int)gpr;
    }
    return true;
}

// Window mode of the wave kernel (bdx_wave_win.hip) for a batch: single-pass known-score configs whose ref_search_range
// window — resolved at the planned read length (classification.jl:795-800) — is at most half the read: the tiles are
// scattered, every read's slot holds just its window (+ up to 15 positions in front: the loads are aligned 16-byte vectors).
// This is what lets 10 kbp reads with a 200-column window (BASELINE config 5) take the wave kernel at all: a tile's bytes
// wait in registers, which bounds a contiguous tile at 10 KB.
bool size_wave_win(bdx_ctx *ctx, BdxWavePlan &wp, int read_len, long long n_reads) {
    wp.winm = 0;
    if (!wp.enabled || ctx->dev.vlen || !wp.ranged || wp.split || wp.kend || ctx->dev.is_dual || ctx->tune.no_wave || ctx->tune.no_win) return false;
    if (read_len < 1) read_len = 1;
    const BdxDevRange &dr = ctx->dev.pass[0].ref_search;
    {
        const long long LIM = 1LL << 28;  // (the kernel resolves the windows in 32-bit arithmetic)
        if (dr.start_offset < -LIM || dr.start_offset > LIM || dr.end_offset < -LIM || dr.end_offset > LIM) return false;
    }
    long long f = dr.start_from_end ? read_len + dr.start_offset : dr.start_offset;
    long long l = dr.end_from_end ? read_len + dr.end_offset : dr.end_offset;
    if (f < 1) f = 1;
    if (l > read_len) l = read_len;
    const long long wlen = l >= f ? l - f + 1 : 0;
    if (wlen < 1 || wlen * 2 > read_len) return false;
    const int slot = (int)((wlen + 15 + 15) & ~15LL);
    const size_t tables = bdx_wave_table_bytes(wp, ctx->plan.hist_entries);
    const double chance = wp.chance * (double)wlen / 150.0;
    int best_waves = 0;
    const int rws[2] = {32, 16};
    for (int rw : rws) {
        if (ctx->tune.wave_rw && rw != ctx->tune.wave_rw) continue;
        if (!ctx->tune.wave_rw && rw > 16 && n_reads / rw < (long long)ctx->n_cu * 16) continue;  // small batches: a tile per resident wave first
        const int vecs = rw * (slot >> 4);
        if (!((rw == 32 && vecs <= 64 * 7) || (rw == 16 && vecs <= 64 * 4))) continue;  // (instantiated register budgets)
        const int span = rw * slot + 16;
        const int hq_cap = rw * (int)std::ceil(std::max(6.0, 4.0 + 2.5 * chance));
        const int sq_cap = rw * (int)std::ceil(std::max(3.0, 1.8 + 1.6 * chance));
        const size_t area = bdx_wave_area_bytes(rw, span, false, hq_cap, sq_cap, 0, true);
        const int shapes[3] = {16, 8, 4};  // (tried: 32-read tiles on 12 waves per CU — C5 0.329 vs 0.314 ms with 16-read tiles on 16 waves)
        for (int w : shapes) {
            if (ctx->tune.wave_waves && w != ctx->tune.wave_waves) continue;
            const size_t lds = tables + (size_t)w * area;
            if (lds > LDS_MAX) continue;
            int per_cu = (int)(LDS_MAX / (((lds + 1279) / 1280) * 1280));
            if (per_cu * w > 16) per_cu = 16 / w;
            if (per_cu * w > best_waves) {
                best_waves = per_cu * w;
                wp.rw = rw;
                wp.waves = w;
                wp.blocks = per_cu * ctx->n_cu;
                wp.span_cap = span;
                wp.hq_cap = hq_cap;
                wp.sq_cap = sq_cap;
            }
        }
        if (best_waves >= 12) break;
    }
    if (best_waves < 4) return false;
    wp.slot = slot;
    wp.read_len_hint = read_len;
    wp.scan_gpr = 0;
    wp.winm = 1;
    return true;
}

// ---- pairs mode of the wave kernel (bdx_pairs.hip): tables of the FULL-budget set --------------------------------
// Between tier 1 and the general kernel of a tiered config: the reads tier 1 lists are gathered into slots and filtered
// by the two-intact-pieces lemma on barcode masks (one table entry per (piece, 4-base key): the barcodes whose piece has
// that key).  Eligible: the conditions of build_wave_tables on alphabet, lengths and ranges; every barcode's budget kb
// at most 4 with 4 (kb + 2) <= m (kb + 2 disjoint 4-base pieces at offsets 0, 4, ..); at most 128 barcodes.
int build_pair_tables(bdx_ctx *ctx) {
    const bdx_config_t &c = ctx->cfg;
    BdxFilterSet &F = ctx->F();
    BdxWavePlan &wp = F.pplan;
    wp = BdxWavePlan{};
    const BdxBitparPlan &bp = F.bplan;
    const int npass = c.is_dual ? 2 : 1;
    // (a capped set gets pair tables only as the pairs tier: budgets capped at tier_cap_fixed operations)
    if (ctx->tune.no_wave || ctx->tune.no_pairs || !bp.enabled || bp.word_bytes != 4 || (bp.tier_capped && ctx->tier_cap_fixed < 0) || c.filter != BDX_FILTER_AUTO ||
        (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel))
        return BDX_OK;
    bool split = false;
    for (int k = 0; k < npass; ++k) split |= !bp.known_ok[k];
    const bool sgm = c.algorithm == BDX_ALG_SEMIGLOBAL;
    const bool kclass = (sgm && !c.has_nindel && c.match == 0 && c.mismatch == 1 && c.indel == 1) || (c.algorithm == BDX_ALG_EXACT && !getenv("BDX_NO_KNOWN_EXACT"));
    const int cmin = sgm ? (c.mismatch < c.indel ? c.mismatch : c.indel) : 1;
    if (cmin < 1 || (sgm && c.match < 0)) return BDX_OK;
    // the pairs tier (capped set): the filter only has to be lossless for alignments of COST <= cap x cmin — a barcode it does not
    // flag costs more, i.e. at least (cap + 1) cmin = the tier's slo (costs are integers; the tier exists for mismatch = cmin = 1)
    const long long cost_cap = (ctx->cur == 1 && ctx->tier_cap_fixed >= 0) ? (long long)ctx->tier_cap_fixed * cmin : (1LL << 40);
    const auto whole = [](const bdx_range_t &r) { return !r.start_from_end && r.start_offset <= 1 && r.end_from_end && r.end_offset >= 0; };
    int Btot = 0, cwt = 0;
    bool ranged = false;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        // (a ref_search_range is allowed for :semiglobal: the kernel resolves every read's column window itself, classification.jl:795-807;
        // start / end ranges that could bind stay on the general kernel)
        if (p.explicit_window != 0 || !whole(p.barcode_start_range) || !whole(p.barcode_end_range)) return BDX_OK;
        if (!whole(p.ref_search_range)) {
            if (c.algorithm != BDX_ALG_SEMIGLOBAL) return BDX_OK;
            ranged = true;
        }
        if (p.n_barcodes < 1) return BDX_OK;
        for (uint32_t i = 0; i < p.bc_off[p.n_barcodes]; ++i) {
            const uint8_t ch = p.bc_bytes[i];
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') return BDX_OK;
        }
        Btot += p.n_barcodes;
        cwt += (p.n_barcodes + 31) / 32;
    }
    // more than 128 barcodes (known-score configs only: split mode keeps four mask words per read): groups of 128 barcodes,
    // each with its own piece tables of four-word masks
    if (Btot > 512 || (split && (cwt > 4 || Btot > 128))) return BDX_OK;
    const int groups = (Btot + 127) / 128;
    int nw = groups > 1 ? 4 : (Btot + 31) / 32;
    if (const char *e = getenv("BDX_PAIRS_NW")) nw = atoi(e) > nw ? atoi(e) : nw;  // (tuning experiment)
    if (nw > 4) nw = 4;
    const int estride = nw <= 2 ? 8 : 16;
    std::vector<uint32_t> meta((size_t)Btot, 0u), peq8((size_t)Btot * 9, 0u), settle((size_t)Btot, 0u), peq8r((size_t)Btot * 9, 0u);
    int kmax = 0, track = 1 << 20, mmin = 1 << 20, g = 0;
    struct Bc { int g, m, kb; const uint8_t *bc; };
    std::vector<Bc> bcs;
    // SAME-DIAGONAL variants (split configs whose indels cost more than their mismatches — the reference's demo2 options:
    // mismatch 1, indel 2, budget 6 of 24): an alignment with g indels lies on at most g + 1 diagonals and has at most
    // e(g) = g + floor((ae - g indel) / mismatch) operations; with P disjoint pieces, P - e(g) >= g + 2 for every possible g
    // puts two intact pieces on ONE diagonal (an intact piece cannot span an indel) — far more selective than "two pieces
    // within kb diagonals", and valid beyond the classic variant's 4 (kb + 2) <= m.  Tried per piece length: 4 bases
    // (six pieces), then 3 (eight).  The alignment then lies within g_max columns of that diagonal (`spread`).
    // Order of preference: six 4-base pieces on one diagonal (strictly more selective than the classic variant), the classic
    // variant (two 4-base pieces within kb diagonals) where its conditions hold, eight 3-base pieces on one diagonal.
    bool classic_ok = true;
    for (int k = 0; k < npass; ++k)
        for (int b = 0; b < c.pass[k].n_barcodes; ++b) {
            const int m = (int)(c.pass[k].bc_off[b + 1] - c.pass[k].bc_off[b]);
            const long long ae = std::min(cost_cap, c.algorithm == BDX_ALG_EXACT ? 0 : (long long)std::floor(c.max_error_rate * (double)m));
            if (ae >= 0 && (ae / cmin > 4 || 4 * (ae / cmin + 2) > m)) classic_ok = false;
        }
    int sd_pl = 0, sd_spread = 0;
    if (split && sgm && c.mismatch >= 1 && c.indel >= 1 && groups == 1) {
        for (int pl = 4; pl >= 3 && !sd_pl; --pl) {
            if (pl == 3 && classic_ok) break;
            bool ok = true;
            int spread = 0;
            for (int k = 0; k < npass && ok; ++k) {
                const bdx_pass_t &p = c.pass[k];
                for (int b = 0; b < p.n_barcodes && ok; ++b) {
                    const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
                    const long long ae = std::min(cost_cap, (long long)std::floor(c.max_error_rate * (double)m));
                    if (ae < 0) continue;
                    const int P = std::min(pl == 4 ? 6 : 8, m / pl);
                    const long long gmax = ae / c.indel;
                    for (long long gg = 0; gg <= gmax && ok; ++gg) {
                        const long long e = gg + (ae - gg * c.indel) / c.mismatch;
                        ok = (long long)P - e >= gg + 2;
                    }
                    if (gmax > spread) spread = (int)gmax;
                    if (ae / cmin > 15 || m - (int)(ae / cmin) - 1 < 12) ok = false;  // (sweep budget field / score tracking from column 12)
                }
            }
            if (ok && spread <= 8) {
                sd_pl = pl;
                sd_spread = spread;
            }
        }
    }
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c.pass[k];
        for (int b = 0; b < p.n_barcodes; ++b, ++g) {
            const int m = (int)(p.bc_off[b + 1] - p.bc_off[b]);
            if (m < 1 || m > 32) return BDX_OK;
            const uint8_t *bc = p.bc_bytes + p.bc_off[b];
            const int shift = 32 - m;
            const uint32_t rows = m == 32 ? 0xFFFFFFFFu : (((1u << m) - 1u) << shift);
            const uint32_t pad = ~rows;
            for (int code = 0; code < 8; ++code) {
                uint32_t mask = pad;
                if (code < 4)
                    for (int i = 0; i < m; ++i)
                        if (((bc[i] >> 1) & 3) == code) mask |= 1u << (shift + i);
                peq8[(size_t)g * 9 + code] = mask;
                uint32_t maskr = pad;  // (the reversed barcode: known-trim class, see build_wave_tables)
                if (code < 4)
                    for (int i = 0; i < m; ++i)
                        if (((bc[m - 1 - i] >> 1) & 3) == code) maskr |= 1u << (shift + i);
                peq8r[(size_t)g * 9 + code] = maskr;
            }
            const long long ae = std::min(cost_cap, c.algorithm == BDX_ALG_EXACT ? 0 : (long long)std::floor(c.max_error_rate * (double)m));
            if (ae < 0) {  // can never be recorded: in no table entry, never swept
                meta[(size_t)g] = (uint32_t)m | (255u << 8) | (255u << 16);
                continue;
            }
            const long long kb = ae / cmin;
            if (!sd_pl && (kb > 4 || 4 * (kb + 2) > m)) return BDX_OK;
            int dmax = 255;
            for (long long d = 0; d <= kb; ++d)  // lone-survivor accept threshold of the replay, as in build_wave_tables
                if (d <= ae && (double)d / (double)m <= c.max_error_rate) dmax = (int)d;
            meta[(size_t)g] = (uint32_t)m | ((uint32_t)kb << 8) | ((uint32_t)dmax << 16) | ((uint32_t)sd_spread << 24);
            if ((int)kb > kmax) kmax = (int)kb;
            if (m - (int)kb - 1 < track) track = m - (int)kb - 1;
            if (m < mmin) mmin = m;
            bcs.push_back(Bc{g, m, (int)kb, bc});
        }
    }
    if (bcs.empty() || track < 12) return BDX_OK;
    const int KB = sd_pl == 4 ? 8 : sd_pl == 3 ? 9 : kmax <= 3 ? 3 : 4;  // (the kernel's variant number)
    const int PL = sd_pl ? sd_pl : 4, P = sd_pl == 4 ? 6 : sd_pl == 3 ? 8 : KB + 2, NK = 1 << (2 * PL);
    std::vector<uint32_t> tab((size_t)groups * P * NK * (size_t)(estride / 4), 0u);
    for (const Bc &x : bcs) {
        const int np = sd_pl ? std::min(P, x.m / PL) : x.kb + 2;  // pieces of this barcode
        for (int t = 0; t < np; ++t) {
            uint32_t key = 0;
            for (int i = 0; i < PL; ++i) key |= (uint32_t)((x.bc[PL * t + i] >> 1) & 3) << (2 * i);
            const int grp = x.g >> 7, gl = x.g & 127;
            tab[(((size_t)grp * P + (size_t)t) * NK + key) * (size_t)(estride / 4) + (size_t)(gl >> 5)] |= 1u << (gl & 31);
        }
    }
    wp.q = 4;
    wp.n_barcodes = Btot;
    wp.b0 = c.pass[0].n_barcodes;
    wp.split = split ? 1 : 0;
    wp.bm_bytes = (int)(tab.size() * 4);
    wp.n_ent = 0;
    wp.track_from = track > 28 ? 28 : track;
    wp.pairs_kb = KB;
    wp.pairs_spread = sd_pl ? sd_spread : KB;
    wp.nw = nw;
    wp.groups = groups;
    wp.ranged = ranged ? 1 : 0;
    wp.cand_words = split ? cwt : (c.is_dual ? 4 : 0);  // (known-score dual configs: the survivor slots of pass 1)
    if (bdx_wave_table_bytes(wp, ctx->plan.hist_entries) > 112 * 1024) return BDX_OK;  // (at least four waves' work areas must fit beside the tables)
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_tab = 0, o_peq = al(tab.size() * 4), o_meta = o_peq + al(peq8.size() * 4), o_settle = o_meta + al(meta.size() * 4),
                 o_peqr = o_settle + al(settle.size() * 4), bytes = o_peqr + al(peq8r.size() * 4);
    std::vector<uint8_t> blob(bytes, 0);
    memcpy(blob.data() + o_tab, tab.data(), tab.size() * 4);
    memcpy(blob.data() + o_peq, peq8.data(), peq8.size() * 4);
    memcpy(blob.data() + o_meta, meta.data(), meta.size() * 4);
    memcpy(blob.data() + o_settle, settle.data(), settle.size() * 4);
    memcpy(blob.data() + o_peqr, peq8r.data(), peq8r.size() * 4);
    HIP_TRY(ctx, F.pair_tables.ensure(bytes));
    HIP_TRY(ctx, hipMemcpy(F.pair_tables.p, blob.data(), bytes, hipMemcpyHostToDevice));
    const uint8_t *base = (const uint8_t *)F.pair_tables.p;
    wp.d_bitmap = base + o_tab;
    wp.d_rank = (const uint16_t *)base;  // (never read)
    wp.d_ent = (const uint32_t *)base;
    wp.d_peq8 = (const uint32_t *)(base + o_peq);
    wp.d_meta = (const uint32_t *)(base + o_meta);
    wp.d_settle = (const uint32_t *)(base + o_settle);
    wp.d_peq8r = (const uint32_t *)(base + o_peqr);
    ctx->pair_mmin = mmin;
    wp.enabled = 1;
    // known-trim class (see build_wave_tables): the listed reads of a config with trim sides get verdict and keep range from
    // the pairs mode too, in its non-split form
    F.pplan_k = BdxWavePlan{};
    bool trims_ok = true;
    for (int k = 0; k < npass; ++k) trims_ok = trims_ok && c.pass[k].explicit_window != BDX_WINDOW_ALIGN_ONE;
    if (split && kclass && trims_ok && !c.need_traceback &&
        !ctx->tune.no_known && !ctx->tune.no_kend && groups == 1 && wp.pairs_kb <= 4) {  // (the same-diagonal variants only exist in split mode)
        F.pplan_k = wp;
        F.pplan_k.split = 0;
        F.pplan_k.cand_words = c.is_dual ? 4 : 0;
        F.pplan_k.kend = 1;
        for (int k = 0; k < npass; ++k)
            if (c.pass[k].trim_side == 3) F.pplan_k.kend = 2;
    }
    F.pplan_a = BdxWavePlan{};  // known-alignment class (build_wave_tables): the same with `summary` allowed
    if (split && kclass && trims_ok && !ctx->tune.no_known &&
        !ctx->tune.no_kend && !getenv("BDX_NO_KALN") && groups == 1 && wp.pairs_kb <= 4) {
        F.pplan_a = wp;
        F.pplan_a.split = 0;
        F.pplan_a.cand_words = c.is_dual ? 4 : 0;
        F.pplan_a.kend = 3;
    }
    return BDX_OK;
}

// Geometry of the pairs mode for a batch: 16-read tiles of slots of `read_len` rounded up to 16 bytes.
bool size_pairs(bdx_ctx *ctx, BdxWavePlan &wp, int read_len);
bool size_pairs(bdx_ctx *ctx, int read_len) { return size_pairs(ctx, ctx->fs[0].pplan, read_len); }
bool size_pairs(bdx_ctx *ctx, BdxWavePlan &wp, int read_len) {
    if (!wp.enabled || ctx->dev.vlen) return false;
    if (read_len < 1) read_len = 1;
    // a read's slot in the tile's images: its bytes are fetched as aligned 16-byte vectors, so it starts up to 15 positions in
    const int slot = (read_len + 15 + 15) & ~15;
    const int rw = 16;
    const int span = rw * slot + 16;
    if (span > 6 * 1024 + 16) return false;  // (instantiated: three and six 16-byte vectors per lane)
    int cpr = ((15 + read_len - ctx->pair_mmin + wp.pairs_spread + 8) >> 4) + 1;  // (diagonals are counted from the slot's start)
    if (read_len < ctx->pair_mmin) cpr = 1;
    if (cpr > slot / 16) cpr = slot / 16;
    if (cpr < 1) cpr = 1;
    const size_t tables = bdx_wave_table_bytes(wp, ctx->plan.hist_entries);
    // (31 chance flags per read at 96 barcodes and kb = 4: the queue holds a 16-read tile's worth; with more barcodes it is
    // drained several times per tile; a tile whose queue runs over between two drains is handed on / swept whole)
    // (same-diagonal variants: ~80 chance flags per read at 96 barcodes of eight 3-base pieces — the queue is drained inside the scan)
    wp.hq_cap = wp.groups > 1 ? 1024 : wp.pairs_kb >= 8 ? 1280 : 56 * rw;
    wp.sq_cap = 0;
    const size_t area = bdx_wave_area_bytes(rw, span, true, wp.hq_cap, 0, wp.cand_words + (wp.ranged ? 4 : 0));
    int best = 0;
    const int shapes[3] = {16, 8, 4};
    for (int w : shapes) {
        if (ctx->tune.wave_waves && w != ctx->tune.wave_waves) continue;
        const size_t lds = tables + (size_t)w * area;
        if (lds > LDS_MAX) continue;
        int per_cu = (int)(LDS_MAX / (((lds + 1279) / 1280) * 1280));
        if (per_cu * w > 16) per_cu = 16 / w;
        if (per_cu * w > best) {
            best = per_cu * w;
            wp.waves = w;
            wp.blocks = per_cu * ctx->n_cu;
        }
    }
    if (best < 4) return false;
    wp.rw = rw;
    wp.span_cap = span;
    wp.slot = slot;
    wp.cpr = cpr;
    wp.read_len_hint = read_len;
    return true;
}

// Geometry of the fused kernel for a given typical read length: the largest R whose LDS
// footprint still lets two workgroups share a CU (8 waves/CU), else whatever fits.
// force_slot: list mode (tier 0 of the tiered budgets) — the reads are scattered, every read is staged into a slot
bool size_bitpar(bdx_ctx *ctx, int read_len, long long n_reads, bool force_slot = false) {
    BdxBitparPlan &bp = ctx->F().bplan;
    if (!bp.enabled) return false;
    if (read_len < 1) read_len = 1;
    if (ctx->dev.vlen) force_slot = true;  // window upload: only each read's window is there
    // small batches: keep >= ~1024 tiles in flight (4 per CU) before growing the tile
    int r_cap = 256;
    while (r_cap > 16 && n_reads / r_cap < 4LL * ctx->n_cu) r_cap >>= 1;
    if (bp.read_len_hint == read_len && bp.r_cap == r_cap && bp.reads_per_block > 0 && (bp.slot_bytes > 0 || !force_slot)) return true;
    bp.r_cap = r_cap;
    const int forced = ctx->tune.bitpar_r;
    // Pick the R that keeps the most waves resident per CU (the sweep is latency-bound):
    // workgroups/CU = min(8, floor(160 KiB / LDS(R))) with 4 waves each; ties -> larger R
    // (fewer table reloads).  R = 16 is only taken when nothing larger fits.
    // Column-window bound for this read length: the union over the passes of
    // final_search_range (classification.jl:799-800), resolved exactly like the device does.
    // Window lengths are non-decreasing in n, so the bound at the hint covers shorter reads.
    int wmax = read_len;
    {
        long long ulo = (1LL << 40), uhi = 0;
        const int npass = ctx->dev.is_dual ? 2 : 1;
        for (int k = 0; k < npass; ++k) {
            const BdxDevPass &P = ctx->dev.pass[k];
            long long f, l;
            if (P.explicit_window) {
                f = P.win_first;
                l = P.win_last;
            } else {
                auto res = [&](const BdxDevRange &dr, long long &a, long long &b) {
                    long long s = dr.start_from_end ? read_len + dr.start_offset : dr.start_offset;
                    long long e = dr.end_from_end ? read_len + dr.end_offset : dr.end_offset;
                    a = s > 1 ? s : 1;
                    b = e < read_len ? e : read_len;
                    if (b < a) b = a - 1;
                };
                long long rf, rl, bf, bl, ef, el;
                res(P.ref_search, rf, rl);
                res(P.bc_start, bf, bl);
                res(P.bc_end, ef, el);
                f = rf > bf ? rf : bf;
                l = rl < el ? rl : el;
            }
            if (f < 1) f = 1;
            if (l > read_len) l = read_len;
            if (l < f) continue;
            long long h = ctx->dev.algorithm == BDX_ALG_SEMIGLOBAL ? l : l + ctx->dev.max_m - 1;
            if (h > read_len) h = read_len;
            if (f - 1 < ulo) ulo = f - 1;
            if (h > uhi) uhi = h;
        }
        if (uhi > ulo) wmax = (int)(uhi - ulo);
        else wmax = 16;
    }
    const bool slot_mode = force_slot || ((long long)wmax * 2 + 96 <= (long long)read_len && !ctx->tune.no_slot);
    const int slot = slot_mode ? ((wmax + 15 + 16 + 15) & ~15) : 0;
    bp.slot_bytes = slot;
    bp.seed_span = slot_mode ? wmax : read_len;
    if (ctx->F().splan.enabled && ctx->F().splan.diag) {
        // index width for this read length, and the sweep queue for the expected number of flagged pairs
        if (bp.seed_span > 312) {  // the widest index holds 320 positions: weak single seeds if they apply, else the plain sweep
            ctx->F().splan = ctx->F().splan_alt;  // (disabled if weak seeds do not apply either)
            ctx->F().splan_alt = BdxSeedPlan{};
            return size_bitpar(ctx, read_len, n_reads, force_slot);
        }
        bp.diag_nw = bp.seed_span <= 152 ? 5 : 10;
        const double L = (double)(bp.seed_span < 32 ? 32 : bp.seed_span);
        const double flagged = ctx->F().splan.diag_flag_coef * ((L - 3.0) / 256.0) * ((L - 3.0) / 256.0) / (L + 24.0) +
                               (double)(ctx->F().splan.n_always[0] + ctx->F().splan.n_always[1]);
        bp.diag_qcap = (int)(flagged * 1.3) + 12;  // per read (a sub-batch shares 4..8 reads' worth)
    }
    const bool diag = ctx->F().splan.enabled && ctx->F().splan.diag;
    const int tries[7] = {256, 128, 64, 32, 16, 8, 4};
    int best_R = 0, best_blocks = 0, best_stage = 0;
    for (int R : tries) {
        if (diag ? R > 32 : R < 16) continue;  // the diagonal variant indexes 8 reads at a time (40 KiB): small tiles
        if (forced && R != forced) continue;
        if (!forced && R > r_cap) continue;
        if (bp.word_bytes == 16 && (R > 64 || R < 16)) continue;  // (128-bit sweep words: instantiated for tiles of 64 / 32 / 16 reads)
        if (!forced && !ctx->F().splan.enabled && R > 64 && read_len <= 1024) continue;  // sweep-all: 64-read tiles measured best
        size_t st = slot_mode ? (size_t)R * (size_t)slot : (size_t)R * (size_t)read_len + 64;
        st = (st + 15) & ~(size_t)15;
        if (st > (size_t)1 << 20) continue;
        bp.reads_per_block = R;
        bp.stage_bytes = (int)st;
        bp.read_len_hint_for_lds = read_len;
        const size_t lds = bdx_bitpar_lds_bytes(ctx->dev, bp, ctx->plan, &ctx->F().splan);
        if (lds > LDS_MAX) continue;
        int blocks = (int)(LDS_MAX / (((lds + 1279) / 1280) * 1280));  // LDS is allocated in 1280-byte granules (measured: 54128 B -> 2 per CU, 51872 B -> 3)
        // Measured on MI355X (tools/probe.py): tile size matters more than residency once 3
        // workgroups (12 waves) share a CU — larger tiles fill the 256 lanes of the sparse
        // sweep / exact stages better.  Rank: >= 3 resident (largest R wins), then 2, then 1.
        const int rank = blocks >= 3 ? 3 : blocks;
        if (!diag && R == 16 && best_R) continue;
        if (rank > best_blocks) {
            best_blocks = rank;
            best_R = R;
            best_stage = (int)st;
        }
    }
    if (best_R && diag && best_blocks < 2) {
        // the index leaves room for one workgroup per CU only (very many barcodes): weak single seeds if they
        // apply, else the plain sweep
        ctx->F().splan = ctx->F().splan_alt;  // (disabled if weak seeds do not apply either)
        ctx->F().splan_alt = BdxSeedPlan{};
        return size_bitpar(ctx, read_len, n_reads, force_slot);
    }
    if (best_R) {
        bp.reads_per_block = best_R;
        bp.stage_bytes = best_stage;
        bp.read_len_hint = read_len;
        bp.read_len_hint_for_lds = read_len;
        return true;
    }
    if (ctx->F().splan.enabled) {
        // the seed tables do not fit next to everything else (very many barcodes): the two-intact-pieces index
        // gives way to the weak single seeds kept beside it; those give way to the plain sweep; plan again
        if (ctx->F().splan.diag && ctx->F().splan_alt.enabled) {
            ctx->F().splan = ctx->F().splan_alt;
            ctx->F().splan_alt = BdxSeedPlan{};
        } else {
            ctx->F().splan.enabled = 0;
        }
        return size_bitpar(ctx, read_len, n_reads, force_slot);
    }
    bp.reads_per_block = 0;
    bp.read_len_hint = 0;
    return false;
}

int validate(const bdx_config_t *c) {
    if (!c) return fail(nullptr, BDX_E_INVALID, "config is NULL");
    if (c->abi_version != BDX_ABI_VERSION)
        return fail(nullptr, BDX_E_INVALID, "ABI version mismatch: header %u, library %u", c->abi_version,
                    (unsigned)BDX_ABI_VERSION);
    if (c->struct_size != sizeof(bdx_config_t))
        return fail(nullptr, BDX_E_INVALID, "bdx_config_t size mismatch: caller %u, library %zu", c->struct_size,
                    sizeof(bdx_config_t));
    if (c->algorithm < BDX_ALG_SEMIGLOBAL || c->algorithm > BDX_ALG_EXACT)
        return fail(nullptr, BDX_E_INVALID, "unknown matching_algorithm %d", c->algorithm);
    if (!std::isfinite(c->max_error_rate) || std::fabs(c->max_error_rate) > BDX_MAX_RATE)
        return fail(nullptr, BDX_E_INVALID, "max_error_rate must be finite and |rate| <= %g", BDX_MAX_RATE);
    if (!std::isfinite(c->min_delta)) return fail(nullptr, BDX_E_INVALID, "min_delta must be finite");
    const int costs[4] = {c->match, c->mismatch, c->indel, c->has_nindel ? c->nindel : 1};
    for (int v : costs)
        if (v > BDX_MAX_COST || v < -BDX_MAX_COST)
            return fail(nullptr, BDX_E_INVALID, "scoring costs must lie in [-%d, %d]", BDX_MAX_COST, BDX_MAX_COST);
    if (c->algorithm == BDX_ALG_SEMIGLOBAL) {
        // the reference divides by indel / min(indel, nindel) (classification.jl:170-176);
        // a zero divisor raises DivideError there for every read.
        const int div = c->has_nindel ? (c->indel < c->nindel ? c->indel : c->nindel) : c->indel;
        if (div == 0) return fail(nullptr, BDX_E_INVALID, "indel (and nindel) must be non-zero (DivideError in the reference)");
    }
    const int npass = c->is_dual ? 2 : 1;
    for (int k = 0; k < npass; ++k) {
        const bdx_pass_t &p = c->pass[k];
        // core.jl:308-313
        if (p.trim_side != 0 && p.trim_side != 3 && p.trim_side != 5)
            return fail(nullptr, BDX_E_INVALID, "trim_side%s must be 3 or 5, got %d", k ? "2" : "", p.trim_side);
        if (p.n_barcodes < 1) return fail(nullptr, BDX_E_INVALID, "pass %d has no barcodes", k + 1);
        if (!p.bc_bytes || !p.bc_off || !p.bc_len_no_N)
            return fail(nullptr, BDX_E_INVALID, "pass %d barcode tables are NULL", k + 1);
        if (p.bc_off[0] != 0) return fail(nullptr, BDX_E_INVALID, "bc_off[0] must be 0");
        for (int i = 0; i < p.n_barcodes; ++i) {
            if (p.bc_off[i + 1] < p.bc_off[i]) return fail(nullptr, BDX_E_INVALID, "bc_off must be non-decreasing");
            const uint32_t m = p.bc_off[i + 1] - p.bc_off[i];
            if (m == 0)
                return fail(nullptr, BDX_E_INVALID, "barcode %d of pass %d is empty (outside the supported domain)", i + 1, k + 1);
            if (m > BDX_MAX_M)
                return fail(nullptr, BDX_E_INVALID, "barcode %d of pass %d is longer than %d", i + 1, k + 1, BDX_MAX_M);
            if (p.bc_len_no_N[i] < 0 || p.bc_len_no_N[i] > (int32_t)m)
                return fail(nullptr, BDX_E_INVALID, "bc_len_no_N[%d] out of range", i);
        }
        if (p.explicit_window < 0 || p.explicit_window > BDX_WINDOW_ALIGN_ONE)
            return fail(nullptr, BDX_E_INVALID, "explicit_window must be 0, 1 or 2");
        if (p.explicit_window) {
            const int64_t lim = (int64_t)1 << 30;
            const int64_t w[4] = {p.win_first, p.win_last, p.win_max_start_pos, p.win_min_end_pos};
            for (int64_t v : w)
                if (v > lim || v < -lim) return fail(nullptr, BDX_E_INVALID, "explicit window values must be within +-2^30");
        }
    }
    return BDX_OK;
}

int upload_tables(bdx_ctx *ctx) {
    const bdx_config_t &c = ctx->cfg;
    BdxDevCfg &d = ctx->dev;
    d.algorithm = c.algorithm;
    d.is_dual = c.is_dual != 0;
    d.max_error_rate = c.max_error_rate;
    d.min_delta = c.min_delta;
    d.match = c.match;
    d.mismatch = c.mismatch;
    d.indel = c.indel;
    d.has_nindel = c.has_nindel != 0;
    d.nindel = c.has_nindel ? c.nindel : 0;
    d.need_traceback = c.need_traceback != 0;
    d.force_lds_dp = ctx->tune.lds_dp;
    d.band_kb[0] = d.band_kb[1] = -1;
    d.band_m = 0;
    d.dense_w = 0;
    d.band_lb[0] = d.band_lb[1] = 0;
    d.max_m = 1;
    d.any_traceback = d.need_traceback;
    const int npass = d.is_dual ? 2 : 1;
    for (int k = 0; k < 2; ++k) {
        BdxDevPass &P = d.pass[k];
        memset(&P, 0, sizeof P);
        if (k >= npass) continue;
        const bdx_pass_t &p = c.pass[k];
        P.ref_search = cvt_range(p.ref_search_range);
        P.bc_start = cvt_range(p.barcode_start_range);
        P.bc_end = cvt_range(p.barcode_end_range);
        P.trim_side = p.trim_side;
        P.n_barcodes = p.n_barcodes;
        P.cand_words = (p.n_barcodes + 31) / 32;
        P.explicit_window = p.explicit_window;
        P.win_first = p.win_first;
        P.win_last = p.win_last;
        P.win_max_start = p.win_max_start_pos;
        P.win_min_end = p.win_min_end_pos;
        if (p.trim_side != 0) d.any_traceback = 1;
        const size_t nbytes = p.bc_off[p.n_barcodes];
        for (int i = 0; i < p.n_barcodes; ++i) {
            const int m = (int)(p.bc_off[i + 1] - p.bc_off[i]);
            if (m > d.max_m) d.max_m = m;
        }
        HIP_TRY(ctx, ctx->bc_bytes[k].ensure(nbytes + 16));
        HIP_TRY(ctx, ctx->bc_off[k].ensure((size_t)(p.n_barcodes + 1) * 4));
        HIP_TRY(ctx, ctx->bc_nn[k].ensure((size_t)p.n_barcodes * 4));
        HIP_TRY(ctx, hipMemcpy(ctx->bc_bytes[k].p, p.bc_bytes, nbytes, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->bc_off[k].p, p.bc_off, (size_t)(p.n_barcodes + 1) * 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->bc_nn[k].p, p.bc_len_no_N, (size_t)p.n_barcodes * 4, hipMemcpyHostToDevice));
        P.bc_bytes = (const uint8_t *)ctx->bc_bytes[k].p;
        P.bc_off = (const uint32_t *)ctx->bc_off[k].p;
        P.bc_len_no_N = (const int32_t *)ctx->bc_nn[k].p;
    }
    d.counts_stride2 = d.is_dual ? (d.pass[1].n_barcodes > 1 ? d.pass[1].n_barcodes : 1) : 1;
    const long long nc = 4LL + (long long)d.pass[0].n_barcodes * d.counts_stride2;
    if (nc > (1LL << 28)) return fail(ctx, BDX_E_INVALID, "sample_counts table too large (%lld entries)", nc);
    d.n_counts = (int)nc;
    // allowed_error = floor(rate * normalisation) must stay inside the int32 DP domain
    if (std::fabs(c.max_error_rate) * (double)d.max_m >= (double)(1 << 27))
        return fail(ctx, BDX_E_INVALID, "max_error_rate * barcode length exceeds the supported range");
    HIP_TRY(ctx, ctx->counts_own.ensure((size_t)d.n_counts * 8));
    HIP_TRY(ctx, hipMemset(ctx->counts_own.p, 0, (size_t)d.n_counts * 8));
    ctx->counts = (unsigned long long *)ctx->counts_own.p;
    return BDX_OK;
}

}  // namespace

int bdx_stats_reserve(bdx_ctx *ctx, long long rows, bool exact) {
    if (!ctx->dev.need_traceback || rows <= ctx->st_rows) return BDX_OK;
    long long want = rows;
    if (!exact) {  // generous steps: a batch with slightly longer reads must not re-allocate every time
        want = ctx->st_rows ? ctx->st_rows + ctx->st_rows / 2 : 0;
        if (want < rows) want = rows;
        want = (want + 63) & ~63LL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // zeroing and copying run on the context's stream (the kernels that add to the tables do, and a non-blocking or
    // caller-supplied stream has no implicit order with the NULL stream); old tables are freed after one more sync
    const int npass = ctx->dev.is_dual ? 2 : 1;
    std::vector<DevBuf> retired;
    auto retire_all = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (DevBuf &b : retired) b.release();
        retired.clear();
    };
    if (!ctx->st_len_fixed) {
        // the transposed len table grows by its key stride: every barcode's run of counters moves to the new pitch
        const int old_stride = bdx_stats_stride(ctx, 1);
        const int old_rows = ctx->st_len_rows;
        ctx->st_len_rows = (int)(want > 0x3FFFFFF0LL ? 0x3FFFFFF0LL : want);
        const int new_stride = bdx_stats_stride(ctx, 1);
        for (int p = 0; p < npass; ++p) {
            const size_t B = (size_t)ctx->dev.pass[p].n_barcodes;
            DevBuf nb;
            hipError_t e1 = nb.ensure((size_t)new_stride * B * 8);
            if (e1 == hipSuccess) e1 = hipMemsetAsync(nb.p, 0, (size_t)new_stride * B * 8, ctx->stream);
            if (e1 == hipSuccess && old_rows > 0 && ctx->st_tab[p][1].p)
                e1 = hipMemcpy2DAsync(nb.p, (size_t)new_stride * 8, ctx->st_tab[p][1].p, (size_t)old_stride * 8, (size_t)old_stride * 8, B,
                                      hipMemcpyDeviceToDevice, ctx->stream);
            if (e1 != hipSuccess) {
                retire_all();
                nb.release();
                ctx->st_len_rows = old_rows;
                return fail(ctx, BDX_E_DEVICE, "growing the statistics tables failed: %s", hipGetErrorString(e1));
            }
            retired.push_back(ctx->st_tab[p][1]);
            ctx->st_tab[p][1] = nb;
        }
    }
    for (int p = 0; p < npass; ++p)
        for (int w = 0; w < 1; ++w) {  // pos (raw has a fixed height; len: above)
            const size_t old_bytes = bdx_stats_phys_words(ctx, p, w, ctx->st_rows) * 8;
            const size_t new_bytes = bdx_stats_phys_words(ctx, p, w, want) * 8;
            DevBuf nb;
            hipError_t e1 = nb.ensure(new_bytes);
            if (e1 == hipSuccess) e1 = hipMemsetAsync(nb.p, 0, new_bytes, ctx->stream);
            if (e1 == hipSuccess && old_bytes) e1 = hipMemcpyAsync(nb.p, ctx->st_tab[p][w].p, old_bytes, hipMemcpyDeviceToDevice, ctx->stream);
            if (e1 != hipSuccess) {
                retire_all();
                nb.release();
                return fail(ctx, BDX_E_DEVICE, "growing the statistics tables failed: %s", hipGetErrorString(e1));
            }
            retired.push_back(ctx->st_tab[p][w]);
            ctx->st_tab[p][w] = nb;  // row-major by key: the old table is a prefix of the new one
        }
    retire_all();
    ctx->st_rows = want;
    return BDX_OK;
}

namespace {

// statistics tables whose size is known from the config: the raw-score table (and the overflow flag)
int init_stats(bdx_ctx *ctx) {
    if (!ctx->dev.need_traceback) return BDX_OK;
    const bdx_config_t &c = ctx->cfg;
    const int npass = c.is_dual ? 2 : 1;
    double top = 0.0;  // the largest recordable numerator: floor(rate * normalisation) at the initial threshold
    for (int k = 0; k < npass; ++k)
        for (int b = 0; b < c.pass[k].n_barcodes; ++b) {
            const int m = (int)(c.pass[k].bc_off[b + 1] - c.pass[k].bc_off[b]);
            const double norm = (c.algorithm == BDX_ALG_SEMIGLOBAL && c.has_nindel) ? (double)c.pass[k].bc_len_no_N[b] : (double)m;
            const double ae = c.algorithm == BDX_ALG_EXACT ? 0.0 : std::floor(c.max_error_rate * norm);
            if (ae > top) top = ae;
        }
    if (top > (double)(1 << 20))
        return fail(ctx, BDX_E_INVALID, "summary statistics support scores up to %d; max_error_rate * barcode length gives %g", 1 << 20, top);
    ctx->st_raw_rows = (int)top + 1;
    // Clean-class configs (and :hamming / :exact): an alignment spans at most m columns plus its insertions (<= m),
    // leading deletions count through the origin: end - start + 1 <= 2 m — a fixed height.  Otherwise (start / end
    // ranges that bind: the band's seeded cells carry origins of their own) only start >= 1 - m and end <= n hold:
    // the table grows with the reads like the position table.
    // (:semiglobal with a ref_search_range that starts inside the read: an alignment out of the reference's initial column keeps the
    // origin 1 - i whatever the window's first column is, classification.jl:278-283 — its length end - start + 1 reaches n + m:
    // found by round 4's known-alignment tests, "a statistics key fell outside its table" on ref_search_range = "end-90:end")
    bool sg_window = false;
    if (c.algorithm == BDX_ALG_SEMIGLOBAL)
        for (int p = 0; p < npass; ++p) {
            const bdx_range_t &r = c.pass[p].ref_search_range;
            sg_window = sg_window || r.start_from_end || r.start_offset > 1 || c.pass[p].explicit_window != 0;
        }
    ctx->st_len_fixed = (ctx->plan.clean || ctx->plan.band_roll || c.algorithm != BDX_ALG_SEMIGLOBAL) && !sg_window;
    ctx->st_len_rows = ctx->st_len_fixed ? 2 * ctx->dev.max_m + 2 : 0;
    for (int p = 0; p < npass; ++p)
        for (int w = (ctx->st_len_fixed ? 1 : 2); w < 3; ++w) {
            const size_t bytes = bdx_stats_phys_words(ctx, p, w, 0) * 8;
            HIP_TRY(ctx, ctx->st_tab[p][w].ensure(bytes));
            HIP_TRY(ctx, hipMemsetAsync(ctx->st_tab[p][w].p, 0, bytes, ctx->stream));
        }
    HIP_TRY(ctx, ctx->st_flag.ensure(256));
    HIP_TRY(ctx, hipMemsetAsync(ctx->st_flag.p, 0, 256, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (ordered on the stream the kernels use, and done before create returns)
    return BDX_OK;
}

}  // namespace



// ---- host entry point: result vectors back to the caller -------------------------------------------------------------
// A device-to-host copy into PAGEABLE memory is staged by the runtime and, when the caller's arrays are fresh (the usual
// case: a result vector allocated per call), page-faulted in by that one copying thread: 13 of the 40 ms of a 10 M-read
// call.  Large downloads into pageable memory therefore go through a page-locked staging buffer of the context's own —
// one asynchronous DMA per vector at PCIe speed — and a few host threads copy each vector out (and fault the caller's
// pages in, in parallel) while the next one is still in flight.  Page-locked destinations (bdx_host_alloc) and small
// downloads keep the direct copies.  (Measured and dropped: populating the caller's pages with MADV_POPULATE_WRITE from a few
// threads while the reads go up — it slows the runtime's pageable upload down by more than the download gains: 293 -> 251 M reads/s.)
struct BackItem {
    void *h;
    const void *d;
    size_t bytes;
};

static bool host_is_page_locked(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (an unregistered pointer is an error for older runtimes: not sticky)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int download_items(bdx_ctx *ctx, const BackItem *items, int n_items) {
    size_t total = 0;
    const void *first = nullptr;
    for (int k = 0; k < n_items; ++k)
        if (items[k].h && items[k].d && items[k].bytes) {
            total += (items[k].bytes + 255) & ~(size_t)255;
            if (!first) first = items[k].h;
        }
    const bool staged = total >= ((size_t)16 << 20) && n_items <= 10 && !ctx->tune.no_staged_download && first && !host_is_page_locked(first);
    if (!staged) {
        for (int k = 0; k < n_items; ++k)
            if (items[k].h && items[k].d && items[k].bytes)
                HIP_TRY(ctx, hipMemcpyAsync(items[k].h, items[k].d, items[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return BDX_OK;
    }
    if (ctx->h_back_bytes < total) {
        if (ctx->h_back) (void)hipHostFree(ctx->h_back);
        ctx->h_back = nullptr;
        ctx->h_back_bytes = 0;
        if (hipHostMalloc(&ctx->h_back, total + (total >> 3), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();  // (page-locked memory is a limited resource: fall back to the direct copies)
            ctx->h_back = nullptr;
            for (int k = 0; k < n_items; ++k)
                if (items[k].h && items[k].d && items[k].bytes)
                    HIP_TRY(ctx, hipMemcpyAsync(items[k].h, items[k].d, items[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return BDX_OK;
        }
        ctx->h_back_bytes = total + (total >> 3);
    }
    if (!ctx->back_events_made) {
        for (hipEvent_t &e : ctx->back_events) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->back_events_made = true;
    }
    size_t offs[10];
    size_t off = 0;
    for (int k = 0; k < n_items; ++k) {
        offs[k] = off;
        if (!(items[k].h && items[k].d && items[k].bytes)) continue;
        HIP_TRY(ctx, hipMemcpyAsync((char *)ctx->h_back + off, items[k].d, items[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->back_events[k], ctx->stream));
        off += (items[k].bytes + 255) & ~(size_t)255;
    }
    int T = (int)std::thread::hardware_concurrency();
    T = T < 1 ? 1 : (T > 8 ? 8 : T);
    std::atomic<int> failed{0};
    std::atomic<int> next_slice[10];
    for (auto &x : next_slice) x.store(0);
    const int dev = ctx->device;
    // (every vector is cut into T slices claimed by whoever is there: with fewer helper threads than planned — the host is out
    // of threads — the caller simply copies more of them itself)
    auto work = [&](const bool helper) {
        if (helper && hipSetDevice(dev) != hipSuccess) {
            failed.store(1);
            return;
        }
        for (int k = 0; k < n_items; ++k) {
            if (!(items[k].h && items[k].d && items[k].bytes)) continue;
            if (hipEventSynchronize(ctx->back_events[k]) != hipSuccess) {
                failed.store(1);
                return;
            }
            const size_t per = ((items[k].bytes + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
            for (;;) {
                const int sl = next_slice[k].fetch_add(1);
                if (sl >= T) break;
                const size_t a = per * (size_t)sl, b = a + per < items[k].bytes ? a + per : items[k].bytes;
                if (a < b) memcpy((char *)items[k].h + a, (const char *)ctx->h_back + offs[k] + a, b - a);
            }
        }
    };
    {
        std::vector<std::thread> th;
        try {
            for (int t = 1; t < T; ++t) th.emplace_back(work, true);
        } catch (...) {  // (no more threads to be had: the ones that started and the caller share the slices)
        }
        work(false);
        for (auto &x : th) x.join();
    }
    if (failed.load()) return fail(ctx, BDX_E_DEVICE, "device-to-host copy of the results failed");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->staged_downloads += 1;
    return BDX_OK;
}


// ---- host entry point: shared tail (device outputs, launch, download) and the window upload ----------
static int run_and_download(bdx_ctx *ctx, const uint8_t *d_seq, const int64_t *d_off, int64_t n_reads, const bdx_outputs_t *out,
                            const bool mapped_outputs = false) {
    // int32 outputs: bc1 bc2 keep_start keep_end (n each), pass_start pass_end pass_raw pass_bc (2n each)
    const size_t n = (size_t)n_reads;
    if (mapped_outputs && n <= (size_t)(256 * 1024) && !out->pass_start && !out->pass_end && !out->pass_raw && !out->pass_bc &&
        !out->pass_score && !out->pass_delta) {
        // small batches: the kernels write the four verdict vectors straight into page-locked host memory (posted
        // writes over PCIe, 16 bytes per read) — no device-to-host copy call at all
        if (ctx->h_stage_bytes < n * 16) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->h_stage, n * 16 + 4096, hipHostMallocDefault));
            ctx->h_stage_bytes = n * 16 + 4096;
        }
        void *hs_dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&hs_dev, ctx->h_stage, 0));
        int32_t *bm = (int32_t *)hs_dev;
        bdx_outputs_t dm{};
        dm.bc1 = bm;
        dm.bc2 = out->bc2 ? bm + n : nullptr;
        dm.keep_start = out->keep_start ? bm + 2 * n : nullptr;
        dm.keep_end = out->keep_end ? bm + 3 * n : nullptr;
        int rcm = bdx_classify_device(ctx, d_seq, d_off, n_reads, &dm);
        if (rcm != BDX_OK) return rcm;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int32_t *hs = (const int32_t *)ctx->h_stage;
        if (out->bc1) memcpy(out->bc1, hs, n * 4);
        if (out->bc2) memcpy(out->bc2, hs + n, n * 4);
        if (out->keep_start) memcpy(out->keep_start, hs + 2 * n, n * 4);
        if (out->keep_end) memcpy(out->keep_end, hs + 3 * n, n * 4);
        return BDX_OK;
    }
    HIP_TRY(ctx, ctx->d_out_i32.ensure(n * 4 * 12));
    HIP_TRY(ctx, ctx->d_out_f64.ensure(n * 8 * 4));
    if (ctx->tune.poison) {  // test switch: an output element no kernel writes comes back as garbage, never as a stale right answer
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_out_i32.p, 0xA5, n * 4 * 12, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_out_f64.p, 0xA5, n * 8 * 4, ctx->stream));
    }
    int32_t *bi = (int32_t *)ctx->d_out_i32.p;
    bdx_outputs_t d{};
    d.bc1 = bi;
    d.bc2 = out->bc2 ? bi + n : nullptr;
    d.keep_start = out->keep_start ? bi + 2 * n : nullptr;
    d.keep_end = out->keep_end ? bi + 3 * n : nullptr;
    d.pass_start = out->pass_start ? bi + 4 * n : nullptr;
    d.pass_end = out->pass_end ? bi + 6 * n : nullptr;
    d.pass_raw = out->pass_raw ? bi + 8 * n : nullptr;
    d.pass_bc = out->pass_bc ? bi + 10 * n : nullptr;
    d.pass_score = out->pass_score ? (double *)ctx->d_out_f64.p : nullptr;
    d.pass_delta = out->pass_delta ? (double *)ctx->d_out_f64.p + 2 * n : nullptr;
    int rc = bdx_classify_device(ctx, d_seq, d_off, n_reads, &d);
    if (rc != BDX_OK) return rc;
    // Small batches (the reference hands over chunks of 4000 reads, core.jl:5-10): the four verdict vectors sit side
    // by side on the device — ONE copy into a page-locked staging buffer and four host memcpys instead of four
    // pageable copies with their fixed cost each.
    const bool one_copy = n <= (size_t)(256 * 1024) && !out->pass_start && !out->pass_end && !out->pass_raw && !out->pass_bc &&
                          !out->pass_score && !out->pass_delta;
    if (one_copy) {
        if (ctx->h_stage_bytes < n * 16) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->h_stage, n * 16 + 4096, hipHostMallocDefault));
            ctx->h_stage_bytes = n * 16 + 4096;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_stage, bi, n * 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int32_t *hs = (const int32_t *)ctx->h_stage;
        if (out->bc1) memcpy(out->bc1, hs, n * 4);
        if (out->bc2) memcpy(out->bc2, hs + n, n * 4);
        if (out->keep_start) memcpy(out->keep_start, hs + 2 * n, n * 4);
        if (out->keep_end) memcpy(out->keep_end, hs + 3 * n, n * 4);
        return BDX_OK;
    }
    const BackItem items[10] = {{out->bc1, d.bc1, n * 4},           {out->bc2, d.bc2, n * 4},           {out->keep_start, d.keep_start, n * 4},
                                {out->keep_end, d.keep_end, n * 4}, {out->pass_start, d.pass_start, n * 8}, {out->pass_end, d.pass_end, n * 8},
                                {out->pass_raw, d.pass_raw, n * 8}, {out->pass_bc, d.pass_bc, n * 8},     {out->pass_score, d.pass_score, n * 16},
                                {out->pass_delta, d.pass_delta, n * 16}};
    return download_items(ctx, items, 10);
}

// Large batches through the host entry point: the reads go up in a few chunks on a copy stream of the context's own
// while the kernels of the previous chunk run (one classify call per chunk on the context's stream, tied to its copy
// by an event); the verdict vectors come back once at the end.  With pageable host memory hipMemcpyAsync returns when
// the chunk is staged, so the launch of chunk i's kernels falls exactly between the copies of chunks i and i + 1.
static int classify_host_pipelined(bdx_ctx *ctx, const uint8_t *seq_bytes, const int64_t *seq_off, int64_t n_reads,
                                   const bdx_outputs_t *out, int n_chunks) {
    const size_t n = (size_t)n_reads;
    const int64_t base = seq_off[0];
    const int64_t total = seq_off[n_reads] - base;
    if (!ctx->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->copy_events) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_TRY(ctx, ctx->d_seq.ensure((size_t)total + 64));
    HIP_TRY(ctx, ctx->d_off.ensure((n + 1) * 8));
    HIP_TRY(ctx, ctx->d_out_i32.ensure(n * 4 * 12));
    HIP_TRY(ctx, ctx->d_out_f64.ensure(n * 8 * 4));
    int32_t *bi = (int32_t *)ctx->d_out_i32.p;
    double *bf = (double *)ctx->d_out_f64.p;
    // (the longest read is known: bdx_classify_host has scanned the offsets — a device-side measurement per chunk would
    // synchronise the stream and undo the overlap)
    struct Reset {
        bdx_ctx *c;
        ~Reset() { c->host_maxlen = 0; }
    } reset{ctx};
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off.p, seq_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->copy_stream));
    const uint8_t *d_seq = (const uint8_t *)ctx->d_seq.p - base;
    for (int c = 0; c < n_chunks; ++c) {
        // (earlier chunks take the remainder: no work buffer has to grow while kernels run)
        const size_t r0 = n / n_chunks * c + ((size_t)c < n % n_chunks ? c : n % n_chunks);
        const size_t r1 = r0 + n / n_chunks + ((size_t)c < n % n_chunks ? 1 : 0);
        const int64_t b0 = seq_off[r0], b1 = seq_off[r1];
        if (b1 > b0)
            HIP_TRY(ctx, hipMemcpyAsync((uint8_t *)ctx->d_seq.p + (b0 - base), seq_bytes + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(ctx, hipEventRecord(ctx->copy_events[c], ctx->copy_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_events[c], 0));
        bdx_outputs_t d{};
        d.bc1 = bi + r0;
        d.bc2 = out->bc2 ? bi + n + r0 : nullptr;
        d.keep_start = out->keep_start ? bi + 2 * n + r0 : nullptr;
        d.keep_end = out->keep_end ? bi + 3 * n + r0 : nullptr;
        d.pass_start = out->pass_start ? bi + 4 * n + 2 * r0 : nullptr;
        d.pass_end = out->pass_end ? bi + 6 * n + 2 * r0 : nullptr;
        d.pass_raw = out->pass_raw ? bi + 8 * n + 2 * r0 : nullptr;
        d.pass_bc = out->pass_bc ? bi + 10 * n + 2 * r0 : nullptr;
        d.pass_score = out->pass_score ? bf + 2 * r0 : nullptr;
        d.pass_delta = out->pass_delta ? bf + 2 * n + 2 * r0 : nullptr;
        const int rc = bdx_classify_device(ctx, d_seq, (const int64_t *)ctx->d_off.p + r0, (int64_t)(r1 - r0), &d);
        if (rc != BDX_OK) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            return rc;
        }
    }
    const BackItem items[10] = {{out->bc1, bi, n * 4},
                                {out->bc2, out->bc2 ? bi + n : nullptr, n * 4},
                                {out->keep_start, out->keep_start ? bi + 2 * n : nullptr, n * 4},
                                {out->keep_end, out->keep_end ? bi + 3 * n : nullptr, n * 4},
                                {out->pass_start, out->pass_start ? bi + 4 * n : nullptr, n * 8},
                                {out->pass_end, out->pass_end ? bi + 6 * n : nullptr, n * 8},
                                {out->pass_raw, out->pass_raw ? bi + 8 * n : nullptr, n * 8},
                                {out->pass_bc, out->pass_bc ? bi + 10 * n : nullptr, n * 8},
                                {out->pass_score, out->pass_score ? (const void *)bf : nullptr, n * 16},
                                {out->pass_delta, out->pass_delta ? (const void *)(bf + 2 * n) : nullptr, n * 16}};
    const int rcd = download_items(ctx, items, 10);
    if (rcd != BDX_OK) return rcd;
    ctx->pipelined_calls += 1;
    return BDX_OK;
}

// Host mirror of the device's per-read window arithmetic (bdx_core.h resolve_range / pass_window and the per-read
// setup of bdx_bitpar.hip): the 0-based half-open byte range [ulo, uhi) of a read of n code units that ANY pass may
// touch — final_search_range first:last per pass (classification.jl:795-809), + max_m - 1 beyond the last start
// position for :hamming / :exact.
// Longest read and monotonicity of a host offset vector (threads for large batches: the pass is memory-bound).
static bool scan_offsets(const int64_t *seq_off, int64_t n_reads, int64_t &mx_out, bool &monotone_out) {
    const int nt = n_reads > 262144 ? 8 : 1;
    std::vector<int64_t> mx((size_t)nt, 0);
    std::vector<char> bad((size_t)nt, 0);
    auto work = [&](int t) {
        const int64_t lo = n_reads * t / nt, hi = n_reads * (t + 1) / nt;
        int64_t m = 0;
        bool neg = false;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t d = seq_off[i + 1] - seq_off[i];
            neg |= d < 0;
            m = d > m ? d : m;
        }
        mx[(size_t)t] = m;
        bad[(size_t)t] = neg ? 1 : 0;
    };
    try {
        if (nt == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
    } catch (...) {
        return false;
    }
    mx_out = 0;
    monotone_out = true;
    for (int t = 0; t < nt; ++t) {
        mx_out = mx[(size_t)t] > mx_out ? mx[(size_t)t] : mx_out;
        monotone_out = monotone_out && !bad[(size_t)t];
    }
    return true;
}

static void host_union_window(const BdxDevCfg &cfg, long long n_ll, long long &ulo, long long &uhi) {
    const long long n = n_ll > (1LL << 30) ? (1LL << 30) : n_ll;
    const auto resolve = [&](const BdxDevRange &dr, long long &first, long long &last) {
        const long long s = dr.start_from_end ? n + dr.start_offset : dr.start_offset;
        const long long e = dr.end_from_end ? n + dr.end_offset : dr.end_offset;
        const long long a = s > 1 ? s : 1;
        long long b = e < n ? e : n;
        if (b < a) b = a - 1;
        first = a;
        last = b;
    };
    ulo = (1LL << 40);
    uhi = 0;
    const bool sgm = cfg.algorithm == BDX_ALG_SEMIGLOBAL;
    for (int p = 0; p < (cfg.is_dual ? 2 : 1); ++p) {
        const BdxDevPass &P = cfg.pass[p];
        long long first, last;
        bool ok = true;
        if (P.explicit_window) {
            first = P.win_first > 1 ? P.win_first : 1;
            last = P.win_last < n ? P.win_last : n;
        } else {
            long long rf, rl, bf, bl, ef, el;
            resolve(P.ref_search, rf, rl);
            resolve(P.bc_start, bf, bl);
            resolve(P.bc_end, ef, el);
            first = rf > bf ? rf : bf;
            if (first < 1) first = 1;
            last = rl < el ? rl : el;
            if (n < last) last = n;
            if (first > last || first > bl || last < ef) ok = false;  // :805-807
        }
        const long long f = ok ? (first > 1 ? first : 1) : 1;
        const long long l = ok ? (last < n ? last : n) : 0;
        if (l >= f) {
            long long h = sgm ? l : l + cfg.max_m - 1;
            if (h > n) h = n;
            if (f - 1 < ulo) ulo = f - 1;
            if (h > uhi) uhi = h;
        }
    }
    if (uhi <= ulo) ulo = uhi = 0;
}

// 1: not worth it (the caller uploads the whole reads); 0: classified through the window upload; < 0: error
static int classify_host_windows(bdx_ctx *ctx, const uint8_t *seq_bytes, const int64_t *seq_off, int64_t n_reads,
                                 const bdx_outputs_t *out) {
    if (ctx->tune.no_window_upload) return 1;
    const int64_t total = seq_off[n_reads] - seq_off[0];
    if (total < (int64_t)n_reads * 512) return 1;  // short reads: nothing to save
    const size_t n = (size_t)n_reads;
    ctx->h_coff.resize(n + 1);
    ctx->h_vlen.resize(n);
    ctx->h_vlo.resize(n);
    // two passes over the reads, both on a few host threads (the gather touches one cache line or two of every
    // 10 kbp read: latency-bound on one core): windows, a serial prefix sum of their sizes, gather
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthr = n < 65536 ? 1 : (hw >= 8 ? 8 : (hw >= 2 ? hw : 1));
    const auto parallel = [&](const auto &body) {
        if (nthr == 1) {
            body((size_t)0, n, (size_t)0);
            return;
        }
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthr; ++t) th.emplace_back([&, t]() { body(n * t / nthr, n * (t + 1) / nthr, t); });
        for (std::thread &x : th) x.join();
    };
    std::vector<long long> t_max(nthr, 0);
    std::vector<int> t_bad(nthr, 0);
    parallel([&](const size_t i0, const size_t i1, const size_t t) {
        long long mx = 0;
        for (size_t i = i0; i < i1; ++i) {
            const long long len = seq_off[i + 1] - seq_off[i];
            if (len < 0) {
                t_bad[t] = 1;
                return;
            }
            long long ulo, uhi;
            host_union_window(ctx->dev, len, ulo, uhi);
            ctx->h_coff[i + 1] = uhi - ulo;  // (sizes now, offsets after the prefix sum)
            ctx->h_vlen[i] = (int32_t)(len > (1LL << 30) ? (1LL << 30) : len);
            ctx->h_vlo[i] = (int32_t)ulo;
            if (len > mx) mx = len;
        }
        t_max[t] = mx;
    });
    long long maxlen = 0;
    for (size_t t = 0; t < nthr; ++t) {
        if (t_bad[t]) return fail(ctx, BDX_E_INVALID, "seq_off is not non-decreasing");
        if (t_max[t] > maxlen) maxlen = t_max[t];
    }
    ctx->h_coff[0] = 0;
    for (size_t i = 0; i < n; ++i) ctx->h_coff[i + 1] += ctx->h_coff[i];
    const int64_t wbytes = ctx->h_coff[n];
    if (wbytes * 2 + (int64_t)n_reads * 16 > total) return 1;  // the windows are most of the reads anyway
    ctx->h_win.resize((size_t)wbytes + 64);
    parallel([&](const size_t i0, const size_t i1, const size_t) {
        for (size_t i = i0; i < i1; ++i) {
            const int64_t len_w = ctx->h_coff[i + 1] - ctx->h_coff[i];
            if (len_w > 0) memcpy(ctx->h_win.data() + ctx->h_coff[i], seq_bytes + seq_off[i] + ctx->h_vlo[i], (size_t)len_w);
        }
    });
    HIP_TRY(ctx, ctx->d_seq.ensure((size_t)wbytes + 64));
    HIP_TRY(ctx, ctx->d_off.ensure((n + 1) * 8));
    HIP_TRY(ctx, ctx->d_vlen.ensure(n * 4));
    HIP_TRY(ctx, ctx->d_vlo.ensure(n * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_seq.p, ctx->h_win.data(), (size_t)wbytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off.p, ctx->h_coff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_vlen.p, ctx->h_vlen.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_vlo.p, ctx->h_vlo.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->dev.vlen = (const int32_t *)ctx->d_vlen.p;
    ctx->dev.vlo = (const int32_t *)ctx->d_vlo.p;
    ctx->virt_maxlen = (int)(maxlen > (1LL << 30) ? (1LL << 30) : (maxlen < 1 ? 1 : maxlen));
    const int rc = run_and_download(ctx, (const uint8_t *)ctx->d_seq.p, (const int64_t *)ctx->d_off.p, n_reads, out);
    ctx->dev.vlen = nullptr;  // (run_and_download has synchronised the stream)
    ctx->dev.vlo = nullptr;
    ctx->virt_maxlen = 0;
    for (BdxFilterSet &f : ctx->fs) f.bplan.read_len_hint = 0;  // the slot geometry was forced: plan afresh for ordinary batches
    if (rc == BDX_OK) ctx->window_uploads += 1;
    return rc == BDX_OK ? 0 : rc;
}

extern "C" {

int32_t bdx_abi_version(void) { return BDX_ABI_VERSION; }

const char *bdx_last_error(const bdx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t bdx_create(const bdx_config_t *config, bdx_ctx **out) {
    if (!out) return fail(nullptr, BDX_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate(config);
    if (rc != BDX_OK) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, BDX_E_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (config->device < 0 || config->device >= ndev)
        return fail(nullptr, BDX_E_INVALID, "device ordinal %d out of range (0..%d)", config->device, ndev - 1);
    bdx_ctx *ctx = new (std::nothrow) bdx_ctx();
    if (!ctx) return fail(nullptr, BDX_E_DEVICE, "out of host memory");
    ctx->cfg = *config;
    ctx->device = config->device;
    ctx->tune = read_tuning();  // the environment is consulted here and nowhere else
    auto bail = [&](int code) {
        g_create_error = ctx->err;
        bdx_destroy(ctx);
        return code;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) {
        ctx->err = "hipSetDevice failed";
        return bail(BDX_E_DEVICE);
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        ctx->err = "hipStreamCreate failed";
        return bail(BDX_E_DEVICE);
    }
    ctx->stream = ctx->own_stream;
    {
        // the device's shape comes from the device (a partitioned part has fewer compute units than a full MI355X)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
        ctx->n_cu = ctx->tune.cu_count > 0 ? ctx->tune.cu_count : cus;
    }
    rc = upload_tables(ctx);
    if (rc != BDX_OK) return bail(rc);
    if (ctx->d_dbg.ensure(256) != hipSuccess || hipMemset(ctx->d_dbg.p, 0, 256) != hipSuccess) {
        ctx->err = "hipMalloc failed";
        return bail(BDX_E_DEVICE);
    }
    ctx->dev.dbg_rejected = (unsigned int *)ctx->d_dbg.p;
    rc = plan_generic(ctx);  // needs the caller's host tables: run before they are dropped
    if (rc != BDX_OK) return bail(rc);
    ctx->plan.n_cu = ctx->n_cu;
    rc = init_stats(ctx);
    if (rc != BDX_OK) return bail(rc);
    rc = build_bitpar_tables(ctx);
    if (rc != BDX_OK) return bail(rc);
    if (ctx->plan.band_roll && !ctx->F().bplan.enabled) {  // no filter, no hand-over windows: the rolling band would walk whole windows
        ctx->band_roll_off = true;
        rc = plan_generic(ctx);
        if (rc != BDX_OK) return bail(rc);
        ctx->plan.n_cu = ctx->n_cu;
    }
    rc = build_seed_tables(ctx, true);
    if (rc != BDX_OK) return bail(rc);
    rc = build_diag_tables(ctx);
    if (rc != BDX_OK) return bail(rc);
    if (!ctx->F().splan.enabled) {  // neither: moderately selective single seeds still beat sweeping every pair
        rc = build_seed_tables(ctx, false);
        if (rc != BDX_OK) return bail(rc);
    } else if (ctx->F().splan.diag) {  // the fallback of size_bitpar when the index does not fit a batch
        rc = build_seed_tables(ctx, false, true);
        if (rc != BDX_OK) return bail(rc);
    }
    rc = build_wave_tables(ctx);
    if (rc != BDX_OK) return bail(rc);
    rc = build_pair_tables(ctx);
    if (rc != BDX_OK) return bail(rc);
    // tier 1 (capped budgets, strict single seeds) beside a full-budget set that is NOT already strict single seeds
    {
        const BdxFilterSet &full = ctx->fs[0];
        const bool strict_full = full.splan.enabled && !full.splan.diag && full.splan.q >= 7;
        // (the unit-level API's hand-made windows stay on the plain path)
        bool plain_windows = true;
        for (int k = 0; k < (config->is_dual ? 2 : 1); ++k) {
            plain_windows = plain_windows && config->pass[k].explicit_window == 0;
            // N-scoring with real wildcards: position-dependent indel costs — "an alignment's result does not depend on
            // the running threshold" is only argued (and fuzzed) for uniform costs; such configs stay on one tier
            if (config->algorithm == BDX_ALG_SEMIGLOBAL && config->has_nindel)
                for (uint32_t i = 0; i < config->pass[k].bc_off[config->pass[k].n_barcodes]; ++i)
                    if (config->pass[k].bc_bytes[i] == 'N') plain_windows = false;
        }
        if (full.bplan.enabled && plain_windows && !strict_full && !ctx->tune.no_tier && config->filter == BDX_FILTER_AUTO) {
            // Piece length behind tier 1's capped budgets, cap(m) = m / q - 1: 8-base seeds are the most selective; 7- or
            // 6-base pieces raise the cap of some lengths by one (14-15 and 21-23 bases with q = 7, 12-13 with q = 6), so
            // that tier 1 settles reads with one more error and tier 0 — a plain sweep or the two-intact-pieces kernel —
            // sees far fewer reads (m = 14, B = 96, rate 0.2: 444 -> 724 M reads/s), as long as the chance hits per
            // read stay few and no cap goes beyond 2 (measured: caps of 3 — m = 24 with q = 6, m = 28 with q = 7 — cost
            // more in tier 1 than they save in tier 0, at 24 and at 96 barcodes).
            {
                int cmin = 1;
                if (config->algorithm == BDX_ALG_SEMIGLOBAL) {
                    cmin = config->mismatch < config->indel ? config->mismatch : config->indel;
                    if (config->has_nindel && config->nindel < cmin) cmin = config->nindel;
                    if (cmin < 1) cmin = 1;
                }
                long long best_caps = -1;
                int best_q = 8;
                const double limit[9] = {0, 0, 0, 0, 0, 0, 8.0, 4.0, 1e30};
                for (int q = 8; q >= 6; --q) {
                    long long caps = 0, pieces = 0, cap_max = 0;
                    for (int k = 0; k < (config->is_dual ? 2 : 1); ++k)
                        for (int b = 0; b < config->pass[k].n_barcodes; ++b) {
                            const int m = (int)(config->pass[k].bc_off[b + 1] - config->pass[k].bc_off[b]);
                            const double norm = (config->algorithm == BDX_ALG_SEMIGLOBAL && config->has_nindel) ? (double)config->pass[k].bc_len_no_N[b] : (double)m;
                            const long long ae = config->algorithm == BDX_ALG_EXACT ? 0 : (long long)std::floor(config->max_error_rate * norm);
                            if (ae < 0) continue;
                            long long cap = m / q - 1;
                            if (cap < 0) cap = 0;
                            if (cap > ae / cmin) cap = ae / cmin;
                            caps += cap;
                            pieces += cap + 1;
                            if (cap > cap_max) cap_max = cap;
                        }
                    const double chance = 150.0 * (double)pieces / std::pow(4.0, (double)q);
                    // a cap lifted from 0 to 1 pays at once, 1 -> 2 a little, 2 -> 3 never did (measured, B = 24 and 96)
                    if (q < 8 && cap_max > (q == 6 ? 1 : 2)) continue;
                    if (chance <= limit[q] && caps > best_caps) {
                        best_caps = caps;
                        best_q = q;
                    }
                }
                ctx->tier_q = best_q;
            }
            if (ctx->tune.tier_q >= 5 && ctx->tune.tier_q <= 8) ctx->tier_q = ctx->tune.tier_q;
            ctx->cur = 1;
            rc = build_bitpar_tables(ctx);
            if (rc == BDX_OK && ctx->fs[1].bplan.enabled && ctx->fs[1].bplan.tier_capped) {
                rc = build_seed_tables(ctx, true);
                // very many barcodes: moderately selective 8-base seeds (a dozen chance pairs per read) still beat
                // the full-budget filter by far
                if (rc == BDX_OK && !ctx->fs[1].splan.enabled) rc = build_seed_tables(ctx, false);
                if (rc == BDX_OK && ctx->fs[1].splan.enabled && ctx->fs[1].splan.q < ctx->tier_q) ctx->fs[1].splan.enabled = 0;
                if (rc == BDX_OK) rc = build_wave_tables(ctx);
            }
            ctx->cur = 0;
            if (rc != BDX_OK) return bail(rc);
            ctx->tiered = ctx->fs[1].bplan.enabled && ctx->fs[1].bplan.tier_capped && ctx->fs[1].splan.enabled;
            // with_delta and a min_delta beyond the score of an unseen barcode: not even a perfect match can be
            // proven unambiguous at the capped budgets (only a visible runner-up could settle a read) — tier 1
            // would be a pass over the whole batch for next to nothing
            if (ctx->tiered && config->min_delta != 0.0)
                for (int k = 0; k < (config->is_dual ? 2 : 1); ++k)
                    if (!(ctx->fs[1].bplan.tier_slo[k] >= config->min_delta)) ctx->tiered = 0;
        }
    }
    // The PAIRS TIER: a split config with min_delta whose seed tier proves nothing (above), mismatch = cmin = 1 and indels dearer —
    // the reference's demo2 options (mismatch 1, indel 2, rate 0.25, min_delta 0.15): tier 1 = the same-diagonal pairs mode with
    // six 4-base pieces over the WHOLE batch at budgets capped at 4 (3) operations — ~3 chance flags per read instead of the ~90 of
    // the full-budget variant — followed by the exact kernel, which settles every read whose winner leaves min_delta of room below
    // slo = (cap + 1) / m (a perfect match or one mismatch under demo2's options: ~75 % of the reads) and lists the rest for tier 0.
    if (!ctx->tiered && ctx->fs[0].bplan.enabled && ctx->fs[0].pplan.enabled && ctx->fs[0].pplan.split && !ctx->tune.no_tier && !ctx->tune.no_pairs &&
        config->filter == BDX_FILTER_AUTO && config->algorithm == BDX_ALG_SEMIGLOBAL && !config->has_nindel && config->mismatch == 1 &&
        config->indel >= 2 && config->match == 0 && config->min_delta != 0.0) {
        bool plain = true;
        for (int k = 0; k < (config->is_dual ? 2 : 1); ++k) plain = plain && config->pass[k].explicit_window == 0;
        for (int cap = 4; cap >= 3 && plain && !ctx->tiered; --cap) {
            ctx->tier_cap_fixed = cap;
            ctx->cur = 1;
            ctx->fs[1].splan = BdxSeedPlan{};
            ctx->fs[1].wplan = BdxWavePlan{};
            rc = build_bitpar_tables(ctx);
            bool ok = rc == BDX_OK && ctx->fs[1].bplan.enabled && ctx->fs[1].bplan.tier_capped;
            for (int k = 0; ok && k < (config->is_dual ? 2 : 1); ++k) ok = ctx->fs[1].bplan.tier_slo[k] >= config->min_delta;
            if (ok) rc = build_pair_tables(ctx);
            ok = ok && rc == BDX_OK && ctx->fs[1].pplan.enabled && ctx->fs[1].pplan.pairs_kb == 8 && ctx->fs[1].pplan.split;
            ctx->cur = 0;
            if (rc != BDX_OK) return bail(rc);
            if (ok) {
                ctx->tiered = 1;
                ctx->pairs_tier = 1;
            } else {
                ctx->tier_cap_fixed = -1;
                ctx->fs[1].bplan.enabled = 0;
            }
        }
    }
    ctx->path = ctx->F().bplan.enabled ? (ctx->F().splan.enabled ? (ctx->F().splan.diag ? "qgram2+bitpar+verify" : "qgram+bitpar+verify") : "bitpar+verify") : "generic";
    if (ctx->tiered) ctx->path = "tier1:qgram+bitpar > " + ctx->path;
    ctx->filter_used = ctx->F().bplan.enabled ? (ctx->F().splan.enabled ? BDX_FILTER_QGRAM : BDX_FILTER_BITPAR) : BDX_FILTER_OFF;
    if (ctx->F().bplan.enabled) {
        if (ctx->d_maxlen.ensure(1024) != hipSuccess) {
            ctx->err = "hipMalloc failed";
            return bail(BDX_E_DEVICE);
        }
    }
    // the copied config must not keep pointing at caller memory
    for (int k = 0; k < 2; ++k) {
        ctx->cfg.pass[k].bc_bytes = nullptr;
        ctx->cfg.pass[k].bc_off = nullptr;
        ctx->cfg.pass[k].bc_len_no_N = nullptr;
    }
    if (bdx_generic_set_lds_limit(LDS_MAX) != hipSuccess) {
        ctx->err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        return bail(BDX_E_DEVICE);
    }
    *out = ctx;
    return BDX_OK;
}

void bdx_destroy(bdx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->own_stream);
        (void)hipStreamDestroy(ctx->own_stream);
    }
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        for (hipEvent_t &e : ctx->copy_events)
            if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->d_dbg.p) {
        unsigned int rej[2] = {0, 0};
        if (hipMemcpy(rej, ctx->d_dbg.p, sizeof rej, hipMemcpyDeviceToHost) == hipSuccess) g_rejected_windows += (long long)rej[0] + rej[1];
    }
    ctx->d_dbg.release();
    for (int k = 0; k < 2; ++k) {
        ctx->bc_bytes[k].release();
        ctx->bc_off[k].release();
        ctx->bc_nn[k].release();
        ctx->d_cand[k].release();
        ctx->d_wins[k].release();
        ctx->d_wcnt[k].release();
    }
    ctx->counts_own.release();
    for (BdxFilterSet &f : ctx->fs) {
        f.bp_tables.release();
        f.seed_tables.release();
        f.seed_tables_alt.release();
        f.wave_tables.release();
        f.pair_tables.release();
    }
    ctx->d_wlist.release();
    ctx->d_tier.release();
    ctx->d_carry.release();
    ctx->d_maxlen.release();
    ctx->d_exc.release();
    ctx->d_seq.release();
    ctx->d_off.release();
    ctx->d_out_i32.release();
    ctx->d_out_f64.release();
    ctx->d_vlen.release();
    ctx->d_vlo.release();
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    ctx->h_in = nullptr;
    if (ctx->h_back) (void)hipHostFree(ctx->h_back);
    ctx->h_back = nullptr;
    if (ctx->back_events_made)
        for (hipEvent_t &e : ctx->back_events)
            if (e) (void)hipEventDestroy(e);
    ctx->back_events_made = false;
    bdx_comm_release(ctx);
    ctx->counts_sum.release();
    for (int p = 0; p < 2; ++p)
        for (int w = 0; w < 3; ++w) {
            ctx->st_tab[p][w].release();
            ctx->st_sum[p][w].release();
        }
    ctx->st_flag.release();
    delete ctx;
}

int32_t bdx_set_stream(bdx_ctx *ctx, void *hip_stream) {
    if (!ctx) return BDX_E_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return BDX_OK;
}

int32_t bdx_set_read_length_hint(bdx_ctx *ctx, int32_t typical_read_length) {
    if (!ctx) return BDX_E_INVALID;
    ctx->user_len_hint = typical_read_length > 0 ? typical_read_length : 0;
    return BDX_OK;
}

int32_t bdx_sync(bdx_ctx *ctx) {
    if (!ctx) return BDX_E_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDX_OK;
}

int32_t bdx_classify_device(bdx_ctx *ctx, const uint8_t *d_seq_bytes, const int64_t *d_seq_off, int64_t n_reads,
                            const bdx_outputs_t *d_out) {
    if (!ctx) return BDX_E_INVALID;
    if (n_reads < 0) return fail(ctx, BDX_E_INVALID, "n_reads is negative");
    if (n_reads == 0) return BDX_OK;
    if (!d_seq_bytes || !d_seq_off || !d_out) return fail(ctx, BDX_E_INVALID, "NULL device pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    BdxDevOut o;
    o.bc1 = d_out->bc1;
    o.bc2 = d_out->bc2;
    o.keep_start = d_out->keep_start;
    o.keep_end = d_out->keep_end;
    o.pass_start = d_out->pass_start;
    o.pass_end = d_out->pass_end;
    o.pass_raw = d_out->pass_raw;
    o.pass_score = d_out->pass_score;
    o.pass_bc = d_out->pass_bc;
    o.pass_delta = d_out->pass_delta;
    BdxDevStats st{};
    const BdxDevStats *stp = nullptr;
    int measured_len = -1;
    if (ctx->virt_maxlen > 0) {
        measured_len = ctx->virt_maxlen;  // window upload: the host has seen every length
    } else if (ctx->host_maxlen > 0) {
        measured_len = ctx->host_maxlen;  // host entry point: the offsets were on the host anyway
    } else if (ctx->dev.need_traceback) {
        // statistics tables are sized from the batch's true maximum read length (a hint is only a hint)
        HIP_TRY(ctx, ctx->d_maxlen.ensure(1024));
        HIP_TRY(ctx, bdx_launch_maxlen((const long long *)d_seq_off, n_reads, (int *)ctx->d_maxlen.p, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&measured_len, ctx->d_maxlen.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (ctx->dev.need_traceback) {
        int rc = bdx_stats_reserve(ctx, (long long)measured_len + ctx->dev.max_m + 2);
        if (rc != BDX_OK) return rc;
        for (int p = 0; p < 2; ++p) {
            st.pos[p] = (unsigned long long *)ctx->st_tab[p][0].p;
            st.len[p] = (unsigned long long *)ctx->st_tab[p][1].p;
            st.raw[p] = (unsigned long long *)ctx->st_tab[p][2].p;
        }
        st.rows = ctx->st_rows;
        st.raw_rows = ctx->st_raw_rows;
        st.len_rows = ctx->st_len_rows;
        st.len_stride = bdx_stats_stride(ctx, 1);
        st.raw_stride = bdx_stats_stride(ctx, 2);
        st.pos_bias = ctx->dev.max_m;
        st.overflow = (unsigned int *)ctx->st_flag.p;
        stp = &st;
    }
    bool filtered = false;
    int tier_len = 0;  // > 0: tiered budgets apply to this batch (the read length both tiers were planned for)
    int batch_len = 0;  // the read length the filtered launches were planned for
    if (ctx->F().bplan.enabled) {
        int len = ctx->virt_maxlen > 0 ? ctx->virt_maxlen : ctx->user_len_hint;
        if (len <= 0 && measured_len >= 0) len = measured_len > 0 ? measured_len : 1;
        if (len <= 0) {  // measure the batch: one tiny kernel + a 4-byte copy
            int host_len = 0;
            HIP_TRY(ctx, bdx_launch_maxlen((const long long *)d_seq_off, n_reads, (int *)ctx->d_maxlen.p, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&host_len, ctx->d_maxlen.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            len = host_len;
        }
        batch_len = len;
        filtered = size_bitpar(ctx, len, n_reads);
        if (filtered && ctx->tiered) {  // both tiers must be plannable for this batch, else the full budget alone
            ctx->cur = 1;
            const bool ok1 = size_bitpar(ctx, len, n_reads);
            ctx->cur = 0;
            tier_len = ok1 ? len : 0;
        }
    }
    if (filtered) {
        // The fused kernel filters; the exact DP runs at full width in the generic kernel:
        //  * split (trimming / summary / weighted costs / N-scoring / Hamming / exact): every read's
        //    candidate mask (+ column windows) goes through HBM, the generic kernel gives every verdict;
        //  * known-score configs: the fused kernel also gives the verdict of (nearly) every read by
        //    replaying the reducer; the few it cannot settle are listed and evaluated by the generic
        //    kernel in list mode;
        //  * tiered budgets (known-score configs whose full budget is too large for selective single seeds):
        //    tier 1 — capped budgets, single seeds — runs over the whole batch and settles every read whose
        //    verdict cannot depend on a barcode beyond the cap; tier 0 — the full budget — then runs in list
        //    mode over the rest.
        const int npass = ctx->dev.is_dual ? 2 : 1;
        bool split = false;
        for (int k = 0; k < npass; ++k) split |= !ctx->F().bplan.known_ok[k];
        // (:exact returns the occurrence's start and end whatever the output policy: a caller that wants them gets the launch in
        // its split form — known-alignment class first, exact kernel for what that lists)
        if (ctx->dev.algorithm == BDX_ALG_EXACT && (o.pass_start != nullptr || o.pass_end != nullptr)) split = true;
        const bool tiered = tier_len > 0;
        if (n_reads > 0xFFFFFFF0LL) return fail(ctx, BDX_E_INVALID, "more than 2^32 reads in one batch");
        uint32_t *c0 = nullptr, *c1 = nullptr, *w0 = nullptr, *w1 = nullptr;
        uint8_t *n0 = nullptr, *n1 = nullptr;
        const bool windows = split && !ctx->tune.no_windows;
        // dense window table of the plain-sweep kernel (few barcodes, many genuine candidates per read; columns fit 16 bits)
        const bool dense_w = windows && ctx->fs[0].bplan.dense_d && !ctx->fs[0].splan.enabled && batch_len <= 60000 && !ctx->tune.no_dense;
        for (int k = 0; k < npass; ++k) {
            HIP_TRY(ctx, ctx->d_cand[k].ensure((size_t)n_reads * ctx->dev.pass[k].cand_words * 4 + 64));
            if (windows) {
                size_t per_read = (size_t)BDX_WCAP * 3;
                if (dense_w && (size_t)ctx->dev.pass[k].n_barcodes > per_read) per_read = (size_t)ctx->dev.pass[k].n_barcodes;
                HIP_TRY(ctx, ctx->d_wins[k].ensure((size_t)n_reads * per_read * 4 + 64));
                HIP_TRY(ctx, ctx->d_wcnt[k].ensure((size_t)n_reads + 64));
            }
        }
        c0 = (uint32_t *)ctx->d_cand[0].p;
        c1 = npass > 1 ? (uint32_t *)ctx->d_cand[1].p : c0;
        if (windows) {
            w0 = (uint32_t *)ctx->d_wins[0].p;
            n0 = (uint8_t *)ctx->d_wcnt[0].p;
            w1 = npass > 1 ? (uint32_t *)ctx->d_wins[1].p : w0;
            n1 = npass > 1 ? (uint8_t *)ctx->d_wcnt[1].p : n0;
        }
        // scratch words (d_maxlen, 1 KiB): +64 tile queue, +128 hand-over count (+132.. tuning statistics),
        // +192 tier-0 list length, +256 tile queue of the second launch; one memset clears them all
        // (the block's two halves alternate between calls: this call's last launch clears the other half for the next call,
        // which then needs no memset of its own — 5 us of fill + a launch gap per call, 1 % of a 10 M-read C2 step)
        const int spar = ctx->scratch_par & 1;
        char *scratch = (char *)ctx->d_maxlen.p + 512 * spar;
        uint32_t *zero_next = (uint32_t *)((char *)ctx->d_maxlen.p + 512 * (1 - spar) + 64);
        uint32_t *exc_list = nullptr;
        unsigned int *exc_count = (unsigned int *)(scratch + 128);
        if (!split) {
            HIP_TRY(ctx, ctx->d_exc.ensure((size_t)n_reads * 4 + 64));
            exc_list = (uint32_t *)ctx->d_exc.p;
        }
        if (ctx->tune.poison) {
            // test switch: whatever a consumer reads without a producer having written it is garbage on EVERY run
            if (tiered) HIP_TRY(ctx, ctx->d_tier.ensure((size_t)n_reads * 4 + 64));
            if (ctx->fs[0].wplan.enabled) HIP_TRY(ctx, ctx->d_wlist.ensure((size_t)n_reads * 4 + 64));
            // every hand-over buffer is filled with 0xA5; between each producer and its consumer a checker kernel
            // (bdx_poison_check_kernel) then looks at exactly the elements the consumer is going to read — an element that
            // still holds the fill was never written: counted (bdx_rejected_windows) and made harmless
            DevBuf *bufs[] = {&ctx->d_cand[0], &ctx->d_cand[1], &ctx->d_wins[0], &ctx->d_wins[1], &ctx->d_wcnt[0], &ctx->d_wcnt[1],
                              &ctx->d_exc, &ctx->d_tier, &ctx->d_wlist, &ctx->d_carry};
            for (DevBuf *b : bufs)
                if (b->p) HIP_TRY(ctx, hipMemsetAsync(b->p, 0xA5, b->cap, ctx->stream));
        }
        if (!ctx->scratch_zeroed && !ctx->scratch_clean[spar]) HIP_TRY(ctx, hipMemsetAsync(scratch + 64, 0, 4 * BDX_SCRATCH_WORDS, ctx->stream));
        ctx->scratch_zeroed = false;
        ctx->scratch_clean[spar] = false;  // (a call that fails half-way leaves it that way: the next one clears it itself)
        // restricted runs of passes that only report score (+ end) through the clean-class DP start m + kb columns before
        // the first end column (orc_selftest_clean_short_lookback); everything else keeps 2 (m + kb) + 1
        int short_lb[2] = {0, 0};
        for (int k = 0; k < npass; ++k) {
            const int ts = ctx->dev.pass[k].trim_side;
            short_lb[k] = (ctx->plan.clean || ctx->plan.band_roll) && ctx->dev.algorithm == BDX_ALG_SEMIGLOBAL && !ctx->dev.need_traceback &&
                          (ts == 0 || (ts == 5 && o.pass_start == nullptr));
        }
        for (BdxFilterSet &f : ctx->fs) {
            f.bplan.short_lb[0] = short_lb[0];
            f.bplan.short_lb[1] = short_lb[1];
            f.bplan.n_cu = ctx->n_cu;
        }
        // diagonal-band DP of the exact kernel (sg_core_band): clean class, every barcode of the config with the same
        // number of rows and of the pass with the same budget, column windows handed over by tracked sweeps
        const auto band_cfg = [&](const BdxFilterSet &f) {
            BdxDevCfg dv = ctx->dev;
            dv.dense_w = (&f == &ctx->fs[0]) && dense_w;
            for (int k = 0; k < npass; ++k) {
                const int kb = f.bplan.kb_uniform[k];
                // (rolling band, barcodes beyond 32 rows: any uniform budget; the first end column is rebuilt from the hand-over row
                // when every barcode has the same length — band_lb is made of max_m)
                const bool on = ctx->plan.band_roll ? (windows && ctx->plan.same_len && kb >= 0)
                                                    : (windows && ctx->plan.clean && ctx->plan.uniform_len > 0 && !ctx->tune.no_band &&
                                                       ctx->dev.algorithm == BDX_ALG_SEMIGLOBAL && kb >= 0 && kb <= 4);
                dv.band_m = ctx->plan.uniform_len;
                dv.band_kb[k] = on ? kb : -1;
                if (on && !ctx->plan.band_roll) ctx->band_launches += 1;
                dv.band_lb[k] = short_lb[k] ? ctx->dev.max_m + kb : 2 * (ctx->dev.max_m + kb) + 1;
            }
            return dv;
        };
        // test switch BDX_POISON: between a producer and its consumer, every element the consumer will read must have
        // been written (bdx_poison_check_kernel); `with_windows`: also the window hand-over of both passes for these reads
        const auto poison_check = [&](uint32_t *list, const unsigned int *count, bool check_list, bool with_windows, bool packed = false) -> hipError_t {
            if (!ctx->tune.poison) return hipSuccess;
            unsigned int *dbg = (unsigned int *)ctx->d_dbg.p;
            if (!with_windows || !windows)
                return list ? bdx_launch_poison_check(list, count, n_reads, nullptr, nullptr, nullptr, 0, 0, (check_list ? 1 : 0) | (packed ? 2 : 0), dbg, ctx->stream) : hipSuccess;
            for (int k = 0; k < npass; ++k) {
                hipError_t e = bdx_launch_poison_check(list, count, n_reads, k ? w1 : w0, k ? n1 : n0, k ? c1 : c0, ctx->dev.pass[k].cand_words,
                                                       ctx->dev.pass[k].n_barcodes, (check_list && k == 0) ? 1 : 0, dbg, ctx->stream);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        BdxTierArgs t0{0, nullptr, nullptr, nullptr, nullptr};
        // Wave-autonomous kernel (bdx_wave.hip) in front of the general one: it answers the reads of the known-score
        // class and lists the rest — as tier 1 of a tiered config, or (plain configs) ahead of the same filter set
        // in list mode.  The list-mode plan is made first: if it cannot be made, the general kernel runs alone.
        bool wave1 = false, wave0 = false;
        if (!split && !ctx->dev.vlen) {
            // (window mode first: reads much longer than their column window — only the windows are fetched)
            if (tiered)
                wave1 = size_wave_win(ctx, ctx->fs[1].wplan, batch_len, n_reads) || size_wave(ctx, 1, batch_len, n_reads);
            else if (size_wave_win(ctx, ctx->fs[0].wplan, batch_len, n_reads) || size_wave(ctx, 0, batch_len, n_reads))
                wave0 = size_bitpar(ctx, batch_len, n_reads, true);
            if (!tiered && !wave0) (void)size_bitpar(ctx, batch_len, n_reads);  // (restore the dense plan)
        }
        // Known-end class (trim_side = 5, single pass, no start positions or statistics wanted): the same kernel in its
        // known-end form answers the reads it can settle, trimmed keep range included; the listed rest goes through the
        // split path (filter in list mode -> exact kernel in list mode).
        bool wave1k = false, wave0k = false;
        bool trim3 = false;  // (a trim_side = 3 pass of the known-trim class knows its start only)
        for (int k = 0; k < npass; ++k) trim3 |= ctx->dev.pass[k].trim_side == 3;
        const bool kend_ok = split && windows && !ctx->dev.vlen && !dense_w && o.pass_start == nullptr && stp == nullptr && !(trim3 && o.pass_end != nullptr);
        // known-alignment class: the caller wants positions the known-trim class does not know, or the statistics tables
        const bool aln_ok = split && windows && !ctx->dev.vlen && !dense_w && !kend_ok && ctx->fs[tiered ? 1 : 0].wplan_a.enabled;
        BdxWavePlan &wk1 = aln_ok ? ctx->fs[1].wplan_a : ctx->fs[1].wplan_k, &wk0 = aln_ok ? ctx->fs[0].wplan_a : ctx->fs[0].wplan_k;
        if (kend_ok || aln_ok) {
            if (tiered)
                wave1k = size_wave(ctx, wk1, batch_len, n_reads);
            else if (size_wave(ctx, wk0, batch_len, n_reads)) {
                wave0k = size_bitpar(ctx, batch_len, n_reads, true);
                if (!wave0k) (void)size_bitpar(ctx, batch_len, n_reads);  // (restore the dense plan)
            }
        }
        // split configs (trimming, summary, weighted costs): the wave kernel as the FILTER of a dense launch — candidate
        // masks and column windows in the formats of the general kernel's split mode, every verdict from the exact
        // kernel as before.  Tiered: tier 1 (all reads); plain: the only filter launch.
        bool wsplit1 = false, wsplit0 = false, pairs_t1 = false;
        BdxWaveSplit wsp{};
        for (int k = 0; k < 2; ++k) {
            wsp.cw[k] = k < npass ? ctx->dev.pass[k].cand_words : 0;
            wsp.cand_out[k] = k ? c1 : c0;
            wsp.wins_out[k] = k ? w1 : w0;
            wsp.wcnt_out[k] = k ? n1 : n0;
            wsp.short_lb[k] = short_lb[k];
        }
        const BdxWaveSplit &wsp_all = wsp;
        if (split && windows && !ctx->dev.vlen && !wave1k && !wave0k) {
            if (tiered)
                wsplit1 = ctx->fs[1].wplan.split && size_wave(ctx, 1, batch_len, n_reads);
            else
                wsplit0 = ctx->fs[0].wplan.split && !dense_w && size_wave(ctx, 0, batch_len, n_reads);
        }
        if (wave0) {
            HIP_TRY(ctx, ctx->d_wlist.ensure((size_t)n_reads * 4 + 64));
            if (ctx->fs[0].wplan.winm)
                HIP_TRY(ctx, bdx_launch_wave_win(ctx->dev, ctx->fs[0].wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                                 ctx->counts, 0, 0.0, (uint32_t *)ctx->d_wlist.p, (unsigned int *)(scratch + 192), ctx->stream, ctx->tune.debug));
            else
            HIP_TRY(ctx, bdx_launch_wave(ctx->dev, ctx->fs[0].wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                         ctx->counts, (int *)(scratch + 256), 0, 0.0, (uint32_t *)ctx->d_wlist.p, (unsigned int *)(scratch + 192),
                                         ctx->stream, ctx->tune.debug));
            ctx->wave_launches += 1;
            t0.in_list = (const uint32_t *)ctx->d_wlist.p;
            t0.in_count = (const unsigned int *)(scratch + 192);
        }
        if (wave0k) {
            HIP_TRY(ctx, ctx->d_wlist.ensure((size_t)n_reads * 4 + 64));
            HIP_TRY(ctx, bdx_launch_wave_end(ctx->dev, wk0, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                             ctx->counts, 0, 0.0, (uint32_t *)ctx->d_wlist.p, (unsigned int *)(scratch + 192), ctx->stream, ctx->tune.debug, 0.0,
                                             aln_ok ? stp : nullptr));
            ctx->wave_launches += 1;
            t0.in_list = (const uint32_t *)ctx->d_wlist.p;
            t0.in_count = (const unsigned int *)(scratch + 192);
        }
        // Carried passes: tier 1 of a dual known-class config lists a read when ONE of its passes is open; the pass it settled goes
        // along (two state bits on the list entry + the pass's winning survivor in d_carry[read]) and the pairs mode only looks for
        // the other pass's barcodes — about half of its sweeps for C4.  Only when the pairs mode in its known form is what reads
        // tier 1's list (nothing else understands the state bits), reads fit 30 bits and min_delta = 0 (a lone carried winner
        // then IS the pass's result).
        bool carry_on = false;
        for (BdxFilterSet &f : ctx->fs)
            f.wplan.d_carry = f.pplan.d_carry = f.wplan_k.d_carry = f.wplan_a.d_carry = f.pplan_k.d_carry = f.pplan_a.d_carry = nullptr;
        // (the known-trim / known-alignment forms — wave1k — and the plain known-score form of a dual config without trimming — wave1)
        const bool carry_tier = tiered && (wave1k || (wave1 && !split && !ctx->fs[1].wplan.winm));
        if (carry_tier && ctx->dev.is_dual && ctx->dev.min_delta == 0.0 && !ctx->tune.no_carry && n_reads < (1LL << 30) && o.pass_start == nullptr &&
            o.pass_end == nullptr && o.pass_raw == nullptr && o.pass_bc == nullptr && o.pass_score == nullptr && o.pass_delta == nullptr) {
            BdxWavePlan &pp = wave1k ? (aln_ok ? ctx->fs[0].pplan_a : ctx->fs[0].pplan_k) : ctx->fs[0].pplan;
            BdxWavePlan &t1p = wave1k ? wk1 : ctx->fs[1].wplan;
            if (size_pairs(ctx, pp, tier_len) && pp.groups <= 1 && pp.pairs_kb <= 4 && !pp.split) {
                HIP_TRY(ctx, ctx->d_carry.ensure((size_t)n_reads * 4 + 64));
                carry_on = true;
                t1p.d_carry = (uint32_t *)ctx->d_carry.p;
                pp.d_carry = (uint32_t *)ctx->d_carry.p;
            }
        }
        if (tiered) {
            HIP_TRY(ctx, ctx->d_tier.ensure((size_t)n_reads * 4 + 64));
            BdxTierArgs t1{1, (uint32_t *)ctx->d_tier.p, (unsigned int *)(scratch + 192), nullptr, nullptr};
            BdxFilterSet &f1 = ctx->fs[1];
            f1.bplan.d_tile_counter = (int *)(scratch + 256);
            f1.bplan.dense_w = 0;
            f1.bplan.grid_override = ctx->tune.grid;
            f1.bplan.dbg = ctx->tune.debug;
            if (wave1k) {  // tier 1 as the known-end form of the wave kernel: verdicts + trimmed keep range of what it settles, the rest listed
                HIP_TRY(ctx, bdx_launch_wave_end(ctx->dev, wk1, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                                 ctx->counts, 1, f1.bplan.tier_slo[0], t1.out_list, t1.out_count, ctx->stream, ctx->tune.debug, f1.bplan.tier_slo[1],
                                                 aln_ok ? stp : nullptr));
                ctx->wave_launches += 1;
            } else if (ctx->pairs_tier && split && windows && !dense_w && !ctx->dev.vlen && size_pairs(ctx, f1.pplan, batch_len)) {
                // the pairs tier: tier 1's filter is the same-diagonal pairs mode over every read of the batch
                HIP_TRY(ctx, bdx_launch_pairs(ctx->dev, f1.pplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, nullptr,
                                              nullptr, o, nullptr, nullptr, nullptr, ctx->stream, ctx->tune.debug >> 8, &wsp));
                ctx->pair_launches += 1;
                pairs_t1 = true;
            } else if (wsplit1) {  // tier 1's filter as the wave-autonomous kernel (split mode: the exact kernel settles and lists)
                HIP_TRY(ctx, bdx_launch_wave(ctx->dev, f1.wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                             nullptr, (int *)(scratch + 256), 0, 0.0, nullptr, nullptr, ctx->stream, ctx->tune.debug, &wsp));
                ctx->wave_launches += 1;
            } else if (wave1 && f1.wplan.winm) {  // ... in window mode
                HIP_TRY(ctx, bdx_launch_wave_win(ctx->dev, f1.wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                                 ctx->counts, 1, f1.bplan.tier_slo[0], t1.out_list, t1.out_count, ctx->stream, ctx->tune.debug));
                ctx->wave_launches += 1;
            } else if (wave1) {  // tier 1 as the wave-autonomous kernel: same budgets, same settle rule, same list
                HIP_TRY(ctx, bdx_launch_wave(ctx->dev, f1.wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                             ctx->counts, (int *)(scratch + 256), 1, f1.bplan.tier_slo[0], t1.out_list, t1.out_count, ctx->stream, ctx->tune.debug,
                                             nullptr, f1.bplan.tier_slo[1]));
                ctx->wave_launches += 1;
            } else
            HIP_TRY(ctx, bdx_launch_bitpar(ctx->dev, ctx->plan, f1.bplan, f1.splan, d_seq_bytes, (const long long *)d_seq_off, n_reads,
                                           o, ctx->counts, c0, c1, ctx->stream, w0, w1, n0, n1, split ? 1 : 0, exc_list, exc_count, &t1));
            if (split && !wave1k) HIP_TRY(ctx, poison_check(nullptr, nullptr, false, true));
            if (split && !wave1k)  // the exact kernel answers what tier 1 settles and lists the rest (known-score / known-end configs: the filter kernel did)
                HIP_TRY(ctx, bdx_launch_generic(band_cfg(f1), ctx->plan, d_seq_bytes, (const long long *)d_seq_off, n_reads, o, ctx->counts,
                                                c0, npass > 1 ? c1 : nullptr, ctx->stream, w0, npass > 1 ? w1 : nullptr, n0,
                                                npass > 1 ? n1 : nullptr, nullptr, nullptr, stp, &t1, f1.bplan.tier_slo));
            // tier 0 walks the list: scattered reads -> slot staging
            // (tier 0 sees a fraction of the batch — 10..25 % in the bench configs: its tile size is planned for a sixteenth of
            // the batch, so that the list of a small batch still spreads over the device — C5, 400 k reads: tiles of 16 instead
            // of 128 reads, 0.42 -> 0.38 ms; batches of millions of reads keep their tiles)
            long long n_list_est = n_reads / (ctx->tune.tier0_div > 0 ? ctx->tune.tier0_div : 16);
            if (n_list_est < 1) n_list_est = 1;
            if (!size_bitpar(ctx, tier_len, n_list_est, true)) return fail(ctx, BDX_E_DEVICE, "internal: tier 0 cannot be planned in list mode");
            t0.in_list = (const uint32_t *)ctx->d_tier.p;
            t0.in_count = (const unsigned int *)(scratch + 192);
        }
        if (tiered || wave0 || wave0k) HIP_TRY(ctx, poison_check((uint32_t *)t0.in_list, t0.in_count, true, false, carry_on));
        // Pairs mode of the wave kernel between tier 1 and the general kernel: the listed reads are gathered into slots and
        // filtered at the full budgets by the two-intact-pieces lemma.  Known-score configs: it answers them (what it cannot
        // answer goes on to the general kernel in list mode); split configs: it is tier 0's filter (masks + windows of the
        // listed reads for the exact kernel).
        bool pairs = false, pairs_k = false;
        if (tiered && (kend_ok || aln_ok) && size_pairs(ctx, aln_ok ? ctx->fs[0].pplan_a : ctx->fs[0].pplan_k, tier_len)) {
            // known-end class: the pairs mode answers the listed reads itself (verdict + trimmed keep range); what it cannot
            // answer goes on to the split path in list mode
            const BdxWavePlan &pp = aln_ok ? ctx->fs[0].pplan_a : ctx->fs[0].pplan_k;
            HIP_TRY(ctx, ctx->d_wlist.ensure((size_t)n_reads * 4 + 64));
            if (ctx->tune.poison) HIP_TRY(ctx, hipMemsetAsync(ctx->d_wlist.p, 0xA5, ctx->d_wlist.cap, ctx->stream));
            uint32_t *list2 = (uint32_t *)ctx->d_wlist.p;
            unsigned int *count2 = (unsigned int *)(scratch + 320);
            HIP_TRY(ctx, bdx_launch_pairs(ctx->dev, pp, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, t0.in_list,
                                          t0.in_count, o, ctx->counts, list2, count2, ctx->stream, ctx->tune.debug >> 8, nullptr, aln_ok ? stp : nullptr));
            ctx->pair_launches += 1;
            pairs = pairs_k = true;
            t0.in_list = list2;
            t0.in_count = count2;
            HIP_TRY(ctx, poison_check(list2, count2, true, false));
        } else
        if (tiered && (!split || windows) && !dense_w && size_pairs(ctx, tier_len)) {
            const BdxWavePlan &pp = ctx->fs[0].pplan;
            if (!split) HIP_TRY(ctx, ctx->d_wlist.ensure((size_t)n_reads * 4 + 64));
            if (ctx->tune.poison && !split) HIP_TRY(ctx, hipMemsetAsync(ctx->d_wlist.p, 0xA5, ctx->d_wlist.cap, ctx->stream));
            uint32_t *list2 = split ? nullptr : (uint32_t *)ctx->d_wlist.p;
            unsigned int *count2 = split ? nullptr : (unsigned int *)(scratch + 320);
            HIP_TRY(ctx, bdx_launch_pairs(ctx->dev, pp, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, t0.in_list,
                                          t0.in_count, o, split ? nullptr : ctx->counts, list2, count2, ctx->stream, ctx->tune.debug >> 8, split ? &wsp_all : nullptr));
            ctx->pair_launches += 1;
            pairs = true;
            if (!split) {  // the general kernel (list mode) evaluates what is left
                t0.in_list = list2;
                t0.in_count = count2;
                HIP_TRY(ctx, poison_check(list2, count2, true, false));
            }
        }
        // Same-diagonal pairs mode as the ONLY filter of a split config without tiers (weighted costs whose full budget is beyond
        // every seeded variant — the reference's demo2 options): every read of the batch is laid out in slots and scanned;
        // masks + windows of all reads go to the exact kernel's dense launch.
        bool pairs_all = false;
        if (!tiered && split && windows && !dense_w && !wsplit0 && !wave0k && !ctx->dev.vlen && ctx->fs[0].pplan.enabled &&
            ctx->fs[0].pplan.pairs_kb >= 8 && ctx->fs[0].pplan.split && size_pairs(ctx, batch_len)) {
            const BdxWavePlan &pp = ctx->fs[0].pplan;
            HIP_TRY(ctx, bdx_launch_pairs(ctx->dev, pp, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, nullptr,
                                          nullptr, o, nullptr, nullptr, nullptr, ctx->stream, ctx->tune.debug >> 8, &wsp_all));
            ctx->pair_launches += 1;
            pairs_all = true;
        }
        ctx->F().bplan.d_tile_counter = (int *)(scratch + 64);
        ctx->F().bplan.dense_w = dense_w;
        ctx->F().bplan.grid_override = ctx->tune.grid;
        ctx->F().bplan.dbg = ctx->tune.debug;
        if ((pairs && split && !pairs_k) || pairs_all) {
            // (tier 0's filter already ran: the pairs mode wrote the listed reads' masks and windows)
        } else if (wsplit0) {
            HIP_TRY(ctx, bdx_launch_wave(ctx->dev, ctx->fs[0].wplan, ctx->plan.hist_entries, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                         nullptr, (int *)(scratch + 256), 0, 0.0, nullptr, nullptr, ctx->stream, ctx->tune.debug, &wsp));
            ctx->wave_launches += 1;
        } else
        HIP_TRY(ctx, bdx_launch_bitpar(ctx->dev, ctx->plan, ctx->F().bplan, ctx->F().splan, d_seq_bytes,
                                       (const long long *)d_seq_off, n_reads, o, ctx->counts, c0, c1, ctx->stream, w0, w1, n0,
                                       n1, split ? 1 : 0, exc_list, exc_count, (tiered || wave0 || wave0k) ? &t0 : nullptr));
        const bool listed = tiered || wave0k;  // split configs: the exact kernel's last launch walks a list
        if (split)
            HIP_TRY(ctx, poison_check(listed ? (uint32_t *)t0.in_list : nullptr, listed ? t0.in_count : nullptr, false, true));
        else
            HIP_TRY(ctx, poison_check(exc_list, exc_count, true, false));
        if (split)  // (tiered: list mode over the reads tier 1 handed on)
            HIP_TRY(ctx, bdx_launch_generic(band_cfg(ctx->F()), ctx->plan, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                            ctx->counts, c0, npass > 1 ? c1 : nullptr, ctx->stream, w0,
                                            npass > 1 ? w1 : nullptr, n0, npass > 1 ? n1 : nullptr, listed ? t0.in_list : nullptr,
                                            listed ? t0.in_count : nullptr, stp, nullptr, nullptr, zero_next));
        else
            HIP_TRY(ctx, bdx_launch_generic(ctx->dev, ctx->plan, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                            ctx->counts, c0, npass > 1 ? c1 : nullptr, ctx->stream, nullptr, nullptr, nullptr,
                                            nullptr, exc_list, exc_count, stp, nullptr, nullptr, zero_next));
        // (the call's last launch is enqueued: the other half will hold zeros when the next call's kernels start)
        ctx->scratch_clean[1 - spar] = true;
        ctx->scratch_par = 1 - spar;
#ifdef BDX_TUNING
        if (ctx->tune.debug & 128) {  // tuning statistics of the fused kernel (see bdx_bitpar.hip)
                unsigned int st[4] = {0, 0, 0, 0}, tl = 0;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                HIP_TRY(ctx, hipMemcpy(st, exc_count, sizeof(st), hipMemcpyDeviceToHost));
                HIP_TRY(ctx, hipMemcpy(&tl, scratch + 192, sizeof(tl), hipMemcpyDeviceToHost));
                fprintf(stderr, "[bdx] handed over %u reads; %u windowed sweeps, %u columns, %u tiles with a fallback read; tier 0 list %u (of %lld reads)\n",
                        st[0], st[1], st[2], st[3], tl, (long long)n_reads);
        }
#endif
        ctx->last_blocks = (n_reads + ctx->F().bplan.reads_per_block - 1) / ctx->F().bplan.reads_per_block;
        ctx->path = ctx->F().splan.enabled ? (ctx->F().splan.diag ? "qgram2+bitpar+verify" : "qgram+bitpar+verify") : "bitpar+verify";
        if (wsplit0) ctx->path = "wave+verify";
        if (pairs) ctx->path = pairs_k ? (aln_ok ? "pairs(aln) > " : "pairs(end) > ") + ctx->path : split ? "pairs+verify" : "pairs > " + ctx->path;
        if (pairs_all) ctx->path = "pairs(diag)+verify";
        if (tiered) ctx->path = (pairs_t1 ? "tier1:pairs(diag) > " : (wave1k && aln_ok) ? "tier1:wave(aln) > " : wave1k ? "tier1:wave(end) > " : (wave1 && ctx->fs[1].wplan.winm) ? "tier1:wave(win) > " : (wave1 || wsplit1) ? "tier1:wave > " : "tier1:qgram+bitpar > ") + ctx->path;
        if (wave0) ctx->path = (ctx->fs[0].wplan.winm ? "wave(win) > " : "wave > ") + ctx->path;
        if (wave0k) ctx->path = (aln_ok ? "wave(aln) > " : "wave(end) > ") + ctx->path;
        ctx->filter_used = ctx->F().splan.enabled ? BDX_FILTER_QGRAM : BDX_FILTER_BITPAR;
    } else {
        HIP_TRY(ctx, bdx_launch_generic(ctx->dev, ctx->plan, d_seq_bytes, (const long long *)d_seq_off, n_reads, o,
                                        ctx->counts, nullptr, nullptr, ctx->stream, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, stp));
        ctx->last_blocks = (n_reads + ctx->plan.threads - 1) / ctx->plan.threads;
        ctx->path = "generic";
        ctx->filter_used = BDX_FILTER_OFF;
    }
    ctx->launches += 1;
    return BDX_OK;
}

int32_t bdx_classify_host(bdx_ctx *ctx, const uint8_t *seq_bytes, const int64_t *seq_off, int64_t n_reads,
                          const bdx_outputs_t *out) {
    if (!ctx) return BDX_E_INVALID;
    if (n_reads < 0) return fail(ctx, BDX_E_INVALID, "n_reads is negative");
    if (n_reads == 0) return BDX_OK;
    if (!seq_bytes || !seq_off || !out) return fail(ctx, BDX_E_INVALID, "NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t base = seq_off[0];
    const int64_t total = seq_off[n_reads] - base;
    if (total < 0) return fail(ctx, BDX_E_INVALID, "seq_off is not non-decreasing");
    {
        // Window upload: when the passes only look at a short column window of long reads (ONT-style reads with the
        // barcodes at an end, C5), copy just each read's window — the union over the passes of final_search_range
        // (+ m - 1 for :hamming / :exact), resolved exactly like the device does — instead of the whole read:
        // 10 kbp reads with "1:200" move 212 B per read over PCIe instead of 10 KB.
        int rcw = classify_host_windows(ctx, seq_bytes, seq_off, n_reads, out);
        if (rcw != 1) return rcw;  // 0 done, < 0 error, 1: ordinary upload below
    }
    // The batch's longest read, for the launch plan and the statistics tables: the offsets are on the host anyway
    // (saves the device-side measurement — a tiny kernel, a 4-byte copy and a stream synchronisation per call, which
    // matters at the reference's chunk size of 4000 reads)
    {
        // one pass over the offsets: they must be non-decreasing (a negative length would reach the kernels' address
        // arithmetic), and the longest read comes out of the same pass; large batches are scanned by a few threads
        int64_t mx = 0;
        bool monotone = true;
        if (!scan_offsets(seq_off, n_reads, mx, monotone)) return fail(ctx, BDX_E_DEVICE, "out of host memory");
        if (!monotone) return fail(ctx, BDX_E_INVALID, "seq_off is not non-decreasing");
        ctx->host_maxlen = (int)(mx > (1LL << 30) ? (1LL << 30) : (mx < 1 ? 1 : mx));
    }
    struct HostLenReset {
        bdx_ctx *c;
        ~HostLenReset() { c->host_maxlen = 0; }
    } host_len_reset{ctx};
    if ((size_t)total + (size_t)(n_reads + 1) * 8 <= ((size_t)2 << 20)) {  // (beyond ~2 MB the extra host copy costs more than the second transfer)
        // small batches: bytes and offsets through ONE page-locked staging buffer and ONE asynchronous copy
        const size_t o_off = ((size_t)total + 64 + 255) & ~(size_t)255;
        const size_t bytes = o_off + (size_t)(n_reads + 1) * 8;
        if (ctx->h_in_bytes < bytes) {
            if (ctx->h_in) (void)hipHostFree(ctx->h_in);
            ctx->h_in = nullptr;
            ctx->h_in_bytes = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->h_in, bytes + (1 << 16), hipHostMallocDefault));
            ctx->h_in_bytes = bytes + (1 << 16);
        }
        memcpy(ctx->h_in, seq_bytes + base, (size_t)total);
        memcpy((char *)ctx->h_in + o_off, seq_off, (size_t)(n_reads + 1) * 8);
        HIP_TRY(ctx, ctx->d_seq.ensure(bytes + 64));
        void *h_in_dev = nullptr;  // the staging buffer as the device sees it
        HIP_TRY(ctx, hipHostGetDevicePointer(&h_in_dev, ctx->h_in, 0));
        const bool zero_scratch = ctx->d_maxlen.p != nullptr;  // (allocated at bdx_create when a filter is in use)
        HIP_TRY(ctx, bdx_launch_copy(ctx->d_seq.p, h_in_dev, bytes, ctx->stream, zero_scratch ? (char *)ctx->d_maxlen.p + 512 * (ctx->scratch_par & 1) + 64 : nullptr, 4 * BDX_SCRATCH_WORDS));
        ctx->scratch_zeroed = zero_scratch;
        const int rcs = run_and_download(ctx, (const uint8_t *)ctx->d_seq.p - base, (const int64_t *)((const char *)ctx->d_seq.p + o_off),
                                         n_reads, out, /*mapped_outputs=*/true);
        ctx->scratch_zeroed = false;  // (also when the batch took a path that never looked at the flag)
        return rcs;
    }
    // (worth it when the kernels take a noticeable part of the call — tiered budgets, split mode, no filter; the
    // single fused launch of a plain known-score config is 3 ms per 10 M reads, chunking it costs more than it hides:
    // measured C4 202 -> 242 M reads/s from pageable and 235 -> 295 M from page-locked buffers, C2 299 -> 260 M)
    bool heavy = ctx->tiered || !ctx->fs[0].bplan.enabled;
    for (int k = 0; k < (ctx->dev.is_dual ? 2 : 1); ++k) heavy = heavy || !ctx->fs[0].bplan.known_ok[k];
    if (heavy && total >= ((int64_t)96 << 20) && n_reads >= 8 * 65536 && !ctx->tune.no_pipeline) {
        int k = (int)(total / ((int64_t)48 << 20));
        k = k < 2 ? 2 : (k > 8 ? 8 : k);
        return classify_host_pipelined(ctx, seq_bytes, seq_off, n_reads, out, k);
    }
    // The offsets are uploaded as given; the byte pointer is rebased so that off[0] indexes it.
    HIP_TRY(ctx, ctx->d_seq.ensure((size_t)total + 64));
    HIP_TRY(ctx, ctx->d_off.ensure((size_t)(n_reads + 1) * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_seq.p, seq_bytes + base, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_off.p, seq_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    // kernel sees the byte base shifted by -base so that off[i] addresses read i
    return run_and_download(ctx, (const uint8_t *)ctx->d_seq.p - base, (const int64_t *)ctx->d_off.p, n_reads, out);
}

// Page-locked host memory for the buffers handed to bdx_classify_host (reads, offsets, outputs): the
// copies then run as asynchronous DMA at PCIe speed instead of being staged through the driver.
void *bdx_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0) bytes = 1;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void bdx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int64_t bdx_counts_len(const bdx_ctx *ctx) { return ctx ? ctx->dev.n_counts : 0; }

int32_t bdx_get_counts(bdx_ctx *ctx, int64_t *out, int64_t n) {
    if (!ctx || !out) return BDX_E_INVALID;
    if (n < ctx->dev.n_counts) return fail(ctx, BDX_E_INVALID, "counts buffer too small: need %d", ctx->dev.n_counts);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->counts, (size_t)ctx->dev.n_counts * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDX_OK;
}

int32_t bdx_reset_counts(bdx_ctx *ctx) {
    if (!ctx) return BDX_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counts, 0, (size_t)ctx->dev.n_counts * 8, ctx->stream));
    if (ctx->dev.need_traceback)
        for (int p = 0; p < (ctx->dev.is_dual ? 2 : 1); ++p)
            for (int w = 0; w < 3; ++w)
                if (ctx->st_tab[p][w].p)
                    HIP_TRY(ctx, hipMemsetAsync(ctx->st_tab[p][w].p, 0, bdx_stats_phys_words(ctx, p, w, ctx->st_rows) * 8, ctx->stream));
    return BDX_OK;
}

int32_t bdx_stats_shape(const bdx_ctx *ctx, int32_t pass, int32_t which, int64_t *rows, int64_t *key0, int64_t *n_barcodes) {
    if (!ctx) return BDX_E_INVALID;
    if (pass < 0 || pass > 1 || which < BDX_STATS_POS || which > BDX_STATS_RAW) return BDX_E_INVALID;
    const bool on = ctx->dev.need_traceback && (pass == 0 || ctx->dev.is_dual);
    if (rows) *rows = !on ? 0 : (which == BDX_STATS_RAW ? ctx->st_raw_rows : which == BDX_STATS_LEN ? ctx->st_len_rows : ctx->st_rows);
    if (key0) *key0 = which == BDX_STATS_POS ? 1 - (int64_t)ctx->dev.max_m : 0;
    if (n_barcodes) *n_barcodes = on ? ctx->dev.pass[pass].n_barcodes : 0;
    return BDX_OK;
}

int32_t bdx_get_stats(bdx_ctx *ctx, int32_t pass, int32_t which, int32_t reduced, int64_t *out, int64_t n_words) {
    if (!ctx || !out) return BDX_E_INVALID;
    if (pass < 0 || pass > 1 || which < BDX_STATS_POS || which > BDX_STATS_RAW) return fail(ctx, BDX_E_INVALID, "bad statistics table selector");
    if (!ctx->dev.need_traceback || (pass == 1 && !ctx->dev.is_dual)) return fail(ctx, BDX_E_STATE, "the config collects no statistics for this pass (need_traceback = 0)");
    // the shape handed out is the CURRENT one (bdx_stats_shape); the summed twins were filled at the shape of the last
    // all-reduce — fewer pos rows, and for a growing length table fewer keys at a smaller pitch: rows / keys appended
    // since then read as zero
    const size_t B = (size_t)ctx->dev.pass[pass].n_barcodes;
    const size_t words = bdx_stats_words(ctx, pass, which, ctx->st_rows);
    if ((size_t)n_words < words) return fail(ctx, BDX_E_INVALID, "statistics buffer too small: need %zu words", words);
    const DevBuf &src = reduced ? ctx->st_sum[pass][which] : ctx->st_tab[pass][which];
    if (reduced && !src.p) return fail(ctx, BDX_E_STATE, "bdx_allreduce_counts has not been called");
    const long long pos_rows = reduced ? ctx->st_sum_rows : ctx->st_rows;
    const size_t keys_now = which == BDX_STATS_RAW ? (size_t)ctx->st_raw_rows : (size_t)ctx->st_len_rows;
    const size_t keys_src = which == BDX_STATS_LEN && reduced ? (size_t)ctx->st_sum_len_rows : keys_now;
    const size_t stride_src = (keys_src + 15) & ~(size_t)15;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned int flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->st_flag.p, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int64_t> phys;  // len / raw: [barcode][key stride] on the device
    if (which == BDX_STATS_POS) {
        const size_t have = (size_t)pos_rows * B;
        if (have) HIP_TRY(ctx, hipMemcpyAsync(out, src.p, have * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (words > have) memset(out + have, 0, (words - have) * 8);
    } else {
        phys.resize(stride_src * B);
        if (!phys.empty()) HIP_TRY(ctx, hipMemcpyAsync(phys.data(), src.p, phys.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) return fail(ctx, BDX_E_STATE, "a statistics key fell outside its table (internal sizing error)");
    if (which != BDX_STATS_POS) {
        for (size_t k = 0; k < keys_now; ++k)
            for (size_t b = 0; b < B; ++b) out[k * B + b] = k < keys_src ? phys[b * stride_src + k] : 0;
    }
    return BDX_OK;
}

void *bdx_counts_device_ptr(bdx_ctx *ctx) { return ctx ? (void *)ctx->counts : nullptr; }

int32_t bdx_set_counts_buffer(bdx_ctx *ctx, void *d_counts) {
    if (!ctx) return BDX_E_INVALID;
    ctx->counts = d_counts ? (unsigned long long *)d_counts : (unsigned long long *)ctx->counts_own.p;
    return BDX_OK;
}

const char *bdx_kernel_path(const bdx_ctx *ctx) { return ctx ? ctx->path.c_str() : ""; }

int64_t bdx_window_uploads(const bdx_ctx *ctx) { return ctx ? ctx->window_uploads : 0; }

int64_t bdx_band_launches(const bdx_ctx *ctx) { return ctx ? ctx->band_launches : 0; }

int64_t bdx_wave_launches(const bdx_ctx *ctx) { return ctx ? ctx->wave_launches : 0; }
int64_t bdx_pair_launches(const bdx_ctx *ctx) { return ctx ? ctx->pair_launches : 0; }

int64_t bdx_pipelined_calls(const bdx_ctx *ctx) { return ctx ? ctx->pipelined_calls : 0; }
int64_t bdx_staged_downloads(const bdx_ctx *ctx) { return ctx ? ctx->staged_downloads : 0; }
int64_t bdx_last_list_reads(bdx_ctx *ctx) {
    if (!ctx || !ctx->d_maxlen.p) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    unsigned int v = 0;
    // (the scratch half of the LAST call: the halves alternate, the last launch of a call clears the other one)
    const char *last = (const char *)ctx->d_maxlen.p + 512 * (1 - (ctx->scratch_par & 1)) + 192;
    if (hipMemcpy(&v, last, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

int64_t bdx_rejected_windows(bdx_ctx *ctx) {
    if (!ctx || !ctx->d_dbg.p) return 0;
    unsigned int rej[2] = {0, 0};  // [0] refused by the exact kernel, [1] found unwritten by the poison checker
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(rej, ctx->d_dbg.p, sizeof rej, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (int64_t)rej[0] + (int64_t)rej[1];
}

int64_t bdx_debug_rejected_windows_total(void) { return (int64_t)g_rejected_windows.load(); }

int32_t bdx_launch_info(const bdx_ctx *ctx, bdx_launch_info_t *out) {
    if (!ctx || !out) return BDX_E_INVALID;
    const bool f = ctx->filter_used != BDX_FILTER_OFF && ctx->F().bplan.reads_per_block > 0;
    out->threads_per_block = f ? 256 : ctx->plan.threads;
    out->lds_bytes_per_block = f ? (int32_t)bdx_bitpar_lds_bytes(ctx->dev, ctx->F().bplan, ctx->plan, &ctx->F().splan) : (int32_t)ctx->plan.lds_bytes;
    out->blocks = ctx->last_blocks;
    out->reads_per_block = f ? ctx->F().bplan.reads_per_block : ctx->plan.threads;
    out->filter_used = ctx->filter_used;
    out->max_m = ctx->dev.max_m;
    out->launches = ctx->launches;
    return BDX_OK;
}

}  // extern "C"
