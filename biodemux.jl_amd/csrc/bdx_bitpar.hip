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
#include <cstdlib>

#include "bdx_core.h"

namespace {

struct BitparArgs {
    BdxDevCfg cfg;
    const uint8_t *seq;
    const long long *off;
    long long n_reads;
    BdxDevOut out;
    unsigned long long *counts;
    int dp_rows;
    int stage_bytes;     // capacity of EACH of the two staging areas (raw bytes, codes)
    int bc_stage_bytes;
    int hist_entries;
    const uint8_t *lut;  // 256 bytes: byte -> symbol code
    const uint32_t *peq[2];
    const uint32_t *pvinit[2];
    const int32_t *kb[2];
    int ncodes;
    int bpad[2];
    int bshift[2];
    int known_ok[2];  // config-level eligibility of the known-score class per pass
    int dbg;  // timing experiments only (env BDX_DEBUG): 1 = skip stage 2, 2 = skip stage 1 sweep
};

template <int BS, int R>
__global__ __launch_bounds__(BS) void bdx_bitpar_kernel(const BitparArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LDS unsigned char *smem = (LDS unsigned char *)smem_raw;
    const BdxDevCfg &cfg = a.cfg;
    const int tid = threadIdx.x;
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
    LDS int *DPbase = (LDS int *)take((size_t)a.dp_rows * R * 4);
    LDS int *OGbase = (LDS int *)take(cfg.any_traceback ? (size_t)a.dp_rows * R * 4 : 0);
    LDS uint32_t *off0 = (LDS uint32_t *)take((size_t)(B0 + 1) * 4);
    LDS uint32_t *off1 = (LDS uint32_t *)take((size_t)(B1 + 1) * 4);
    LDS int *nn0 = (LDS int *)take((size_t)B0 * 4);
    LDS int *nn1 = (LDS int *)take((size_t)B1 * 4);
    LDS unsigned char *bcs = take((size_t)a.bc_stage_bytes);
    LDS int *hist = (LDS int *)take((size_t)a.hist_entries * 4);
    LDS unsigned char *lut = take(256);
    LDS uint32_t *peq0 = (LDS uint32_t *)take((size_t)a.ncodes * a.bpad[0] * 4);
    LDS uint32_t *peq1 = (LDS uint32_t *)take(cfg.is_dual ? (size_t)a.ncodes * a.bpad[1] * 4 : 0);
    LDS uint32_t *pv0 = (LDS uint32_t *)take((size_t)B0 * 4);
    LDS uint32_t *pv1 = (LDS uint32_t *)take((size_t)B1 * 4);
    LDS int *kb0 = (LDS int *)take((size_t)B0 * 4);
    LDS int *kb1 = (LDS int *)take((size_t)B1 * 4);
    LDS uint32_t *cand = (LDS uint32_t *)take((size_t)R * (cw0 + cw1) * 4);
    LDS int *roff = (LDS int *)take((size_t)(R + 1) * 4);   // byte offset of each read in the stage
    LDS int *win = (LDS int *)take((size_t)R * 4 * 4);      // [pass][first|last][R]
    LDS uint32_t *slots = (LDS uint32_t *)take((size_t)2 * R * 4 * 4);  // [pass][R][4] (barcode << 8 | d)
    LDS int *scnt = (LDS int *)take((size_t)2 * R * 4);                 // [pass][R] entries pushed
    LDS unsigned char *full = take((size_t)2 * R);                      // [pass][R] read is in the known-score class
    LDS unsigned char *rstage = take((size_t)a.stage_bytes);
    LDS unsigned char *codes = take((size_t)a.stage_bytes);

    // ---- tables -> LDS ----
    for (int i = tid; i <= B0; i += BS) off0[i] = cfg.pass[0].bc_off[i];
    for (int i = tid; i < B0; i += BS) {
        nn0[i] = cfg.pass[0].bc_len_no_N[i];
        pv0[i] = a.pvinit[0][i];
        kb0[i] = a.kb[0][i];
    }
    for (int i = tid; i < a.ncodes * a.bpad[0]; i += BS) peq0[i] = a.peq[0][i];
    if (cfg.is_dual) {
        for (int i = tid; i <= B1; i += BS) off1[i] = cfg.pass[1].bc_off[i];
        for (int i = tid; i < B1; i += BS) {
            nn1[i] = cfg.pass[1].bc_len_no_N[i];
            pv1[i] = a.pvinit[1][i];
            kb1[i] = a.kb[1][i];
        }
        for (int i = tid; i < a.ncodes * a.bpad[1]; i += BS) peq1[i] = a.peq[1][i];
    }
    for (int i = tid; i < 256; i += BS) lut[i] = a.lut[i];
    for (int i = tid; i < a.hist_entries; i += BS) hist[i] = 0;
    for (int i = tid; i < R * (cw0 + cw1); i += BS) cand[i] = 0;
    for (int i = tid; i < 2 * R; i += BS) scnt[i] = 0;
    __syncthreads();
    const int bytes0 = (int)off0[B0];
    const int bytes1 = cfg.is_dual ? (int)off1[B1] : 0;
    for (int i = tid; i < bytes0; i += BS) bcs[i] = cfg.pass[0].bc_bytes[i];
    for (int i = tid; i < bytes1; i += BS) bcs[bytes0 + i] = cfg.pass[1].bc_bytes[i];

    // ---- this workgroup's reads [r0, r1): one contiguous span of the packed batch ----
    const long long r0 = (long long)blockIdx.x * R;
    long long r1 = r0 + R;
    if (r1 > a.n_reads) r1 = a.n_reads;
    const int nr = (int)(r1 - r0);
    const long long span0 = a.off[r0];
    const long long span1 = a.off[r1];
    const uintptr_t g0 = (uintptr_t)(a.seq + span0);
    const uintptr_t g0a = g0 & ~(uintptr_t)15;
    const int head = (int)(g0 - g0a);
    const long long need = (span1 - span0) + head;
    const bool staged = need + 16 <= (long long)a.stage_bytes;  // wave-uniform (whole workgroup)
    if (staged) {
        const int nvec = (int)((need + 15) >> 4);
        const u32x4 *src = (const u32x4 *)g0a;
        LDS u32x4 *dst = (LDS u32x4 *)rstage;
        for (int k = tid; k < nvec; k += BS) dst[k] = __builtin_nontemporal_load(src + k);
    }
    // per-read stage offsets and column windows of both passes
    for (int t = tid; t <= nr; t += BS) roff[t] = head + (int)(a.off[r0 + t] - span0);
    for (int t = tid; t < nr; t += BS) {
        const long long rn = a.off[r0 + t + 1] - a.off[r0 + t];
        const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
        for (int p = 0; p < npass; ++p) {
            PassWindow w;
            const bool ok = pass_window(cfg.pass[p], n, w);
            int f = ok ? (w.first > 1 ? w.first : 1) : 1;
            int l = ok ? (w.last < n ? w.last : n) : 0;
            win[(p * 2 + 0) * R + t] = f;
            win[(p * 2 + 1) * R + t] = l;
            // known-score class, per read: every column 1..n is swept and neither the start nor
            // the end range binds (band_offset = m-n-steps, min_valid_start <= 1, j >= min_end always)
            full[p * R + t] = (unsigned char)(a.known_ok[p] && ok && n > 0 && w.first == 1 && w.last == n &&
                                              w.max_start >= n && w.min_end <= 1);
        }
    }
    __syncthreads();

    if (staged) {
        // ---- transcode bytes -> symbol codes (4 per lane per step) ----
        const int nvec4 = (int)((need + 3) >> 2);
        for (int k = tid; k < nvec4; k += BS) {
            const uint32_t w = ((LDS uint32_t *)rstage)[k];
            const uint32_t c = (uint32_t)lut[w & 255] | ((uint32_t)lut[(w >> 8) & 255] << 8) |
                               ((uint32_t)lut[(w >> 16) & 255] << 16) | ((uint32_t)lut[w >> 24] << 24);
            ((LDS uint32_t *)codes)[k] = c;
        }
        __syncthreads();

        // ---- stage 1: Myers bit-vector sweep, one lane per (read, barcode) pair ----
        const bool sg = cfg.algorithm == BDX_ALG_SEMIGLOBAL;
        for (int p = 0; p < npass; ++p) {
            const int B = p ? B1 : B0;
            const int cw = p ? cw1 : cw0;
            const LDS uint32_t *peq = p ? peq1 : peq0;
            const LDS uint32_t *pv = p ? pv1 : pv0;
            const LDS int *kb = p ? kb1 : kb0;
            const int sh = a.bshift[p];  // log2(bytes per code row of peq)
            LDS uint32_t *cnd = cand + (p ? R * cw0 : 0);
            const LDS int *wf = win + (p * 2 + 0) * R;
            const LDS int *wl = win + (p * 2 + 1) * R;
            const int total = nr * B;
            // Two independent pairs per lane per trip (pair, pair + BS): the recurrence is a serial
            // chain of ~14 dependent VALU ops per column, so a second chain fills the issue slots
            // the first one leaves while waiting (the sweep is latency-bound at 2-4 waves/SIMD).
            struct Sweep {
                const LDS unsigned char *c;
                const LDS unsigned char *pq;
                uint32_t Pv, Mv;
                int score, best, ncol, r, b;
            };
            auto setup = [&](int pair, Sweep &w) {
                w.ncol = 0;
                w.r = 0;
                w.b = 0;
                w.c = codes;
                w.pq = (const LDS unsigned char *)peq;
                w.Pv = 0;
                w.Mv = 0;
                w.score = 0;
                w.best = 0x7FFFFFFF;
                if (pair >= total) return;
                const int r = pair / B;
                const int b = pair - r * B;
                const int jf = wf[r];
                int jl = wl[r];
                if (jl < jf) return;
                w.r = r;
                w.b = b;
                w.Pv = pv[b];
                w.score = __builtin_popcount(w.Pv);  // = barcode length m
                w.best = w.score;
                if (!sg) {
                    // :hamming / :exact bound the START positions by the window (SURVEY Q11,
                    // classification.jl:490-491, :570-571); the occurrence itself reaches m-1 further
                    const int nread = roff[r + 1] - roff[r];
                    jl = jl + w.score - 1 < nread ? jl + w.score - 1 : nread;
                }
                w.c = codes + roff[r] + (jf - 1);
                w.ncol = jl - jf + 1;
                w.pq = (const LDS unsigned char *)(peq + b);
            };
            auto step = [&](Sweep &w, int j) {
                const uint32_t Eq = *(const LDS uint32_t *)(w.pq + ((uint32_t)w.c[j] << sh));
                const uint32_t Xv = Eq | w.Mv;
                const uint32_t Xh = (((Eq & w.Pv) + w.Pv) ^ w.Pv) | Eq;
                uint32_t Ph = w.Mv | ~(Xh | w.Pv);
                uint32_t Mh = w.Pv & Xh;
                w.score += (int)(Ph >> 31) - (int)(Mh >> 31);
                Ph <<= 1;
                Mh <<= 1;
                w.Pv = Mh | ~(Xv | Ph);
                w.Mv = Ph & Xv;
                w.best = w.score < w.best ? w.score : w.best;
            };
            auto finish = [&](const Sweep &w) {
                if (w.ncol > 0 && w.best <= kb[w.b]) {
                    __hip_atomic_fetch_or(&cnd[w.r * cw + (w.b >> 5)], 1u << (w.b & 31), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (full[p * R + w.r]) {
                        const int k = __hip_atomic_fetch_add(&scnt[p * R + w.r], 1, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (k < 4) slots[(p * R + w.r) * 4 + k] = ((uint32_t)w.b << 8) | (uint32_t)w.best;
                    }
                }
            };
            for (int pair = tid; pair < ((a.dbg & 2) ? 0 : total); pair += 2 * BS) {
                Sweep A, Bw;
                setup(pair, A);
                setup(pair + BS, Bw);
                const int common = A.ncol < Bw.ncol ? A.ncol : Bw.ncol;
                int j = 0;
#pragma unroll 4
                for (; j < common; ++j) {
                    step(A, j);
                    step(Bw, j);
                }
                for (int ja = j; ja < A.ncol; ++ja) step(A, ja);
                for (int jb = j; jb < Bw.ncol; ++jb) step(Bw, jb);
                finish(A);
                finish(Bw);
            }
        }
        __syncthreads();
    }

    // ---- stage 2: exact evaluation, one lane per read ----
    const bool active = tid < nr;
    const long long ridx = r0 + tid;
    Verdict v{0, 0, -1, -1};
    PassOut p1{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()}, p2{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    if (active && !(a.dbg & 1)) {
        const long long ro = a.off[ridx];
        const long long rn = a.off[ridx + 1] - ro;
        const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
        LDS int *DP = DPbase + tid;
        LDS int *OG = OGbase + tid;
        if (staged) {
            const uint32_t *c0 = (const uint32_t *)(cand + tid * cw0);
            const uint32_t *c1 = (const uint32_t *)(cand + R * cw0 + tid * cw1);
            Bytes<true> r{rstage + roff[tid]};
            Bytes<true> q0{bcs}, q1{bcs + bytes0};
            KnownPass kn[2];
            for (int p = 0; p < 2; ++p) {
                const int cnt = scnt[p * R + tid];
                const LDS uint32_t *e = slots + (p * R + tid) * 4;
                // more than four survivors (or a read outside the class): exact evaluation instead
                kn[p] = KnownPass{p < npass && full[p * R + tid] && cnt <= 4, e[0], e[1], e[2], e[3], cnt};
            }
            classify_one<true>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, R, c0, c1, v, p1, p2, kn[0], kn[1]);
        } else {  // span larger than the staging area: unfiltered evaluation straight from HBM/L2
            Bytes<false> r{a.seq + ro};
            Bytes<false> q0{cfg.pass[0].bc_bytes}, q1{cfg.pass[1].bc_bytes};
            classify_one<false>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, R, nullptr, nullptr, v, p1, p2);
        }
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

    // ---- DemuxStats scalar counters ----
    if (a.counts) {
        int slot = -1;
        if (active) {
            if (v.bc1 > 0) slot = 4 + (v.bc1 - 1) * cfg.counts_stride2 + (v.bc2 > 0 ? v.bc2 - 1 : 0);
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

template <int BS, int R>
hipError_t launch_one(const BitparArgs &a, size_t lds, long long n_reads, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)bdx_bitpar_kernel<BS, R>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long long blocks = (n_reads + R - 1) / R;
    if (blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((bdx_bitpar_kernel<BS, R>), dim3((unsigned)blocks), dim3(BS), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

size_t bdx_bitpar_lds_bytes(const BdxDevCfg &cfg, const BdxBitparPlan &bp, const BdxGenericPlan &gp) {
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const int R = bp.reads_per_block;
    const int B0 = cfg.pass[0].n_barcodes, B1 = cfg.is_dual ? cfg.pass[1].n_barcodes : 0;
    const int cw0 = cfg.pass[0].cand_words, cw1 = cfg.is_dual ? cfg.pass[1].cand_words : 0;
    size_t o = 0;
    o += al((size_t)gp.dp_rows * R * 4);
    o += al(cfg.any_traceback ? (size_t)gp.dp_rows * R * 4 : 0);
    o += al((size_t)(B0 + 1) * 4) + al((size_t)(B1 + 1) * 4) + al((size_t)B0 * 4) + al((size_t)B1 * 4);
    o += al((size_t)gp.bc_stage_bytes) + al((size_t)gp.hist_entries * 4) + al(256);
    o += al((size_t)bp.ncodes * bp.bpad[0] * 4) + al(cfg.is_dual ? (size_t)bp.ncodes * bp.bpad[1] * 4 : 0);
    o += 2 * (al((size_t)B0 * 4) + al((size_t)B1 * 4));
    o += al((size_t)R * (cw0 + cw1) * 4) + al((size_t)(R + 1) * 4) + al((size_t)R * 16);
    o += al((size_t)2 * R * 16) + al((size_t)2 * R * 4) + al((size_t)2 * R);
    o += 2 * al((size_t)bp.stage_bytes);
    return o;
}

hipError_t bdx_launch_bitpar(const BdxDevCfg &cfg, const BdxGenericPlan &gp, const BdxBitparPlan &bp,
                             const uint8_t *d_seq, const long long *d_off, long long n_reads, const BdxDevOut &out,
                             unsigned long long *d_counts, hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    BitparArgs a;
    a.cfg = cfg;
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.out = out;
    a.counts = d_counts;
    a.dp_rows = gp.dp_rows;
    a.stage_bytes = bp.stage_bytes;
    a.bc_stage_bytes = gp.bc_stage_bytes;
    a.hist_entries = gp.hist_entries;
    a.lut = bp.d_lut;
    for (int k = 0; k < 2; ++k) {
        a.peq[k] = bp.d_peq[k];
        a.pvinit[k] = bp.d_pvinit[k];
        a.kb[k] = bp.d_kb[k];
        a.bpad[k] = bp.bpad[k];
        a.bshift[k] = 2;
        while ((4 << (a.bshift[k] - 2)) < bp.bpad[k] * 4) a.bshift[k]++;
    }
    a.ncodes = bp.ncodes;
    a.dbg = 0;
    a.known_ok[0] = bp.known_ok[0];
    a.known_ok[1] = bp.known_ok[1];
    if (const char *e = getenv("BDX_DEBUG")) a.dbg = atoi(e);
    const size_t lds = bdx_bitpar_lds_bytes(cfg, bp, gp);
    switch (bp.reads_per_block) {
        case 256:
            return launch_one<256, 256>(a, lds, n_reads, stream);
        case 128:
            return launch_one<256, 128>(a, lds, n_reads, stream);
        case 64:
            return launch_one<256, 64>(a, lds, n_reads, stream);
        case 32:
            return launch_one<256, 32>(a, lds, n_reads, stream);
        case 16:
            return launch_one<256, 16>(a, lds, n_reads, stream);
        default:
            return hipErrorInvalidValue;
    }
}
