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
#include "bdx_core.h"

#ifndef BDX_CLEAN_WAVES
#define BDX_CLEAN_WAVES 2  // waves per SIMD the clean-class exact kernel is compiled for: 2 = 208 VGPRs and NO scratch; 3 (<= 168 VGPRs)
                           // spills 176-432 B per lane (1.8 GB each way per 10 M reads of C4) for the same speed within 2 %
#endif

namespace {

struct GenericArgs {
    BdxDevCfg cfg;
    const uint8_t *seq;
    const long long *off;
    long long n_reads;
    BdxDevOut out;
    unsigned long long *counts;
    const uint32_t *cand0;  // [n_reads][cand_words] or null
    const uint32_t *cand1;
    const uint32_t *wins[2];  // [n_reads][BDX_WCAP][3] column-window entries per pass, or null
    const uint8_t *wcnt[2];   // [n_reads] entries valid (255: none -> whole window)
    int dp_rows;
    int stage_bytes;    // capacity of the read staging area (0: never stage)
    const uint32_t *list;             // list mode: the reads to evaluate (indices into the batch) ...
    const unsigned int *list_count;  // ... and how many (device-resident; the grid strides over them)
    int bc_stage_bytes; // bytes of the barcode staging area (both passes; 0: barcodes not staged)
    int hist_entries;
    BdxDevStats stats;  // rows == 0: no histograms
    // tiered budgets, split mode (bdx_abi.cpp): this launch evaluated the candidates of tier 1 (capped budgets);
    // reads whose verdict could depend on a barcode beyond the cap are appended to tier_list instead of being
    // answered
    int tier1;
    double tier_slo[2];
    uint32_t *tier_list;
    unsigned int *tier_count;
    // the LAST launch of a classify call clears the scratch words (tile queues, list lengths) the NEXT call will use — the other
    // half of the context's scratch block, which no kernel of this call touches (bdx_abi.cpp: ping-pong): no memset per call
    uint32_t *zero_words;
};

// Tier settle rule for one pass evaluated over the barcodes tier 1 can see (every b with unit distance <= its
// capped budget).  A barcode it cannot see scores >= slo; the alignment of a barcode does not depend on the
// running threshold beyond being accepted or not (in-domain costs, start / end ranges that do not bind for this
// read — checked by the caller), so the reducers (:632-713) over the visible set give the reference's answer iff
//   * a winner exists and min_score < slo strictly (no unseen barcode can win, or tie and come first), and
//   * no_delta, or the visible runner-up is <= slo (then it IS sub_min), or the bound delta >= fl(slo - min) already
//     proves "not ambiguous" and nobody asked for the delta value.
__device__ __forceinline__ bool tier_pass_settled(const BdxDevCfg &cfg, const PassOut &po, const double slo, const bool want_delta) {
    if (po.bc <= 0 || !(po.score < slo)) return false;
    if (cfg.min_delta == 0.0) return true;
    if (po.sub <= slo) return true;
    return !want_delta && po.status == 1 && (slo - po.score) >= cfg.min_delta;
}

// LDS carve-up (all 16-byte aligned):
//   [DP: dp_rows*BS int][OG: dp_rows*BS int if any_traceback][off0|off1: uint32][nn0|nn1: int]
//   [barcode bytes pass0|pass1][hist: int[hist_entries]][read bytes: stage_bytes]
// REGM > 0: the exact DP of SimpleScoring barcodes up to REGM rows runs register-resident
// (sg_core_reg); the kernel then needs no DP/origin columns in LDS.
// CLEAN: the clean-class register DP (sg_core_clean) — no predicated rows, <= 168 VGPRs, three workgroups per CU;
// UM: every barcode has exactly REGM rows.
template <int BS, int REGM, bool CLEAN = false, bool UM = false>
__global__ __launch_bounds__(BS, (BS == 256 ? (CLEAN ? BDX_CLEAN_WAVES : 2) : 1)) void bdx_generic_kernel(const GenericArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LDS unsigned char *smem = (LDS unsigned char *)smem_raw;
    const BdxDevCfg &cfg = a.cfg;
    const int tid = threadIdx.x;
    const int B0 = cfg.pass[0].n_barcodes;
    const int B1 = cfg.is_dual ? cfg.pass[1].n_barcodes : 0;
    if (a.zero_words && blockIdx.x == 0)
        for (int i = tid; i < BDX_SCRATCH_WORDS; i += BS) a.zero_words[i] = 0u;

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

    // Two ways to walk the batch:
    //  * dense (a.list == nullptr): workgroup b evaluates reads [b*BS, (b+1)*BS), staged as one span;
    //  * list: the reads named by a.list[0 .. *a.list_count) (the fused filter kernel's hand-overs) —
    //    scattered, so they are read straight from HBM/L2; the grid strides over the list.
    const bool listed = a.list != nullptr;
    long long total = a.n_reads;
    if (listed) {
        const long long c = (long long)*a.list_count;
        total = c < a.n_reads ? c : a.n_reads;
    }
    for (long long base = (long long)blockIdx.x * BS; base < total; base += listed ? (long long)gridDim.x * BS : total) {
    const long long r0 = base;
    long long r1 = r0 + BS;
    if (r1 > total) r1 = total;
    long long span0 = 0;
    int head = 0;
    bool staged = false;
    if (!listed) {
        span0 = a.off[r0];
        const long long span1 = a.off[r1];
        const uintptr_t g0 = (uintptr_t)(a.seq + span0);
        const uintptr_t g0a = g0 & ~(uintptr_t)15;
        head = (int)(g0 - g0a);
        const long long need = (span1 - span0) + head;
        staged = bc_staged && a.stage_bytes > 0 && need + 16 <= (long long)a.stage_bytes && cfg.vlen == nullptr;
        if (staged) {  // coalesced 16-B copy of the contiguous span, each HBM byte fetched once
            const int nvec = (int)((need + 15) >> 4);
            const GlobalVec16 src = (GlobalVec16)g0a;
            LDS u32x4 *dst = (LDS u32x4 *)rstage;
            for (int k = tid; k < nvec; k += BS) dst[k] = __builtin_nontemporal_load(src + k);
        }
        __syncthreads();
    }

    const bool active = r0 + tid < r1;
    const long long ridx = listed ? (active ? (long long)a.list[r0 + tid] : 0) : r0 + tid;
    Verdict v{0, 0, -1, -1};
    PassOut p1{0, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()}, p2{2, 0, -1, -1, -1, __builtin_inf(), __builtin_inf()};
    if (active) {
        // (window upload: ro is where position 0 of the read WOULD be; only window columns are ever touched)
        const long long ro = a.off[ridx] - (cfg.vlen ? (long long)cfg.vlo[ridx] : 0);
        const long long rn = cfg.vlen ? (long long)cfg.vlen[ridx] : a.off[ridx + 1] - a.off[ridx];
        const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
        const uint32_t *c0 = a.cand0 ? a.cand0 + ridx * cfg.pass[0].cand_words : nullptr;
        const uint32_t *c1 = a.cand1 ? a.cand1 + ridx * cfg.pass[1].cand_words : nullptr;
        LDS int *DP = DPbase + tid;
        LDS int *OG = OGbase + tid;
        const uint32_t *we0 = a.wins[0] ? a.wins[0] + ridx * (cfg.dense_w ? (long long)B0 : (long long)(BDX_WCAP * 3)) : nullptr;
        const uint32_t *we1 = a.wins[1] ? a.wins[1] + ridx * (cfg.dense_w ? (long long)B1 : (long long)(BDX_WCAP * 3)) : nullptr;
        const int wc0 = a.wcnt[0] ? (int)a.wcnt[0][ridx] : 255;
        const int wc1 = a.wcnt[1] ? (int)a.wcnt[1][ridx] : 255;
        const KnownPass nokn{false, 0, 0, 0, 0, 0};
        if (staged) {
            Bytes<true> r{rstage + head + (ro - span0)};
            Bytes<true> q0{bcs}, q1{bcs + bytes0};
            classify_one<true, REGM, CLEAN, UM>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, BS, c0, c1, v, p1, p2, nokn, nokn,
                                                0x4E, we0, wc0, we1, wc1);
        } else {
            Bytes<false> r{a.seq + ro};
            Bytes<false> q0{cfg.pass[0].bc_bytes}, q1{cfg.pass[1].bc_bytes};
            classify_one<false, REGM, CLEAN, UM>(cfg, q0, q1, off0, off1, nn0, nn1, r, n, DP, OG, BS, c0, c1, v, p1, p2, nokn, nokn,
                                                 0x4E, we0, wc0, we1, wc1);
        }
    }
    bool answer = active;
    if (a.tier1) {  // (workgroup-uniform)
        bool settled = false;
        if (active) {
            const long long rn = cfg.vlen ? (long long)cfg.vlen[ridx] : a.off[ridx + 1] - a.off[ridx];
            const int n = (int)(rn > (1LL << 30) ? (1LL << 30) : rn);
            const bool sgm = cfg.algorithm == BDX_ALG_SEMIGLOBAL;
            const auto free_ranges = [&](const BdxDevPass &P) {  // neither the start nor the end range binds for this read
                PassWindow w;
                return !sgm || (pass_window(P, n, w) && w.max_start >= n && w.min_end <= 1);
            };
            const bool want_delta = a.out.pass_delta != nullptr;
            settled = n > 0 && free_ranges(cfg.pass[0]) && tier_pass_settled(cfg, p1, a.tier_slo[0], want_delta);
            if (settled && cfg.is_dual && p1.status == 1)
                settled = free_ranges(cfg.pass[1]) && tier_pass_settled(cfg, p2, a.tier_slo[1], want_delta);
        }
        const bool hand = active && !settled;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(hand);
        if (mk) {  // one queue reservation per wave
            const int lane = tid & 63;
            const int leader = __builtin_ctzll(mk);
            unsigned int basek = 0;
            if (lane == leader) basek = atomicAdd(a.tier_count, (unsigned int)__builtin_popcountll(mk));
            basek = (unsigned int)__shfl((int)basek, leader, 64);
            if (hand) a.tier_list[basek + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (uint32_t)ridx;
        }
        answer = active && settled;
    }
    if (answer) {
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
        if (a.stats.rows > 0) {
            stats_update(a.stats, 0, B0, p1);
            if (cfg.is_dual) stats_update(a.stats, 1, B1, p2);
        }
    }

    // DemuxStats scalar counters (classification.jl:942-978), merged like reporting.jl:1-9
    if (a.counts) {
        int slot = -1;
        if (answer) {
            if (v.bc1 > 0)
                slot = 4 + (v.bc1 - 1) * cfg.counts_stride2 + (v.bc2 > 0 ? v.bc2 - 1 : 0);
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
    }
    }  // dense: one trip; list: grid stride
    if (a.counts) {
        __syncthreads();
        for (int i = tid; i < a.hist_entries; i += BS) {
            const int h = hist[i];
            if (h) atomicAdd(&a.counts[i], (unsigned long long)h);
        }
    }
}

}  // namespace

namespace {
__global__ void bdx_maxlen_kernel(const long long *off, long long n, int *out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int v = 0;
    for (; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long d = off[i + 1] - off[i];
        const int x = d > 0x3FFFFFFF ? 0x3FFFFFFF : (int)d;
        v = x > v ? x : v;
    }
    for (int s = 32; s > 0; s >>= 1) {
        const int o = __shfl_xor(v, s, 64);
        v = o > v ? o : v;
    }
    // one atomic per workgroup: atomics on ONE address retire one every ~10 ns (8192 of them were 80 of this kernel's 97 us)
    __shared__ int wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) v = wmax[w] > v ? wmax[w] : v;
        if (v > 0) atomicMax(out, v);
    }
}
}  // namespace

namespace {
// page-locked host memory -> device memory, 16 bytes per lane and trip.  The host entry point uses it for small
// batches instead of hipMemcpyAsync: copies of a few hundred KB through the copy engines cost ~35 us each and
// serialise between the contexts of concurrent worker threads; a kernel runs on the context's own stream.
typedef uint32_t bdx_copy_v4 __attribute__((ext_vector_type(4)));
__global__ void bdx_copy_kernel(bdx_copy_v4 *dst, const bdx_copy_v4 *src, long long n16, uint32_t *zero, int zero_words) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x)
        dst[i] = __builtin_nontemporal_load(src + i);
    // (the launch also clears the filter kernels' scratch words: one stream operation less per small batch)
    if (blockIdx.x == 0 && zero)
        for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zero[i] = 0u;
}
// TEST SWITCH (BDX_POISON): between a producer and its consumer, look at what the producer handed over — every
// element the consumer is going to read must have been WRITTEN (the buffers were filled with 0xA5 before the call):
//   * a list: entries [0, *count) — an unwritten one is counted and replaced by read 0 (a wild index would be followed);
//   * window hand-over of one pass, for the reads of this launch (all of them, or the listed ones): the count byte
//     itself, the entries it announces (barcode field), or — dense table (254) — the entry of every candidate barcode;
//     a read with an unwritten element is counted and set to "no windows" (255).
// Violations are added to dbg[1]; the suite fails when it is not 0 (bdx_rejected_windows).
__global__ void bdx_poison_check_kernel(uint32_t *list, const unsigned int *list_count, long long n_reads, const uint32_t *wins, uint8_t *wcnt,
                                        const uint32_t *cand, int cand_words, int n_barcodes, int check_list, unsigned int *dbg) {
    const long long n = list ? (long long)(*list_count < (unsigned long long)n_reads ? *list_count : n_reads) : n_reads;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long rid = i;
        if (list) {
            uint32_t e = list[i];
            // (check_list bit 1: the entries carry two state bits above a 30-bit read number — carried passes, bdx_wave.hip)
            const bool unwritten = e == 0xA5A5A5A5u;
            if ((check_list & 2) && !unwritten) e &= 0x3FFFFFFFu;
            if ((check_list & 1) && (unwritten || (long long)e >= n_reads)) {
                atomicAdd(dbg + 1, 1u);
                list[i] = 0u;
                e = 0u;
            }
            rid = e;
        }
        if (!wcnt || !wins) continue;
        const unsigned w = wcnt[rid];
        bool bad = false;
        if (w == 0xA5u) {
            bad = true;
        } else if (w <= (unsigned)BDX_WCAP) {
            for (unsigned k = 0; k < w; ++k) bad |= wins[((size_t)rid * BDX_WCAP + k) * 3] == 0xA5A5A5A5u;
        } else if (w == 254u && cand) {
            for (int cw = 0; cw < cand_words; ++cw) {
                uint32_t bits = cand[(size_t)rid * cand_words + cw];
                while (bits) {
                    const int b = cw * 32 + __builtin_ctz(bits);
                    bits &= bits - 1u;
                    if (b < n_barcodes) bad |= wins[(size_t)rid * n_barcodes + b] == 0xA5A5A5A5u;
                }
            }
        }
        if (bad) {
            atomicAdd(dbg + 1, 1u);
            wcnt[rid] = 255;
        }
    }
}
}  // namespace

hipError_t bdx_launch_poison_check(uint32_t *list, const unsigned int *list_count, long long n_reads, const uint32_t *wins, uint8_t *wcnt,
                                   const uint32_t *cand, int cand_words, int n_barcodes, int check_list, unsigned int *dbg, hipStream_t stream) {
    if (n_reads <= 0 || !dbg) return hipSuccess;
    long long blocks = (n_reads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bdx_poison_check_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, list, list_count, n_reads, wins, wcnt, cand, cand_words,
                       n_barcodes, check_list, dbg);
    return hipGetLastError();
}

hipError_t bdx_launch_copy(void *d_dst, const void *src_mapped, size_t bytes, hipStream_t stream, void *d_zero, int zero_bytes) {
    const long long n16 = (long long)((bytes + 15) / 16);
    if (n16 <= 0) return hipSuccess;
    long long blocks = (n16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bdx_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bdx_copy_v4 *)d_dst, (const bdx_copy_v4 *)src_mapped, n16,
                       (uint32_t *)d_zero, zero_bytes / 4);
    return hipGetLastError();
}

hipError_t bdx_launch_maxlen(const long long *d_off, long long n_reads, int *d_out, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    if (n_reads <= 0) return hipSuccess;
    long long blocks = (n_reads + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(bdx_maxlen_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_off, n_reads, d_out);
    return hipGetLastError();
}

template <int BS, int REGM, bool CLEAN = false, bool UM = false>
static hipError_t generic_attr(size_t bytes) {
    return hipFuncSetAttribute((const void *)bdx_generic_kernel<BS, REGM, CLEAN, UM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t bdx_generic_set_lds_limit(size_t bytes) {
    hipError_t e;
    if ((e = generic_attr<256, 0>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<128, 0>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<64, 0>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 0, true>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<128, 0, true>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<64, 0, true>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 24>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 32>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 24, true, true>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 24, true, false>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 32, true, true>(bytes)) != hipSuccess) return e;
    if ((e = generic_attr<256, 32, true, false>(bytes)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t bdx_launch_generic(const BdxDevCfg &cfg, const BdxGenericPlan &plan, const uint8_t *d_seq,
                              const long long *d_off, long long n_reads, const BdxDevOut &out,
                              unsigned long long *d_counts, const uint32_t *d_cand0, const uint32_t *d_cand1,
                              hipStream_t stream, const uint32_t *d_wins0, const uint32_t *d_wins1,
                              const uint8_t *d_wcnt0, const uint8_t *d_wcnt1, const uint32_t *d_list,
                              const unsigned int *d_list_count, const BdxDevStats *stats, const BdxTierArgs *tier,
                              const double *tier_slo, uint32_t *zero_words) {
    if (n_reads <= 0) return hipSuccess;
    GenericArgs a;
    a.zero_words = zero_words;
    a.cfg = cfg;
    a.seq = d_seq;
    a.off = d_off;
    a.n_reads = n_reads;
    a.out = out;
    a.counts = d_counts;
    a.cand0 = d_cand0;
    a.cand1 = d_cand1;
    a.wins[0] = d_wins0;
    a.wins[1] = d_wins1;
    a.wcnt[0] = d_wcnt0;
    a.wcnt[1] = d_wcnt1;
    a.list = d_list;
    a.list_count = d_list_count;
    if (stats)
        a.stats = *stats;
    else
        a.stats = BdxDevStats{};
    a.tier1 = tier && tier->tier1 ? 1 : 0;
    a.tier_slo[0] = tier_slo ? tier_slo[0] : 0.0;
    a.tier_slo[1] = tier_slo ? tier_slo[1] : 0.0;
    a.tier_list = tier ? tier->out_list : nullptr;
    a.tier_count = tier ? tier->out_count : nullptr;
    a.cfg.end_only_ok = out.pass_start == nullptr ? 1 : 0;
    a.dp_rows = plan.dp_rows;
    a.stage_bytes = plan.stage_bytes;
    a.bc_stage_bytes = plan.bc_stage_bytes;
    a.hist_entries = plan.hist_entries;
    long long blocks = (n_reads + plan.threads - 1) / plan.threads;
    if (blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const long long list_grid = 4LL * (plan.n_cu > 0 ? plan.n_cu : 256);
    if (d_list && blocks > list_grid) blocks = list_grid;  // list mode: the hand-overs are few; the grid strides
    const dim3 grid((unsigned)blocks), block((unsigned)plan.threads);
    if (plan.reg_rows == 24 && plan.threads == 256 && plan.clean) {
        if (plan.uniform_m)
            hipLaunchKernelGGL((bdx_generic_kernel<256, 24, true, true>), grid, block, plan.lds_bytes, stream, a);
        else
            hipLaunchKernelGGL((bdx_generic_kernel<256, 24, true, false>), grid, block, plan.lds_bytes, stream, a);
    } else if (plan.reg_rows == 32 && plan.threads == 256 && plan.clean) {
        if (plan.uniform_m)
            hipLaunchKernelGGL((bdx_generic_kernel<256, 32, true, true>), grid, block, plan.lds_bytes, stream, a);
        else
            hipLaunchKernelGGL((bdx_generic_kernel<256, 32, true, false>), grid, block, plan.lds_bytes, stream, a);
    } else if (plan.reg_rows == 24 && plan.threads == 256) {
        hipLaunchKernelGGL((bdx_generic_kernel<256, 24>), grid, block, plan.lds_bytes, stream, a);
    } else if (plan.reg_rows == 32 && plan.threads == 256) {
        hipLaunchKernelGGL((bdx_generic_kernel<256, 32>), grid, block, plan.lds_bytes, stream, a);
    } else if (plan.band_roll) {  // barcodes beyond 32 rows in the clean class: the rolling diagonal band (sg_band_roll)
        switch (plan.threads) {
            case 256:
                hipLaunchKernelGGL((bdx_generic_kernel<256, 0, true>), grid, block, plan.lds_bytes, stream, a);
                break;
            case 128:
                hipLaunchKernelGGL((bdx_generic_kernel<128, 0, true>), grid, block, plan.lds_bytes, stream, a);
                break;
            case 64:
                hipLaunchKernelGGL((bdx_generic_kernel<64, 0, true>), grid, block, plan.lds_bytes, stream, a);
                break;
            default:
                return hipErrorInvalidValue;
        }
    } else {
        switch (plan.threads) {
            case 256:
                hipLaunchKernelGGL((bdx_generic_kernel<256, 0>), grid, block, plan.lds_bytes, stream, a);
                break;
            case 128:
                hipLaunchKernelGGL((bdx_generic_kernel<128, 0>), grid, block, plan.lds_bytes, stream, a);
                break;
            case 64:
                hipLaunchKernelGGL((bdx_generic_kernel<64, 0>), grid, block, plan.lds_bytes, stream, a);
                break;
            default:
                return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}
