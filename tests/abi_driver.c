/*
 * abi_driver.c — a plain C11 host of libbiodemux_hip.so (test infrastructure).
 *
 * What a non-Python host sees: it includes include/biodemux_hip.h, nothing else of this repository.
 *   1. compile time: the struct layouts the Julia shim in INTEGRATION.md §2 spells out by hand
 *      (BdxRange / BdxPass / BdxConfig / BdxOutputs, natural alignment) are pinned with offsetof;
 *   2. `abi_driver --layout`: prints the sizes, checks bdx_abi_version and that bdx_create refuses a bad
 *      config with the reference's message (no GPU needed);
 *   3. `abi_driver <barcodes.txt> <reads.fastq> <max_error_rate> [trim_side]`: classifies a FASTQ file
 *      through bdx_create / bdx_classify_host / bdx_get_counts, then runs the merge_stats sequence of a
 *      one-process host (bdx_comm_init_all over its single context -> RCCL, bdx_allreduce_counts_all,
 *      bdx_get_reduced_counts) and prints one line per read: "bc1 bc2 keep_start keep_end", followed by
 *      "counts: ..." and "reduced: ...".  barcodes.txt holds one preprocessed barcode per line.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "biodemux_hip.h"

/* ---- layout the Julia structs of INTEGRATION.md assume ---- */
_Static_assert(sizeof(bdx_range_t) == 24, "BdxRange");
_Static_assert(offsetof(bdx_range_t, start_offset) == 0 && offsetof(bdx_range_t, end_offset) == 8 &&
               offsetof(bdx_range_t, start_from_end) == 16 && offsetof(bdx_range_t, end_from_end) == 20, "BdxRange fields");
_Static_assert(offsetof(bdx_pass_t, ref_search_range) == 0 && offsetof(bdx_pass_t, barcode_start_range) == 24 &&
               offsetof(bdx_pass_t, barcode_end_range) == 48 && offsetof(bdx_pass_t, trim_side) == 72 &&
               offsetof(bdx_pass_t, n_barcodes) == 76 && offsetof(bdx_pass_t, bc_bytes) == 80 &&
               offsetof(bdx_pass_t, bc_off) == 88 && offsetof(bdx_pass_t, bc_len_no_N) == 96 &&
               offsetof(bdx_pass_t, explicit_window) == 104 && offsetof(bdx_pass_t, win_first) == 112 &&
               offsetof(bdx_pass_t, win_last) == 120 && offsetof(bdx_pass_t, win_max_start_pos) == 128 &&
               offsetof(bdx_pass_t, win_min_end_pos) == 136, "BdxPass fields");
_Static_assert(sizeof(bdx_pass_t) == 144, "BdxPass");
_Static_assert(offsetof(bdx_config_t, abi_version) == 0 && offsetof(bdx_config_t, struct_size) == 4 &&
               offsetof(bdx_config_t, algorithm) == 8 && offsetof(bdx_config_t, is_dual) == 12 &&
               offsetof(bdx_config_t, max_error_rate) == 16 && offsetof(bdx_config_t, min_delta) == 24 &&
               offsetof(bdx_config_t, match) == 32 && offsetof(bdx_config_t, mismatch) == 36 &&
               offsetof(bdx_config_t, indel) == 40 && offsetof(bdx_config_t, has_nindel) == 44 &&
               offsetof(bdx_config_t, nindel) == 48 && offsetof(bdx_config_t, need_traceback) == 52 &&
               offsetof(bdx_config_t, filter) == 56 && offsetof(bdx_config_t, device) == 60 &&
               offsetof(bdx_config_t, pass) == 64, "BdxConfig fields");
_Static_assert(sizeof(bdx_config_t) == 64 + 2 * 144, "BdxConfig");
_Static_assert(sizeof(bdx_outputs_t) == 10 * sizeof(void *), "BdxOutputs");
_Static_assert(offsetof(bdx_outputs_t, bc1) == 0 && offsetof(bdx_outputs_t, keep_end) == 24 &&
               offsetof(bdx_outputs_t, pass_score) == 56 && offsetof(bdx_outputs_t, pass_delta) == 72, "BdxOutputs fields");
_Static_assert(BDX_COMM_ID_BYTES == 128, "unique id size");

typedef struct {
    uint8_t *bytes;
    size_t nbytes, cap;
    int64_t *off;
    size_t n, ncap;
} packed_t;

static void push(packed_t *p, const char *s, size_t len) {
    if (p->nbytes + len + 1 > p->cap) {
        p->cap = (p->cap + len + 1) * 2;
        p->bytes = realloc(p->bytes, p->cap);
    }
    if (p->n + 2 > p->ncap) {
        p->ncap = (p->ncap + 2) * 2;
        p->off = realloc(p->off, p->ncap * sizeof(int64_t));
    }
    if (p->n == 0) p->off[0] = 0;
    memcpy(p->bytes + p->nbytes, s, len);
    p->nbytes += len;
    p->off[++p->n] = (int64_t)p->nbytes;
}

static size_t chomp(char *line) {
    size_t len = strlen(line);
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    return len;
}

static bdx_range_t full_range(void) { /* "1:end" (classification.jl:61-94) */
    bdx_range_t r = {1, 0, 0, 1};
    return r;
}

static void fill_config(bdx_config_t *c, const packed_t *bc, const uint32_t *off32, const int32_t *nn, double rate, int trim) {
    memset(c, 0, sizeof *c);
    c->abi_version = BDX_ABI_VERSION;
    c->struct_size = (uint32_t)sizeof *c;
    c->algorithm = BDX_ALG_SEMIGLOBAL;
    c->max_error_rate = rate;
    c->min_delta = 0.0;
    c->match = 0;
    c->mismatch = 1;
    c->indel = 1;
    c->filter = BDX_FILTER_AUTO;
    for (int k = 0; k < 2; ++k) {
        c->pass[k].ref_search_range = full_range();
        c->pass[k].barcode_start_range = full_range();
        c->pass[k].barcode_end_range = full_range();
    }
    c->pass[0].trim_side = trim;
    c->pass[0].n_barcodes = (int32_t)bc->n;
    c->pass[0].bc_bytes = bc->bytes;
    c->pass[0].bc_off = off32;
    c->pass[0].bc_len_no_N = nn;
}

static int layout(void) {
    printf("sizeof: range %zu pass %zu config %zu outputs %zu launch_info %zu\n", sizeof(bdx_range_t), sizeof(bdx_pass_t),
           sizeof(bdx_config_t), sizeof(bdx_outputs_t), sizeof(bdx_launch_info_t));
    if (bdx_abi_version() != BDX_ABI_VERSION) {
        printf("abi version mismatch\n");
        return 1;
    }
    /* validation happens before any device is touched: the reference's trim_side check (core.jl:308-313) */
    packed_t bc = {0};
    push(&bc, "ACGT", 4);
    uint32_t off32[2] = {0, 4};
    int32_t nn[1] = {4};
    bdx_config_t c;
    fill_config(&c, &bc, off32, nn, 0.2, 4);
    bdx_ctx *ctx = (bdx_ctx *)0x1;
    int32_t rc = bdx_create(&c, &ctx);
    printf("bdx_create(trim_side=4) -> %d, ctx %s, \"%s\"\n", rc, ctx ? "non-null" : "NULL", bdx_last_error(NULL));
    if (rc != BDX_E_INVALID || ctx != NULL || !strstr(bdx_last_error(NULL), "trim_side must be 3 or 5")) return 1;
    c.struct_size = 8;
    c.pass[0].trim_side = 0;
    rc = bdx_create(&c, &ctx);
    printf("bdx_create(struct_size=8) -> %d, \"%s\"\n", rc, bdx_last_error(NULL));
    if (rc != BDX_E_INVALID) return 1;
    printf("layout ok\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "--layout") == 0) return layout();
    if (argc < 4) {
        fprintf(stderr, "usage: %s --layout | <barcodes.txt> <reads.fastq> <max_error_rate> [trim_side]\n", argv[0]);
        return 2;
    }
    char line[1 << 16];
    packed_t bc = {0}, rd = {0};
    FILE *f = fopen(argv[1], "r");
    if (!f) return perror(argv[1]), 2;
    while (fgets(line, sizeof line, f)) {
        size_t len = chomp(line);
        if (len) push(&bc, line, len);
    }
    fclose(f);
    f = fopen(argv[2], "r");
    if (!f) return perror(argv[2]), 2;
    for (long ln = 0; fgets(line, sizeof line, f); ++ln)
        if (ln % 4 == 1) push(&rd, line, chomp(line)); /* the sequence line of every FASTQ record */
    fclose(f);
    uint32_t *off32 = malloc((bc.n + 1) * sizeof *off32);
    int32_t *nn = malloc(bc.n * sizeof *nn);
    for (size_t i = 0; i <= bc.n; ++i) off32[i] = (uint32_t)bc.off[i];
    for (size_t i = 0; i < bc.n; ++i) {
        nn[i] = 0;
        for (int64_t j = bc.off[i]; j < bc.off[i + 1]; ++j) nn[i] += bc.bytes[j] != 'N';
    }
    bdx_config_t c;
    fill_config(&c, &bc, off32, nn, atof(argv[3]), argc > 4 ? atoi(argv[4]) : 0);
    bdx_ctx *ctx = NULL;
    if (bdx_create(&c, &ctx) != BDX_OK) {
        fprintf(stderr, "bdx_create: %s\n", bdx_last_error(NULL));
        return 1;
    }
    const size_t n = rd.n;
    int32_t *out = malloc(4 * n * sizeof *out + 16);
    bdx_outputs_t o;
    memset(&o, 0, sizeof o);
    o.bc1 = out;
    o.bc2 = out + n;
    o.keep_start = out + 2 * n;
    o.keep_end = out + 3 * n;
    if (bdx_classify_host(ctx, rd.bytes, rd.off, (int64_t)n, &o) != BDX_OK) {
        fprintf(stderr, "bdx_classify_host: %s\n", bdx_last_error(ctx));
        return 1;
    }
    for (size_t i = 0; i < n; ++i) printf("%d %d %d %d\n", o.bc1[i], o.bc2[i], o.keep_start[i], o.keep_end[i]);
    const int64_t nc = bdx_counts_len(ctx);
    int64_t *counts = malloc((size_t)nc * sizeof *counts), *sum = malloc((size_t)nc * sizeof *sum);
    if (bdx_get_counts(ctx, counts, nc) != BDX_OK) return 1;
    printf("counts:");
    for (int64_t i = 0; i < nc; ++i) printf(" %lld", (long long)counts[i]);
    printf("\n");
    /* merge_stats (reporting.jl:1-9) of a one-process host: communicator over its contexts, grouped all-reduce */
    bdx_ctx *all[1] = {ctx};
    if (bdx_comm_init_all(all, 1) != BDX_OK || bdx_allreduce_counts_all(all, 1) != BDX_OK ||
        bdx_get_reduced_counts(ctx, sum, nc) != BDX_OK) {
        fprintf(stderr, "merge_stats over RCCL: %s\n", bdx_last_error(ctx));
        return 1;
    }
    printf("reduced (%d rank, rank %d):", bdx_comm_size(ctx), bdx_comm_rank(ctx));
    for (int64_t i = 0; i < nc; ++i) printf(" %lld", (long long)sum[i]);
    printf("\n");
    bdx_destroy(ctx);
    return 0;
}
