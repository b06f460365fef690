/* ASan/UBSan driver for the CPU-side native code (tests/test_sanitizers.py builds and runs it):
 * the oracle (self-test + batch classify over ragged synthetic reads, threads) and the synthetic
 * generator.  Exit code 0 = no sanitizer report and the internal checks hold. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../oracle/bdx_oracle.h"

void bdx_synth_chunk(uint64_t seed, uint64_t tag, uint64_t chunk_id, int64_t n_reads, int32_t read_len,
                     const uint8_t *bc_bytes, const int64_t *bc_off, int32_t n_bc, double plant_frac, double sub,
                     double ins, double del, double n_rate, int32_t plant_lo, int32_t plant_hi, int32_t plant_only,
                     uint8_t *reads, int32_t *truth);

int main(void) {
    int64_t fb[6];
    if (orc_selftest_known_class(99, 60000, fb) != 0) {
        fprintf(stderr, "selftest mismatch\n");
        return 2;
    }
    enum { B = 12, M = 18, N = 4000, L = 90 };
    uint8_t bcs[B * M];
    int64_t off[B + 1], nn[B];
    uint64_t s = 12345;
    for (int i = 0; i < B * M; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        bcs[i] = (uint8_t)"ACGT"[(s >> 33) & 3];
    }
    for (int i = 0; i <= B; i++) off[i] = (int64_t)i * M;
    for (int i = 0; i < B; i++) nn[i] = M;
    uint8_t *reads = malloc((size_t)N * L);
    int32_t *truth = malloc(sizeof(int32_t) * N);
    bdx_synth_chunk(7, 1, 0, N, L, bcs, off, B, 0.9, 0.03, 0.01, 0.01, 0.002, 0, -1, 0, reads, truth);
    /* ragged: cut every read to a pseudo-random length (incl. 0) by building offsets into a packed copy */
    uint8_t *packed = malloc((size_t)N * L + 1);
    int64_t *roff = malloc(sizeof(int64_t) * (N + 1));
    roff[0] = 0;
    for (int i = 0; i < N; i++) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        int len = (int)((s >> 33) % (L + 1));
        memcpy(packed + roff[i], reads + (size_t)i * L, (size_t)len);
        roff[i + 1] = roff[i] + len;
    }
    for (int variant = 0; variant < 6; variant++) {
        orc_config_t c;
        memset(&c, 0, sizeof c);
        c.algorithm = variant == 4 ? ORC_ALG_HAMMING : (variant == 5 ? ORC_ALG_EXACT : ORC_ALG_SEMIGLOBAL);
        c.max_error_rate = variant == 1 ? 0.25 : 0.2;
        c.min_delta = variant == 2 ? 0.1 : 0.0;
        c.match = 0;
        c.mismatch = 1;
        c.indel = variant == 1 ? 2 : 1;
        c.has_nindel = variant == 3;
        c.nindel = 1;
        c.is_dual = variant == 2;
        c.summary = variant == 3;
        for (int p = 0; p < 2; p++) {
            orc_range_t full = {1, 0, 0, 1};
            c.pass[p].ref_search_range = full;
            c.pass[p].barcode_start_range = full;
            c.pass[p].barcode_end_range = full;
            c.pass[p].trim_side = (variant == 0) ? 0 : (p == 0 ? 5 : 3);
            c.pass[p].n_barcodes = B;
            c.pass[p].bc_bytes = bcs;
            c.pass[p].bc_off = off;
            c.pass[p].bc_len_no_N = nn;
        }
        if (variant == 1) {
            orc_range_t w = {-40, 1, 0, 1}; /* end-40:end */
            c.pass[0].ref_search_range = w;
        }
        int32_t *bc1 = malloc(sizeof(int32_t) * N), *bc2 = malloc(sizeof(int32_t) * N);
        int32_t *ks = malloc(sizeof(int32_t) * N), *ke = malloc(sizeof(int32_t) * N);
        int32_t *ps = malloc(sizeof(int32_t) * 2 * N), *pe = malloc(sizeof(int32_t) * 2 * N), *pb = malloc(sizeof(int32_t) * 2 * N);
        double *psc = malloc(sizeof(double) * 2 * N), *pd = malloc(sizeof(double) * 2 * N);
        int64_t counts[4 + B * B];
        memset(counts, 0, sizeof counts);
        orc_classify_batch(&c, packed, roff, N, bc1, bc2, ks, ke, ps, pe, psc, pb, pd, counts, 4);
        if (counts[0] != N || counts[1] + counts[2] + counts[3] != N) {
            fprintf(stderr, "counter invariant broken (variant %d)\n", variant);
            return 3;
        }
        free(bc1); free(bc2); free(ks); free(ke); free(ps); free(pe); free(pb); free(psc); free(pd);
    }
    free(reads); free(truth); free(packed); free(roff);
    printf("sanitize driver ok\n");
    return 0;
}
