/*
 * bdx_synth.c — seeded synthetic read generator (host utility for tests and bench.py; not on
 * the hot path).  Shapes follow SURVEY.md §8(d): uniform {A,C,G,T} background, a planted,
 * per-base mutated barcode in `plant_frac` of the reads, a sprinkle of 'N'.
 *
 * Determinism: one xoshiro256** stream per (seed, tag, chunk_id); chunks are independent, so
 * any shard of the global stream can be generated on its own (multi-GPU ranks).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    uint64_t s[4];
} rng_t;

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t splitmix(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void rng_init(rng_t *r, uint64_t seed, uint64_t tag, uint64_t chunk) {
    uint64_t x = seed ^ (tag * 0xD1342543DE82EF95ULL) ^ (chunk * 0xA24BAED4963EE407ULL);
    for (int i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}

static inline uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s;
    const uint64_t result = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static inline uint32_t rng_below(rng_t *r, uint32_t n) { /* unbiased enough for test data */
    return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};

static inline int code_of(uint8_t c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

/*
 * Fills (or, with plant_only != 0, only plants into) n_reads fixed-length reads.
 *   reads      uint8[n_reads * read_len]
 *   truth      int32[n_reads] (1-based planted barcode, 0 = none); may be NULL
 *   plant_hi   < 0: no upper bound on the (0-based) start other than fitting the read
 */
void bdx_synth_chunk(uint64_t seed, uint64_t tag, uint64_t chunk_id, int64_t n_reads, int32_t read_len,
                     const uint8_t *bc_bytes, const int64_t *bc_off, int32_t n_bc, double plant_frac, double sub,
                     double ins, double del, double n_rate, int32_t plant_lo, int32_t plant_hi,
                     int32_t plant_only, uint8_t *reads, int32_t *truth) {
    rng_t R;
    rng_init(&R, seed, tag, chunk_id);
    const int n = read_len;
    uint8_t buf[4096];
    const double log1mp = (n_rate > 0 && n_rate < 1) ? log(1.0 - n_rate) : 0.0;
    for (int64_t i = 0; i < n_reads; i++) {
        uint8_t *rd = reads + i * (int64_t)n;
        if (!plant_only) {
            int j = 0;
            while (j < n) {
                uint64_t w = rng_next(&R);
                for (int k = 0; k < 32 && j < n; k++, j++, w >>= 2) rd[j] = ACGT[w & 3];
            }
        }
        int32_t t = 0;
        if (n_bc > 0 && rng_unit(&R) < plant_frac) {
            const int b = (int)rng_below(&R, (uint32_t)n_bc);
            t = b + 1;
            const uint8_t *bc = bc_bytes + bc_off[b];
            const int m = (int)(bc_off[b + 1] - bc_off[b]);
            int L = 0;
            for (int k = 0; k < m && L < (int)sizeof(buf) - 2; k++) {
                const double u = rng_unit(&R);
                uint8_t c = bc[k];
                if (u < sub) { /* substitute by a different base */
                    const int cc = code_of(c);
                    c = ACGT[((cc < 0 ? 0 : cc) + 1 + (int)rng_below(&R, 3)) & 3];
                    buf[L++] = c;
                } else if (u < sub + del) {
                    /* deleted */
                } else {
                    buf[L++] = c;
                }
                if (rng_unit(&R) < ins) buf[L++] = ACGT[rng_below(&R, 4)];
            }
            int hi = n - L;
            if (plant_hi >= 0 && plant_hi < hi) hi = plant_hi;
            int lo = plant_lo;
            if (hi < 0) hi = 0;
            if (lo > hi) lo = hi;
            int start = lo + (int)rng_below(&R, (uint32_t)(hi - lo + 1));
            for (int k = 0; k < L && start + k < n; k++) rd[start + k] = buf[k];
        }
        if (truth) truth[i] = t;
        if (!plant_only && n_rate > 0) { /* geometric gaps between 'N's */
            double pos = -1.0;
            for (;;) {
                double u = rng_unit(&R);
                if (u <= 0.0) u = 1e-300;
                pos += 1.0 + floor(log(u) / log1mp);
                if (pos >= (double)n) break;
                rd[(int)pos] = 'N';
            }
        }
    }
}
