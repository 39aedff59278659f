/*
 * vfind_synth.h -- seeded synthetic read generator used by bench.py and the parity tests
 * (SURVEY.md 8d).  Bench/test tooling in benchlib/libvfind_synth.so; not part of the product
 * library or of the drop-in boundary.  Every read is a pure function of (cfg, read index): read i is produced from the
 * stream splitmix64(seed ^ i), so any read is reproducible on any host, rank or thread count.
 *
 *   read = flank5 | prefix' | variant | suffix' | filler, cut / padded with uniform ACGT to L
 */
#ifndef VFIND_SYNTH_H
#define VFIND_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vf_synth_cfg {
    uint64_t seed;
    uint32_t read_len;        /* L */
    uint32_t flank_max;       /* 5' flank length uniform in [0, flank_max] */
    const uint8_t *prefix; uint32_t prefix_len;
    const uint8_t *suffix; uint32_t suffix_len;
    uint64_t n_variants;      /* V: library size; variant v is itself a pure function of (seed, v) */
    double zipf_s;            /* variant popularity ~ rank^-s */
    double sub_rate;          /* per adapter base substitution probability */
    double indel_rate;        /* per adapter base insertion-or-deletion probability */
    double no_adapter_frac;   /* reads that are pure random sequence */
    double non_codon_frac;    /* library entries whose length is not a multiple of 3 */
} vf_synth_cfg;

typedef struct vf_synth vf_synth;

/* two random m-mers with 40-60 % GC and no self-overlap (no proper border); NUL-terminated */
int vf_synth_adapters(uint64_t seed, uint32_t m, char *prefix_out, char *suffix_out);

/* builds the Zipf sampler (V doubles) */
int vf_synth_create(const vf_synth_cfg *cfg, vf_synth **out);
void vf_synth_destroy(vf_synth *g);

/* reads [first, first+n): ascii (n*L bytes, may be NULL), rows (n*row_dwords, may be NULL),
 * lens (n uint16, may be NULL) */
int vf_synth_generate(const vf_synth *g, uint64_t first, uint64_t n, uint8_t *ascii, uint32_t *rows,
                      uint32_t row_dwords, uint16_t *lens, uint32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
