/*
 * philox_normal.c — specification (CPU side) of the fmhip normal-increment generator
 * (TEST INFRASTRUCTURE, see fm_oracle.h).
 *
 * The reference draws its increments with cuRAND's default XORWOW generator, one curandGenerateNormal
 * call per (step, factor) (BrownianMotionCudaWithRandomVariableCuda.java:159-178).  cuRAND bit streams are
 * not reproducible outside CUDA, and the reference's own tests only pin the increments statistically
 * (BrownianMotionTest.java:120-121).  This repository therefore defines its own counter-based generator;
 * this file is its normative definition and the HIP kernel must match it BIT FOR BIT:
 *
 *   key  = { lo32(seed), hi32(seed) }
 *   ctr  = { lo32(pb), hi32(pb), stream, 0x464D4850 }        pb = global path index / 4,
 *                                                            stream = step * n_factors + factor
 *   r[0..3] = Philox4x32-10(ctr, key)                        (Salmon et al., SC'11; Random123)
 *   z[i]    = Normal(r[i])                                   → paths 4pb … 4pb+3
 *
 * Normal(w) — round 2 re-specification for CDNA4 (round 1: Box–Muller restricted to IEEE-correct fp32 operations, ≈ 72
 * VALU instructions per normal, 2.3–2.8 TB/s written; the write stream can take ≈ 25): inverse normal CDF by HIERARCHICAL
 * SEGMENTATION (Lee, Luk, Villasenor, Cheung 2006): one 32-bit word → one normal,
 *   sign = bit 31;  k = (w << 1) | 1  (odd);  p = k·2^-33 in (0, 1/2)           never 0 or 1/2
 *   lz = clz(k): the octave of the tail;  norm = k << lz;  idx = bits 28..30 of norm;  tf = (float)(norm & 0x0FFFFFFF)
 *   |z| = fma(fma(fma(c3, tf, c2), tf, c1), tf, c0),  {c0..c3} = TABLE[lz·8 + idx]   (normal_table.h, tools/normal_table.py:
 *          cubic per segment fitted to -Φ^-1(p); max abs error 5.4e-7 ≈ one fp32 rounding at |z| ≈ 4…6)
 *   z = copysign(|z|, sign)
 * Integer operations, one exact int→float conversion with rounding to nearest, three fp32 FMAs: CPU and GPU produce
 * identical bits.  |z| <= 6.36 (p >= 2^-33).
 *
 * Philox4x32-10 is pinned by the Random123 known-answer vectors (tests/test_oracle_known_answers.py).
 */
#include "fm_oracle.h"
#include "normal_table.h"
#include <math.h>
#include <string.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
#define FMHIP_RNG_DOMAIN 0x464D4850u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float bits_to_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* One standard normal from one 32-bit word (see the header). */
static float spec_normal(uint32_t w) {
    const uint32_t k = (w << 1) | 1u;
    const int lz = __builtin_clz(k);                               /* k is odd: never 0 */
    const uint32_t norm = k << lz;
    const uint32_t idx = (norm >> 28) & 7u;
    const float tf = (float)(norm & 0x0FFFFFFFu);                  /* round to nearest even (default mode) */
    const float* c = ORC_NORMAL_TABLE + ((uint32_t)lz * 8u + idx) * 4u;
    float m = fmaf(c[3], tf, c[2]);
    m = fmaf(m, tf, c[1]);
    m = fmaf(m, tf, c[0]);
    return bits_to_float((float_to_bits(m) & 0x7FFFFFFFu) | (w & 0x80000000u));
}

void orc_normal4(int64_t seed, uint64_t pb, uint32_t stream, float z[4]) {
    const uint32_t key[2] = { (uint32_t)(uint64_t)seed, (uint32_t)((uint64_t)seed >> 32) };
    const uint32_t ctr[4] = { (uint32_t)pb, (uint32_t)(pb >> 32), stream, FMHIP_RNG_DOMAIN };
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    for (int i = 0; i < 4; i++) z[i] = spec_normal(r[i]);
}

void orc_bm_increment(int64_t seed, uint32_t stream, int64_t path_offset, int64_t n, float sqrt_dt, float* out) {
    int64_t p = 0;
    while (p < n) {
        const uint64_t gp = (uint64_t)(path_offset + p);
        float z[4];
        orc_normal4(seed, gp >> 2, stream, z);
        for (unsigned k = (unsigned)(gp & 3); k < 4 && p < n; k++, p++) out[p] = sqrt_dt * z[k];
    }
}
