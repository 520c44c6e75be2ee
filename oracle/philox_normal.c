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
 *   (z0,z1) = BoxMuller(r0, r1),  (z2,z3) = BoxMuller(r2, r3)   → paths 4pb … 4pb+3
 *
 * BoxMuller uses only IEEE-754 correctly rounded fp32 operations (+, *, /, sqrt, fma) in a fixed order,
 * so CPU and GPU produce identical bits:
 *   u1 = fma((float)ra, 2^-32, 2^-33)            in (0, 1]
 *   t  = (float)(rb >> 8) * 2^-22                 = 4*u2 in [0, 4), exact
 *   radius = sqrt(-2 ln u1)   with ln from an atanh series on the reduced mantissa
 *   (cos, sin)(2π u2)         by quadrant + octant reflection and Taylor polynomials on [0, π/4]
 *
 * Philox4x32-10 is pinned by the Random123 known-answer vectors (tests/test_oracle_known_answers.py).
 */
#include "fm_oracle.h"
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

/* ln(u) for u in (0,1], fp32, fixed operation order. */
static float spec_logf(float u) {
    uint32_t b = float_to_bits(u);
    int e = (int)(b >> 23) - 127;                                  /* u is normal: u >= 2^-33 */
    float f = bits_to_float((b & 0x007FFFFFu) | 0x3F800000u);      /* [1,2) */
    if (f > 1.41421354f) { f = f * 0.5f; e += 1; }                 /* (0.7071, 1.4142] */
    const float s = (f - 1.0f) / (f + 1.0f);
    const float z = s * s;
    float p = 0.222222224f;                                        /* 2/9 */
    p = fmaf(p, z, 0.285714298f);                                  /* 2/7 */
    p = fmaf(p, z, 0.400000006f);                                  /* 2/5 */
    p = fmaf(p, z, 0.666666687f);                                  /* 2/3 */
    p = p * z;
    const float lnf = fmaf(s, p, s + s);
    const float ef = (float)e;
    return fmaf(ef, 0.693145751953125f, fmaf(ef, 1.42860677e-06f, lnf));
}

static void spec_box_muller(uint32_t ra, uint32_t rb, float* za, float* zb) {
    const float u1 = fmaf((float)ra, 0x1p-32f, 0x1p-33f);
    const float radius = sqrtf(-2.0f * spec_logf(u1));
    const float t = (float)(rb >> 8) * 0x1p-22f;
    const int q = (int)t;
    const float fr = t - (float)q;
    const int swap = fr > 0.5f;
    const float g = swap ? 1.0f - fr : fr;
    const float x = g * 1.57079637f;                               /* π/2 */
    const float x2 = x * x;
    float ps = 2.75573188e-06f;                                    /*  1/9!  */
    ps = fmaf(ps, x2, -1.98412701e-04f);                           /* -1/7!  */
    ps = fmaf(ps, x2, 8.33333377e-03f);                            /*  1/5!  */
    ps = fmaf(ps, x2, -1.66666672e-01f);                           /* -1/3!  */
    ps = ps * x2;
    const float sinx = fmaf(x, ps, x);
    float pc = -2.75573192e-07f;                                   /* -1/10! */
    pc = fmaf(pc, x2, 2.48015876e-05f);                            /*  1/8!  */
    pc = fmaf(pc, x2, -1.38888892e-03f);                           /* -1/6!  */
    pc = fmaf(pc, x2, 4.16666679e-02f);                            /*  1/4!  */
    pc = fmaf(pc, x2, -0.5f);
    const float cosx = fmaf(pc, x2, 1.0f);
    const float sp = swap ? cosx : sinx;                           /* sin, cos of fr*π/2 */
    const float cp = swap ? sinx : cosx;
    float c, s;
    switch (q & 3) {
    case 0:  c =  cp; s =  sp; break;
    case 1:  c = -sp; s =  cp; break;
    case 2:  c = -cp; s = -sp; break;
    default: c =  sp; s = -cp; break;
    }
    *za = radius * c;
    *zb = radius * s;
}

void orc_normal4(int64_t seed, uint64_t pb, uint32_t stream, float z[4]) {
    const uint32_t key[2] = { (uint32_t)(uint64_t)seed, (uint32_t)((uint64_t)seed >> 32) };
    const uint32_t ctr[4] = { (uint32_t)pb, (uint32_t)(pb >> 32), stream, FMHIP_RNG_DOMAIN };
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    spec_box_muller(r[0], r[1], &z[0], &z[1]);
    spec_box_muller(r[2], r[3], &z[2], &z[3]);
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
