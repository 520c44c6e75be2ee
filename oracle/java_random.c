/*
 * java_random.c — java.util.Random restated from its published specification (TEST INFRASTRUCTURE).
 *
 * The reference's differential test feeds `new Random(31415).nextDouble()` into both factories
 * (RandomVariableGPUTest.java:194-201).  No JVM exists in this environment, so the generator is
 * restated from the java.util.Random Javadoc: a 48-bit LCG
 *     seed0 = (seed ^ 0x5DEECE66D) & (2^48 - 1)
 *     next(bits): seed = (seed * 0x5DEECE66D + 0xB) & (2^48 - 1);  return (int)(seed >>> (48 - bits))
 *     nextInt()    = next(32)
 *     nextDouble() = (((long)next(26) << 27) + next(27)) * 2^-53
 * Pinned in tests/test_oracle_known_answers.py by the widely published values
 * new Random(42).nextInt() == -1170105035 and new Random(0).nextInt() == -1155484576.
 */
#include "fm_oracle.h"

#define JR_MULT 0x5DEECE66DULL
#define JR_MASK ((1ULL << 48) - 1)

static inline uint64_t jr_scramble(int64_t seed) { return ((uint64_t)seed ^ JR_MULT) & JR_MASK; }
static inline int32_t jr_next(uint64_t* state, int bits) {
    *state = (*state * JR_MULT + 0xBULL) & JR_MASK;
    return (int32_t)(int64_t)(*state >> (48 - bits));
}

void orc_java_random_doubles(int64_t seed, int64_t n, double* out) {
    uint64_t st = jr_scramble(seed);
    for (int64_t i = 0; i < n; i++) {
        const int64_t hi = (int64_t)jr_next(&st, 26);
        const int64_t lo = (int64_t)jr_next(&st, 27);
        out[i] = (double)((hi << 27) + lo) * 0x1.0p-53;
    }
}

int32_t orc_java_random_next_int(int64_t seed, int skip) {
    uint64_t st = jr_scramble(seed);
    int32_t v = 0;
    for (int k = 0; k <= skip; k++) v = jr_next(&st, 32);
    return v;
}
