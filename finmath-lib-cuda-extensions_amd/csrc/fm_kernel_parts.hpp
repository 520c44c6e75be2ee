// fm_kernel_parts.hpp — device building blocks shared by the interpreter (kernels.hip) and by the specialised kernels the
// JIT tier generates (jit.cpp → hiprtc): global-address-space vector pointers, wave64 DPP reduction, the per-pass
// {Σ, Σ², min, max} accumulation and the workgroup combine.  Sharing them is what makes the two tiers produce
// BIT-IDENTICAL results (same per-lane accumulation order, same wave/LDS combine, same final combine of the partials),
// so a program may switch tier between two launches without any observable difference.
#pragma once
#include "fm_program.h"
#include "fm_device_math.hpp"

namespace fm {

// Vector pointers arrive as 64-bit integers in the row block; casting them to the GLOBAL address space keeps the
// data path on global_load/global_store_dwordx4 (a plain C++ pointer would be "generic" → flat_load, which also
// ties up lgkmcnt).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 __attribute__((address_space(1))) gfloat4;

// Vector data is streamed: every element of an input is read once per launch and every output written once, and a launch
// usually moves far more than the 4 MiB L2 of an XCD.  The non-temporal hint (`nt` on global_load/store_dwordx4) keeps
// the stream from displacing what is still useful in the caches; measured on a 3-read-1-write triad of this shape
// (benchmarks/stream_nontemporal.hip): 5.86 → 6.35 TB/s.
__device__ __forceinline__ f32x4 load_stream(const gfloat4* __restrict__ p, uint32_t i) { return __builtin_nontemporal_load(p + i); }
__device__ __forceinline__ void store_stream(gfloat4* __restrict__ p, uint32_t i, const f32x4 v) { __builtin_nontemporal_store(v, p + i); }

// wave64 data movement without LDS: v_mov_b32 with a DPP control (quad_perm / row_mirror / row_bcast).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float x) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(x), (int)__float_as_uint(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)b, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(b >> 32), (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// Full wave64 reduction of {Σ, Σ², min, max}; the result is valid in lane 63.  Fixed combination order ⇒ deterministic.
__device__ __forceinline__ void wave_reduce(double& s1, double& s2, float& mn, float& mx) {
#define FM_STEP(CTRL, MASK)                                                                             \
    { const double t1 = dpp_d<CTRL, MASK>(s1), t2 = dpp_d<CTRL, MASK>(s2);                              \
      const float tn = dpp_f<CTRL, MASK>(mn), tx = dpp_f<CTRL, MASK>(mx);                               \
      s1 += t1; s2 += t2; mn = jmin(mn, tn); mx = jmax(mx, tx); }
    FM_STEP(0xB1, 0xf)      // quad_perm [1,0,3,2]
    FM_STEP(0x4E, 0xf)      // quad_perm [2,3,0,1]
    FM_STEP(0x141, 0xf)     // row_half_mirror
    FM_STEP(0x140, 0xf)     // row_mirror  → every lane of a row holds the row total
    // across the four rows of 16: row_bcast15 into rows 1,3 then row_bcast31 into rows 2,3 (gfx9 DPP)
    { const double t1 = dpp_d<0x142, 0xa>(s1), t2 = dpp_d<0x142, 0xa>(s2);
      const float tn = dpp_f<0x142, 0xa>(mn), tx = dpp_f<0x142, 0xa>(mx);
      const bool on = ((threadIdx.x >> 4) & 1) != 0;                       // rows 1 and 3 received data
      s1 = on ? s1 + t1 : s1; s2 = on ? s2 + t2 : s2; mn = on ? jmin(mn, tn) : mn; mx = on ? jmax(mx, tx) : mx; }
    { const double t1 = dpp_d<0x143, 0xc>(s1), t2 = dpp_d<0x143, 0xc>(s2);
      const float tn = dpp_f<0x143, 0xc>(mn), tx = dpp_f<0x143, 0xc>(mx);
      const bool on = ((threadIdx.x >> 5) & 1) != 0;                       // rows 2 and 3 received data
      s1 = on ? s1 + t1 : s1; s2 = on ? s2 + t2 : s2; mn = on ? jmin(mn, tn) : mn; mx = on ? jmax(mx, tx) : mx; }
#undef FM_STEP
}

// One pass of a fused reduction: E values per lane into the lane's fp64 accumulators (fp64 accumulation of fp32 values,
// as the twin does: RandomVariableFromFloatArray.java:325-333, :373-381).  min/max use the hardware v_min_f32 /
// v_max_f32 (which order -0 < +0 like java.lang.Math.min/max) and track NaN separately in a wave-level ballot mask
// (scalar registers): NaN anywhere ⇒ the reduction result is NaN.
template <int E>
__device__ __forceinline__ void red_accumulate(const float (&x)[E], const double shift, const bool pass_full,
                                               const uint32_t (&i4)[E / FM_VEC], const int64_t n,
                                               double& acc_sum, double& acc_sq, float& acc_min, float& acc_max,
                                               unsigned long long& nan_mask)
{
    if (pass_full && shift == 0.0) {        // getAverage / first pass of getVariance: no subtraction
        // No per-element NaN test here: Σx² is NaN exactly when some x is NaN (inf² = +inf, and a sum of non-negative
        // terms cannot produce one otherwise); red_unit_end() reads that off the accumulator once per unit.
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double dv = (double)x[j];
            acc_sum += dv;
            acc_sq = __builtin_fma(dv, dv, acc_sq);
        }
#pragma unroll
        for (int j = 0; j < E; j += 2) { acc_min = hw_min3(acc_min, x[j], x[j + 1]); acc_max = hw_max3(acc_max, x[j], x[j + 1]); }
    } else if (pass_full) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double dv = (double)x[j] - shift;
            acc_sum += dv;
            acc_sq = __builtin_fma(dv, dv, acc_sq);
            nan_mask |= __ballot(x[j] != x[j]);
        }
#pragma unroll
        for (int j = 0; j < E; j += 2) { acc_min = hw_min3(acc_min, x[j], x[j + 1]); acc_max = hw_max3(acc_max, x[j], x[j + 1]); }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool ok = (int64_t)i4[j / 4] * FM_VEC + (j & 3) < n;
            const double dv = ok ? (double)x[j] - shift : 0.0;
            acc_sum += dv;
            acc_sq = __builtin_fma(dv, dv, acc_sq);
            acc_min = ok ? hw_min(acc_min, x[j]) : acc_min;
            acc_max = ok ? hw_max(acc_max, x[j]) : acc_max;
            nan_mask |= __ballot(ok && (x[j] != x[j]));
        }
    }
}

// ---- The tree of a fused reduction.  A vector is cut into UNITS of FM_UNIT_ELEMS = 2048 consecutive elements (256 lanes x two
// float4, lane l holding the elements 4l…4l+3 of each half) and into SPANS of FM_SPAN_UNITS = 4 consecutive units (more for vectors
// beyond 2^29 elements: runtime.cpp, launch).  Its moments are, by definition,
//   lane value   = the lane's 8 elements, accumulated in element order                                (red_accumulate)
//   unit partial = lanes l, l+64, l+128, l+192 added in this order, then the wave64 DPP tree over l    (red_span_fold)
//   span partial = its unit partials added in unit order                                             (red_span_fold / wave_sum_span_partials)
//   row          = span partials, lane l of one wave taking the spans l, l+64, …, then the DPP tree   (block_combine)
// — a function of the data and of the vector's length, NOT of the launch: a workgroup may take a whole span (the kernels of large
// launches: fewest partials, fewest arrivals) or a single unit (loop kernels, which keep one tile in registers; small launches, which
// want four times the workgroups); 4 or 8 elements per lane and pass; either execution tier.  Until round 3 a lane accumulated across
// the passes of its workgroup first, which tied the value to the workgroup's span; the transposition below — lane values of a unit
// through LDS, one WAVE per unit instead of one wave per quarter of every unit — costs two LDS stores per unit and lane.
// LDS: 24 bytes per lane, unit slot and reduction; 4 slots for one reduction, 2 for two (24 KB either way).  The workgroup adds unit
// partials to its running value one after the other, so how many units it collects before it does so changes nothing.
template <int NRED> struct RedShared {
    static constexpr int SLOTS = FM_SPAN_UNITS / NRED >= 1 ? FM_SPAN_UNITS / NRED : 1;
    f64x2 lane_sums[NRED][SLOTS][FM_BLOCK];         // {Σ, Σ²} of a lane over one unit
    f32x2 lane_ext[NRED][SLOTS][FM_BLOCK];          // {min, max}
    f64x2 unit_sums[NRED][SLOTS];
    f32x2 unit_ext[NRED][SLOTS];
    f64x2 wg_sums[NRED];                            // the workgroup's running value (red_span_fold → block_combine), kept by ONE lane
    f32x2 wg_ext[NRED];
};
// The wave of a workgroup that stays for the hand-off (block_combine) — it also keeps the workgroup's running value.  It rotates with the
// workgroup index: the waves of a workgroup sit on different SIMDs, and a new workgroup needs a free slot on every one of them —
// lingering waves all on the same SIMD would block it just the same.
__device__ __forceinline__ uint32_t red_keeper_wave() { return blockIdx.x & 3u; }
template <int NRED> __device__ __forceinline__ RedShared<NRED>& red_shared() { __shared__ RedShared<NRED> s; return s; }

// End of a unit: the lane's values go to LDS slot `slot` (the unit's place among those the workgroup is collecting) and the accumulators
// start again.  NaN: with shift 0 the unmasked path of red_accumulate does not test elements — Σx² of a lane is NaN exactly when
// one of its x is (inf² = +inf, and a sum of non-negative terms cannot produce one otherwise); the other paths collect a wave-level
// ballot.  Either way min and max of the lane become NaN here and stay NaN through every jmin / jmax above.
template <int NRED>
__device__ __forceinline__ void red_unit_end(double (&acc_sum)[NRED], double (&acc_sq)[NRED], float (&acc_min)[NRED], float (&acc_max)[NRED],
                                             unsigned long long (&nan_mask)[NRED], const double (&shift)[NRED], const uint32_t slot)
{
    RedShared<NRED>& S = red_shared<NRED>();
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        const bool nan = nan_mask[r] != 0ull || (shift[r] == 0.0 && acc_sq[r] != acc_sq[r]);
        const float q = __builtin_nanf("");
        S.lane_sums[r][slot][threadIdx.x] = f64x2{ acc_sum[r], acc_sq[r] };
        S.lane_ext[r][slot][threadIdx.x] = f32x2{ nan ? q : acc_min[r], nan ? q : acc_max[r] };
        acc_sum[r] = 0.0; acc_sq[r] = 0.0; nan_mask[r] = 0ull;
        acc_min[r] = __builtin_huge_valf(); acc_max[r] = -__builtin_huge_valf();
    }
}

// `units` slots are filled (a span's worth, or what the workgroup has).  Wave w turns slot w into the unit partial; ONE lane (lane 0 of
// the wave that stays for the hand-off) then adds the unit partials, in unit order, to the workgroup's running value in LDS (`first`:
// it starts with them) — in LDS, not in registers: six registers per reduction through the whole kernel were the difference between
// three and four waves per SIMD for the interpreter.  Workgroup-uniform arguments; two barriers.
template <int NRED>
__device__ __forceinline__ void red_span_fold(const uint32_t units, const bool first)
{
    RedShared<NRED>& S = red_shared<NRED>();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads();
    if (wave < units) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) {
            f64x2 a[FM_BLOCK / 64]; f32x2 x[FM_BLOCK / 64];
#pragma unroll
            for (int k = 0; k < FM_BLOCK / 64; ++k) { a[k] = S.lane_sums[r][wave][lane + 64u * k]; x[k] = S.lane_ext[r][wave][lane + 64u * k]; }
            double s1 = a[0].x, s2 = a[0].y; float mn = x[0].x, mx = x[0].y;
#pragma unroll
            for (int k = 1; k < FM_BLOCK / 64; ++k) { s1 += a[k].x; s2 += a[k].y; mn = jmin(mn, x[k].x); mx = jmax(mx, x[k].y); }
            wave_reduce(s1, s2, mn, mx);
            if (lane == 63u) { S.unit_sums[r][wave] = f64x2{ s1, s2 }; S.unit_ext[r][wave] = f32x2{ mn, mx }; }
        }
    }
    __syncthreads();
    if (wave == red_keeper_wave() && lane == 0u) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) {
            f64x2 w = first ? S.unit_sums[r][0] : S.wg_sums[r];
            f32x2 e = first ? S.unit_ext[r][0] : S.wg_ext[r];
#pragma unroll
            for (uint32_t u = 0; u < (uint32_t)RedShared<NRED>::SLOTS; ++u) {
                if (u < units && !(first && u == 0u)) {
                    const f64x2 a = S.unit_sums[r][u]; const f32x2 x = S.unit_ext[r][u];
                    w.x += a.x; w.y += a.y; e.x = jmin(e.x, x.x); e.y = jmax(e.y, x.y);
                }
            }
            S.wg_sums[r] = w; S.wg_ext[r] = e;
        }
    }
}

// The bookkeeping of the three calls above for a kernel that walks its tiles in order: call after the red_accumulate of every tile.
// TPU = tiles per unit (1 at 8 elements per lane, 2 at 4); rel = the tile's index within the workgroup's stretch; last = it is the
// workgroup's last tile.
template <int NRED, int E>
__device__ __forceinline__ void red_tile_end(const uint32_t rel, const bool last,
                                             double (&acc_sum)[NRED], double (&acc_sq)[NRED], float (&acc_min)[NRED], float (&acc_max)[NRED],
                                             unsigned long long (&nan_mask)[NRED], const double (&shift)[NRED])
{
    constexpr uint32_t TPU = (uint32_t)(FM_UNIT_ELEMS / (FM_BLOCK * E));
    static_assert(TPU == 1u || TPU == 2u, "a reduction unit is one pass at 8 elements per lane, two at 4");
    if ((rel % TPU) != TPU - 1u && !last) return;
    constexpr uint32_t SLOTS = (uint32_t)RedShared<NRED>::SLOTS;
    const uint32_t unit = rel / TPU, slot = unit % SLOTS;
    red_unit_end<NRED>(acc_sum, acc_sq, acc_min, acc_max, nan_mask, shift, slot);
    if (slot == SLOTS - 1u || last) red_span_fold<NRED>(slot + 1u, unit < SLOTS);
}

// FM_HANDOFF_RELEASE: 1 = the arrival counter's add is an agent-scope release (see block_combine); 0 = relaxed behind sc1 stores
// and an explicit drain.  Measured on the bench launch (profiles/round03_handoff_ab.txt) before choosing the default.
#ifndef FM_HANDOFF_RELEASE
#define FM_HANDOFF_RELEASE 0
#endif
#if FM_HANDOFF_RELEASE
#define FM_HANDOFF_ADD_ORDER __ATOMIC_RELEASE
#else
#define FM_HANDOFF_ADD_ORDER __ATOMIC_RELAXED
#endif

// Device-coherent accesses for the hand-off between workgroups (possibly on different XCDs, whose L2s are not coherent
// with each other): `sc1` stores / loads go through to memory, no cache-wide write-back or invalidate is needed.
__device__ __forceinline__ void store_coherent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_coherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Number of arrival-counting groups of a row with `blocks` workgroups (see block_combine): 1 below 512, up to 7.
__device__ __forceinline__ uint32_t combine_groups(uint32_t blocks) { return blocks >= 512u ? (blocks >= 1792u ? 7u : blocks >> 8) : 1u; }
// Slots of the partial array per (row, reduction): one per workgroup + one per group (FM_COMBINE_GROUP_SLOTS >= 7).
constexpr uint32_t FM_COMBINE_GROUP_SLOTS = 8;

// One wave adds the partials p[first], p[first + stride], … (count of them; 4 doubles {Σ, Σ², min, max} each) in a FIXED
// order: lane l takes the elements l, l + 64, … (the loads of 8 of them issued before the first is used — one at a time
// every element is a dependent memory round trip), then the wave64 DPP tree.  The result is valid in lane 63.
__device__ __forceinline__ void wave_sum_partials(const double* __restrict__ p, uint32_t first, uint32_t stride, uint32_t count,
                                                  double& s1, double& s2, float& mn, float& mx)
{
    const uint32_t lane = threadIdx.x & 63u;
    s1 = 0.0; s2 = 0.0; mn = __builtin_huge_valf(); mx = -__builtin_huge_valf();
    for (uint32_t k0 = lane; k0 < count; k0 += 64u * 8u) {
        double v[8][4];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t k = k0 + u * 64u;
            const double* q = p + (size_t)(first + (k < count ? k : k0) * stride) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = load_coherent(q + c);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u)
            if (k0 + u * 64u < count) { s1 += v[u][0]; s2 += v[u][1]; mn = jmin(mn, (float)v[u][2]); mx = jmax(mx, (float)v[u][3]); }      // min / max are floats, widened
    }
    wave_reduce(s1, s2, mn, mx);
}

// As wave_sum_partials, for a launch whose workgroups take a QUARTER (or half) of a span each: the partials of span k of the group are
// p[(first + k * stride) * Q + q], q = 0 … Q-1 (as many as the row has: `blocks` workgroups), added in this order before the span
// joins the lane's sum.  Small launches only (a row of at most a few thousand workgroups): two spans in flight per lane.
__device__ __forceinline__ void wave_sum_span_partials(const double* __restrict__ p, uint32_t first, uint32_t stride, uint32_t count,
                                                       uint32_t Q, uint32_t blocks, double& s1, double& s2, float& mn, float& mx)
{
    const uint32_t lane = threadIdx.x & 63u;
    s1 = 0.0; s2 = 0.0; mn = __builtin_huge_valf(); mx = -__builtin_huge_valf();
    for (uint32_t k = lane; k < count; k += 64u) {
        const uint32_t b0 = (first + k * stride) * Q;
        double v[FM_SPAN_UNITS][4];
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)FM_SPAN_UNITS; ++q) {
            const double* src = p + (size_t)((q < Q && b0 + q < blocks) ? b0 + q : b0) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[q][c] = load_coherent(src + c);
        }
        double t1 = v[0][0], t2 = v[0][1]; float tn = (float)v[0][2], tx = (float)v[0][3];
#pragma unroll
        for (uint32_t q = 1; q < (uint32_t)FM_SPAN_UNITS; ++q)
            if (q < Q && b0 + q < blocks) { t1 += v[q][0]; t2 += v[q][1]; tn = jmin(tn, (float)v[q][2]); tx = jmax(tx, (float)v[q][3]); }
        s1 += t1; s2 += t2; mn = jmin(mn, tn); mx = jmax(mx, tx);
    }
    wave_reduce(s1, s2, mn, mx);
}

// Workgroup combine.  The workgroup's value is in LDS (red_span_fold); one partial per workgroup and reduction:
// partials[row][r][blockIdx.x] = {Σ, Σ², min, max}.  Q = workgroups per span (1: a workgroup took a whole span; 4: one unit each).
// Waves other than one are finished here: the rest — the coherent stores, the wait for them, the arrival counter and its round
// trip, ≈ 3-4 µs — keeps ONE wave slot of the workgroup busy, not four (measured on the bench launch: see DESIGN.md §4.2).
// Arrival counting: device-scope atomics execute at the memory side, ≈ 11-13 ns apiece on one cache line, so a row of
// thousands of workgroups (one vector of 2^26 paths: 8192) would queue on a single counter for longer than the kernel runs
// (measured: 128 µs instead of 45).  A row therefore counts in G = combine_groups(spans) groups (span s belongs to group
// s mod G; each group counter in a cache line — a plane — of its own: 32 B apart was not enough).  The LAST workgroup
// of a group adds the group's span partials in a fixed order (wave_sum_partials) and, if G > 1, stores the group partial behind
// the row's workgroup partials and moves the second-level counter; the last of those adds the G group partials.  The final
// moments are a deterministic function of the data and of the number of spans: no float atomics, no second launch.
// (block_combine_values: the same with the workgroup's values handed in — wg_sums / wg_ext in LDS, written by lane 0 of the keeper wave; the
// merged loop kernels reduce their chains one after the other through RedShared<1> and keep the values in a ChainValues<K>)
template <int NRED>
__device__ __forceinline__ void block_combine_values(const f64x2* wg_sums, const f32x2* wg_ext,
                                                     double* __restrict__ partials, const uint32_t row,
                                                     double* __restrict__ results, uint32_t* __restrict__ counter,       // counter: this row's arrival counter
                                                     uint64_t* done_flag, const uint64_t done_value,                       // see DevProgramArgs::done_flag
                                                     const uint32_t Q)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (wave != red_keeper_wave()) return;

    const uint32_t slots = gridDim.x + FM_COMBINE_GROUP_SLOTS;             // per (row, reduction)
    const uint32_t spans = (gridDim.x + Q - 1u) / Q;
    const uint32_t G = combine_groups(spans);
    // A row of one group whose workgroups all run at once (ONE row of 10^6 paths, a unit per workgroup: 489) would still queue 489 adds
    // on one counter — 6 µs behind a kernel of 5 (benchmarks/single_row_stream.cpp: 13.0 µs with the reduction, 4.6 without).  Such a row
    // COUNTS in seven groups as well; the arithmetic stays that of one group (the row's last workgroup adds every span partial in the
    // order it always did), so the moments do not change.
    const uint32_t C = (G == 1u && gridDim.x >= FM_COUNT_IN_GROUPS_FROM) ? 7u : G;
    const uint32_t cg = (blockIdx.x / Q) % C;
    const uint32_t counted_spans = (spans - cg + C - 1u) / C;
    // workgroups of the group: Q per span, except that the row's last span may have fewer
    const uint32_t members = counted_spans * Q - (((spans - 1u) % C == cg) ? spans * Q - gridDim.x : 0u);
    const uint32_t g = (G == 1u) ? 0u : cg;
    const uint32_t group_spans = (G == 1u) ? spans : counted_spans;
    uint32_t group_last = 0u;
    if (lane == 0u) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) {
            const f64x2 w = wg_sums[r]; const f32x2 e = wg_ext[r];          // (written by this very lane: red_span_fold)
            const double s1 = w.x, s2 = w.y;
            const float mn = e.x, mx = e.y;
            double* out = partials + (((size_t)row * NRED + r) * slots + blockIdx.x) * 4;
            store_coherent(out + 0, s1); store_coherent(out + 1, s2); store_coherent(out + 2, (double)mn); store_coherent(out + 3, (double)mx);
        }
        // the partial is in memory before this workgroup is counted (the stores are drained, then the counter moves)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t arrived = __hip_atomic_fetch_add(counter + (size_t)cg * FM_COUNTER_PLANE, 1u, FM_HANDOFF_ADD_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        // Ordering of this hand-off.  Producer: every partial is an sc1 (write-through) store, drained by the s_waitcnt above before the
        // counter moves (MI355X_MICROARCH.md, "Valid forms", sc1 table; FM_HANDOFF_RELEASE = 1 makes the add an agent-scope RELEASE on
        // top: buffer_wbl2 + wait in front of it).  Consumer: the last arriver learns that it is last from the value its own add
        // RETURNED and only then loads the partials — all sc1 loads (wave_sum_partials), which do not hit in the L1 or in a foreign
        // XCD's L2 — and, since round 3, behind an agent-scope ACQUIRE fence (buffer_inv sc1: one wave per row and launch, free), so
        // the hand-off no longer rests on the measured envelope of the sc1-only form alone (that table is for one workgroup per CU;
        // these kernels run four or five).  The signal fences keep the COMPILER from moving the relaxed accesses across the add,
        // whichever compiler (hipcc today, hiprtc at run time) builds this header.
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (arrived == members - 1u) {
            group_last = 1u;
            if (C != G) {       // counted in groups, added as one: the last of the seven last arrivers goes on (the partials of every group are in memory:
                                // each of its members drained its stores before it was counted, and this add follows the group's last count)
                const uint32_t groups_counted = __hip_atomic_fetch_add(counter + (size_t)7 * FM_COUNTER_PLANE, 1u, FM_HANDOFF_ADD_ORDER, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                group_last = (groups_counted == C - 1u) ? 1u : 0u;
            }
            if (group_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counter + (size_t)cg * FM_COUNTER_PLANE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            if (C != G && group_last) __hip_atomic_store(counter + (size_t)7 * FM_COUNTER_PLANE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (__builtin_amdgcn_readfirstlane(group_last) == 0u) return;           // lane 0 is the first active lane

    // ---- this wave belongs to the last workgroup of group g: add the group's partials
    uint32_t row_last = (G == 1u) ? 1u : 0u;
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        double* base = partials + ((size_t)row * NRED + r) * slots * 4;
        double s1, s2; float mn, mx;
        if (Q == 1u) wave_sum_partials(base, g, G, group_spans, s1, s2, mn, mx);
        else wave_sum_span_partials(base, g, G, group_spans, Q, gridDim.x, s1, s2, mn, mx);
        if (lane == 63u) {
            if (G == 1u) {
                double* o = results + ((size_t)row * NRED + r) * 4;
                // NaN results are canonicalised: which NaN (sign, payload) an fp64 add of two NaNs returns depends on the operand
                // order the compiler picked, and the two execution tiers must agree bit for bit
                o[0] = (s1 != s1) ? __builtin_nan("") : s1; o[1] = (s2 != s2) ? __builtin_nan("") : s2;
                o[2] = (mn != mn) ? __builtin_nan("") : (double)mn; o[3] = (mx != mx) ? __builtin_nan("") : (double)mx;
            } else {
                double* out = base + (size_t)(gridDim.x + g) * 4;
                store_coherent(out + 0, s1); store_coherent(out + 1, s2); store_coherent(out + 2, (double)mn); store_coherent(out + 3, (double)mx);
            }
        }
    }
    if (G == 1u) {
        // results in pinned host memory, then the flag the host polls (release at system scope: the results are visible before it)
        if (done_flag && lane == 63u) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (lane == 63u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t groups_done = __hip_atomic_fetch_add(counter + (size_t)7 * FM_COUNTER_PLANE, 1u, FM_HANDOFF_ADD_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (groups_done == G - 1u) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(counter + (size_t)7 * FM_COUNTER_PLANE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); row_last = 1u; }
    }
    row_last = __builtin_amdgcn_readlane(row_last, 63);
    if (row_last == 0u) return;

    // ---- last group of the row: add the G group partials
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        const double* base = partials + ((size_t)row * NRED + r) * slots * 4;
        double s1, s2; float mn, mx;
        wave_sum_partials(base, gridDim.x, 1u, G, s1, s2, mn, mx);
        if (lane == 63u) {
            double* o = results + ((size_t)row * NRED + r) * 4;
            o[0] = (s1 != s1) ? __builtin_nan("") : s1; o[1] = (s2 != s2) ? __builtin_nan("") : s2;
            o[2] = (mn != mn) ? __builtin_nan("") : (double)mn; o[3] = (mx != mx) ? __builtin_nan("") : (double)mx;
        }
    }
    if (done_flag && lane == 63u) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int NRED>
__device__ __forceinline__ void block_combine(double* __restrict__ partials, const uint32_t row,
                                              double* __restrict__ results, uint32_t* __restrict__ counter,
                                              uint64_t* done_flag, const uint64_t done_value, const uint32_t Q)
{
    RedShared<NRED>& S = red_shared<NRED>();
    block_combine_values<NRED>(S.wg_sums, S.wg_ext, partials, row, results, counter, done_flag, done_value, Q);
}

// ---- Several reductions of ONE tile taken one after the other (merged loop kernels: K chains, K up to 16).  RedShared<K> would hold K lane
// transpositions at once (24 bytes per lane and reduction: 86 KB for 14); here every chain goes through the one slot of RedShared<1> — its
// tile is one unit of the tree, folded at once — and leaves its value in ChainValues<K>.  Same order of additions per chain as a kernel
// that reduces that chain alone with one tile per workgroup (red_tile_end<1, 8> with rel = 0, last = true).
template <int K> struct ChainValues { f64x2 wg_sums[K]; f32x2 wg_ext[K]; };
template <int K> __device__ __forceinline__ ChainValues<K>& chain_values() { __shared__ ChainValues<K> v; return v; }
template <int K, int E>
__device__ __forceinline__ void red_chain_unit(const int k, const float (&x)[E], const double shift_value, const bool pass_full,
                                               const uint32_t (&i4)[E / FM_VEC], const int64_t n)
{
    static_assert(FM_BLOCK * E == FM_UNIT_ELEMS, "the workgroup's tile is one unit of the reduction tree");
    double acc_sum[1] = { 0.0 }, acc_sq[1] = { 0.0 }, shift[1] = { shift_value }; unsigned long long nan_mask[1] = { 0ull };
    float acc_min[1] = { __builtin_huge_valf() }, acc_max[1] = { -__builtin_huge_valf() };
    red_accumulate<E>(x, shift[0], pass_full, i4, n, acc_sum[0], acc_sq[0], acc_min[0], acc_max[0], nan_mask[0]);
    red_unit_end<1>(acc_sum, acc_sq, acc_min, acc_max, nan_mask, shift, 0u);
    red_span_fold<1>(1u, true);
    if ((threadIdx.x >> 6) == red_keeper_wave() && (threadIdx.x & 63u) == 0u) {
        RedShared<1>& S = red_shared<1>();
        ChainValues<K>& V = chain_values<K>();
        V.wg_sums[k] = S.wg_sums[0]; V.wg_ext[k] = S.wg_ext[0];
    }
}

} // namespace fm
