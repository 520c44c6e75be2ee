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
        // terms cannot produce one otherwise); red_finish() reads that off the accumulator once per workgroup.
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

// After the last pass of a workgroup: NaN detection of the unshifted fast path (see red_accumulate).  With a shift the
// accumulator can also be NaN from inf - inf, so those paths keep their per-element test.
__device__ __forceinline__ void red_finish(const double shift, const double acc_sq, unsigned long long& nan_mask)
{
    if (shift == 0.0) nan_mask |= __ballot(acc_sq != acc_sq);
}

// fp64 min/max with java.lang.Math semantics for the final combine of the per-workgroup partials
__device__ __forceinline__ double jmin_d(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && (__double_as_longlong(b) < 0)) return b;
    return (a <= b) ? a : b;
}
__device__ __forceinline__ double jmax_d(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && (__double_as_longlong(a) < 0)) return b;
    return (a >= b) ? a : b;
}

// Device-coherent accesses for the hand-off between workgroups (possibly on different XCDs, whose L2s are not coherent
// with each other): `sc1` stores / loads go through to memory, no cache-wide write-back or invalidate is needed.
__device__ __forceinline__ void store_coherent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_coherent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Workgroup combine: wave64 DPP reduction, then 4 waves through LDS, one partial per workgroup and reduction:
// partials[row][r][blockIdx.x] = {Σ, Σ², min, max}.  The LAST workgroup of a row to arrive (device-scope counter) then sums
// the row's partials in a FIXED order (thread t takes workgroups t, t+256, …, then an LDS tree) and writes the final
// moments: deterministic, no float atomics, and no second launch.
template <int NRED>
__device__ __forceinline__ void block_combine(const double (&acc_sum)[NRED], const double (&acc_sq)[NRED],
                                              const float (&acc_min)[NRED], const float (&acc_max)[NRED],
                                              const unsigned long long (&nan_mask)[NRED],
                                              double* __restrict__ partials, const uint32_t row,
                                              double* __restrict__ results, uint32_t* __restrict__ counter)      // counter: this row's arrival counter
{
    __shared__ double lds_sum[NRED][FM_BLOCK / 64], lds_sq[NRED][FM_BLOCK / 64];
    __shared__ float  lds_min[NRED][FM_BLOCK / 64], lds_max[NRED][FM_BLOCK / 64];
    __shared__ double sh[4][FM_BLOCK];
    __shared__ uint32_t last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        double s1 = acc_sum[r], s2 = acc_sq[r];
        float mn = acc_min[r], mx = acc_max[r];
        if (nan_mask[r] != 0ull) { mn = __builtin_nanf(""); mx = mn; }      // wave-uniform
        wave_reduce(s1, s2, mn, mx);
        if (lane == 63) { lds_sum[r][wave] = s1; lds_sq[r][wave] = s2; lds_min[r][wave] = mn; lds_max[r][wave] = mx; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) {
            double s1 = lds_sum[r][0], s2 = lds_sq[r][0];
            float mn = lds_min[r][0], mx = lds_max[r][0];
#pragma unroll
            for (int wv = 1; wv < FM_BLOCK / 64; ++wv) {
                s1 += lds_sum[r][wv]; s2 += lds_sq[r][wv];
                mn = jmin(mn, lds_min[r][wv]); mx = jmax(mx, lds_max[r][wv]);
            }
            double* out = partials + (((size_t)row * NRED + r) * gridDim.x + blockIdx.x) * 4;
            store_coherent(out + 0, s1); store_coherent(out + 1, s2); store_coherent(out + 2, (double)mn); store_coherent(out + 3, (double)mx);
        }
        // the partial is in memory before this workgroup is counted (the stores are drained, then the counter moves)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Arrival counting.  Device-scope atomics execute at the memory side, ≈ 11-13 ns apiece on one address: a row of
        // thousands of workgroups (one vector of 2^26 paths: 8192) would queue there for longer than the kernel runs
        // (measured: 128 µs instead of 45).  Rows with many workgroups therefore count in up to 7 groups (workgroup b
        // belongs to group b mod G), the last arrival of each group moves a second-level counter, and the last of those
        // is the last workgroup of the row.  The queue is per cache line, not per address (7 group counters 32 B apart:
        // still 72 µs), so every group counter lives in a plane of its own: counter + plane·FM_COUNTER_PLANE.
        const uint32_t G = gridDim.x >= 512u ? (gridDim.x >= 1792u ? 7u : gridDim.x >> 8) : 1u;
        const uint32_t g = blockIdx.x % G;
        const uint32_t members = (gridDim.x - g + G - 1u) / G;
        uint32_t last = 0u;
        const uint32_t arrived = __hip_atomic_fetch_add(counter + (size_t)g * FM_COUNTER_PLANE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == members - 1u) {
            __hip_atomic_store(counter + (size_t)g * FM_COUNTER_PLANE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            if (G == 1u) last = 1u;
            else {
                const uint32_t groups_done = __hip_atomic_fetch_add(counter + (size_t)7 * FM_COUNTER_PLANE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (groups_done == G - 1u) { __hip_atomic_store(counter + (size_t)7 * FM_COUNTER_PLANE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1u; }
            }
        }
        last_flag = last;
    }
    __syncthreads();
    if (last_flag == 0u) return;                                             // workgroup-uniform
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
        const double* p = partials + ((size_t)row * NRED + r) * gridDim.x * 4;
        double s1 = 0.0, s2 = 0.0, mn = __builtin_huge_val(), mx = -__builtin_huge_val();
        // Thread t adds the partials of workgroups t, t+256, … in that order.  The loads of 8 of them are issued before
        // the first is used: left one at a time (the compiler keeps coherent loads in program order and waits for each
        // before the dependent add), a row of 8192 workgroups cost 32 memory round trips here — ≈ 40 µs of a 45 µs kernel.
        for (uint32_t b0 = threadIdx.x; b0 < gridDim.x; b0 += FM_BLOCK * 8u) {
            double v[8][4];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t b = b0 + u * FM_BLOCK;
                const double* q = p + (size_t)(b < gridDim.x ? b : b0) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = load_coherent(q + k);
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u)
                if (b0 + u * FM_BLOCK < gridDim.x) { s1 += v[u][0]; s2 += v[u][1]; mn = jmin_d(mn, v[u][2]); mx = jmax_d(mx, v[u][3]); }
        }
        sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2; sh[2][threadIdx.x] = mn; sh[3][threadIdx.x] = mx;
        __syncthreads();
        for (int stride = FM_BLOCK / 2; stride > 0; stride >>= 1) {
            if ((int)threadIdx.x < stride) {
                sh[0][threadIdx.x] += sh[0][threadIdx.x + stride];
                sh[1][threadIdx.x] += sh[1][threadIdx.x + stride];
                sh[2][threadIdx.x] = jmin_d(sh[2][threadIdx.x], sh[2][threadIdx.x + stride]);
                sh[3][threadIdx.x] = jmax_d(sh[3][threadIdx.x], sh[3][threadIdx.x + stride]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            double* o = results + ((size_t)row * NRED + r) * 4;
            // NaN results are canonicalised: which NaN (sign, payload) an fp64 add of two NaNs returns depends on the operand
            // order the compiler picked, and the two execution tiers must agree bit for bit
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double v = sh[k][0]; o[k] = (v != v) ? __builtin_nan("") : v; }
        }
        __syncthreads();
    }
}

} // namespace fm
