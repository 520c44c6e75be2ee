// merged_chains_harness.hip — PROTOTYPE of a loop kernel that serves SEVERAL product chains which read the same per-iteration inputs
// (the swaptions of ONE exercise date: tenors 1…10, 15, 20, 25, 30 years read L_e[e + p] for p below their own length).
// One load of L feeds every chain that has started; the denominator 1 + δL, its reciprocal and the Newton step are shared (same bits: the
// in-range division chain of fm_device_math.hpp is a function of (a, b) and b is the same number).  Compared with the engine's one launch
// per chain length: 61 vectors read per exercise date instead of 304.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I finmath-lib-cuda-extensions_amd/csrc -DK_=14 benchmarks/merged_chains_harness.hip -o /tmp/merged
//   run:   merged [rows=7] [paths=1000000] [launches=10] [elems=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "fm_kernel_parts.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#ifndef K_
#define K_ 14
#endif
namespace fm {
struct MergedArgs { int64_t n; uint32_t row_words, iterations; uint64_t dump; double* partials; double* results; uint32_t* counters; };

template <int K> struct MergedShared { f64x2 wg_sums[K]; f32x2 wg_ext[K]; };

// row: [numeraire][R loop inputs][K x {swap rate, first iteration}] (floats behind the pointers)
template <int K, int E>
__global__ void __launch_bounds__(FM_BLOCK) merged_kernel(const MergedArgs A, const uint64_t* __restrict__ rows)
{
    constexpr int T = E / FM_VEC;
    __shared__ MergedShared<K> M;
    const uint64_t* __restrict__ rowp = rows + (size_t)blockIdx.y * A.row_words;
    const uint32_t R = A.iterations;
    const int64_t n = A.n;
    const float* __restrict__ scal = reinterpret_cast<const float*>(rowp + 1 + R);
    const uint32_t tile = blockIdx.x;
    uint32_t i4[T], i4c[T]; bool lane_valid[T];
    _Pragma("unroll") for (int t = 0; t < T; ++t) { i4[t] = (tile * T + t) * FM_BLOCK + threadIdx.x; lane_valid[t] = (int64_t)i4[t] * FM_VEC < n; i4c[t] = lane_valid[t] ? i4[t] : 0u; }
    f32x4 nx[T];
    { const gfloat4* __restrict__ p = reinterpret_cast<const gfloat4*>(rowp[1]); _Pragma("unroll") for (int t = 0; t < T; ++t) nx[t] = load_stream(p, i4c[t]); }
    float c[K][E];
    _Pragma("unroll") for (int k = 0; k < K; ++k) _Pragma("unroll") for (int j = 0; j < E; ++j) c[k][j] = -0.0f;       // -0 + x = x for every x: a chain's first payoff enters by the same addition
    const float delta = 0.5f;
    for (uint32_t it = 0; it < R; ++it) {
        float l0[E];
        _Pragma("unroll") for (int t = 0; t < T; ++t) { l0[4 * t] = nx[t].x; l0[4 * t + 1] = nx[t].y; l0[4 * t + 2] = nx[t].z; l0[4 * t + 3] = nx[t].w; }
        if (it + 1u < R) { const gfloat4* __restrict__ p = reinterpret_cast<const gfloat4*>(rowp[2 + it]); _Pragma("unroll") for (int t = 0; t < T; ++t) nx[t] = load_stream(p, i4c[t]); }
        // shared by all chains: the denominator of discount(·, L, δ), its reciprocal after one Newton step, the range test of the denominator
        f32x2 den[E / 2], y1[E / 2], nd[E / 2];
        DivRange drange;
        _Pragma("unroll") for (int p = 0; p < E / 2; ++p) {
            const f32x2 l = { l0[2 * p], l0[2 * p + 1] }, sv = { delta, delta }, one = { 1.0f, 1.0f };
            const f32x2 pr = l * sv; den[p] = one + pr;
            f32x2 y; y.x = __builtin_amdgcn_rcpf(den[p].x); y.y = __builtin_amdgcn_rcpf(den[p].y);
            nd[p] = -den[p];
            const f32x2 e = __builtin_elementwise_fma(nd[p], y, one);
            y1[p] = __builtin_elementwise_fma(e, y, y);
            drange.take(den[p]);
        }
        _Pragma("unroll") for (int k = 0; k < K; ++k) {
            const uint32_t first = __builtin_amdgcn_readfirstlane(__float_as_uint(scal[2 * k + 1]));
            if (it >= first) {
                const float sr = scal[2 * k];
                DivRange range = drange;
                f32x2 num[E / 2], q[E / 2];
                _Pragma("unroll") for (int p = 0; p < E / 2; ++p) {
                    const f32x2 l = { l0[2 * p], l0[2 * p + 1] }, srv = { sr, sr }, sv = { delta, delta };
                    const f32x2 d = l - srv, pay = d * sv;
                    const f32x2 cc = { c[k][2 * p], c[k][2 * p + 1] };
                    num[p] = cc + pay;
                    const f32x2 q0 = num[p] * y1[p];
                    const f32x2 r0 = __builtin_elementwise_fma(nd[p], q0, num[p]);
                    q[p] = __builtin_elementwise_fma(r0, y1[p], q0);
                    range.take(num[p]);
                }
                if (__builtin_amdgcn_ballot_w64(range.outside()) != 0ull) {
                    asm volatile("; division: IEEE expansion with scaling and fix-up for operands outside [2^-48, 2^48)");
                    _Pragma("unroll") for (int p = 0; p < E / 2; ++p) q[p] = div_pair(num[p], den[p]);
                }
                _Pragma("unroll") for (int p = 0; p < E / 2; ++p) { c[k][2 * p] = q[p].x; c[k][2 * p + 1] = q[p].y; }
            }
        }
    }
    // tail: floor(0) / numeraire, moments of every chain
    float x2[E];
    { const gfloat4* __restrict__ p = reinterpret_cast<const gfloat4*>(rowp[0]); _Pragma("unroll") for (int t = 0; t < T; ++t) { const f32x4 x = load_stream(p, i4c[t]); x2[4 * t] = x.x; x2[4 * t + 1] = x.y; x2[4 * t + 2] = x.z; x2[4 * t + 3] = x.w; } }
    const bool pass_full = ((int64_t)tile * T + T) * (FM_BLOCK * FM_VEC) <= n;
    _Pragma("unroll") for (int k = 0; k < K; ++k) {
        float q0[E], q1[E];
        _Pragma("unroll") for (int j = 0; j < E; ++j) q0[j] = ueval<U_FLOOR_S>(c[k][j], 0.f, 0.f, 0.0f);
        ueval_div_all<U_DIV, E>(q1, q0, x2, nullptr, 0.f);
        double acc_sum[1] = { 0.0 }, acc_sq[1] = { 0.0 }, shift[1] = { 0.0 }; unsigned long long nan_mask[1] = { 0ull };
        float acc_min[1] = { __builtin_huge_valf() }, acc_max[1] = { -__builtin_huge_valf() };
        red_accumulate<E>(q1, shift[0], pass_full, i4, n, acc_sum[0], acc_sq[0], acc_min[0], acc_max[0], nan_mask[0]);
        red_unit_end<1>(acc_sum, acc_sq, acc_min, acc_max, nan_mask, shift, 0u);
        red_span_fold<1>(1u, true);
        if ((threadIdx.x >> 6) == red_keeper_wave() && (threadIdx.x & 63u) == 0u) { M.wg_sums[k] = red_shared<1>().wg_sums[0]; M.wg_ext[k] = red_shared<1>().wg_ext[0]; }
    }
    // hand-off: K partials per workgroup, one arrival; the last workgroup of the row adds them per chain (one counter: prototype)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (wave != red_keeper_wave()) return;
    const uint32_t slots = gridDim.x;
    uint32_t last = 0u;
    if (lane == 0u) {
        _Pragma("unroll") for (int k = 0; k < K; ++k) {
            double* out = A.partials + (((size_t)blockIdx.y * K + k) * slots + blockIdx.x) * 4;
            store_coherent(out + 0, M.wg_sums[k].x); store_coherent(out + 1, M.wg_sums[k].y); store_coherent(out + 2, (double)M.wg_ext[k].x); store_coherent(out + 3, (double)M.wg_ext[k].y);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t* counter = A.counters + (size_t)blockIdx.y * FM_COUNTER_STRIDE + (size_t)(blockIdx.x % 7u) * FM_COUNTER_PLANE;
        const uint32_t members = (gridDim.x - (blockIdx.x % 7u) + 6u) / 7u;
        const uint32_t arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (arrived == members - 1u) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t* second = A.counters + (size_t)blockIdx.y * FM_COUNTER_STRIDE + (size_t)7 * FM_COUNTER_PLANE;
            const uint32_t groups = __hip_atomic_fetch_add(second, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            if (groups == 6u) { last = 1u; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __hip_atomic_store(second, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
    if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
    for (int k = 0; k < K; ++k) {
        double s1, s2; float mn, mx;
        wave_sum_span_partials(A.partials + ((size_t)blockIdx.y * K + k) * slots * 4, 0u, 1u, (gridDim.x + 3u) / 4u, 4u, gridDim.x, s1, s2, mn, mx);
        if (lane == 63u) { double* o = A.results + ((size_t)blockIdx.y * K + k) * 4; o[0] = s1; o[1] = s2; o[2] = mn; o[3] = mx; }
    }
}
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 7, launches = argc > 3 ? atoi(argv[3]) : 10, elems = argc > 4 ? atoi(argv[4]) : 8;
    const long n = argc > 2 ? atol(argv[2]) : 1000000;
    constexpr int K = K_;
    const int periods14[14] = { 60, 50, 40, 30, 20, 18, 16, 14, 12, 10, 8, 6, 4, 2 };
    const int R = periods14[14 - K];                                        // K = 14: all tenors; fewer: the shorter ones (an exercise date beyond 10 years)
    const size_t stride = ((size_t)n + 63) / 64 * 64;
    const size_t per_row = 1 + (size_t)R;
    float* slab = nullptr;
    CK(hipMalloc(&slab, (size_t)rows * per_row * stride * 4));
    {
        std::vector<float> h(stride);
        for (size_t i = 0; i < stride; ++i) h[i] = 0.01f + 0.02f * (float)((i * 2654435761u) % 100000) * 1e-5f;
        for (size_t v = 0; v < (size_t)rows * per_row; ++v) CK(hipMemcpy(slab + v * stride, h.data(), stride * 4, hipMemcpyHostToDevice));
    }
    const size_t rw = 1 + (size_t)R + K;                                      // K x two floats
    std::vector<uint64_t> table((size_t)rows * rw, 0);
    size_t next = 0;
    auto vec = [&]() { return (uint64_t)(uintptr_t)(slab + ((next++ * 7919u) % ((size_t)rows * per_row)) * stride); };
    for (int r = 0; r < rows; ++r) {
        uint64_t* row = table.data() + (size_t)r * rw;
        for (size_t i = 0; i < 1 + (size_t)R; ++i) row[i] = vec();
        float* sc = reinterpret_cast<float*>(row + 1 + R);
        for (int k = 0; k < K; ++k) { sc[2 * k] = 0.02f; uint32_t first = (uint32_t)(R - periods14[14 - K + k]); std::memcpy(&sc[2 * k + 1], &first, 4); }
    }
    uint64_t* dev_rows = nullptr; void* dump = nullptr;
    CK(hipMalloc(&dev_rows, table.size() * 8)); CK(hipMalloc(&dump, fm::FM_DUMP_BYTES));
    CK(hipMemcpy(dev_rows, table.data(), table.size() * 8, hipMemcpyHostToDevice));
    fm::MergedArgs a{};
    const long per_pass = (long)fm::FM_BLOCK * elems;
    const uint32_t tiles = (uint32_t)((n + per_pass - 1) / per_pass);
    a.n = n; a.row_words = (uint32_t)rw; a.iterations = (uint32_t)R; a.dump = (uint64_t)(uintptr_t)dump;
    CK(hipMalloc(&a.partials, (size_t)rows * K * (tiles + 8) * 32)); CK(hipMalloc(&a.results, (size_t)rows * K * 32));
    CK(hipMalloc(&a.counters, fm::FM_COUNTER_PLANES * fm::FM_COUNTER_PLANE * sizeof(uint32_t))); CK(hipMemset(a.counters, 0, fm::FM_COUNTER_PLANES * fm::FM_COUNTER_PLANE * sizeof(uint32_t)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 4.0 * n * rows * (double)per_row;
    long reads_unmerged = 0; for (int k = 0; k < K; ++k) reads_unmerged += periods14[14 - K + k] + 1;
    float best = 1e30f, sum = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; ++l) {
            if (elems == 8) fm::merged_kernel<K, 8><<<dim3(tiles, rows), fm::FM_BLOCK>>>(a, dev_rows);
            else fm::merged_kernel<K, 4><<<dim3(tiles, rows), fm::FM_BLOCK>>>(a, dev_rows);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) { sum += ms; if (ms < best) best = ms; }
    }
    hipFuncAttributes attr;
    if (elems == 8) CK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&fm::merged_kernel<K, 8>)));
    else CK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&fm::merged_kernel<K, 4>)));
    std::vector<double> res((size_t)rows * K * 4);
    CK(hipMemcpy(res.data(), a.results, res.size() * 8, hipMemcpyDeviceToHost));
    printf("K %d rows %d iterations %d paths %ld elems %d, %d VGPRs: %.1f us per launch (best of 3 x %d; mean %.1f), %.0f GB/s of its own reads; the %ld reads of one launch per chain at 6.5 TB/s: %.1f us; mean[0][0] %.9g mean[0][K-1] %.9g\n",
           K, rows, R, n, elems, attr.numRegs, best * 1e3 / launches, launches, sum / 3 * 1e3 / launches, bytes / (best * 1e-3 / launches) / 1e9,
           reads_unmerged, 4.0 * n * rows * reads_unmerged / 6.5e12 * 1e6, res[0] / n, res[(size_t)(K - 1) * 4] / n);
    return 0;
}
