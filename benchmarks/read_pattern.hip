// read_pattern.hip — is the valuation chains' READ PATTERN slower than a linear sweep?  A workgroup (tile, row) reads its 8 KB tile
// from each of R different 4 MB vectors, one per iteration, the next iteration's loads in flight (what a peeled valuation kernel does:
// 56-88 rows x 61 vectors per launch), with F dependent packed multiply-adds per element pair and iteration standing in for the
// arithmetic.  Compared with: the same bytes as ONE vector per row swept tile after tile (rows x 1 vector of R x 4 MB).
//   run: read_pattern [rows=88] [R=61] [launches=5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// TAIL: what a valuation kernel does behind its loop — one more vector read when the loop is over (the numeraire), a barrier, one
// lane's device-scope atomic and its round trip (the hand-off of the fused expectation).  lds_pad: dynamic LDS that limits the
// workgroups a CU holds (160 KB per CU).
template <int F, int DEPTH, bool TAIL, bool STORE = false>
__global__ void __launch_bounds__(256) chains(const float* const* __restrict__ table, int R, size_t stride_tiles, float* __restrict__ out, float s, unsigned* __restrict__ counters) {
    extern __shared__ float lds_pad[];
    // table[row * R + k] = vector k of the row; tile t of it = 2048 floats at t * 2048 (scattered layout) — or, linear layout, the
    // caller passes pointers k * tiles * 2048 apart inside one allocation per row
    const float* const* row = table + (size_t)blockIdx.y * R;
    const size_t off = (size_t)blockIdx.x * 2048 / 4;
    f4 nx[DEPTH][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < R) { const f4* p = reinterpret_cast<const f4*>(row[d]) + off; nx[d][0] = __builtin_nontemporal_load(p + threadIdx.x); nx[d][1] = __builtin_nontemporal_load(p + 256 + threadIdx.x); }
    f2 acc[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
    const f2 c = { s, s };
    for (int k = 0; k < R; ++k) {
        f4 v0 = nx[0][0], v1 = nx[0][1];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) { nx[d][0] = nx[d + 1][0]; nx[d][1] = nx[d + 1][1]; }
        if (k + DEPTH < R) { const f4* p = reinterpret_cast<const f4*>(row[k + DEPTH]) + off; nx[DEPTH - 1][0] = __builtin_nontemporal_load(p + threadIdx.x); nx[DEPTH - 1][1] = __builtin_nontemporal_load(p + 256 + threadIdx.x); }
        f2 x[4] = { { v0.x, v0.y }, { v0.z, v0.w }, { v1.x, v1.y }, { v1.z, v1.w } };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int f = 0; f < F; ++f) x[j] = __builtin_elementwise_fma(x[j], c, acc[j]);
            acc[j] = F ? x[j] : acc[j] + x[j];
        }
    }
    f2 t = acc[0] + acc[1] + acc[2] + acc[3];
    if constexpr (TAIL) {
        const f4* p = reinterpret_cast<const f4*>(row[0]) + off;       // (a vector read before: any would do, it comes from memory again)
        const f4 a = __builtin_nontemporal_load(p + threadIdx.x), b = __builtin_nontemporal_load(p + 256 + threadIdx.x);
        t.x += a.x * b.y; t.y += a.z + b.w;
        lds_pad[threadIdx.x] = t.x + t.y;
        __syncthreads();
        if (threadIdx.x == (blockIdx.x & 3u) * 64u) {
            const unsigned arrived = __hip_atomic_fetch_add(counters + blockIdx.y * 64u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == 0xffffffffu) out[1] = lds_pad[7];
        }
    }
    if constexpr (STORE) {        // one value per element of the tile into the row's result vector (the vector behind the row's last input: nobody reads it here)
        f4* q = reinterpret_cast<f4*>(const_cast<float*>(row[R - 1])) + off;
        const f4 v0 = { t.x, t.y, acc[0].x, acc[1].y }, v1 = { acc[2].x, acc[3].y, t.y, t.x };
        __builtin_nontemporal_store(v0, q + threadIdx.x); __builtin_nontemporal_store(v1, q + 256 + threadIdx.x);
    }
    if (t.x + t.y == 12345.678f) out[blockIdx.x] = t.x;
}

static unsigned* g_counters = nullptr;
template <int F, int DEPTH, bool TAIL = false, bool STORE = false>
static int measure(const char* what, const float* const* dtab, int rows, int R, int tiles, float* out, int launches, double bytes, int wgs_per_cu = 8) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int l = 0; l < launches + 1; ++l) {
        CK(hipEventRecord(e0));
        const size_t lds = wgs_per_cu >= 8 ? 1024 : (size_t)(160 * 1024 / wgs_per_cu - 1024) & ~size_t(255);
        chains<F, DEPTH, TAIL, STORE><<<dim3(tiles, rows), 256, lds>>>(dtab, R, 0, out, 1.0000001f, g_counters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (l > 0 && ms < best) best = ms;
    }
    printf("  %-30s %2d multiply-adds per pair, %d ahead, %d workgroups per CU%s: %8.1f us  %6.0f GB/s\n", what, F, DEPTH, wgs_per_cu, TAIL ? (STORE ? ", tail (read + barrier + atomic), one tile stored" : ", tail (read + barrier + atomic)") : (STORE ? ", one tile stored" : ""), best * 1e3, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 88, R = argc > 2 ? atoi(argv[2]) : 61, launches = argc > 3 ? atoi(argv[3]) : 5;
    const size_t n = 1000000, stride = 1000064;           // floats per vector as the pool lays them out (4,000,256 bytes apart)
    const int tiles = (int)((n + 2047) / 2048);
    const size_t vecs = (size_t)rows * R;
    float* slab = nullptr; float* out = nullptr;
    CK(hipMalloc(&slab, vecs * stride * 4 + 65536)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(slab, 0x3c, vecs * stride * 4));
    const double bytes = 4.0 * 2048.0 * tiles * (double)vecs;
    printf("%d rows x %d vectors of %zu floats (%.1f GB), %d tiles per vector, best of %d launches\n", rows, R, n, vecs * stride * 4 / 1e9, tiles, launches);
    std::vector<const float*> h(vecs);
    const float** dtab = nullptr; CK(hipMalloc(&dtab, vecs * 8));
    // (a) as the engine has them: vector k of row r somewhere in the pool — here shuffled over the slab
    for (size_t v = 0; v < vecs; ++v) h[v] = slab + ((v * 7919u) % vecs) * stride;
    CK(hipMemcpy(dtab, h.data(), vecs * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_counters, 65536 * 4)); CK(hipMemset(g_counters, 0, 65536 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chains<8, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chains<8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chains<8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    printf("scattered: every (row, iteration) a vector of its own, anywhere in a %.1f GB slab\n", vecs * stride * 4 / 1e9);
    printf(" occupancy and the tail of a valuation kernel (8 multiply-adds per pair, one iteration ahead):\n");
    for (int w : { 8, 6, 5, 4, 3 }) { if (measure<8, 1, false>("", dtab, rows, R, tiles, out, launches, bytes, w)) return 1; }
    for (int w : { 8, 6, 5, 4, 3 }) { if (measure<8, 1, true>("", dtab, rows, R, tiles, out, launches, bytes, w)) return 1; }
    for (int w : { 6, 5, 4 }) { if (measure<8, 2, true>("", dtab, rows, R, tiles, out, launches, bytes, w)) return 1; }
    printf(" one 8 KB store per workgroup at its end (1 written vector per %d read):\n", R);
    if (measure<8, 1, false, true>("", dtab, rows, R, tiles, out, launches, bytes, 8)) return 1;
    if (measure<8, 1, true, true>("", dtab, rows, R, tiles, out, launches, bytes, 8)) return 1;
    if (measure<8, 1, true, true>("", dtab, rows, R, tiles, out, launches, bytes, 6)) return 1;
    printf(" look-ahead and arithmetic at full occupancy:\n");
    if (measure<0, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<0, 2>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<0, 4>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<8, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<8, 2>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<16, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<16, 2>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    // (b) consecutive: vector k of row r directly behind vector k-1 of the same row
    for (size_t v = 0; v < vecs; ++v) h[v] = slab + v * stride;
    CK(hipMemcpy(dtab, h.data(), vecs * 8, hipMemcpyHostToDevice));
    printf("consecutive: the vectors of a row one behind the other\n");
    if (measure<0, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<8, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    // (c) few distinct vectors: every row reads the SAME R vectors (244 MB in all: the memory-side cache holds them) — the TLB reach question
    for (size_t v = 0; v < vecs; ++v) h[v] = slab + (v % (size_t)R) * stride;
    CK(hipMemcpy(dtab, h.data(), vecs * 8, hipMemcpyHostToDevice));
    printf("shared: every row reads the same %d vectors (%.0f MB)\n", R, R * stride * 4 / 1e6);
    if (measure<0, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    if (measure<8, 1>("", dtab, rows, R, tiles, out, launches, bytes)) return 1;
    return 0;
}
