// peeled_harness.hip — ONE peeled (head + loop + tail) kernel of the kernel pack in isolation, no engine: synthetic vectors and row
// tables of the shape the LMM calibration launches it with, timed with HIP events; the target of source-level experiments (generator
// knobs, hand-edited variants of the generated source) that the engine's own launches cannot isolate.
//   source: finmath-lib-cuda-extensions_amd/build/jit_pack_tool --source finmath-lib-cuda-extensions_amd/csrc/kernel_pack.txt fm_jit_<hash> > k.hip
//           (its first comment line names the counts below)
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -structurizecfg-skip-uniform-regions \
//                 -I finmath-lib-cuda-extensions_amd/csrc -DKERNEL_SOURCE='"k.hip"' -DNX_=3 -DG_=0 -DCO_=1 -DNXO_=1 -DLI_=1 -DLO_=0 [-DREDUCE_] benchmarks/peeled_harness.hip -o …
//   run:    peeled_harness [rows=88] [iterations=58] [paths=1000000] [launches=10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include KERNEL_SOURCE
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 88, R = argc > 2 ? atoi(argv[2]) : 58, launches = argc > 4 ? atoi(argv[4]) : 10;
    const long n = argc > 3 ? atol(argv[3]) : 1000000;
    const size_t NX = NX_, G = G_, CO = CO_, NXO = NXO_, LI = LI_, LO = LO_;
    const size_t stride = ((size_t)n + 63) / 64 * 64;                       // floats per vector, 256-byte aligned
    const size_t n_in = NX + G + (size_t)R * LI, n_out = CO + NXO + (size_t)R * LO, per_row = n_in + n_out;
    float* slab = nullptr;
    CK(hipMalloc(&slab, (size_t)rows * per_row * stride * 4));
    {   // forward rates around 2 %
        std::vector<float> h(stride);
        for (size_t i = 0; i < stride; ++i) h[i] = 0.01f + 0.02f * (float)((i * 2654435761u) % 100000) * 1e-5f;
        for (size_t v = 0; v < (size_t)rows * per_row; ++v) CK(hipMemcpy(slab + v * stride, h.data(), stride * 4, hipMemcpyHostToDevice));
    }
    const size_t n_scal = 256 + (size_t)R * 64;                             // more than any kernel reads; all 0.5
    const size_t rw = NX + G + CO + NXO + (size_t)R * (LI + LO) + (n_scal + 1) / 2;
    std::vector<uint64_t> table((size_t)rows * rw, 0);
    size_t next = 0;
    auto vec = [&]() { return (uint64_t)(uintptr_t)(slab + ((next++ * 7919u) % ((size_t)rows * per_row)) * stride); };      // anywhere in the slab, as the pool hands them out
    for (int r = 0; r < rows; ++r) {
        uint64_t* row = table.data() + (size_t)r * rw;
        size_t k = 0;
        for (size_t i = 0; i < NX + G + CO + NXO; ++i) row[k++] = vec();
        for (int it = 0; it < R; ++it) for (size_t i = 0; i < LI + LO; ++i) row[k++] = vec();
        float* sc = reinterpret_cast<float*>(row + k);
        for (size_t i = 0; i < n_scal; ++i) sc[i] = 0.5f;
    }
    uint64_t* dev_rows = nullptr; void* dump = nullptr;
    CK(hipMalloc(&dev_rows, table.size() * 8)); CK(hipMalloc(&dump, fm::FM_DUMP_BYTES));
    CK(hipMemcpy(dev_rows, table.data(), table.size() * 8, hipMemcpyHostToDevice));
    fm::DevRolledArgs a{};
    const long per_pass = (long)fm::FM_BLOCK * 8;
    a.n = n; a.tiles_per_row = (uint32_t)((n + per_pass - 1) / per_pass); a.row_words = (uint32_t)rw; a.iterations = (uint32_t)R; a.dump = (uint64_t)(uintptr_t)dump;
    CK(hipMalloc(&a.partials, (size_t)rows * (a.tiles_per_row + 8) * 32)); CK(hipMalloc(&a.results, (size_t)rows * 32));
    CK(hipMalloc(&a.counters, fm::FM_COUNTER_PLANES * fm::FM_COUNTER_PLANE * sizeof(uint32_t))); CK(hipMemset(a.counters, 0, fm::FM_COUNTER_PLANES * fm::FM_COUNTER_PLANE * sizeof(uint32_t)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 4.0 * n * rows * (double)(n_in + NXO + (size_t)R * LO);        // what the engine counts: inputs read + values stored
    float best = 1e30f, sum = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; ++l) fm_jit_table<<<dim3(a.tiles_per_row, rows), fm::FM_BLOCK>>>(a, dev_rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) { sum += ms; if (ms < best) best = ms; }
    }
    hipFuncAttributes attr; CK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&fm_jit_table)));
    printf("rows %d iterations %d paths %ld, %d VGPRs: %.1f us per launch (best of 3 x %d; mean %.1f), %.0f GB/s algorithmic\n", rows, R, n, attr.numRegs, best * 1e3 / launches, launches,
           sum / 3 * 1e3 / launches, bytes / (best * 1e-3 / launches) / 1e9);
    return 0;
}
