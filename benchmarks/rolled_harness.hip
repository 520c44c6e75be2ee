// rolled_harness.hip — ONE rolled-loop kernel of the kernel pack in isolation (no engine): synthetic vectors and row tables of the shape
// the LMM calibration launches it with, timed with HIP events; the target of rocprofv3 --pmc passes and of source-level experiments.
//   build:  <dump the kernel's source>  then
//           hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -structurizecfg-skip-uniform-regions \
//                 -I finmath-lib-cuda-extensions_amd/csrc -DKERNEL_SOURCE='"<file>"' -DG_=4 -DCI_=4 -DCO_=4 -DLI_=1 -DLO_=1 -DLS_=24 -DELEMS_=8 benchmarks/rolled_harness.hip -o …
//   run:    rolled_harness [rows=8] [iterations=75] [paths=1000000] [launches=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include KERNEL_SOURCE
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 8, R = argc > 2 ? atoi(argv[2]) : 75, launches = argc > 4 ? atoi(argv[4]) : 20;
    const long n = argc > 3 ? atol(argv[3]) : 1000000;
    const size_t G = G_, CI = CI_, CO = CO_, LI = LI_, LO = LO_, LS = LS_;
    const size_t stride = ((size_t)n + 63) / 64 * 64;                       // floats per vector, 256-byte aligned
    const size_t per_row = G + CI + CO + (size_t)R * (LI + LO);
    float* slab = nullptr;
    CK(hipMalloc(&slab, (size_t)rows * per_row * stride * 4));
    {   // forward rates around 2 %, increments around ±0.7, running sums from zero
        std::vector<float> h(stride);
        for (size_t v = 0; v < (size_t)rows * per_row; ++v) {
            const size_t k = v % per_row;
            for (size_t i = 0; i < stride; ++i) {
                const float u = (float)((i * 2654435761u + v * 40503u) % 100000) * 1e-5f;
                h[i] = k < G ? (u - 0.5f) * 1.4f : k < G + CI ? 0.0f : 0.01f + 0.02f * u;
            }
            CK(hipMemcpy(slab + v * stride, h.data(), stride * 4, hipMemcpyHostToDevice));
        }
    }
    const size_t rw = per_row + ((size_t)R * LS + 1) / 2;
    std::vector<uint64_t> table((size_t)rows * rw, 0);
    for (int r = 0; r < rows; ++r) {
        uint64_t* row = table.data() + (size_t)r * rw;
        for (size_t k = 0; k < per_row; ++k) row[k] = (uint64_t)(uintptr_t)(slab + ((size_t)r * per_row + k) * stride);
        float* sc = reinterpret_cast<float*>(row + per_row);
        for (size_t i = 0; i < (size_t)R * LS; ++i) { static const float six[6] = { 0.5f, 1.0f, 0.0025f, 0.005f, 0.5f, 0.005f }; sc[i] = six[i % 6]; }
    }
    uint64_t* dev_rows = nullptr; void* dump = nullptr;
    CK(hipMalloc(&dev_rows, table.size() * 8)); CK(hipMalloc(&dump, fm::FM_DUMP_BYTES));
    CK(hipMemcpy(dev_rows, table.data(), table.size() * 8, hipMemcpyHostToDevice));
    fm::DevRolledArgs a{};
    const long per_pass = (long)fm::FM_BLOCK * ELEMS_;
    a.n = n; a.tiles_per_row = (uint32_t)((n + per_pass - 1) / per_pass); a.row_words = (uint32_t)rw; a.iterations = (uint32_t)R; a.dump = (uint64_t)(uintptr_t)dump;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int l = 0; l < launches; ++l) fm_jit_table<<<dim3(a.tiles_per_row, rows), fm::FM_BLOCK>>>(a, dev_rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 4.0 * n * rows * (double)per_row;
        printf("rows %d iterations %d paths %ld: %.1f us per launch, %.0f GB/s algorithmic, %.2f us per iteration and workgroup round\n", rows, R, n, ms * 1e3 / launches,
               bytes / (ms * 1e-3 / launches) / 1e9, ms * 1e3 / launches / R / ((double)a.tiles_per_row * rows / 1024.0));
    }
    return 0;
}
