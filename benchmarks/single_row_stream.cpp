// single_row_stream.cpp — stream S (bench.py's program: 12 fused path-ops + 4 fused reductions) on ONE row and on 8 rows, through the C-ABI:
// what a launch costs alone (one HIP event pair around it: what fmhip_profile_* and a tracing profiler report) against what it costs in a
// stream of launches (K launches between ONE pair of events on the runtime stream, and by the host's clock).  benchmarks/launch_latency.hip
// says what the difference is made of.
//   build: hipcc -O2 -std=c++17 -I include benchmarks/single_row_stream.cpp -L finmath-lib-cuda-extensions_amd/lib -lfmhip \
//                -Wl,-rpath,'$ORIGIN/../../finmath-lib-cuda-extensions_amd/lib' -o benchmarks/build/single_row_stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fmhip.h"
#define FM(x) do { if ((x) != FMHIP_OK) { printf("fmhip error: %s (line %d)\n", fmhip_last_error(), __LINE__); return 1; } } while (0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    FM(fmhip_init(-1));
    FM(fmhip_set_jit(FMHIP_JIT_SYNC, nullptr));
    // t = (x + 4) / 2 * y − z; u = sqrt|log exp t|; v = floor(cap(u, 1.5), 0.25) + y·z; w = t >= 0 ? v : x   (bench.py: build_stream_s)
    const fmhip_prog_op ops[] = {
        { FMHIP_OP_ADD_S, 0, -1, -1, 4.0 }, { FMHIP_OP_DIV_S, 3, -1, -1, 2.0 }, { FMHIP_OP_MULT, 4, 1, -1, 0 }, { FMHIP_OP_SUB, 5, 2, -1, 0 },
        { FMHIP_OP_EXP, 6, -1, -1, 0 }, { FMHIP_OP_LOG, 7, -1, -1, 0 }, { FMHIP_OP_ABS, 8, -1, -1, 0 }, { FMHIP_OP_SQRT, 9, -1, -1, 0 },
        { FMHIP_OP_CAP_S, 10, -1, -1, 1.5 }, { FMHIP_OP_FLOOR_S, 11, -1, -1, 0.25 }, { FMHIP_OP_ADDPRODUCT, 12, 1, 2, 0 }, { FMHIP_OP_CHOOSE, 6, 13, 0, 0 } };
    const int32_t out_value = 14;
    fmhip_program prog;
    FM(fmhip_program_create(ops, 12, 3, &out_value, 1, &out_value, 1, &prog));
    void* stream_v = nullptr; FM(fmhip_get_stream(&stream_v));
    hipStream_t stream = (hipStream_t)stream_v;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"workload\": \"stream S + fused reductions, specialised tier, one launch per fmhip_program_run_into\", \"results\": [\n");
    const long sizes[] = { 10000, 100000, 1000000 };
    const int batches[] = { 1, 8 };
    bool first = true;
    for (long n : sizes) for (int B : batches) {
        std::vector<float> h((size_t)n);
        std::vector<fmhip_vec> in((size_t)B * 3), out((size_t)B);
        for (int k = 0; k < B * 3; ++k) {
            for (long i = 0; i < n; ++i) h[(size_t)i] = 0.5f + 0.001f * (float)((i * 2654435761u + (unsigned)k * 40503u) % 1000);
            FM(fmhip_vec_create_from_float(h.data(), n, &in[(size_t)k]));
        }
        for (int k = 0; k < B; ++k) FM(fmhip_vec_create_from_float(h.data(), n, &out[(size_t)k]));
        std::vector<fmhip_moments> m((size_t)B);
        for (int r = 0; r < 200; ++r) FM(fmhip_program_run_into(prog, B, in.data(), out.data(), nullptr, nullptr, nullptr));
        FM(fmhip_synchronize());
        // alone: one event pair per launch
        FM(fmhip_profile_enable(1));
        for (int r = 0; r < 500; ++r) FM(fmhip_program_run_into(prog, B, in.data(), out.data(), nullptr, nullptr, nullptr));
        double ms = 0; int64_t k = 0; FM(fmhip_profile_read(&ms, &k)); FM(fmhip_profile_enable(0));
        const double alone_us = ms * 1e3 / (double)k;
        // in a stream: K launches between one pair of events, and by the host's clock
        const int K = 4000;
        FM(fmhip_synchronize());
        const auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, stream));
        for (int r = 0; r < K; ++r) FM(fmhip_program_run_into(prog, B, in.data(), out.data(), nullptr, nullptr, nullptr));
        CK(hipEventRecord(e1, stream));
        const auto h1 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(e1));
        const auto h2 = std::chrono::steady_clock::now();
        float dev_ms = 0; CK(hipEventElapsedTime(&dev_ms, e0, e1));
        // with the expectations read on the host every time (what getAverage() waits for)
        FM(fmhip_synchronize());
        const auto g0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 1000; ++r) FM(fmhip_program_run_into(prog, B, in.data(), out.data(), nullptr, m.data(), nullptr));
        const auto g1 = std::chrono::steady_clock::now();
        const double stream_us = dev_ms * 1e3 / K, bytes = 16.0 * (double)n * B;
        printf("%s {\"n\": %ld, \"batch\": %d, \"alone_between_two_events_us\": %.2f, \"in_a_stream_us\": %.2f, \"in_a_stream_GBps\": %.0f, \"in_a_stream_frac\": %.3f, "
               "\"host_enqueue_us\": %.2f, \"host_wall_us_per_launch\": %.2f, \"launch_and_read_expectations_us\": %.2f}",
               first ? "" : ",\n", n, B, alone_us, stream_us, bytes / (stream_us * 1e-6) / 1e9, bytes / (stream_us * 1e-6) / 8e12,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / K, std::chrono::duration<double, std::micro>(h2 - h0).count() / K,
               std::chrono::duration<double, std::micro>(g1 - g0).count() / 1000);
        first = false;
        fflush(stdout);
        for (fmhip_vec v : in) FM(fmhip_vec_release(v));
        for (fmhip_vec v : out) FM(fmhip_vec_release(v));
    }
    printf("\n],\n\"what_the_kernel_time_of_one_row_is_made_of\": [\n");
    {   // the same row with parts of the program taken away: K launches between one pair of events
        const fmhip_prog_op triad[] = { { FMHIP_OP_ADDPRODUCT, 0, 1, 2, 0 } };
        const int32_t triad_out = 3;
        fmhip_program p_s_nored, p_triad, p_triad_red;
        FM(fmhip_program_create(ops, 12, 3, &out_value, 1, nullptr, 0, &p_s_nored));
        FM(fmhip_program_create(triad, 1, 3, &triad_out, 1, nullptr, 0, &p_triad));
        FM(fmhip_program_create(triad, 1, 3, &triad_out, 1, &triad_out, 1, &p_triad_red));
        struct Case { const char* name; fmhip_program p; } cases[] = { { "a + b*c, no reduction (no log table)", p_triad }, { "a + b*c + reduction", p_triad_red },
                                                                       { "stream S, no reduction (log table in LDS)", p_s_nored }, { "stream S + reduction", prog } };
        bool first2 = true;
        for (long n : sizes) {
            std::vector<float> h((size_t)n, 1.25f);
            fmhip_vec in[3], out;
            for (int k = 0; k < 3; ++k) FM(fmhip_vec_create_from_float(h.data(), n, &in[k]));
            FM(fmhip_vec_create_from_float(h.data(), n, &out));
            for (const Case& c : cases) {
                for (int r = 0; r < 200; ++r) FM(fmhip_program_run_into(c.p, 1, in, &out, nullptr, nullptr, nullptr));
                FM(fmhip_synchronize());
                const int K = 4000;
                CK(hipEventRecord(e0, stream));
                for (int r = 0; r < K; ++r) FM(fmhip_program_run_into(c.p, 1, in, &out, nullptr, nullptr, nullptr));
                CK(hipEventRecord(e1, stream)); CK(hipEventSynchronize(e1));
                float dev_ms = 0; CK(hipEventElapsedTime(&dev_ms, e0, e1));
                printf("%s {\"n\": %ld, \"program\": \"%s\", \"in_a_stream_us\": %.2f}", first2 ? "" : ",\n", n, c.name, dev_ms * 1e3 / K);
                first2 = false;
            }
            for (int k = 0; k < 3; ++k) FM(fmhip_vec_release(in[k]));
            FM(fmhip_vec_release(out));
        }
        FM(fmhip_program_release(p_s_nored)); FM(fmhip_program_release(p_triad)); FM(fmhip_program_release(p_triad_red));
    }
    printf("\n],\n\"sixty_four_rows_of_1000000_paths\": [\n");
    {   // the headline shape: the engine's kernels for a + b*c and for stream S, with and without their fused reduction, K launches between one pair of events
        const fmhip_prog_op triad[] = { { FMHIP_OP_ADDPRODUCT, 0, 1, 2, 0 } };
        const int32_t triad_out = 3;
        fmhip_program p_s_nored, p_triad, p_triad_red;
        FM(fmhip_program_create(ops, 12, 3, &out_value, 1, nullptr, 0, &p_s_nored));
        FM(fmhip_program_create(triad, 1, 3, &triad_out, 1, nullptr, 0, &p_triad));
        FM(fmhip_program_create(triad, 1, 3, &triad_out, 1, &triad_out, 1, &p_triad_red));
        struct Case { const char* name; fmhip_program p; } cases[] = { { "a + b*c", p_triad }, { "a + b*c + reduction", p_triad_red }, { "stream S, no reduction", p_s_nored }, { "stream S + reduction", prog } };
        const long n = 1000000; const int B = 64;
        std::vector<float> h((size_t)n);
        std::vector<fmhip_vec> in((size_t)B * 3), out((size_t)B);
        for (int k = 0; k < B * 3; ++k) {
            for (long i = 0; i < n; ++i) h[(size_t)i] = 0.5f + 0.001f * (float)((i * 2654435761u + (unsigned)k * 40503u) % 1000);
            FM(fmhip_vec_create_from_float(h.data(), n, &in[(size_t)k]));
        }
        for (int k = 0; k < B; ++k) FM(fmhip_vec_create_from_float(h.data(), n, &out[(size_t)k]));
        bool first3 = true;
        for (int round = 0; round < 2; ++round)
        for (const Case& c : cases) {
            for (int r = 0; r < 30; ++r) FM(fmhip_program_run_into(c.p, B, in.data(), out.data(), nullptr, nullptr, nullptr));
            FM(fmhip_synchronize());
            const int K = 200;
            CK(hipEventRecord(e0, stream));
            for (int r = 0; r < K; ++r) FM(fmhip_program_run_into(c.p, B, in.data(), out.data(), nullptr, nullptr, nullptr));
            CK(hipEventRecord(e1, stream)); CK(hipEventSynchronize(e1));
            float dev_ms = 0; CK(hipEventElapsedTime(&dev_ms, e0, e1));
            const double us = dev_ms * 1e3 / K;
            printf("%s {\"program\": \"%s\", \"us\": %.1f, \"GBps\": %.0f, \"frac\": %.4f}", first3 ? "" : ",\n", c.name, us, 16.0 * n * B / (us * 1e-6) / 1e9, 16.0 * n * B / (us * 1e-6) / 8e12);
            first3 = false;
        }
        for (fmhip_vec v : in) FM(fmhip_vec_release(v));
        for (fmhip_vec v : out) FM(fmhip_vec_release(v));
        FM(fmhip_program_release(p_s_nored)); FM(fmhip_program_release(p_triad)); FM(fmhip_program_release(p_triad_red));
    }
    printf("\n]}\n");
    FM(fmhip_program_release(prog));
    FM(fmhip_shutdown());
    return 0;
}
