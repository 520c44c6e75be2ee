// kernels.h — host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "fm_program.h"

namespace fm {

struct DevBmArgs {
    float*       slab;           // n_streams vectors, `stride_floats` apart (stride is a multiple of 64 floats)
    const float* sqrt_dt;        // [n_streams]  (float)sqrt(dt[step of the stream]), one entry per local stream
    int64_t      stride_floats;
    int64_t      n_paths;        // paths held by this process
    int64_t      path_offset;    // global index of local path 0 (path sharding over GPUs)
    uint32_t     key0, key1;     // lo32(seed), hi32(seed)
    uint32_t     n_factors;
    uint32_t     stream0;        // global stream index (step*n_factors+factor) of local stream 0
};

hipError_t launch_program(const DevProgramArgs& a, const uint64_t* rows, double* partials,
                          uint32_t blocks_per_row, uint32_t batch, hipStream_t st);
hipError_t launch_bm(const DevBmArgs& a, uint32_t n_streams, hipStream_t st);
hipError_t launch_fill(float* p, float v, int64_t n_padded, hipStream_t st);
hipError_t preload_kernels();        // makes the device code of every kernel above resident (the runtime would load it at first launch)

// {Σ, Σ², min, max} blocks of 32 bytes collected from wherever the launches that took them left them (slots of the pinned moments arena,
// mapped into the device's address space) into one contiguous device buffer — the send buffer of an RCCL exchange.
constexpr int FM_GATHER_MAX = 384;                 // sources per launch: they travel in the kernel arguments (3 KB of the 4 KB segment)
struct DevGatherArgs { uint32_t count; uint32_t pad; uint64_t src[FM_GATHER_MAX]; };
hipError_t launch_gather_moments(const DevGatherArgs& a, double* out, hipStream_t st);
// gathered[world][count][4] → out[count][4]: the shards' moments combined in shard order by fmhip_expectation_combine's rule (on the device that holds them)
hipError_t launch_combine_moments(const double* gathered, uint32_t world, uint32_t count, double* out, hipStream_t st);

} // namespace fm
