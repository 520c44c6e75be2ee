// sharded.hpp — ONE process driving SEVERAL devices behind the unchanged C-ABI (fmhip_init_devices).
//
// The path shards by Monte-Carlo paths: every vector of n paths is cut into contiguous blocks (boundaries at multiples of four paths:
// one Philox call covers four), block d lives on device d, every element-wise method runs on every block, and only expectations
// couple the blocks.  Design: one ENGINE per device shard (the single-device engine, unchanged: its own stream, pool, row-table ring,
// arrival counters, moments arena, specialised-kernel tier), each driven by a WORKER thread of its own; the caller's C-ABI calls are
// replayed on every worker, in order, through the very same entry points (a worker thread's Engine::get() is its shard's engine).
// The caller's handles are the front's own numbers; each worker keeps the table front handle → its engine's handle.  Calls that
// return nothing but handles are posted and return at once — the front checks what can be checked without the engines (handles,
// sizes, opcode shapes), anything else a shard reports (a device allocation that fails …) surfaces at the next call that waits.  Calls
// that return data wait for all shards: a read gathers the blocks by one device-to-host copy per shard, straight into the caller's
// buffer; host-side moments need NO collective — every shard's launch leaves its 32 bytes in pinned memory, the front adds them in
// shard order by the rule fmhip_expectation_combine implements (sums in shard order, java.lang.Math.min / max).  Nothing is exchanged
// between devices for those.  Expectations wanted ON the devices (fmhip_reduce_moments_batch_devices; the *_device variants deliver to
// the first listed device) are the one exchange: a grouped RCCL all-gather of the shards' moments over the listed devices and a combine
// kernel per device — or, where a device index repeats, a host combine.  Raw device pointers of vectors and the expectation communicator
// are not available with a device list (they name ONE device).
//
// The reference has one device index (RandomVariableCuda.java:161,177).  UNMEASURED on more than one physical GPU: the boxes this was
// built on have one; tests use the device lists {0, 0} and {0, 0, 0} — shards on separate streams of one device.
#pragma once
#include <cstdint>
#include "../../include/fmhip.h"

namespace fm {
// true when a device list is active AND the calling thread is a caller's (not a shard's worker): the C-ABI entry points then hand
// over to the functions below (same signatures and statuses as their fmhip_* namesakes)
bool front_active();
void shard_range(int64_t n, int shards, int shard, int64_t* offset, int64_t* count);
namespace front {
int init_devices(const int* devices, int count);
int shutdown();
int device_info(char* name_buf, int name_buf_len, int* n_compute_units, int64_t* hbm_bytes);
int device_count(int* count);
int synchronize();
int vec_create_from_host(const void* host_values, bool is_double, int64_t n, fmhip_vec* out);
int vec_create_filled(int64_t n, double value, bool initialised, fmhip_vec* out);
int vec_retain(fmhip_vec v);
int vec_release(fmhip_vec v);
int vec_size(fmhip_vec v, int64_t* n_out);
int vec_read(fmhip_vec v, void* host_out, bool as_double, int64_t n);
int call(int opcode, int n_in, const fmhip_vec* in, double scalar, bool has_scalar, fmhip_vec* out);
int set_int(int what, int value, int* previous);          // what: 0 fusion, 1 hold, 2 step grouping, 3 math mode, 4 jit mode
int flush();
int graph_clone(const fmhip_vec* roots, int n_roots, int n_copies, const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map, const double* scalars, int n_scalars, fmhip_vec* out);
int graph_scalars(const fmhip_vec* roots, int n_roots, double* scalars_out, int capacity, int* n_scalars);
int reduce_moments_batch(const fmhip_vec* vectors, int count, const double* shifts, fmhip_moments* out);
int reduce_moments_batch_begin(const fmhip_vec* vectors, int count, const double* shifts, fmhip_ticket* ticket_out);
int reduce_moments_batch_end(fmhip_ticket ticket, fmhip_moments* out, int count);
int reduce_moments_batch_devices(const fmhip_vec* vectors, int count, const double* shifts, void* const* device_out, int n_devices);
int get_stream_of(int shard, void** stream_out);
int expectation_collective(int* kind, char* why, int why_len);
int vec_give_up_values(const fmhip_vec* vectors, int count);
int program_create(const fmhip_prog_op* ops, int n_ops, int n_inputs, const int32_t* out_values, int n_outputs, const int32_t* reduce_values, int n_reduce, fmhip_program* out);
int program_release(fmhip_program p);
int program_shape(fmhip_program p, int* n_inputs, int* n_outputs, int* n_reduce);
int program_tier(fmhip_program p, int* tier, int* vgprs);
int program_run(fmhip_program p, int batch, const fmhip_vec* inputs, fmhip_vec* outputs, bool into, const double* reduce_shift, fmhip_moments* moments);
int jit_wait();
int jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, double* compile_seconds, int64_t* disk_cache_hits);
int bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset, const double* dt, fmhip_vec* out);
int pool(int what);                                        // 0 clean, 1 purge
int pool_stats(fmhip_pool_stats_t* out);
int traffic_stats(int64_t* algorithmic_bytes, int64_t* specialised_launches);
int engine_stats(fmhip_engine_stats_t* out);
int profile_enable(int enabled);
int profile_read(double* kernel_ms_total, int64_t* n_launches);
int unsupported(const char* what);
}
}
