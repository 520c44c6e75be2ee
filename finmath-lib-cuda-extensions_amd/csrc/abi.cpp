// abi.cpp — the extern "C" surface declared in include/fmhip.h.  No logic: argument checks, the engine
// lock, exception → status translation.  A JNI layer maps 1:1 onto these.
#include "runtime.hpp"
#include "sharded.hpp"
#include <cmath>

#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace fm { void mersenne_increments(int32_t, int, int, int64_t, const double*, double*); double inverse_normal_cdf(double); }

using fm::Engine;
using fm::Error;

static thread_local std::string g_last_error;
namespace fm { void set_last_error(const std::string& message) { g_last_error = message; } }
namespace front = fm::front;
#define FRONT(call) do { if (fm::front_active()) return front::call; } while (0)

template <typename F>
static int guarded(F&& f) {
    try {
        std::lock_guard<std::recursive_mutex> lock(Engine::get().mu);
        f();
        return FMHIP_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "host allocation failed";
        return FMHIP_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return FMHIP_ERR_HIP;
    }
}

static void need(const void* p, const char* what) {
    if (!p) throw Error(FMHIP_ERR_INVALID_ARGUMENT, std::string("null pointer: ") + what);
}

extern "C" {

int fmhip_init(int device_index) {
    if (fm::front_active()) { g_last_error = "a device list is active (fmhip_init_devices): fmhip_shutdown first"; return FMHIP_ERR_INVALID_ARGUMENT; }
    return guarded([&] { Engine::get().init(device_index); });
}
// One process, several devices (sharded.hpp): `count` device indices (an index may repeat: shards on separate streams of one device).
// One entry = fmhip_init(devices[0]).
int fmhip_init_devices(const int* devices, int count) {
    // one entry: fmhip_init(devices[0]) — unless FMHIP_WORKER_THREAD=1 asks for the front with ONE shard: the engine's bookkeeping (recording,
    // planning, launching) then runs on a worker thread beside the caller's own (a caller that spends as long in its model classes as the
    // engine spends recording: the two overlap); values and statuses are the same
    static const bool worker_thread = [] { const char* e = std::getenv("FMHIP_WORKER_THREAD"); return e && e[0] == '1'; }();
    if (count == 1 && devices && !worker_thread) return fmhip_init(devices[0]);
    if (fm::front_active()) { g_last_error = "a device list is active already"; return FMHIP_ERR_INVALID_ARGUMENT; }
    return front::init_devices(devices, count);
}
int fmhip_device_count(int* count) {
    FRONT(device_count(count));
    return guarded([&] { need(count, "count"); Engine::get().require_init(); *count = 1; });
}
int fmhip_shutdown(void) {
    FRONT(shutdown());
    // a thread that waits for its moments outside the lock still holds a result slot, a pinned block or an event of this engine: the
    // teardown starts when the last such wait is over (they end by themselves: the device finishes what was launched)
    for (;;) {
        {
            std::lock_guard<std::recursive_mutex> lock(Engine::get().mu);
            if (Engine::get().waits_in_flight.load(std::memory_order_acquire) == 0) return guarded([&] { Engine::get().shutdown(); });
        }
        std::this_thread::yield();
    }
}
int fmhip_is_initialized(void) { return (fm::front_active() || Engine::get().initialized()) ? 1 : 0; }
int fmhip_abi_version(void) { return FMHIP_ABI_VERSION; }
const char* fmhip_last_error(void) { return g_last_error.c_str(); }

int fmhip_device_info(char* name_buf, int name_buf_len, int* n_compute_units, int64_t* hbm_bytes) {
    FRONT(device_info(name_buf, name_buf_len, n_compute_units, hbm_bytes));
    return guarded([&] { Engine::get().device_info(name_buf, name_buf_len, n_compute_units, hbm_bytes); });
}
int fmhip_synchronize(void) { FRONT(synchronize()); return guarded([&] { Engine::get().synchronize(); }); }
int fmhip_get_stream(void** stream_out) {
    FRONT(unsupported("fmhip_get_stream"));
    return guarded([&] { need(stream_out, "stream_out"); Engine::get().require_init(); *stream_out = (void*)Engine::get().stream(); });
}

int fmhip_vec_create_from_double(const double* host_values, int64_t n, fmhip_vec* out) {
    FRONT(vec_create_from_host(host_values, true, n, out));
    return guarded([&] { need(out, "out"); *out = Engine::get().create_from_host(host_values, true, n); });
}
int fmhip_vec_create_from_float(const float* host_values, int64_t n, fmhip_vec* out) {
    FRONT(vec_create_from_host(host_values, false, n, out));
    return guarded([&] { need(out, "out"); *out = Engine::get().create_from_host(host_values, false, n); });
}
int fmhip_vec_create_filled(int64_t n, double value, fmhip_vec* out) {
    FRONT(vec_create_filled(n, value, true, out));
    return guarded([&] { need(out, "out"); *out = Engine::get().create_filled(n, (float)value); });
}
int fmhip_vec_create_uninitialized(int64_t n, fmhip_vec* out) {
    FRONT(vec_create_filled(n, 0.0, false, out));
    return guarded([&] { need(out, "out"); *out = Engine::get().create_uninitialized(n); });
}
int fmhip_vec_retain(fmhip_vec v) { FRONT(vec_retain(v)); return guarded([&] { Engine::get().retain(v); }); }
int fmhip_vec_release(fmhip_vec v) { FRONT(vec_release(v)); return guarded([&] { Engine::get().release(v); }); }
int fmhip_vec_size(fmhip_vec v, int64_t* n_out) {
    FRONT(vec_size(v, n_out));
    return guarded([&] { need(n_out, "n_out"); Engine::get().require_init(); *n_out = Engine::get().node(v)->n; });
}
int fmhip_vec_read_double(fmhip_vec v, double* host_out, int64_t n) {
    FRONT(vec_read(v, host_out, true, n));
    return guarded([&] { Engine::get().read(v, host_out, true, n); });
}
int fmhip_vec_read_float(fmhip_vec v, float* host_out, int64_t n) {
    FRONT(vec_read(v, host_out, false, n));
    return guarded([&] { Engine::get().read(v, host_out, false, n); });
}
int fmhip_vec_device_ptr(fmhip_vec v, void** device_ptr_out) {
    FRONT(unsupported("fmhip_vec_device_ptr"));
    return guarded([&] { need(device_ptr_out, "device_ptr_out"); *device_ptr_out = Engine::get().device_ptr(v); });
}

int fmhip_call_v1s0(int opcode, fmhip_vec a, fmhip_vec* out) {
    if (fm::front_active()) { const fmhip_vec in[1] = { a }; return front::call(opcode, 1, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[1] = { a }; *out = Engine::get().call(opcode, 1, in, 0.0, false); });
}
int fmhip_call_v1s1(int opcode, fmhip_vec a, double s, fmhip_vec* out) {
    if (fm::front_active()) { const fmhip_vec in[1] = { a }; return front::call(opcode, 1, in, s, true, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[1] = { a }; *out = Engine::get().call(opcode, 1, in, s, true); });
}
int fmhip_call_v2s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec* out) {
    if (fm::front_active()) { const fmhip_vec in[2] = { a, b }; return front::call(opcode, 2, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[2] = { a, b }; *out = Engine::get().call(opcode, 2, in, 0.0, false); });
}
int fmhip_call_v2s1(int opcode, fmhip_vec a, fmhip_vec b, double s, fmhip_vec* out) {
    if (fm::front_active()) { const fmhip_vec in[2] = { a, b }; return front::call(opcode, 2, in, s, true, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[2] = { a, b }; *out = Engine::get().call(opcode, 2, in, s, true); });
}
int fmhip_call_v3s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec c, fmhip_vec* out) {
    if (fm::front_active()) { const fmhip_vec in[3] = { a, b, c }; return front::call(opcode, 3, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[3] = { a, b, c }; *out = Engine::get().call(opcode, 3, in, 0.0, false); });
}

int fmhip_set_fusion(int enabled, int* previous) {
    FRONT(set_int(0, enabled, previous));
    return guarded([&] {
        Engine& e = Engine::get();
        e.require_init();
        if (previous) *previous = e.fusion ? 1 : 0;
        if (e.fusion && !enabled) e.flush_all();
        e.fusion = enabled != 0;
    });
}
int fmhip_fusion_hold(int hold, int* previous) {
    FRONT(set_int(1, hold, previous));
    return guarded([&] {
        Engine& e = Engine::get();
        e.require_init();
        if (previous) *previous = e.fusion_hold;
        e.fusion_hold = hold == 2 ? 2 : (hold != 0 ? 1 : 0);
    });
}
int fmhip_set_step_grouping(int steps, int* previous) {
    FRONT(set_int(2, steps, previous));
    return guarded([&] {
        Engine& e = Engine::get();
        e.require_init();
        if (steps < 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "negative number of time steps");
        if (previous) *previous = e.group_steps;
        e.group_steps = steps;
    });
}
int fmhip_graph_clone(const fmhip_vec* roots, int n_roots, int n_copies, const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map,
                      const double* scalars, int n_scalars, fmhip_vec* out) {
    FRONT(graph_clone(roots, n_roots, n_copies, leaf_from, leaf_to, n_map, scalars, n_scalars, out));
    return guarded([&] { Engine::get().graph_clone(roots, n_roots, n_copies, leaf_from, leaf_to, n_map, scalars, n_scalars, out); });
}
int fmhip_graph_scalars(const fmhip_vec* roots, int n_roots, double* scalars_out, int capacity, int* n_scalars) {
    FRONT(graph_scalars(roots, n_roots, scalars_out, capacity, n_scalars));
    return guarded([&] {
        if (!n_scalars) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null count pointer");
        *n_scalars = Engine::get().graph_scalars(roots, n_roots, scalars_out, capacity);
    });
}
int fmhip_set_math_mode(int mode, int* previous) {
    FRONT(set_int(3, mode, previous));
    return guarded([&] {
        Engine& e = Engine::get();
        e.require_init();
        if (mode != FMHIP_MATH_EXACT && mode != FMHIP_MATH_FAST) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown math mode");
        if (previous) *previous = e.math_mode;
        if (e.math_mode != mode) e.flush_all();         // pending nodes were recorded under the old mode
        e.math_mode = mode;
    });
}
int fmhip_flush(void) { FRONT(flush()); return guarded([&] { Engine::get().flush_all(); Engine::get().end_step_group(); }); }

// ---- expectation communicator (include/fmhip.h): the global moments of path-sharded vectors
static void combine_moments(const fmhip_moments* gathered, int world, int count, fmhip_moments* out) {
    for (int k = 0; k < count; ++k) {
        fmhip_moments m = gathered[k];                                   // rank 0, then the others in rank order
        for (int r = 1; r < world; ++r) {
            const fmhip_moments& g = gathered[(size_t)r * count + k];
            m.sum += g.sum; m.sumsq += g.sumsq;
            // java.lang.Math.min / max: NaN-propagating, -0.0 < +0.0 (as the device's reduction, fm_device_math.hpp)
            m.min = (m.min != m.min || g.min != g.min) ? std::nan("") : (g.min < m.min || (g.min == m.min && std::signbit(g.min))) ? g.min : m.min;
            m.max = (m.max != m.max || g.max != g.max) ? std::nan("") : (g.max > m.max || (g.max == m.max && !std::signbit(g.max))) ? g.max : m.max;
        }
        out[k] = m;
    }
}
static void exchange_moments(Engine& e, fmhip_moments* inout, int count) {
    if (e.comm_world <= 1 || !e.comm_gather) return;
    std::vector<fmhip_moments> all((size_t)e.comm_world * count);
    const int st = e.comm_gather(e.comm_context, reinterpret_cast<const double*>(inout), count * 4, reinterpret_cast<double*>(all.data()));
    if (st != 0) throw Error(FMHIP_ERR_HIP, "the expectation communicator's gather failed with status " + std::to_string(st));
    combine_moments(all.data(), e.comm_world, count, inout);
}
int fmhip_set_expectation_comm(int world, int rank, fmhip_gather_fn gather, void* context) {
    if (fm::front_active() && world > 1) return front::unsupported("fmhip_set_expectation_comm");
    return guarded([&] {
        Engine& e = Engine::get();
        if (world < 1 || rank < 0 || rank >= world) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad communicator: world " + std::to_string(world) + ", rank " + std::to_string(rank));
        if (world > 1 && !gather) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a communicator of more than one rank needs a gather function");
        e.comm_world = gather ? world : 1; e.comm_rank = gather ? rank : 0; e.comm_gather = gather; e.comm_context = context;
    });
}
int fmhip_expectation_world(int* world, int* rank) {
    return guarded([&] { Engine& e = Engine::get(); if (world) *world = e.comm_world; if (rank) *rank = e.comm_rank; });
}
int fmhip_expectation_combine(const fmhip_moments* gathered, int world, int count, fmhip_moments* out) {
    return guarded([&] {
        need(gathered, "gathered"); need(out, "out");
        if (world < 1 || count < 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad world or count");
        combine_moments(gathered, world, count, out);
    });
}

// The one call of a caller that values product after product: the engine lock is held for the bookkeeping (graph → launch → commit),
// NOT while the device computes — other threads record and launch meanwhile.  The moments arrive in a slot of pinned memory of their
// own; this thread polls its flag, then takes the lock again to copy them out and give the launch's buffers back.
int fmhip_reduce_moments(fmhip_vec v, double shift, fmhip_moments* out) {
    if (fm::front_active()) { const fmhip_vec one[1] = { v }; const double sh[1] = { shift }; return front::reduce_moments_batch(one, 1, sh, out); }
    Engine::RedLaunch pending;
    int rc = guarded([&] { need(out, "out"); Engine::get().reduce(v, shift, out, nullptr, &pending); if (pending.pending) Engine::get().waits_in_flight.fetch_add(1, std::memory_order_acq_rel); });
    if (rc == FMHIP_OK && pending.pending) {
        const bool arrived = Engine::red_poll(pending);
        rc = guarded([&] { Engine& e = Engine::get(); struct Done { Engine& e; ~Done() { e.waits_in_flight.fetch_sub(1, std::memory_order_acq_rel); } } done{ e }; e.red_complete(pending, arrived); });
    }
    if (rc == FMHIP_OK) rc = guarded([&] { exchange_moments(Engine::get(), out, 1); });
    return rc;
}
int fmhip_reduce_moments_batch(const fmhip_vec* vectors, int count, const double* shifts, fmhip_moments* out) {
    FRONT(reduce_moments_batch(vectors, count, shifts, out));
    return guarded([&] { need(vectors, "vectors"); need(out, "out"); Engine& e = Engine::get(); e.reduce_batch(vectors, count, shifts, out, nullptr); exchange_moments(e, out, count); });
}
int fmhip_reduce_moments_batch_device(const fmhip_vec* vectors, int count, const double* shifts, void* device_out) {
    FRONT(unsupported("fmhip_reduce_moments_batch_device"));
    return guarded([&] { need(vectors, "vectors"); need(device_out, "device_out"); Engine::get().reduce_batch_device(vectors, count, shifts, device_out); });
}
int fmhip_reduce_moments_batch_begin(const fmhip_vec* vectors, int count, const double* shifts, fmhip_ticket* ticket_out) {
    FRONT(reduce_moments_batch_begin(vectors, count, shifts, ticket_out));
    return guarded([&] { need(vectors, "vectors"); need(ticket_out, "ticket_out"); *ticket_out = Engine::get().reduce_batch_begin(vectors, count, shifts); });
}
int fmhip_vec_give_up_values(const fmhip_vec* vectors, int count) {
    FRONT(vec_give_up_values(vectors, count));
    return guarded([&] { need(vectors, "vectors"); Engine::get().give_up_values(vectors, count); });
}
int fmhip_reduce_moments_batch_end(fmhip_ticket ticket, fmhip_moments* out, int count) {
    FRONT(reduce_moments_batch_end(ticket, out, count));
    Engine::MomentsTicket t;
    int rc = guarded([&] {
        need(out, "out");
        Engine& e = Engine::get();
        e.require_init();
        t = e.ticket_take(ticket);
        if (t.count != count) { const int have = t.count; e.ticket_retire(t); throw Error(FMHIP_ERR_SIZE_MISMATCH, "the ticket holds " + std::to_string(have) + " expectations, the caller asks for " + std::to_string(count)); }
        e.waits_in_flight.fetch_add(1, std::memory_order_acq_rel);       // this thread holds the ticket's block and event outside the lock from here on
    });
    if (rc != FMHIP_OK) return rc;
    struct Done { ~Done() { Engine::get().waits_in_flight.fetch_sub(1, std::memory_order_acq_rel); } } done;
    // (a ticket whose moments came with the launches that computed the vectors has them already: ticket_take waited for their slots)
    const hipError_t waited = t.event && t.ready.empty() ? hipEventSynchronize(t.event) : hipSuccess;      // without the engine lock: other threads (and this one's next parameter set) are not held up
    if (waited == hipSuccess) std::memcpy(out, t.ready.empty() ? t.host : (const void*)t.ready.data(), (size_t)count * sizeof(fmhip_moments));
    rc = guarded([&] {
        Engine& e = Engine::get();
        e.ticket_retire(t);
        if (waited != hipSuccess) throw Error(FMHIP_ERR_HIP, std::string("waiting for the expectations failed: ") + hipGetErrorString(waited));
        exchange_moments(e, out, count);
    });
    return rc;
}
int fmhip_reduce_moments_device(fmhip_vec v, double shift, void* device_out_4_doubles) {
    FRONT(unsupported("fmhip_reduce_moments_device"));
    return guarded([&] { need(device_out_4_doubles, "device_out"); Engine::get().reduce(v, shift, nullptr, device_out_4_doubles); });
}

int fmhip_program_create(const fmhip_prog_op* ops, int n_ops, int n_inputs, const int32_t* out_values, int n_outputs,
                         const int32_t* reduce_values, int n_reduce, fmhip_program* out) {
    FRONT(program_create(ops, n_ops, n_inputs, out_values, n_outputs, reduce_values, n_reduce, out));
    return guarded([&] {
        need(out, "out");
        *out = Engine::get().program_create(ops, n_ops, n_inputs, out_values, n_outputs, reduce_values, n_reduce);
    });
}
int fmhip_program_shape(fmhip_program p, int* n_inputs, int* n_outputs, int* n_reduce) {
    FRONT(program_shape(p, n_inputs, n_outputs, n_reduce));
    return guarded([&] {
        Engine& e = Engine::get();
        e.require_init();
        const fm::Program* pr = e.program(p);
        if (n_inputs) *n_inputs = pr->n_in;
        if (n_outputs) *n_outputs = pr->n_out;
        if (n_reduce) *n_reduce = pr->n_red;
    });
}
int fmhip_set_jit(int mode, int* previous) {
    FRONT(set_int(4, mode, previous));
    return guarded([&] {
        if (mode != FMHIP_JIT_OFF && mode != FMHIP_JIT_AUTO && mode != FMHIP_JIT_SYNC) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown JIT mode");
        if (previous) *previous = Engine::get().jit_mode;
        Engine::get().jit_mode = mode;
    });
}
int fmhip_jit_wait(void) {
    FRONT(jit_wait());
    // not under the engine mutex: other threads keep launching while this one waits for the compiler thread
    try { Engine::get().jit_wait(); return FMHIP_OK; } catch (...) { return FMHIP_ERR_HIP; }
}
int fmhip_jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, double* compile_seconds, int64_t* disk_cache_hits) {
    FRONT(jit_stats(compiled, failed, pending, compile_seconds, disk_cache_hits));
    return guarded([&] {
        const fm::JitStats s = Engine::get().jit_stats();
        if (compiled) *compiled = s.compiled;
        if (failed) *failed = s.failed;
        if (pending) *pending = s.pending;
        if (compile_seconds) *compile_seconds = s.seconds;
        if (disk_cache_hits) *disk_cache_hits = s.disk_hits;
    });
}
int fmhip_program_tier(fmhip_program p, int* tier, int* vgprs) {
    FRONT(program_tier(p, tier, vgprs));
    return guarded([&] {
        Engine::get().require_init();
        fm::Program* pr = Engine::get().program(p);
        const bool ready = Engine::get().jit_mode != FMHIP_JIT_OFF && pr->jit && pr->jit->state.load() == fm::JitSlot::READY;
        if (tier) *tier = ready ? 1 : 0;
        if (vgprs) *vgprs = ready ? pr->jit->vgprs : 0;
    });
}
int fmhip_program_source(const fmhip_prog_op* ops, int n_ops, int n_inputs, const int32_t* out_values, int n_outputs,
                         const int32_t* reduce_values, int n_reduce, char* buffer, int64_t capacity, int64_t* needed) {
    return guarded([&] {
        const std::string src = Engine::get().program_source(ops, n_ops, n_inputs, out_values, n_outputs, reduce_values, n_reduce);
        if (needed) *needed = (int64_t)src.size();
        if (buffer && capacity > 0) {
            const size_t k = std::min<size_t>(src.size(), (size_t)capacity - 1);
            std::memcpy(buffer, src.data(), k);
            buffer[k] = 0;
        }
    });
}
int fmhip_program_release(fmhip_program p) { FRONT(program_release(p)); return guarded([&] { Engine::get().program_release(p); }); }
int fmhip_program_launch_count(fmhip_program p, int* n_launches) {
    if (fm::front_active()) { int a = 0; const int st = front::program_shape(p, &a, nullptr, nullptr); if (st == FMHIP_OK && n_launches) *n_launches = 1; return st; }
    return guarded([&] {
        need(n_launches, "n_launches");
        Engine::get().require_init();
        (void)Engine::get().program(p);          // validates the handle
        *n_launches = 1;                         // fused reductions are finished inside the same launch
    });
}
int fmhip_program_run(fmhip_program p, int batch, const fmhip_vec* inputs, fmhip_vec* outputs,
                      const double* reduce_shift, fmhip_moments* moments, void* device_moments) {
    if (fm::front_active()) return device_moments ? front::unsupported("fmhip_program_run with device_moments") : front::program_run(p, batch, inputs, outputs, false, reduce_shift, moments);
    return guarded([&] { Engine::get().program_run(p, batch, inputs, outputs, false, reduce_shift, moments, device_moments); });
}
int fmhip_program_run_into(fmhip_program p, int batch, const fmhip_vec* inputs, const fmhip_vec* outputs,
                           const double* reduce_shift, fmhip_moments* moments, void* device_moments) {
    if (fm::front_active()) return device_moments ? front::unsupported("fmhip_program_run_into with device_moments") : front::program_run(p, batch, inputs, const_cast<fmhip_vec*>(outputs), true, reduce_shift, moments);
    return guarded([&] {
        Engine::get().program_run(p, batch, inputs, const_cast<fmhip_vec*>(outputs), true, reduce_shift, moments, device_moments);
    });
}

int fmhip_bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset,
                      const double* dt, fmhip_vec* out) {
    FRONT(bm_generate(seed, n_steps, n_factors, n_paths, path_offset, dt, out));
    return guarded([&] { Engine::get().bm_generate(seed, n_steps, n_factors, n_paths, path_offset, dt, out); });
}

int fmhip_mersenne_increments(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, double* host_out) {
    try {
        if (n_steps <= 0 || n_factors <= 0 || n_paths < 0 || !dt || (!host_out && n_paths > 0)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad Brownian motion description");
        fm::mersenne_increments(seed, n_steps, n_factors, n_paths, dt, host_out);
        return FMHIP_OK;
    } catch (const Error& e) { g_last_error = e.what(); return e.code; }
}
int fmhip_bm_generate_mersenne(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, fmhip_vec* out) {
    if (fm::front_active()) {           // generated on the host, every increment uploaded block by block through the front
        if (!out || !dt || n_steps <= 0 || n_factors <= 0 || n_paths < 0) { g_last_error = "bad Brownian motion description"; return FMHIP_ERR_INVALID_ARGUMENT; }
        std::vector<double> host((size_t)n_steps * n_factors * (size_t)n_paths);
        fm::mersenne_increments(seed, n_steps, n_factors, n_paths, dt, host.data());
        const size_t count = (size_t)n_steps * n_factors;
        for (size_t k = 0; k < count; ++k) out[k] = 0;
        for (size_t k = 0; k < count; ++k) {
            const int st = front::vec_create_from_host(host.data() + k * (size_t)n_paths, true, n_paths, &out[k]);
            if (st != FMHIP_OK) { for (size_t j = 0; j < k; ++j) { (void)front::vec_release(out[j]); out[j] = 0; } return st; }
        }
        return FMHIP_OK;
    }
    return guarded([&] {
        need(out, "out"); need(dt, "dt");
        if (n_steps <= 0 || n_factors <= 0 || n_paths < 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad Brownian motion description");
        std::vector<double> host((size_t)n_steps * n_factors * (size_t)n_paths);
        fm::mersenne_increments(seed, n_steps, n_factors, n_paths, dt, host.data());
        const size_t count = (size_t)n_steps * n_factors;
        for (size_t k = 0; k < count; ++k) out[k] = 0;
        try { for (size_t k = 0; k < count; ++k) out[k] = Engine::get().create_from_host(host.data() + k * (size_t)n_paths, true, n_paths); }
        catch (...) { for (size_t k = 0; k < count; ++k) if (out[k]) { Engine::get().release(out[k]); out[k] = 0; } throw; }
    });
}
double fmhip_inverse_normal_cdf(double p) { return fm::inverse_normal_cdf(p); }

int fmhip_pool_clean(void) { FRONT(pool(0)); return guarded([&] { Engine::get().pool_clean(); }); }
int fmhip_pool_purge(void) { FRONT(pool(1)); return guarded([&] { Engine::get().pool_purge(); }); }
int fmhip_pool_stats(fmhip_pool_stats_t* out) { FRONT(pool_stats(out)); return guarded([&] { Engine::get().pool_stats(out); }); }

int fmhip_traffic_stats(int64_t* algorithmic_bytes, int64_t* specialised_launches) {
    FRONT(traffic_stats(algorithmic_bytes, specialised_launches));
    return guarded([&] {
        if (algorithmic_bytes) *algorithmic_bytes = Engine::get().algorithmic_bytes();
        if (specialised_launches) *specialised_launches = Engine::get().jit_launches();
    });
}
int fmhip_profile_enable(int enabled) { FRONT(profile_enable(enabled)); return guarded([&] { Engine::get().profile_enable(enabled != 0); }); }
int fmhip_profile_read(double* kernel_ms_total, int64_t* n_launches) {
    FRONT(profile_read(kernel_ms_total, n_launches));
    return guarded([&] { Engine::get().profile_read(kernel_ms_total, n_launches); });
}

} // extern "C"
