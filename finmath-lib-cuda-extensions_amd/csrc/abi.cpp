// abi.cpp — the extern "C" surface declared in include/fmhip.h: argument checks, the engine lock, exception → status translation —
// and the one piece of logic that lives at this level because it is ABOUT the boundary: thread engines (namespace te), i.e. which
// engine a call of which thread on which handle runs on, and how a vector of one engine enters another (import / export, events in both
// directions).  A JNI layer maps 1:1 onto the entry points.
#include "runtime.hpp"
#include "sharded.hpp"
#include <cmath>

#include <chrono>
#include <cstdlib>
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

// ---------------------------------------------------------------- thread engines (fmhip_set_thread_engines)
// An engine per CALLER THREAD on one device: its own stream, pool, pending graph, time-step grouping, lock.  Threads that simulate side by
// side (finmath-lib's optimiser evaluates the columns of a Jacobian on a thread pool, LIBORMarketModelCalibrationATMTest.java:319) record
// without meeting each other, and their launches meet on the device.  A handle carries its engine's number (Engine::OWNER_SHIFT).  What a
// thread does with ANOTHER thread's vector: as an operand, the vector enters this thread's engine as a leaf that aliases the owner's
// storage (the owner computes it if it is still pending, keeps a reference for the import, and the two streams are ordered by events in
// both directions); everything else — release, retain, read, its moments, a program or a ticket of another thread — runs on the owner's
// engine, under the owner's lock, on the calling thread.  No two engine locks are ever held together.
namespace te {
static std::atomic<bool> on{ false };
static std::mutex registry_mu;
static constexpr int MAX_ENGINES = 64;
static std::atomic<Engine*> engines[MAX_ENGINES];
static std::atomic<int> count{ 0 };
static std::vector<Engine*>& retired = *new std::vector<Engine*>();      // engines of earlier generations (threads may still be bound to them: kept, shut down, never destroyed)
static thread_local int inside = 0;                          // > 0: a nested call of an entry point on behalf of this layer (no second look at the handles)
static bool active() { return on.load(std::memory_order_acquire) && inside == 0; }
static bool foreign(int64_t h) { return h > 0 && Engine::owner_of(h) != Engine::get().index(); }
static Engine* owner(int64_t h) {
    const int i = Engine::owner_of(h);
    Engine* e = (i >= 0 && i < count.load(std::memory_order_acquire)) ? engines[i].load(std::memory_order_acquire) : nullptr;
    if (!e) throw Error(FMHIP_ERR_INVALID_HANDLE, "handle " + std::to_string(h) + " names an engine that does not exist");
    return e;
}
struct Inside { Inside() { ++inside; } ~Inside() { --inside; } };
struct Rebind {                                              // the calling thread acts on engine e until the end of the scope
    Engine* previous; Inside nested;
    explicit Rebind(Engine* e) : previous(Engine::thread_is_bound() ? &Engine::get() : nullptr) { Engine::bind_thread(e); }
    ~Rebind() { Engine::bind_thread(previous); }
};
// An engine outlives the thread it served: when the thread ends, the engine (its pool, its code objects, the vectors it owns) waits for
// the next thread without one — a thread pool that replaces its threads does not leave an engine behind per thread it ever had.
static std::vector<int>& idle = *new std::vector<int>();     // (registry_mu)
struct Binding {
    Engine* engine = nullptr;
    ~Binding() {
        if (!engine || engine->retired.load(std::memory_order_acquire) || !on.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lock(registry_mu);
        if (engine->index() > 0 && engines[engine->index()].load(std::memory_order_acquire) == engine) idle.push_back(engine->index());
    }
};
static thread_local Binding binding;
// this thread's engine: the one it is bound to, an idle one, or a new one with the first engine's settings
static void ensure() {
    if (Engine::thread_is_bound()) return;                    // (an engine retired by fmhip_shutdown no longer counts: Engine::get)
    std::unique_lock<std::mutex> lock(registry_mu);
    if (!idle.empty()) {
        Engine* e = engines[idle.back()].load(std::memory_order_acquire);
        idle.pop_back();
        if (e) { (void)hipSetDevice(e->device_index()); Engine::bind_thread(e); binding.engine = e; return; }      // (the current device is a property of the thread)
    }
    const int i = count.load(std::memory_order_acquire);
    if (i >= MAX_ENGINES) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "more caller threads than thread engines (" + std::to_string(MAX_ENGINES) + ")");
    Engine* first = engines[0].load(std::memory_order_acquire);
    const int device = first->device_index();                 // (fixed since fmhip_init)
    Engine* e = Engine::create();
    e->set_index(i);
    // Its lock is taken BEFORE it can be found: whoever finds it in the registry — a setter or fmhip_synchronize walking all engines —
    // waits until it is initialised (until round 5 the engine was published first, and such a caller could lock it uninitialised).
    std::unique_lock<fm::EngineMutex> l(e->mu);
    engines[i].store(e, std::memory_order_release);
    count.store(i + 1, std::memory_order_release);
    lock.unlock();                                            // the other threads' engines start side by side (an engine's start is ≈ 60 ms of allocations)
    e->share_jit_of(*first);
    e->init(device);
    // The settings are the first engine's as they are NOW, after this engine has become visible: a setter that ran before has changed the
    // first engine already, one that comes later finds this engine, waits for its lock and sets it itself — none is missed (copied before
    // the registration, a concurrent fmhip_set_math_mode left one thread on EXACT and the others on FAST).  Lock order e → first: nobody
    // holds the first engine's lock while waiting for another engine's.
    int fusion = 0, math_mode = 0, group_steps = 0, jit_mode = 0; bool profiling = false;
    { std::lock_guard<fm::EngineMutex> l0(first->mu); profiling = first->profiling(); fusion = first->fusion ? 1 : 0; math_mode = first->math_mode; group_steps = first->group_steps; jit_mode = first->jit_mode; }
    e->fusion = fusion != 0; e->math_mode = math_mode; e->group_steps = group_steps; e->jit_mode = jit_mode;
    if (profiling) e->profile_enable(true);
    l.unlock();
    Engine::bind_thread(e);
    binding.engine = e;
}
// the references owners hold for imports that have died in e go back (e's lock is NOT held here)
static void drain(Engine& e) {
    std::vector<Engine::ForeignDone> done;
    { std::lock_guard<fm::EngineMutex> l(e.mu); if (!e.has_foreign_done()) return; done = e.take_foreign_done(); }
    for (const Engine::ForeignDone& d : done) {
        Engine* o = engines[d.owner].load(std::memory_order_acquire);
        if (o) { std::lock_guard<fm::EngineMutex> l(o->mu); try { if (o->initialized()) o->release_exported(d.handle, d.done); } catch (...) {} }
        if (d.done) (void)hipEventDestroy(d.done);
    }
}
// operands of a call that runs on THIS thread's engine: foreign ones become imports (one reference each, given back by the destructor)
struct Localized {
    std::vector<fmhip_vec> local, taken;
    Localized(const fmhip_vec* h, int n) : local(h, h + (n > 0 ? n : 0)) {
        try { import_all(); }
        catch (...) { give_back(); throw; }                   // (a constructor that throws has no destructor: the imports made so far go back here)
    }
    void give_back() {
        if (taken.empty()) return;
        Engine& mine = Engine::get();
        { std::lock_guard<fm::EngineMutex> l(mine.mu); for (fmhip_vec v : taken) { try { mine.release(v); } catch (...) {} } }
        taken.clear();
    }
    void import_all() {
        Engine& mine = Engine::get();
        for (fmhip_vec& v : local) {
            if (!(v > 0 && Engine::owner_of(v) != mine.index())) continue;
            fmhip_vec loc = 0;
            { std::lock_guard<fm::EngineMutex> l(mine.mu); loc = mine.find_import(v); }
            if (!loc) {
                Engine* o = owner(v);
                hipEvent_t ready = nullptr;
                if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) throw Error(FMHIP_ERR_HIP, "hipEventCreate(import)");
                try {
                    Engine::Exported x;
                    { std::lock_guard<fm::EngineMutex> l(o->mu); x = o->export_vector(v, ready); }
                    try { std::lock_guard<fm::EngineMutex> l(mine.mu); loc = mine.import_vector(o->index(), v, x, ready); }
                    catch (...) { std::lock_guard<fm::EngineMutex> l(o->mu); try { o->release(v); } catch (...) {} throw; }
                } catch (...) { (void)hipEventDestroy(ready); throw; }
                (void)hipEventDestroy(ready);                  // (destroyed when the waits on it have been served)
            }
            taken.push_back(loc);
            v = loc;
        }
    }
    ~Localized() { give_back(); }
    bool any() const { return !taken.empty(); }
};
}

// Has this thread ever ENTERED an engine for anything but a release?  A thread that has not — a garbage collector's cleaner thread — is a
// releasing thread: its releases are queued (release_vector).  Thread-local: the driving thread and the releasing thread share nothing here.
static thread_local bool tl_drives = false;

template <typename F>
static int guarded(F&& f) {
    try {
        const bool thread_engines = te::on.load(std::memory_order_acquire);
        if (thread_engines) te::ensure();
        Engine& e = Engine::get();
        {
            std::lock_guard<fm::EngineMutex> lock(e.mu);
            tl_drives = true;
            e.note_driver();
            if (e.late_count() >= Engine::late_eager()) e.drain_late();      // releases other threads have left (Engine::release_later): preferably performed while the device is waited for
            f();
        }
        if (thread_engines) te::drain(e);
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

namespace te {
static constexpr int NOT_MINE_TO_HANDLE = 1;                 // (no status is positive)
template <typename F> static int status_of(F&& f) {
    try { return f(); }
    catch (const Error& e) { g_last_error = e.what(); return e.code; }
    catch (const std::bad_alloc&) { g_last_error = "host allocation failed"; return FMHIP_ERR_OUT_OF_MEMORY; }
    catch (const std::exception& e) { g_last_error = e.what(); return FMHIP_ERR_HIP; }
}
// `call` on the engine that owns h, if that is not this thread's
template <typename F> static int owner_routed(int64_t h, F&& call) {
    // (a thread WITHOUT an engine — a collector thread that only releases, a reader — gets none for this: the handle names its owner)
    return status_of([&]() -> int {
        if (!Engine::thread_is_bound() && h > 0) { Rebind r(owner(h)); return call(); }
        ensure(); if (!foreign(h)) return NOT_MINE_TO_HANDLE; Rebind r(owner(h)); return call(); });
}
// `call(local handles)` on this thread's engine, foreign operands imported
template <typename F> static int with_local(const fmhip_vec* h, int n, F&& call) {
    return status_of([&]() -> int {
        ensure();
        bool any = false;
        for (int i = 0; h && i < n; ++i) any |= foreign(h[i]);
        if (!any) return NOT_MINE_TO_HANDLE;
        Localized L(h, n);
        int rc;
        { Inside nested; rc = call(L.local.data()); }
        return rc;
    });
}
// `call(engine is this thread's)` on every engine, this thread's first; the first failure is the status
template <typename F> static int on_all(F&& call) {
    return status_of([&]() -> int {
        ensure();
        Engine* mine = &Engine::get();
        int rc;
        { Inside nested; rc = call(true); }
        const int n = count.load(std::memory_order_acquire);
        for (int i = 0; i < n; ++i) {
            Engine* e = engines[i].load(std::memory_order_acquire);
            if (!e || e == mine) continue;
            Rebind r(e);
            const int st = call(false);
            if (rc == FMHIP_OK) rc = st;
        }
        return rc;
    });
}
}
#define TE_OWNER(h, call) do { if (te::active()) { const int te_rc = te::owner_routed((h), [&]() -> int { return call; }); if (te_rc != te::NOT_MINE_TO_HANDLE) return te_rc; } } while (0)
#define TE_LOCAL(arr, n, L, call) do { if (te::active()) { const int te_rc = te::with_local((arr), (n), [&](const fmhip_vec* L) -> int { return call; }); if (te_rc != te::NOT_MINE_TO_HANDLE) return te_rc; } } while (0)
#define TE_ALL(mine, call) do { if (te::active()) return te::on_all([&](bool mine) -> int { (void)mine; return call; }); } while (0)

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
    if (te::on.load(std::memory_order_acquire)) { g_last_error = "thread engines are active (fmhip_set_thread_engines): fmhip_shutdown first"; return FMHIP_ERR_INVALID_ARGUMENT; }
    return front::init_devices(devices, count);
}
int fmhip_device_count(int* count) {
    FRONT(device_count(count));
    return guarded([&] { need(count, "count"); Engine::get().require_init(); *count = 1; });
}
// An engine per caller thread (see the top of this file).  Switched on by the thread that initialised the library, which keeps the
// process-wide engine; every other thread gets an engine of its own at its first call.  Off again: fmhip_shutdown.
int fmhip_set_thread_engines(int enabled, int* previous) {
    if (fm::front_active()) return front::unsupported("fmhip_set_thread_engines");
    const bool was = te::on.load(std::memory_order_acquire);
    if (previous) *previous = was ? 1 : 0;
    if ((enabled != 0) == was) return FMHIP_OK;
    if (!enabled) { g_last_error = "thread engines end with fmhip_shutdown"; return FMHIP_ERR_INVALID_ARGUMENT; }
    return te::status_of([&]() -> int {
        std::lock_guard<std::mutex> lock(te::registry_mu);
        Engine& first = Engine::get();
        { std::lock_guard<fm::EngineMutex> l(first.mu); first.require_init(); }
        if (first.index() != 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "thread engines are switched on from the thread that initialised the library");
        te::engines[0].store(&first, std::memory_order_release);
        te::count.store(1, std::memory_order_release);
        Engine::bind_thread(&first);
        te::on.store(true, std::memory_order_release);
        return FMHIP_OK;
    });
}
static int shutdown_thread_engines() {
    // every engine but the first is shut down and destroyed; the caller's thread goes back to the process-wide engine, which the ordinary path below shuts down
    std::lock_guard<std::mutex> lock(te::registry_mu);
    te::on.store(false, std::memory_order_release);
    const int n = te::count.load(std::memory_order_acquire);
    int rc = FMHIP_OK;
    for (int i = 0; i < n; ++i) { Engine* e = te::engines[i].load(std::memory_order_acquire); if (e) te::drain(*e); }      // imports that died: their owners' references go back while the owners exist
    for (int i = n - 1; i >= 1; --i) {
        Engine* e = te::engines[i].load(std::memory_order_acquire);
        if (!e) continue;
        Engine::bind_thread(e);
        for (;;) {
            { std::lock_guard<fm::EngineMutex> l(e->mu); if (e->waits_in_flight.load(std::memory_order_acquire) == 0) { const int st = te::status_of([&]() -> int { if (e->initialized()) e->shutdown(); return FMHIP_OK; }); if (rc == FMHIP_OK) rc = st; break; } }
            std::this_thread::yield();
        }
        te::engines[i].store(nullptr, std::memory_order_release);
        e->retired.store(true, std::memory_order_release);       // the object stays: a thread still bound to it is unbound at its next call (Engine::get)
        te::retired.push_back(e);
    }
    te::idle.clear();
    te::count.store(0, std::memory_order_release);
    te::engines[0].store(nullptr, std::memory_order_release);
    Engine::bind_thread(nullptr);
    return rc;
}
int fmhip_shutdown(void) {
    FRONT(shutdown());
    if (te::on.load(std::memory_order_acquire)) { const int rc = shutdown_thread_engines(); if (rc != FMHIP_OK) return rc; }
    // a thread that waits for its moments outside the lock still holds a result slot, a pinned block or an event of this engine: the
    // teardown starts when the last such wait is over (they end by themselves: the device finishes what was launched)
    for (;;) {
        {
            std::lock_guard<fm::EngineMutex> lock(Engine::get().mu);
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
int fmhip_synchronize(void) { FRONT(synchronize()); TE_ALL(mine, fmhip_synchronize()); return guarded([&] { Engine::get().synchronize(); }); }
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
int fmhip_vec_retain(fmhip_vec v) { FRONT(vec_retain(v)); TE_OWNER(v, fmhip_vec_retain(v)); return guarded([&] { Engine::get().retain(v); }); }
// A release by a thread that drives the engine is performed at once; one by a thread that only ever releases — a collector's cleaner, which hands back
// hundreds of thousands of handles in a burst — is queued and performed by whoever enters the engine next (runtime.hpp: release_later).
// The releasing thread collects them 256 at a time before it touches the engine's queue: one by one, its pushes and the driving thread's
// look at the queue met on the queue's lock for as long as the burst lasted (tens of milliseconds per burst; lmm_hip --finmath-like
// --release-lag 100 between 6 and 11 s from run to run).  What a thread has collected goes to the engine when 256 wait, at its next
// release after a millisecond, and when the thread ends.  Queued releases cannot report an invalid handle.
namespace {
struct LateBuffer {
    Engine* engine = nullptr;
    std::vector<fmhip_vec> handles;
    std::chrono::steady_clock::time_point since{};
    uint64_t seq_seen = ~uint64_t(0);
    std::chrono::steady_clock::time_point seq_since{};
    void flush() {
        if (handles.empty() || !engine) { handles.clear(); return; }
        Engine& e = *engine;
        e.release_later(handles.data(), handles.size());
        handles.clear();
        // nobody has entered the engine for 10 ms (this thread has looked twice and found the same count of entries): the caller has gone
        // quiet, nobody will perform the queue — this thread does
        const uint64_t seq = e.driver_seq();
        const auto now = std::chrono::steady_clock::now();
        if (seq != seq_seen) { seq_seen = seq; seq_since = now; return; }
        if (now - seq_since < std::chrono::milliseconds(10)) return;
        std::lock_guard<fm::EngineMutex> lock(e.mu);
        e.drain_late();
        seq_since = now;
    }
    ~LateBuffer() { try { flush(); } catch (...) {} }
};
}
static int release_vector(fmhip_vec v) {
    try {
        Engine& e = Engine::get();
        static const bool LATE = [] { const char* s = std::getenv("FMHIP_LATE_RELEASES"); return !(s && s[0] == '0'); }();      // =0: every release takes the engine lock (rounds 1–4; A/B measurement)
        if (LATE && !tl_drives) {
            static thread_local LateBuffer mine;
            if (mine.engine != &e) { mine.flush(); mine.engine = &e; }
            if (mine.handles.empty()) mine.since = std::chrono::steady_clock::now();
            mine.handles.push_back(v);
            if (mine.handles.size() >= 256 || ((mine.handles.size() & 15u) == 0 && std::chrono::steady_clock::now() - mine.since > std::chrono::milliseconds(1))) mine.flush();
            return FMHIP_OK;
        }
        std::lock_guard<fm::EngineMutex> lock(e.mu);
        if (e.has_late()) e.drain_late();
        e.release(v);
        return FMHIP_OK;
    } catch (const Error& e) { g_last_error = e.what(); return e.code; }
    catch (const std::exception& e) { g_last_error = e.what(); return FMHIP_ERR_HIP; }
}
int fmhip_vec_release(fmhip_vec v) {
    FRONT(vec_release(v));
    if (te::active()) {
        // (thread engines: the handle names its owner — a thread without an engine of its own, a cleaner, gets none for this)
        const int rc = te::status_of([&]() -> int { if (v <= 0) return te::NOT_MINE_TO_HANDLE; te::Rebind r(te::owner(v)); return release_vector(v); });
        if (rc != te::NOT_MINE_TO_HANDLE) return rc;
    }
    return release_vector(v);
}
int fmhip_vec_size(fmhip_vec v, int64_t* n_out) {
    FRONT(vec_size(v, n_out));
    TE_OWNER(v, fmhip_vec_size(v, n_out));
    return guarded([&] { need(n_out, "n_out"); Engine::get().require_init(); *n_out = Engine::get().node(v)->n; });
}
int fmhip_vec_read_double(fmhip_vec v, double* host_out, int64_t n) {
    FRONT(vec_read(v, host_out, true, n));
    TE_OWNER(v, fmhip_vec_read_double(v, host_out, n));
    return guarded([&] { Engine::get().read(v, host_out, true, n); });
}
int fmhip_vec_read_float(fmhip_vec v, float* host_out, int64_t n) {
    FRONT(vec_read(v, host_out, false, n));
    TE_OWNER(v, fmhip_vec_read_float(v, host_out, n));
    return guarded([&] { Engine::get().read(v, host_out, false, n); });
}
int fmhip_vec_device_ptr(fmhip_vec v, void** device_ptr_out) {
    FRONT(unsupported("fmhip_vec_device_ptr"));
    TE_OWNER(v, fmhip_vec_device_ptr(v, device_ptr_out));
    return guarded([&] { need(device_ptr_out, "device_ptr_out"); *device_ptr_out = Engine::get().device_ptr(v); });
}

int fmhip_call_v1s0(int opcode, fmhip_vec a, fmhip_vec* out) {
    { const fmhip_vec te_in[1] = { a }; TE_LOCAL(te_in, 1, L, fmhip_call_v1s0(opcode, L[0], out)); }
    if (fm::front_active()) { const fmhip_vec in[1] = { a }; return front::call(opcode, 1, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[1] = { a }; *out = Engine::get().call(opcode, 1, in, 0.0, false); });
}
int fmhip_call_v1s1(int opcode, fmhip_vec a, double s, fmhip_vec* out) {
    { const fmhip_vec te_in[1] = { a }; TE_LOCAL(te_in, 1, L, fmhip_call_v1s1(opcode, L[0], s, out)); }
    if (fm::front_active()) { const fmhip_vec in[1] = { a }; return front::call(opcode, 1, in, s, true, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[1] = { a }; *out = Engine::get().call(opcode, 1, in, s, true); });
}
int fmhip_call_v2s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec* out) {
    { const fmhip_vec te_in[2] = { a, b }; TE_LOCAL(te_in, 2, L, fmhip_call_v2s0(opcode, L[0], L[1], out)); }
    if (fm::front_active()) { const fmhip_vec in[2] = { a, b }; return front::call(opcode, 2, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[2] = { a, b }; *out = Engine::get().call(opcode, 2, in, 0.0, false); });
}
int fmhip_call_v2s1(int opcode, fmhip_vec a, fmhip_vec b, double s, fmhip_vec* out) {
    { const fmhip_vec te_in[2] = { a, b }; TE_LOCAL(te_in, 2, L, fmhip_call_v2s1(opcode, L[0], L[1], s, out)); }
    if (fm::front_active()) { const fmhip_vec in[2] = { a, b }; return front::call(opcode, 2, in, s, true, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[2] = { a, b }; *out = Engine::get().call(opcode, 2, in, s, true); });
}
int fmhip_call_v3s0(int opcode, fmhip_vec a, fmhip_vec b, fmhip_vec c, fmhip_vec* out) {
    { const fmhip_vec te_in[3] = { a, b, c }; TE_LOCAL(te_in, 3, L, fmhip_call_v3s0(opcode, L[0], L[1], L[2], out)); }
    if (fm::front_active()) { const fmhip_vec in[3] = { a, b, c }; return front::call(opcode, 3, in, 0.0, false, out); }
    return guarded([&] { need(out, "out"); const fmhip_vec in[3] = { a, b, c }; *out = Engine::get().call(opcode, 3, in, 0.0, false); });
}

int fmhip_set_fusion(int enabled, int* previous) {
    FRONT(set_int(0, enabled, previous));
    TE_ALL(mine, fmhip_set_fusion(enabled, mine ? previous : nullptr));
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
    TE_ALL(mine, fmhip_set_step_grouping(steps, mine ? previous : nullptr));
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
    if (te::active()) {
        const int te_rc = te::status_of([&]() -> int {
            te::ensure();
            bool any = false;
            for (int i = 0; leaf_from && i < n_map; ++i) any |= te::foreign(leaf_from[i]);
            for (int i = 0; leaf_to && i < n_map * n_copies; ++i) any |= te::foreign(leaf_to[i]);
            if (!any) return te::NOT_MINE_TO_HANDLE;
            te::Localized from(leaf_from, n_map), to(leaf_to, n_map * n_copies);
            te::Inside nested;
            return fmhip_graph_clone(roots, n_roots, n_copies, from.local.data(), to.local.data(), n_map, scalars, n_scalars, out);
        });
        if (te_rc != te::NOT_MINE_TO_HANDLE) return te_rc;
    }
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
    TE_ALL(mine, fmhip_set_math_mode(mode, mine ? previous : nullptr));
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
    if (fm::front_active()) {                               // a device list combines its shards' moments itself: a communicator of one rank is what it has
        if (world > 1) return front::unsupported("fmhip_set_expectation_comm");
        if (world < 1 || rank != 0) { g_last_error = "bad communicator: world " + std::to_string(world) + ", rank " + std::to_string(rank); return FMHIP_ERR_INVALID_ARGUMENT; }
        return FMHIP_OK;
    }
    return guarded([&] {
        Engine& e = Engine::get();
        if (world < 1 || rank < 0 || rank >= world) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad communicator: world " + std::to_string(world) + ", rank " + std::to_string(rank));
        if (world > 1 && !gather) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a communicator of more than one rank needs a gather function");
        e.comm_world = gather ? world : 1; e.comm_rank = gather ? rank : 0; e.comm_gather = gather; e.comm_context = context;
    });
}
int fmhip_expectation_world(int* world, int* rank) {
    if (fm::front_active()) { if (world) *world = 1; if (rank) *rank = 0; return FMHIP_OK; }
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
    TE_OWNER(v, fmhip_reduce_moments(v, shift, out));
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
    TE_LOCAL(vectors, count, L, fmhip_reduce_moments_batch(L, count, shifts, out));
    return guarded([&] { need(vectors, "vectors"); need(out, "out"); Engine& e = Engine::get(); e.reduce_batch(vectors, count, shifts, out, nullptr); exchange_moments(e, out, count); });
}
// With a device list the buffer is on the FIRST listed device: it receives the moments of the whole vectors (all shards, combined by the
// one exchange between the devices: fm::front::reduce_moments_batch_devices) — "a single reduce for the final expectations".
static int reduce_to_first_device(const fmhip_vec* vectors, int count, const double* shifts, void* device_out) {
    int shards = 1;
    const int st = front::device_count(&shards);
    if (st != FMHIP_OK) return st;
    std::vector<void*> out((size_t)shards, nullptr);
    out[0] = device_out;
    return front::reduce_moments_batch_devices(vectors, count, shifts, out.data(), shards);
}
int fmhip_reduce_moments_batch_devices(const fmhip_vec* vectors, int count, const double* shifts, void* const* device_out, int n_devices) {
    FRONT(reduce_moments_batch_devices(vectors, count, shifts, device_out, n_devices));
    if (n_devices != 1 || !device_out) { g_last_error = "one device: one output pointer"; return FMHIP_ERR_INVALID_ARGUMENT; }
    return fmhip_reduce_moments_batch_device(vectors, count, shifts, device_out[0]);
}
int fmhip_get_stream_of(int shard, void** stream_out) {
    FRONT(get_stream_of(shard, stream_out));
    if (shard != 0) { g_last_error = "no such device shard: " + std::to_string(shard); return FMHIP_ERR_INVALID_ARGUMENT; }
    return fmhip_get_stream(stream_out);
}
int fmhip_expectation_collective(int* kind, char* why, int why_len) {
    FRONT(expectation_collective(kind, why, why_len));
    if (kind) *kind = 0;
    if (why && why_len > 0) why[0] = 0;
    return FMHIP_OK;
}
int fmhip_reduce_moments_batch_device(const fmhip_vec* vectors, int count, const double* shifts, void* device_out) {
    if (fm::front_active()) return reduce_to_first_device(vectors, count, shifts, device_out);
    TE_LOCAL(vectors, count, L, fmhip_reduce_moments_batch_device(L, count, shifts, device_out));
    return guarded([&] { need(vectors, "vectors"); need(device_out, "device_out"); Engine::get().reduce_batch_device(vectors, count, shifts, device_out); });
}
int fmhip_reduce_moments_batch_begin(const fmhip_vec* vectors, int count, const double* shifts, fmhip_ticket* ticket_out) {
    FRONT(reduce_moments_batch_begin(vectors, count, shifts, ticket_out));
    TE_LOCAL(vectors, count, L, fmhip_reduce_moments_batch_begin(L, count, shifts, ticket_out));
    return guarded([&] { need(vectors, "vectors"); need(ticket_out, "ticket_out"); *ticket_out = Engine::get().reduce_batch_begin(vectors, count, shifts); });
}
int fmhip_vec_give_up_values(const fmhip_vec* vectors, int count) {
    FRONT(vec_give_up_values(vectors, count));
    if (te::active()) {          // a value of another thread's engine: that engine's to give up
        bool any_foreign = false;
        const int st = te::status_of([&]() -> int { te::ensure(); for (int i = 0; vectors && i < count; ++i) any_foreign |= te::foreign(vectors[i]); return FMHIP_OK; });
        if (st != FMHIP_OK) return st;
        if (any_foreign) { for (int i = 0; i < count; ++i) { const fmhip_vec one = vectors[i]; int rc = te::owner_routed(one, [&]() -> int { return fmhip_vec_give_up_values(&one, 1); }); if (rc == te::NOT_MINE_TO_HANDLE) { te::Inside nested; rc = fmhip_vec_give_up_values(&one, 1); } if (rc != FMHIP_OK) return rc; } return FMHIP_OK; }
    }
    return guarded([&] { need(vectors, "vectors"); Engine::get().give_up_values(vectors, count); });
}
int fmhip_reduce_moments_batch_end(fmhip_ticket ticket, fmhip_moments* out, int count) {
    FRONT(reduce_moments_batch_end(ticket, out, count));
    TE_OWNER(ticket, fmhip_reduce_moments_batch_end(ticket, out, count));
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
    if (fm::front_active()) { const double sh[1] = { shift }; return reduce_to_first_device(&v, 1, sh, device_out_4_doubles); }
    TE_OWNER(v, fmhip_reduce_moments_device(v, shift, device_out_4_doubles));
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
    TE_OWNER(p, fmhip_program_shape(p, n_inputs, n_outputs, n_reduce));
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
    TE_ALL(mine, fmhip_set_jit(mode, mine ? previous : nullptr));
    return guarded([&] {
        if (mode != FMHIP_JIT_OFF && mode != FMHIP_JIT_AUTO && mode != FMHIP_JIT_SYNC) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown JIT mode");
        if (previous) *previous = Engine::get().jit_mode;
        Engine::get().jit_mode = mode;
    });
}
int fmhip_jit_wait(void) {
    FRONT(jit_wait());
    TE_ALL(mine, fmhip_jit_wait());
    // not under the engine mutex: other threads keep launching while this one waits for the compiler thread
    try { Engine::get().jit_wait(); return FMHIP_OK; } catch (...) { return FMHIP_ERR_HIP; }
}
int fmhip_jit_stats(int64_t* compiled, int64_t* failed, int64_t* pending, double* compile_seconds, int64_t* disk_cache_hits) {
    FRONT(jit_stats(compiled, failed, pending, compile_seconds, disk_cache_hits));
    if (te::active()) { int64_t c = 0, f = 0, pd = 0, d = 0; double sec = 0; const int rc = te::on_all([&](bool) -> int { int64_t c1 = 0, f1 = 0, p1 = 0, d1 = 0; double s1 = 0; const int st = fmhip_jit_stats(&c1, &f1, &p1, &s1, &d1); c += c1; f += f1; pd += p1; d += d1; sec += s1; return st; });
        if (compiled) *compiled = c; if (failed) *failed = f; if (pending) *pending = pd; if (compile_seconds) *compile_seconds = sec; if (disk_cache_hits) *disk_cache_hits = d; return rc; }
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
    if (!fm::front_active()) TE_OWNER(p, fmhip_program_tier(p, tier, vgprs));
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
int fmhip_program_release(fmhip_program p) { FRONT(program_release(p)); TE_OWNER(p, fmhip_program_release(p)); return guarded([&] { Engine::get().program_release(p); }); }
int fmhip_program_launch_count(fmhip_program p, int* n_launches) {
    if (fm::front_active()) { int a = 0, shards = 1; int st = front::program_shape(p, &a, nullptr, nullptr); if (st == FMHIP_OK) st = front::device_count(&shards); if (st == FMHIP_OK && n_launches) *n_launches = shards; return st; }      // one launch per device shard
    TE_OWNER(p, fmhip_program_launch_count(p, n_launches));
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
    if (te::active()) {
        const int te_rc = te::status_of([&]() -> int {
            te::ensure();
            int n_in = 0;
            { te::Rebind r(te::owner(p)); const int st = fmhip_program_shape(p, &n_in, nullptr, nullptr); if (st != FMHIP_OK) return st; }
            bool any = te::foreign(p);
            for (int i = 0; inputs && i < batch * n_in; ++i) any |= Engine::owner_of(inputs[i]) != Engine::owner_of(p);
            if (!any) return te::NOT_MINE_TO_HANDLE;
            te::Rebind r(te::owner(p));
            te::Localized L(inputs, batch * n_in);
            return fmhip_program_run(p, batch, L.local.data(), outputs, reduce_shift, moments, device_moments);
        });
        if (te_rc != te::NOT_MINE_TO_HANDLE) return te_rc;
    }
    return guarded([&] { Engine::get().program_run(p, batch, inputs, outputs, false, reduce_shift, moments, device_moments); });
}
int fmhip_program_run_into(fmhip_program p, int batch, const fmhip_vec* inputs, const fmhip_vec* outputs,
                           const double* reduce_shift, fmhip_moments* moments, void* device_moments) {
    if (fm::front_active()) return device_moments ? front::unsupported("fmhip_program_run_into with device_moments") : front::program_run(p, batch, inputs, const_cast<fmhip_vec*>(outputs), true, reduce_shift, moments);
    if (te::active()) {
        const int te_rc = te::status_of([&]() -> int {
            te::ensure();
            int n_in = 0, n_out = 0;
            { te::Rebind r(te::owner(p)); const int st = fmhip_program_shape(p, &n_in, &n_out, nullptr); if (st != FMHIP_OK) return st; }
            bool any = te::foreign(p);
            for (int i = 0; inputs && i < batch * n_in; ++i) any |= Engine::owner_of(inputs[i]) != Engine::owner_of(p);
            for (int i = 0; outputs && i < batch * n_out; ++i) if (Engine::owner_of(outputs[i]) != Engine::owner_of(p)) throw Error(FMHIP_ERR_UNSUPPORTED, "fmhip_program_run_into writes into vectors of the program's own engine (thread engines)");
            if (!any) return te::NOT_MINE_TO_HANDLE;
            te::Rebind r(te::owner(p));
            te::Localized L(inputs, batch * n_in);
            return fmhip_program_run_into(p, batch, L.local.data(), outputs, reduce_shift, moments, device_moments);
        });
        if (te_rc != te::NOT_MINE_TO_HANDLE) return te_rc;
    }
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

int fmhip_pool_clean(void) { FRONT(pool(0)); TE_ALL(mine, fmhip_pool_clean()); return guarded([&] { Engine::get().pool_clean(); }); }
int fmhip_pool_purge(void) { FRONT(pool(1)); TE_ALL(mine, fmhip_pool_purge()); return guarded([&] { Engine::get().pool_purge(); }); }
int fmhip_pool_stats(fmhip_pool_stats_t* out) { FRONT(pool_stats(out)); if (te::active()) { if (!out) { g_last_error = "null pointer: out"; return FMHIP_ERR_INVALID_ARGUMENT; } fmhip_pool_stats_t sum; std::memset(&sum, 0, sizeof sum); bool first = true;
        const int rc = te::on_all([&](bool) -> int { fmhip_pool_stats_t one; const int st = fmhip_pool_stats(&one); if (st != FMHIP_OK) return st;
            if (first) { sum = one; first = false; } else { sum.bytes_reserved += one.bytes_reserved; sum.bytes_in_use += one.bytes_in_use; sum.bytes_cached += one.bytes_cached; sum.n_alloc_hits += one.n_alloc_hits; sum.n_alloc_misses += one.n_alloc_misses; sum.n_live_vectors += one.n_live_vectors; sum.n_kernel_launches += one.n_kernel_launches; sum.n_ops_executed += one.n_ops_executed; }
            return FMHIP_OK; });
        if (rc == FMHIP_OK) *out = sum;
        return rc; }
    return guarded([&] { Engine::get().pool_stats(out); }); }

int fmhip_traffic_stats(int64_t* algorithmic_bytes, int64_t* specialised_launches) {
    FRONT(traffic_stats(algorithmic_bytes, specialised_launches));
    if (te::active()) { int64_t b = 0, l = 0; const int rc = te::on_all([&](bool) -> int { int64_t b1 = 0, l1 = 0; const int st = fmhip_traffic_stats(&b1, &l1); b += b1; l += l1; return st; });
        if (algorithmic_bytes) *algorithmic_bytes = b; if (specialised_launches) *specialised_launches = l; return rc; }
    return guarded([&] {
        if (algorithmic_bytes) *algorithmic_bytes = Engine::get().algorithmic_bytes();
        if (specialised_launches) *specialised_launches = Engine::get().jit_launches();
    });
}
int fmhip_engine_stats(fmhip_engine_stats_t* out) {
    FRONT(engine_stats(out));
    if (te::active()) {
        if (!out) { g_last_error = "null pointer: out"; return FMHIP_ERR_INVALID_ARGUMENT; }
        fmhip_engine_stats_t sum; std::memset(&sum, 0, sizeof sum);
        const int rc = te::on_all([&](bool) -> int { fmhip_engine_stats_t one; const int st = fmhip_engine_stats(&one); if (st != FMHIP_OK) return st;
            int64_t* a = &sum.size; const int64_t* b = &one.size; for (size_t i = 1; i < sizeof sum / sizeof(int64_t); ++i) a[i] += b[i]; return FMHIP_OK; });
        sum.size = (int64_t)sizeof sum;
        if (rc == FMHIP_OK) *out = sum;
        return rc;
    }
    return guarded([&] { Engine::get().engine_stats(out); });
}
int fmhip_profile_enable(int enabled) { FRONT(profile_enable(enabled)); TE_ALL(mine, fmhip_profile_enable(enabled)); return guarded([&] { Engine::get().profile_enable(enabled != 0); }); }
int fmhip_profile_read(double* kernel_ms_total, int64_t* n_launches) {
    FRONT(profile_read(kernel_ms_total, n_launches));
    if (te::active()) { double ms = 0; int64_t n = 0; const int rc = te::on_all([&](bool) -> int { double m1 = 0; int64_t n1 = 0; const int st = fmhip_profile_read(&m1, &n1); ms += m1; n += n1; return st; });
        if (kernel_ms_total) *kernel_ms_total = ms; if (n_launches) *n_launches = n; return rc; }
    return guarded([&] { Engine::get().profile_read(kernel_ms_total, n_launches); });
}

} // extern "C"
