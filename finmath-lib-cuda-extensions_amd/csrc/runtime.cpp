// runtime.cpp — see runtime.hpp.  Host-only C++; all device work goes through kernels.h.
#include "runtime.hpp"
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <map>
#include <sstream>
#include <thread>
#include <chrono>
#include <unordered_set>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fm {
static std::atomic<uint64_t> g_te_malloc_ns{ 0 }, g_te_malloc_calls{ 0 }, g_te_init_ns{ 0 };     // FMHIP_TE_TRACE: where an engine's start goes

void EngineMutex::lock_slow(uint64_t me) {
    for (unsigned spins = 0;; ++spins) {
        uint64_t expected = 0;
        if (owner_.load(std::memory_order_relaxed) == 0 && owner_.compare_exchange_weak(expected, me, std::memory_order_acquire, std::memory_order_relaxed)) return;
        if (spins < 256) {
#if defined(__x86_64__)
            _mm_pause();
#endif
        } else if (spins < 320) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

void hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    (void)hipGetLastError();
    const int code = (e == hipErrorOutOfMemory) ? FMHIP_ERR_OUT_OF_MEMORY : FMHIP_ERR_HIP;
    throw Error(code, std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------- opcode table

struct OpInfo { int n_vec; bool scalar; };
static OpInfo op_info(int opcode) {
    if (opcode >= FMHIP_OP_CAP_S && opcode <= FMHIP_OP_POW_S) return {1, true};
    if (opcode >= FMHIP_OP_SQUARED && opcode <= FMHIP_OP_ISNAN) return {1, false};
    if (opcode >= FMHIP_OP_CAP && opcode <= FMHIP_OP_DIV) return {2, false};
    if (opcode >= FMHIP_OP_ACCRUE && opcode <= FMHIP_OP_ADDPRODUCT_VS) return {2, true};
    if (opcode >= FMHIP_OP_ADDPRODUCT && opcode <= FMHIP_OP_CHOOSE) return {3, false};
    return {0, false};
}

// ---------------------------------------------------------------- pool

static size_t round_cap(size_t bytes) {
    size_t cap = (bytes + 255) & ~size_t(255);
    return cap ? cap : 256;
}

void* Pool::alloc(size_t bytes, size_t* cap_out) {
    const size_t cap = round_cap(bytes);
    *cap_out = cap;
    // test hook (tests/test_gpu_replicas.py): the N-th allocation of the process fails like a device that is out of memory
    static const long long FAIL_AT = [] { const char* e = std::getenv("FMHIP_TEST_FAIL_ALLOC_AT"); return e ? std::atoll(e) : 0ll; }();
    static std::atomic<long long> calls{ 0 };            // (several engines — thread engines, device lists — allocate side by side)
    if (FAIL_AT > 0 && ++calls == FAIL_AT) throw Error(FMHIP_ERR_OUT_OF_MEMORY, "device allocation failed (FMHIP_TEST_FAIL_ALLOC_AT)");
    std::vector<void*>& fl = free_[cap];
    if (!fl.empty()) {
        void* p = fl.back();
        fl.pop_back();
        hits++; cached -= (int64_t)cap; in_use += (int64_t)cap;
        return p;
    }
    // Miss.  Device allocations are expensive (≈ 100 µs each) and a Monte-Carlo state is thousands of equally sized vectors:
    // allocate SLABS of 1, 2, 4 … 64 blocks of this size class (≤ 1 GiB per slab) and hand the rest to the free list.
    const size_t grown = slab_blocks_[cap];
    size_t blocks = grown ? std::min<size_t>(grown * 2, 64) : 1;
    while (blocks > 1 && blocks * cap > (size_t(1) << 30)) blocks /= 2;
    void* base = nullptr;
    const auto tm0 = std::chrono::steady_clock::now();
    struct Tm { std::chrono::steady_clock::time_point t0; ~Tm() { g_te_malloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++g_te_malloc_calls; } } tm{ tm0 };
    // The pool never takes the LAST of the device: the HIP runtime itself allocates device memory while the process runs (code objects the
    // JIT tier loads, kernels loaded at their first launch, scratch, signals), and a caller whose handles die late — a garbage-collected
    // one — grows the pool until something fails.  With less than the headroom left a miss counts as out of memory (purge, then the error
    // the caller answers with a collection: RandomVariableCuda.java:311-335 does the same below a free-memory percentage).  Asked on a miss
    // only: a slab allocation costs ≈ 100 µs, the query a few.
    static const size_t HEADROOM = [] { const char* e = std::getenv("FMHIP_POOL_HEADROOM_BYTES"); return e ? (size_t)std::atoll(e) : (size_t(2) << 30); }();
    auto room_for = [&](size_t bytes) { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return true; } return fr >= bytes + HEADROOM; };
    hipError_t e = hipErrorOutOfMemory;
    if (blocks > 1 && !room_for(blocks * cap)) blocks = 1;
    if (room_for(blocks * cap)) {
        e = hipMalloc(&base, blocks * cap);
        if (e != hipSuccess && blocks > 1) { (void)hipGetLastError(); blocks = 1; e = hipMalloc(&base, cap); }
    }
    if (e != hipSuccess) {
        // The device is full.  (1) Cached slabs go back to the driver — as many as this allocation needs and a little more, not all of
        // them (the reference's last resort drops every cached buffer, RandomVariableCuda.java:340; here a caller whose collector had
        // just handed back 290 GB of vectors lost that whole cache to the next 40 KB allocation of an unusual size, and paid for it in
        // device allocations: 75 s instead of 6 for the calibration of lmm_hip --finmath-like --release-lag-bytes 268435456).
        (void)hipGetLastError();
        purge(cap + (size_t(256) << 20));
        blocks = 1;
        if (room_for(cap)) e = hipMalloc(&base, cap);
    }
    if (e != hipSuccess) {
        // (2) Nothing could be given back — slabs are freed whole, and every one of them has a live vector in it — but blocks of a LARGER
        // size class sit cached: one of them is carved into blocks of this class (a 40 KB buffer for a reduction's partials must not fail
        // while gigabytes are cached in 4 MB blocks).
        (void)hipGetLastError();
        if (void* p = borrow(cap)) { misses++; return p; }
        throw Error(FMHIP_ERR_OUT_OF_MEMORY, "device allocation of " + std::to_string(cap) + " bytes failed: " + hipGetErrorString(e) + " (the pool holds " + std::to_string(reserved >> 20) +
                    " MiB, " + std::to_string(in_use >> 20) + " MiB of them in live vectors; " + std::to_string(HEADROOM >> 20) + " MiB of the device are left to the HIP runtime)");
    }
    // (looked up again: purge() above drops the entries of size classes it has emptied — until round 5 a reference taken before the purge
    // was written through here, into a freed map node: heap corruption whenever an allocation of a rarely used size met a full device)
    slab_blocks_[cap] = blocks;
    slabs_.push_back({ base, cap, blocks, 0 });
    std::vector<void*>& fl2 = free_[cap];
    for (size_t i = blocks; i-- > 1;) { fl2.push_back((char*)base + i * cap); cached += (int64_t)cap; }
    misses++; reserved += (int64_t)(blocks * cap); in_use += (int64_t)cap;
    peak_reserved = std::max(peak_reserved, reserved);
    return base;
}

// A cached block of the smallest larger size class becomes a slab of this one (Slab::parent_cap); purge() hands it back when all its
// blocks are free again.
void* Pool::borrow(size_t cap) {
    size_t best = 0;
    for (const auto& kv : free_) if (kv.first > cap && !kv.second.empty() && (best == 0 || kv.first < best)) best = kv.first;
    if (!best) return nullptr;
    void* base = free_[best].back();
    free_[best].pop_back();
    const size_t blocks = best / cap;
    cached -= (int64_t)best;
    slabs_.push_back({ base, cap, blocks, best });
    std::vector<void*>& fl = free_[cap];
    for (size_t i = blocks; i-- > 1;) { fl.push_back((char*)base + i * cap); cached += (int64_t)cap; }
    in_use += (int64_t)cap;
    return base;
}

void Pool::release(void* p, size_t cap) {
    free_[cap].push_back(p);
    in_use -= (int64_t)cap; cached += (int64_t)cap;
}

// Slabs whose blocks are ALL back in the free list go back to the driver (a slab with one live vector stays) — all of them, or as many as
// it takes to give back `need` bytes.  Slabs that were carved out of a cached block (borrow) return that block to its own free list first.
void Pool::purge(size_t need) {
    std::unordered_map<size_t, std::unordered_set<void*>> free_set;
    for (auto& kv : free_) free_set[kv.first].insert(kv.second.begin(), kv.second.end());
    auto all_free = [&](const Slab& sl) {
        const std::unordered_set<void*>& fs = free_set[sl.cap];
        for (size_t i = 0; i < sl.blocks; ++i) if (!fs.count((char*)sl.base + i * sl.cap)) return false;
        return true;
    };
    std::vector<Slab> kept;
    for (const Slab& sl : slabs_) {                          // carved slabs first: their parents' slabs may become free by it
        if (!sl.parent_cap) { kept.push_back(sl); continue; }
        if (!all_free(sl)) { kept.push_back(sl); continue; }
        std::unordered_set<void*>& fs = free_set[sl.cap];
        for (size_t i = 0; i < sl.blocks; ++i) fs.erase((char*)sl.base + i * sl.cap);
        cached -= (int64_t)(sl.blocks * sl.cap);
        free_set[sl.parent_cap].insert(sl.base);
        cached += (int64_t)sl.parent_cap;
    }
    slabs_.swap(kept);
    kept.clear();
    size_t given_back = 0;
    for (size_t k = slabs_.size(); k-- > 0;) {                // youngest first: the old slabs hold the state a caller keeps
        const Slab& sl = slabs_[k];
        if (sl.parent_cap || given_back >= need || !all_free(sl)) { kept.push_back(sl); continue; }
        std::unordered_set<void*>& fs = free_set[sl.cap];
        for (size_t i = 0; i < sl.blocks; ++i) fs.erase((char*)sl.base + i * sl.cap);
        (void)hipFree(sl.base);
        reserved -= (int64_t)(sl.blocks * sl.cap); cached -= (int64_t)(sl.blocks * sl.cap);
        given_back += sl.blocks * sl.cap;
    }
    std::reverse(kept.begin(), kept.end());
    slabs_.swap(kept);
    free_.clear();
    for (auto& kv : free_set) if (!kv.second.empty()) free_[kv.first].assign(kv.second.begin(), kv.second.end());
    for (auto it = slab_blocks_.begin(); it != slab_blocks_.end();) {       // size classes without slabs start small again
        bool any = false;
        for (const Slab& sl : slabs_) any |= sl.cap == it->first && !sl.parent_cap;
        it = any ? std::next(it) : slab_blocks_.erase(it);
    }
}

// ---------------------------------------------------------------- engine lifecycle

// The engine of the calling thread: the process-wide one — or, on a worker thread of a device list (sharded.cpp: one engine per
// device shard, each driven by a thread of its own), that worker's.
static thread_local Engine* tls_engine = nullptr;
Engine& Engine::get() {
    if (tls_engine) { if (!tls_engine->retired.load(std::memory_order_acquire)) return *tls_engine; tls_engine = nullptr; }      // (a thread engine that fmhip_shutdown has retired)
    static Engine* e = new Engine();
    return *e;
}
Engine* Engine::create() { return new Engine(); }
void Engine::bind_thread(Engine* e) { tls_engine = e; }
bool Engine::thread_is_bound() { if (tls_engine && tls_engine->retired.load(std::memory_order_acquire)) tls_engine = nullptr; return tls_engine != nullptr; }

void Engine::require_init() const {
    if (!initialized_) throw Error(FMHIP_ERR_NOT_INITIALIZED, "fmhip_init has not been called");
    static thread_local int bound_device = -1;          // hipSetDevice is per thread; skip it on the hot path once bound
    if (bound_device != device_) { hip_check(hipSetDevice(device_), "hipSetDevice"); bound_device = device_; }
}

void fusion_max_weight_override(int v);      // FMHIP_FUSION_MAX_WEIGHT (experiments)
HostProfile g_host_profile;
void HostProfile::report() const {
    static const char* names[N_SLOTS] = { "call (record one method)", "release", "flush_all (total)", "  build_dag", "  run_dags (total)", "    launch (total)",
                                          "      kernel launch API", "      row table upload", "reduce / reduce_batch (total)",
                                          "  flush: roots + components", "  build_big (walk, schedule, sign)", "  run_big_group (total)",
                                          "graph_clone (total)" };
    std::fprintf(stderr, "[fmhip host profile]\n");
    for (int i = 0; i < N_SLOTS; ++i)
        std::fprintf(stderr, "  %-32s %10lld calls %9.3f s %9.2f us/call\n", names[i], count[i], seconds[i], count[i] ? seconds[i] / count[i] * 1e6 : 0.0);
}

void Engine::init(int device_index) {
    const auto ti0 = std::chrono::steady_clock::now();
    struct Ti { std::chrono::steady_clock::time_point t0; ~Ti() { g_te_init_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } ti{ ti0 };
    if (device_index < 0) {
        const char* e = std::getenv("FMHIP_DEVICE_INDEX");
        if (!e) e = std::getenv("LOCAL_RANK");
        device_index = e ? std::atoi(e) : 0;
    }
    int count = 0;
    hip_check(hipGetDeviceCount(&count), "hipGetDeviceCount");
    if (count <= 0) throw Error(FMHIP_ERR_HIP, "no HIP device visible");
    if (device_index >= count) device_index = device_index % count;
    if (initialized_) {
        if (device_index != device_) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "already initialised on another device");
        return;
    }
    hip_check(hipSetDevice(device_index), "hipSetDevice");
    hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    ring_cap_ = size_t(32) << 20;                // row tables of rolled loops are ≈ 5 KB per row: room for hundreds of launches between wraps
    if (const char* e = std::getenv("FMHIP_RING_BYTES")) {      // tests shrink the ring so that a wrap costs a few launches, not thousands
        const long long v = std::atoll(e);
        if (v >= 4096) ring_cap_ = ((size_t)v + 255) & ~size_t(255);
    }
    hip_check(hipHostMalloc(&ring_host_, ring_cap_, hipHostMallocDefault), "hipHostMalloc(ring)");
    hip_check(hipMalloc(&ring_dev_, ring_cap_ + 256), "hipMalloc(ring)");
    hip_check(hipMalloc(&dump_dev_, FM_DUMP_BYTES), "hipMalloc(dump)");
    hip_check(hipHostMalloc((void**)&result_slots_, (size_t)RESULT_SLOTS * 128, hipHostMallocDefault), "hipHostMalloc(result slots)");
    ARENA_BYTES = size_t(16) << 20;
    if (const char* e = std::getenv("FMHIP_ARENA_BYTES")) { const long long v = std::atoll(e); if (v >= 1024) ARENA_BYTES = ((size_t)v + 31) & ~size_t(31); }      // tests: a wrap after a few dozen expectations
    hip_check(hipHostMalloc((void**)&moments_arena_, ARENA_BYTES, hipHostMallocDefault), "hipHostMalloc(moments arena)");
    arena_off_ = 0; arena_outstanding_.clear();
    free_slots_.clear();
    for (int i = RESULT_SLOTS; i-- > 0;) free_slots_.push_back(i);
    { const char* e = std::getenv("FMHIP_UNIT_WORKGROUPS"); unit_workgroups_ = e ? std::atoll(e) : 512; }
    hip_check(hipMalloc((void**)&counters_dev_, FM_COUNTER_PLANES * FM_COUNTER_PLANE * sizeof(uint32_t)), "hipMalloc(counters)");
    hip_check(hipMemsetAsync(counters_dev_, 0, FM_COUNTER_PLANES * FM_COUNTER_PLANE * sizeof(uint32_t), stream_), "hipMemset(counters)");
    hip_check(hipStreamSynchronize(stream_), "init sync");
    hip_check(preload_kernels(), "loading the device code");
    ring_off_ = 0;
    device_ = device_index;
    if (const char* e = std::getenv("FMHIP_JIT")) {
        const std::string v(e);
        if (v == "off" || v == "0") jit_mode = FMHIP_JIT_OFF;
        else if (v == "sync" || v == "2") jit_mode = FMHIP_JIT_SYNC;
        else if (v == "auto" || v == "1") jit_mode = FMHIP_JIT_AUTO;
    }
    if (!jit_shared_) jit_.start(device_index);              // (a thread engine uses the first engine's compiler and code objects)
    { static std::once_flag once; std::call_once(once, [] { g_host_profile.on = std::getenv("FMHIP_HOST_PROFILE") != nullptr; }); }      // (process-wide: several engines initialise side by side behind a device list)
    for (const char* name : { "FMHIP_BM_GROUP_STEPS", "FMHIP_GROUP_STEPS" })       // (the first: where round 2 had this, in BrownianMotionHip)
        if (const char* e = std::getenv(name)) { const int v = std::atoi(e); if (v >= 0) group_steps = v; }
    group_bm_id_ = 0; group_last_step_ = -1; group_steps_pending_ = 0; group_hold_ = false; group_last_by_bm_.clear();
    { static std::once_flag once; std::call_once(once, [] { if (const char* e = std::getenv("FMHIP_FUSION_MAX_WEIGHT")) { const int v = std::atoi(e); if (v > 0) fusion_max_weight_override(v); } }); }
    initialized_ = true;
}

void Engine::shutdown() {
    if (std::getenv("FMHIP_TE_TRACE")) std::fprintf(stderr, "[fmhip te] engine %d: so far in the process: hipMalloc of the pools %.1f ms in %llu calls, Engine::init %.1f ms\n", index_, g_te_malloc_ns.load() / 1e6, (unsigned long long)g_te_malloc_calls.load(), g_te_init_ns.load() / 1e6);
    if (!initialized_) return;
    drain_late();
    if (g_host_profile.on) g_host_profile.report();
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    if (!jit_shared_) jit_.stop();      // joins the compiler thread, unloads the specialised kernels
    jit_shared_ = nullptr;
    nodes_.for_each([&](Node* nd) {     // leak-safe teardown: free storage of every live vector
        if (nd->buf) buffer_unref(nd->buf);     // back to the pool (blocks belong to slabs); purge() below frees the slabs
        delete nd;
    });
    nodes_.clear();
    for (auto& kv : replicas_) delete kv.second;
    replicas_.clear();
    pend_clear();
    for (Node* nd : node_pool_) delete nd;
    node_pool_.clear();
    for (auto& kv : programs_) if (--kv.second->refs == 0) delete kv.second;
    programs_.clear();
    for (auto& kv : plan_cache_) for (BigPlan::Seg& seg : kv.second.segs) { if (--seg.prog->refs == 0) delete seg.prog; delete seg.prog_red; }
    plan_cache_.clear();
    schedule_cache_.clear(); schedule_cache_bytes_ = 0; dag_policies_.clear(); policy_reset();
    for (auto& kv : program_cache_) if (--kv.second->refs == 0) delete kv.second;
    program_cache_.clear();
    pool_.purge();
    pool_ = Pool();
    if (stage_) (void)hipHostFree(stage_);
    if (result_slots_) (void)hipHostFree(result_slots_);
    if (moments_arena_) (void)hipHostFree(moments_arena_);
    moments_arena_ = nullptr; arena_off_ = 0; arena_outstanding_.clear();
    for (auto& kv : tickets_) free_tickets_.push_back(kv.second);
    tickets_.clear();
    for (MomentsTicket& t : free_tickets_) { if (t.event) (void)hipEventDestroy(t.event); if (t.host) (void)hipHostFree(t.host); }
    free_tickets_.clear();
    result_slots_ = nullptr; free_slots_.clear();
    if (ring_host_) (void)hipHostFree(ring_host_);
    if (ring_dev_) (void)hipFree(ring_dev_);
    if (counters_dev_) (void)hipFree(counters_dev_);
    if (dump_dev_) (void)hipFree(dump_dev_);
    counters_dev_ = nullptr; dump_dev_ = nullptr;
    stage_ = ring_host_ = ring_dev_ = nullptr; stage_cap_ = ring_cap_ = ring_off_ = 0;
    (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
    initialized_ = false;
    device_ = -1;
}

void Engine::synchronize() { require_init(); if (has_late()) drain_late(); hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize"); }

void Engine::device_info(char* name, int len, int* cus, int64_t* hbm) {
    require_init();
    hipDeviceProp_t prop;
    hip_check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
    if (name && len > 0) {          // some boxes report an empty marketing name: fall back to the architecture string
        std::strncpy(name, prop.name[0] ? prop.name : prop.gcnArchName, (size_t)len - 1);
        name[len - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
}

void* Engine::ensure_stage(size_t bytes) {
    if (bytes > stage_cap_) {
        if (stage_) { hip_check(hipStreamSynchronize(stream_), "sync"); (void)hipHostFree(stage_); stage_ = nullptr; stage_cap_ = 0; }
        size_t cap = std::max(bytes, size_t(1) << 20);
        hip_check(hipHostMalloc(&stage_, cap, hipHostMallocDefault), "hipHostMalloc(stage)");
        stage_cap_ = cap;
    }
    return stage_;
}

size_t Engine::ring_reserve(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes > ring_cap_) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "batch too large for the row-table ring");
    if (ring_off_ + bytes > ring_cap_) {        // wrap: earlier tables may still be read by queued kernels
        wait_for_stream("hipStreamSynchronize(ring wrap)");
        ring_off_ = 0;
        ++ring_generation_;                     // device copies of earlier tables are about to be overwritten: nobody may reuse them
    }
    const size_t off = ring_off_;
    ring_off_ += bytes;
    return off;
}

// ---------------------------------------------------------------- buffers / nodes

Buffer* Engine::new_buffer(int64_t n_floats) {
    Buffer* b = new Buffer();
    size_t cap = 0;
    try { b->ptr = (float*)pool_.alloc((size_t)n_floats * 4, &cap); }
    catch (const Error& e) {
        // (out of memory with releases still queued: they are performed — their vectors go back to the pool — and the allocation is tried again)
        if (e.code != FMHIP_ERR_OUT_OF_MEMORY || !has_late()) { delete b; throw; }
        try { drain_late(); b->ptr = (float*)pool_.alloc((size_t)n_floats * 4, &cap); } catch (...) { delete b; throw; }
    }
    catch (...) { delete b; throw; }
    b->cap = cap; b->refs = 1;
    return b;
}

void Engine::buffer_unref(Buffer* b) {
    if (--b->refs > 0) return;
    if (b->foreign >= 0) {          // another engine's storage: whatever this stream still does with it comes before `done`; the owner's reference goes back later, outside this lock
        Foreign& f = foreign_[(size_t)b->foreign];
        hipEvent_t done = nullptr;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(done, stream_);
        foreign_done_.push_back({ f.owner, f.handle, done });
        import_of_.erase(f.handle);
        f.live = false;
        foreign_free_.push_back(b->foreign);
    } else if (b->parent) buffer_unref(b->parent);
    else pool_.release(b->ptr, b->cap);
    delete b;
}

void Engine::set_index(int i) {
    index_ = i;
    next_id_ = ((int64_t)i << OWNER_SHIFT) + 1;
    next_ticket_ = ((int64_t)i << OWNER_SHIFT) + 1;
}

Engine::Exported Engine::export_vector(fmhip_vec h, hipEvent_t ready) {
    require_init();
    Node* nd = node(h);
    touch(nd);
    if (!nd->buf) materialize({ nd });
    make_private(nd);               // (the importer aliases the storage by its address and is kept safe by a reference on THIS vector: storage shared with other vectors — common rows — could be given up by an in-place write here)
    nd->refs_ext++;
    hip_check(hipEventRecord(ready, stream_), "hipEventRecord(export)");
    Exported x;
    x.ptr = nd->buf->ptr; x.n = nd->n;
    if (nd->bm_id) { x.bm_id = ((uint32_t)(index_ + 1) << 24) | (nd->bm_id & 0xffffffu); x.bm_step = nd->bm_step; x.bm_steps = nd->bm_steps; }
    return x;
}

fmhip_vec Engine::import_vector(int owner, fmhip_vec owner_handle, const Exported& x, hipEvent_t ready) {
    require_init();
    hip_check(hipStreamWaitEvent(stream_, ready, 0), "hipStreamWaitEvent(import)");
    int32_t slot;
    if (!foreign_free_.empty()) { slot = foreign_free_.back(); foreign_free_.pop_back(); } else { slot = (int32_t)foreign_.size(); foreign_.push_back({}); }
    foreign_[(size_t)slot] = { owner, owner_handle, true };
    Buffer* b = new Buffer();
    b->ptr = x.ptr; b->cap = 0; b->refs = 1; b->foreign = slot;
    Node* nd = new_node(x.n);
    nd->buf = b;
    nd->moments_blocked = true;                                // (its moments, if anybody asks, are the owner's to keep)
    nd->bm_id = x.bm_id; nd->bm_step = x.bm_step; nd->bm_steps = x.bm_steps;
    import_of_[owner_handle] = nd;
    return nd->id;
}

fmhip_vec Engine::find_import(fmhip_vec foreign_handle) {
    auto it = import_of_.find(foreign_handle);
    if (it == import_of_.end()) return 0;
    Node* nd = it->second;                   // alive: it leaves the map when its storage goes (buffer_unref)
    if (++nd->refs_ext == 1) nodes_.put(nd->id, nd);          // kept by pending consumers only: a handle again
    return nd->id;
}

std::vector<Engine::ForeignDone> Engine::take_foreign_done() { std::vector<ForeignDone> out; out.swap(foreign_done_); return out; }

void Engine::release_exported(fmhip_vec h, hipEvent_t done) {
    require_init();
    if (done) (void)hipStreamWaitEvent(stream_, done, 0);
    release(h);
}

Node* Engine::new_node(int64_t n) {
    Node* nd;
    if (!node_pool_.empty()) { nd = node_pool_.back(); node_pool_.pop_back(); *nd = Node(); }      // recycled: no malloc on the hot path
    else nd = new Node();
    nd->id = next_id_++;
    nd->n = n;
    nd->refs_ext = 1;
    nodes_.put(nd->id, nd);
    return nd;
}

Node* Engine::node(fmhip_vec h) {
    if (owner_of(h) != index_) throw Error(FMHIP_ERR_INVALID_HANDLE, "vector handle " + std::to_string(h) + " belongs to another engine");
    Node* nd = nodes_.get(h);
    if (!nd) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid vector handle " + std::to_string(h));
    return nd;
}

void Engine::drop_expression(Node* nd) {
    for (int i = 0; i < nd->n_in; ++i) {
        Node* in = nd->in[i];
        nd->in[i] = nullptr;
        if (in) node_unref_int(in);
    }
    nd->n_in = 0; nd->opcode = 0; nd->weight = 0;
}

void Engine::node_unref_int(Node* nd) { nd->refs_int--; node_maybe_free(nd); }

void Engine::node_maybe_free(Node* nd) {
    if (nd->refs_ext > 0 || nd->refs_int > 0) return;
    if (nd->buf) buffer_unref(nd->buf);
    else { pend_erase(nd); drop_expression(nd); }
    static const size_t NODE_POOL_CAP = [] { const char* e = std::getenv("FMHIP_NODE_POOL_CAP"); return e ? (size_t)std::atoll(e) : (size_t)65536; }();      // (measured under a lagging collector: recycling ALL the nodes a burst of releases frees — random addresses — is no faster than fresh, consecutive ones)
    if (node_pool_.size() < NODE_POOL_CAP) node_pool_.push_back(nd); else delete nd;      // (a collector's burst frees hundreds of thousands at once: they are the next ones handed out, most recently touched first)
}

void Engine::retain(fmhip_vec h) { require_init(); node(h)->refs_ext++; }

void Engine::release(fmhip_vec h) {
    HostTimer timer(HostProfile::RELEASE);
    require_init();
    Node* nd = node(h);
    if (--nd->refs_ext == 0) { nodes_.erase(h); node_maybe_free(nd); }
}

void Engine::drain_late(size_t at_most) {
    const auto t_begin = std::chrono::steady_clock::now();
    struct Spent { Engine* e; std::chrono::steady_clock::time_point t0; ~Spent() { e->late_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } spent{ this, t_begin };
    std::vector<fmhip_vec> batch;
    {
        std::lock_guard<std::mutex> lock(late_mu_);
        const size_t waiting = late_.size() - late_pos_;
        if (at_most >= waiting) { if (late_pos_ == 0) batch.swap(late_); else { batch.assign(late_.begin() + (std::ptrdiff_t)late_pos_, late_.end()); late_.clear(); } late_pos_ = 0; }
        else { batch.assign(late_.begin() + (std::ptrdiff_t)late_pos_, late_.begin() + (std::ptrdiff_t)(late_pos_ + at_most)); late_pos_ += at_most; }
        late_count_.store(late_.size() - late_pos_, std::memory_order_release);
    }
    if (!initialized_) return;
    (at_most == ~size_t(0) ? n_late_at_once_ : n_late_waiting_) += (int64_t)batch.size();
    // The nodes of handles that died a while ago are COLD, and performing a release touches several lines of several nodes: the node's
    // own three, its neighbours on the deferred list (unlinked), its operands (their counts drop; whatever only the recipe kept alive goes
    // with it).  Two stages of prefetching ahead of the work: the node itself 16 handles ahead, what it points to 8 ahead (by then it is there).
    static const size_t AHEAD = [] { const char* e = std::getenv("FMHIP_DRAIN_PREFETCH"); return e ? (size_t)std::atoll(e) : (size_t)8; }();
    std::vector<Node*> nds(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) nds[i] = owner_of(batch[i]) == index_ ? nodes_.get(batch[i]) : nullptr;     // (a handle that is not one: nobody is left to tell)
    auto fetch = [](const void* p) { __builtin_prefetch(p, 1, 1); };
    for (size_t i = 0; i < batch.size(); ++i) {
        if (AHEAD) {
            if (i + 2 * AHEAD < batch.size()) if (const Node* far = nds[i + 2 * AHEAD]) { fetch(far); fetch((const char*)far + 64); fetch((const char*)far + 128); }
            if (i + AHEAD < batch.size()) if (const Node* near = nds[i + AHEAD]) {
                if (nodes_.get(batch[i + AHEAD]) == near && near->refs_ext == 1) {      // (still the handle's node: a handle released twice in one batch has lost it by now)                            // about to go: what its going touches
                    if (near->pend_prev) fetch(&near->pend_prev->pend_next);
                    if (near->pend_next) fetch(&near->pend_next->pend_prev);
                    if (!near->buf) for (int k = 0; k < near->n_in; ++k) if (near->in[k]) fetch(near->in[k]);
                }
            }
        }
        Node* nd = nds[i];
        if (!nd || nodes_.get(batch[i]) != nd) continue;             // (released twice in one batch: the second is not a handle any more)
        if (--nd->refs_ext == 0) { nodes_.erase(batch[i]); node_maybe_free(nd); }
    }
}

static void check_n(int64_t n) {
    if (n < 0 || n > (int64_t(1) << 31)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "invalid vector size " + std::to_string(n));
}

fmhip_vec Engine::create_uninitialized(int64_t n) {
    require_init(); check_n(n);
    Buffer* b = new_buffer(n);
    Node* nd = new_node(n);
    nd->buf = b;
    return nd->id;
}

fmhip_vec Engine::create_from_host(const void* src, bool is_double, int64_t n) {
    require_init(); check_n(n);
    if (n > 0 && !src) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null host pointer");
    Buffer* b = new_buffer(n);
    try {
        const int64_t chunk = int64_t(16) << 20;            // floats per staging round (64 MiB)
        for (int64_t off = 0; off < n; off += chunk) {
            const int64_t m = std::min(chunk, n - off);
            float* st = (float*)ensure_stage((size_t)m * 4);
            if (is_double) {                                // (float)arrayOfDouble[i], RandomVariableCuda.java:768-774
                const double* s = (const double*)src + off;
                for (int64_t i = 0; i < m; ++i) st[i] = (float)s[i];
            } else std::memcpy(st, (const float*)src + off, (size_t)m * 4);
            hip_check(hipMemcpyAsync(b->ptr + off, st, (size_t)m * 4, hipMemcpyHostToDevice, stream_), "H2D");
            hip_check(hipStreamSynchronize(stream_), "H2D sync");
        }
    } catch (...) { buffer_unref(b); throw; }
    Node* nd = new_node(n);
    nd->buf = b;
    return nd->id;
}

fmhip_vec Engine::create_filled(int64_t n, float v) {
    require_init(); check_n(n);
    Buffer* b = new_buffer(n);
    if (n > 0) {
        hipError_t e = launch_fill(b->ptr, v, (int64_t)(round_cap((size_t)n * 4) / 4), stream_);
        if (e != hipSuccess) { buffer_unref(b); hip_check(e, "fill"); }
        n_launches_++;
    }
    Node* nd = new_node(n);
    nd->buf = b;
    return nd->id;
}

void Engine::read(fmhip_vec h, void* dst, bool as_double, int64_t n) {
    require_init();
    end_step_group();
    Node* nd = node(h);
    if (n != nd->n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "read of " + std::to_string(n) + " elements from a vector of " + std::to_string(nd->n));
    if (n > 0 && !dst) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null host pointer");
    touch(nd);
    if (!nd->buf) materialize({nd});
    const int64_t chunk = int64_t(16) << 20;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        float* st = (float*)ensure_stage((size_t)m * 4);
        hip_check(hipMemcpyAsync(st, nd->buf->ptr + off, (size_t)m * 4, hipMemcpyDeviceToHost, stream_), "D2H");
        wait_for_stream("D2H sync");
        if (as_double) { double* d = (double*)dst + off; for (int64_t i = 0; i < m; ++i) d[i] = st[i]; }
        else std::memcpy((float*)dst + off, st, (size_t)m * 4);
    }
}

// Vectors are immutable but for two doors: a raw device pointer handed out, and fmhip_program_run_into.  Storage that several vectors share
// (rows of a batched launch that were identical and were computed once: run_peeled, merge_families) is copied before either opens.
void Engine::make_private(Node* nd) {
    Buffer* b = nd->buf;
    if (!b || b->refs <= 1 || b->parent || b->foreign >= 0) return;
    Buffer* own = new_buffer(nd->n);
    hipError_t e = hipMemcpyAsync(own->ptr, b->ptr, (size_t)nd->n * 4, hipMemcpyDeviceToDevice, stream_);
    if (e != hipSuccess) { buffer_unref(own); hip_check(e, "copy of a shared vector before it is written in place"); }
    nd->buf = own;
    buffer_unref(b);
}

void* Engine::device_ptr(fmhip_vec h) {
    require_init();
    end_step_group();
    Node* nd = node(h);
    touch(nd);
    if (!nd->buf) materialize({nd});
    // the caller may write through the pointer: whoever still reads this vector — pending expressions, recipes of deferred values — is computed first
    if (nd->refs_int > 0) { flush_all(); materialize_deferred(); }
    nd->has_moments = false; nd->moments_slot = nullptr; nd->moments_blocked = true;
    make_private(nd);
    return nd->buf->ptr;
}

// ---------------------------------------------------------------- program compiler (SSA → accumulator bytecode)
//
// Public opcode + "which operand is in the accumulator" → micro-op and the operand positions fetched from R.
struct UVariant { uint32_t uop; int r1_pos, r2_pos; };      // positions index {a,b,c}; -1 = unused
static bool variant_for(int opcode, int a_pos, UVariant* out) {
    auto set = [&](uint32_t u, int p1, int p2) { *out = { u, p1, p2 }; return true; };
    if (a_pos == 0) {
        switch (opcode) {
        case FMHIP_OP_CAP_S: return set(U_CAP_S, -1, -1);       case FMHIP_OP_FLOOR_S: return set(U_FLOOR_S, -1, -1);
        case FMHIP_OP_ADD_S: return set(U_ADD_S, -1, -1);       case FMHIP_OP_SUB_S: return set(U_SUB_S, -1, -1);
        case FMHIP_OP_BUS_S: return set(U_BUS_S, -1, -1);       case FMHIP_OP_MULT_S: return set(U_MULT_S, -1, -1);
        case FMHIP_OP_DIV_S: return set(U_DIV_S, -1, -1);       case FMHIP_OP_VID_S: return set(U_VID_S, -1, -1);
        case FMHIP_OP_POW_S: return set(U_POW_S, -1, -1);
        case FMHIP_OP_SQUARED: return set(U_SQUARED, -1, -1);   case FMHIP_OP_SQRT: return set(U_SQRT, -1, -1);
        case FMHIP_OP_EXP: return set(U_EXP, -1, -1);           case FMHIP_OP_LOG: return set(U_LOG, -1, -1);
        case FMHIP_OP_INVERT: return set(U_INVERT, -1, -1);     case FMHIP_OP_ABS: return set(U_ABS, -1, -1);
        case FMHIP_OP_SIN: return set(U_SIN, -1, -1);           case FMHIP_OP_COS: return set(U_COS, -1, -1);
        case FMHIP_OP_ISNAN: return set(U_ISNAN, -1, -1);
        case FMHIP_OP_CAP: return set(U_CAP, 1, -1);            case FMHIP_OP_FLOOR: return set(U_FLOOR, 1, -1);
        case FMHIP_OP_ADD: return set(U_ADD, 1, -1);            case FMHIP_OP_MULT: return set(U_MULT, 1, -1);
        case FMHIP_OP_SUB: return set(U_SUB, 1, -1);            case FMHIP_OP_DIV: return set(U_DIV, 1, -1);
        case FMHIP_OP_ACCRUE: return set(U_ACCRUE_A, 1, -1);    case FMHIP_OP_DISCOUNT: return set(U_DISCOUNT_A, 1, -1);
        case FMHIP_OP_ADDPRODUCT_VS: return set(U_ADDPRODUCT_VS_A, 1, -1);
        case FMHIP_OP_ADDPRODUCT: return set(U_ADDPRODUCT_A, 1, 2);
        case FMHIP_OP_ADDRATIO: return set(U_ADDRATIO_A, 1, 2); case FMHIP_OP_SUBRATIO: return set(U_SUBRATIO_A, 1, 2);
        case FMHIP_OP_CHOOSE: return set(U_CHOOSE_T, 1, 2);
        default: return false;
        }
    }
    if (a_pos == 1) {
        switch (opcode) {
        case FMHIP_OP_CAP: return set(U_CAP, 0, -1);            case FMHIP_OP_FLOOR: return set(U_FLOOR, 0, -1);   // min/max are symmetric
        case FMHIP_OP_ADD: return set(U_ADD, 0, -1);            case FMHIP_OP_MULT: return set(U_MULT, 0, -1);
        case FMHIP_OP_SUB: return set(U_BUS, 0, -1);            case FMHIP_OP_DIV: return set(U_VID, 0, -1);
        case FMHIP_OP_ACCRUE: return set(U_ACCRUE_B, 0, -1);    case FMHIP_OP_DISCOUNT: return set(U_DISCOUNT_B, 0, -1);
        case FMHIP_OP_ADDPRODUCT_VS: return set(U_ADDPRODUCT_VS_B, 0, -1);
        case FMHIP_OP_ADDPRODUCT: return set(U_ADDPRODUCT_B, 0, 2);
        case FMHIP_OP_CHOOSE: return set(U_CHOOSE_P, 0, 2);
        default: return false;
        }
    }
    if (a_pos == 2) {
        switch (opcode) {
        case FMHIP_OP_ADDPRODUCT: return set(U_ADDPRODUCT_B, 0, 1);     // a + c*b == a + b*c
        case FMHIP_OP_CHOOSE: return set(U_CHOOSE_N, 0, 1);
        default: return false;
        }
    }
    return false;
}

// Position (0..2) at which value `v` can be consumed from the accumulator by `op`, or -1.
static int acc_position(const SsaOp& op, int n_vec, int v) {
    if (v < 0) return -1;
    const int ids[3] = { op.a, op.b, op.c };
    UVariant uv;
    for (int p = 0; p < n_vec; ++p) if (ids[p] == v && variant_for(op.opcode, p, &uv)) return p;
    return -1;
}

Program* Engine::compile(const std::vector<SsaOp>& ops, int n_in, const std::vector<int>& outs, const std::vector<int>& reds,
                         std::vector<float>* scalars_out, bool fixed_scalars)
{
    // Prefer the 8-elements-per-lane kernel (8 registers); fall back to 4 elements / 16 registers when the program
    // keeps more values alive.
    // fixed_scalars == false: the lazy front-end patches scalars per row, so no value-dependent rewrites are allowed
    bool library_math = false;          // pow / sin / cos live only in the 4-element kernel (see kernels.hip)
    for (const SsaOp& o : ops) library_math |= (o.opcode == FMHIP_OP_POW_S || o.opcode == FMHIP_OP_SIN || o.opcode == FMHIP_OP_COS);
    if (!library_math) {
        try { return compile_variant(ops, n_in, outs, reds, scalars_out, 1, fixed_scalars); }
        catch (const Error& e) { if (e.code != FMHIP_ERR_PROGRAM_LIMIT) throw; }
    }
    return compile_variant(ops, n_in, outs, reds, scalars_out, 0, fixed_scalars);
}

Program* Engine::compile_variant(const std::vector<SsaOp>& ops, int n_in, const std::vector<int>& outs, const std::vector<int>& reds,
                                 std::vector<float>* scalars_out, int variant, bool fixed_scalars)
{
    const int nreg_alloc = FM_VARIANT_NREG[variant] - 1;        // the last register is the "no store" dummy
    const unsigned no_store = (unsigned)nreg_alloc;
    const int n_ops = (int)ops.size();
    if (n_in < 0 || n_in > FM_MAX_IN || n_in > nreg_alloc) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many inputs for one launch");
    if (n_ops > FM_MAX_OPS) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many ops for one launch");
    if ((int)outs.size() > FM_MAX_OUT) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many outputs for one launch");
    if ((int)reds.size() > FM_MAX_RED) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many fused reductions for one launch");
    if (n_in == 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a program needs at least one input vector");
    const int n_val = n_in + n_ops;
    std::vector<int> n_vec(n_ops);
    for (int i = 0; i < n_ops; ++i) {
        const OpInfo inf = op_info(ops[i].opcode);
        if (inf.n_vec == 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown opcode " + std::to_string(ops[i].opcode));
        n_vec[i] = inf.n_vec;
        const int v[3] = { ops[i].a, ops[i].b, ops[i].c };
        for (int k = 0; k < inf.n_vec; ++k)
            if (v[k] < 0 || v[k] >= n_in + i) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "op " + std::to_string(i) + ": bad operand id");
    }
    for (int v : outs) if (v < 0 || v >= n_val) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad output value id");
    for (int v : reds) if (v < 0 || v >= n_val) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad reduce value id");

    // Pass 1: where does each op take its accumulator operand from, and which values must live in R?
    //   The result of op i-1 flows through the accumulator into op i when op i can consume it there;
    //   every other use of a value (and every program output / reduction) reads it from R.
    std::vector<int> apos(n_ops, -1);               // operand position served by the accumulator (-1: needs U_LDA of position 0)
    std::vector<int> last_r_use(n_val, -1);         // last op index reading the value from R (n_ops = program end)
    std::vector<char> stored(n_val, 0);
    for (int k = 0; k < n_in; ++k) stored[k] = 1;
    for (int i = 0; i < n_ops; ++i) {
        const int prev = (i > 0) ? n_in + i - 1 : -1;
        apos[i] = acc_position(ops[i], n_vec[i], prev);
        const int v[3] = { ops[i].a, ops[i].b, ops[i].c };
        for (int p = 0; p < n_vec[i]; ++p) {
            if (p == apos[i]) continue;
            stored[v[p]] = 1;
            last_r_use[v[p]] = i;
        }
    }
    for (int v : outs) { stored[v] = 1; last_r_use[v] = n_ops; }
    for (int v : reds) { stored[v] = 1; last_r_use[v] = n_ops; }

    // Pass 2: emit micro-ops with register allocation (a register is released after the value's last read from R).
    std::vector<int> reg_of(n_val, -1);
    std::vector<int> free_regs;
    for (int r = nreg_alloc - 1; r >= n_in; --r) free_regs.push_back(r);     // lowest register on top
    for (int k = 0; k < n_in; ++k) reg_of[k] = k;
    auto give_back = [&](int r) { free_regs.push_back(r); std::sort(free_regs.begin(), free_regs.end(), std::greater<int>()); };
    for (int k = 0; k < n_in; ++k) if (last_r_use[k] < 0) give_back(k);

    Program* p = new Program();
    p->n_in = n_in; p->n_out = (int)outs.size(); p->n_red = (int)reds.size(); p->n_ops = n_ops;
    std::vector<float> scal;
    int n_uops = 0;
    auto emit = [&](uint32_t word) {
        if (n_uops >= FM_MAX_OPS) { delete p; throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many micro-ops for one launch"); }
        p->proto.ops[n_uops++].w = word;
    };
    for (int i = 0; i < n_ops; ++i) {
        const int v[3] = { ops[i].a, ops[i].b, ops[i].c };
        int ap = apos[i];
        if (ap < 0) {                               // accumulator does not hold an operand: load position 0
            emit(fm_pack_op(U_LDA, (unsigned)reg_of[v[0]], 0, no_store, 0));
            ap = 0;
        }
        UVariant uv{};
        if (!variant_for(ops[i].opcode, ap, &uv)) { delete p; throw Error(FMHIP_ERR_INVALID_ARGUMENT, "internal: no accumulator form"); }
        const unsigned r1 = uv.r1_pos >= 0 ? (unsigned)reg_of[v[uv.r1_pos]] : 0u;
        const unsigned r2 = uv.r2_pos >= 0 ? (unsigned)reg_of[v[uv.r2_pos]] : 0u;
        for (int q = 0; q < n_vec[i]; ++q) {        // values whose last read from R is this op free their register first
            bool dup = false;
            for (int j = 0; j < q; ++j) dup |= (v[j] == v[q]);
            if (!dup && last_r_use[v[q]] == i && reg_of[v[q]] >= 0) { give_back(reg_of[v[q]]); }
        }
        unsigned st = no_store;
        const int res = n_in + i;
        if (stored[res]) {
            if (free_regs.empty()) { delete p; throw Error(FMHIP_ERR_PROGRAM_LIMIT, "register budget of one launch exceeded"); }
            st = (unsigned)free_regs.back(); free_regs.pop_back();
            reg_of[res] = (int)st;
        }
        unsigned slot = 0;
        uint32_t uop = uv.uop;
        if (math_mode == FMHIP_MATH_FAST) { if (uop == U_EXP) uop = U_EXP_FAST; else if (uop == U_LOG) uop = U_LOG_FAST; }
        if (uop == U_LOG) p->proto.flags |= FM_ARGS_LOG_TABLE;
        if (op_info(ops[i].opcode).scalar) {
            if ((int)scal.size() >= FM_MAX_SCAL) { delete p; throw Error(FMHIP_ERR_PROGRAM_LIMIT, "too many scalar operands for one launch"); }
            slot = (unsigned)scal.size();
            float sv = (float)ops[i].scalar;        // "(float)value", RandomVariableCuda.java:521
            if (uop == U_DIV_S && fixed_scalars) {
                // a / (±2^k) == a * (±2^-k) bit for bit (both are the correctly rounded value of the same real number,
                // also for denormal results) as long as the reciprocal is itself a normal float: 1 multiply instead of
                // the 11-instruction IEEE division.  Only when the scalar is fixed at compile time (explicit programs).
                int ex = 0;
                const float mant = std::frexp(sv, &ex);
                if ((mant == 0.5f || mant == -0.5f) && ex > -124 && ex < 126) { uop = U_MULT_S; sv = 1.0f / sv; }
            }
            scal.push_back(sv);
        }
        emit(fm_pack_op(uop, r1, r2, st, slot));
    }
    for (size_t k = 0; k < outs.size(); ++k) p->proto.out_reg[k] = (uint32_t)reg_of[outs[k]];
    for (size_t k = 0; k < reds.size(); ++k) p->proto.red_reg[k] = (uint32_t)reg_of[reds[k]];
    if (scal.empty()) scal.push_back(0.0f);
    p->n_scal = (int)scal.size();
    p->proto.n_ops = (uint32_t)n_uops; p->proto.n_in = (uint32_t)n_in; p->proto.n_out = (uint32_t)outs.size();
    p->proto.n_red = (uint32_t)reds.size(); p->proto.n_scal = (uint32_t)p->n_scal;
    p->proto.variant = (uint32_t)variant;
    p->proto.row_words = (uint32_t)(n_in + (int)outs.size() + (int)reds.size() + (p->n_scal + 1) / 2);
    p->scalars = scal;
    if (scalars_out) *scalars_out = scal;
    return p;
}

// ---------------------------------------------------------------- launch

static const double JIT_HOT_WORK = [] { const char* e = std::getenv("FMHIP_JIT_HOT_WORK"); return e ? std::atof(e) : 2e10; }();    // element-ops on the interpreter before a lazy program is queued for specialisation

void Engine::launch(Program* p, int64_t n, const std::vector<RowSpec>& rows, fmhip_moments* host_moments, void* dev_moments)
{
    HostTimer timer(HostProfile::LAUNCH);
    const int batch = (int)rows.size();
    if (batch <= 0) return;
    if (batch > 65535) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "batch too large");
    const int n_red = p->n_red;
    {   // a row table that does not fit the pinned ring (FMHIP_RING_BYTES can be as small as 4 KB): as many rows per launch as fit
        const size_t row_bytes = (size_t)p->proto.row_words * 8;
        const size_t rows_fit = std::max<size_t>(1, (ring_cap_ - std::min<size_t>(ring_cap_, 256)) / row_bytes);
        if ((size_t)batch * p->proto.row_words > (size_t)FM_INLINE_WORDS && (size_t)batch > rows_fit) {
            for (size_t off = 0; off < (size_t)batch; off += rows_fit) {
                const size_t m = std::min(rows_fit, (size_t)batch - off);
                const std::vector<RowSpec> part(rows.begin() + (std::ptrdiff_t)off, rows.begin() + (std::ptrdiff_t)(off + m));
                launch(p, n, part, host_moments ? host_moments + off * n_red : nullptr, dev_moments ? (char*)dev_moments + off * n_red * 32 : nullptr);
            }
            return;
        }
    }
    if (n == 0) {       // nothing to compute; reductions of an empty vector as the twin's loops leave them (:288,:303,:325)
        if (host_moments) for (int i = 0; i < batch * n_red; ++i) host_moments[i] = { 0.0, 0.0, DBL_MAX, -DBL_MAX };
        if (dev_moments && n_red > 0) {
            std::vector<double> z((size_t)batch * n_red * 4);
            for (int i = 0; i < batch * n_red; ++i) { z[i * 4 + 0] = 0; z[i * 4 + 1] = 0; z[i * 4 + 2] = DBL_MAX; z[i * 4 + 3] = -DBL_MAX; }
            double* st = (double*)ensure_stage(z.size() * 8);
            std::memcpy(st, z.data(), z.size() * 8);
            hip_check(hipMemcpyAsync(dev_moments, st, z.size() * 8, hipMemcpyHostToDevice, stream_), "H2D");
            hip_check(hipStreamSynchronize(stream_), "sync");
        }
        return;
    }
    DevProgramArgs args = p->proto;
    // Tier selection.  Lazy programs are promoted once the interpreter has spent JIT_HOT_WORK element-ops on them
    // (≈10 ms of device time: a compilation costs ≈0.2 s of one background host core); explicit programs at creation.
    // A kernel that EXISTS (the user's code-object cache, the build-time pack) is used from the program's first launch: looked up once, on
    // this thread (two hashes of the generated source and a file).  Round 3's calibration ran 563 launches on the interpreter although
    // every one of those programs had its kernel in the pack.
    if (jit_mode == FMHIP_JIT_AUTO && !p->jit && !p->jit_probed) { p->jit_probed = true; p->jit = jit().request_cached(p->proto); }
    if (jit_mode != FMHIP_JIT_OFF && !p->jit) {
        p->interpreted_work += (double)n * batch * std::max(1, p->n_ops);       // (the stand-alone reduction has no ops: it counts as one)
        if (jit_mode == FMHIP_JIT_SYNC || p->interpreted_work >= JIT_HOT_WORK) p->jit = jit().request(p->proto, jit_mode == FMHIP_JIT_SYNC);
    } else if (jit_mode == FMHIP_JIT_SYNC && p->jit->state.load(std::memory_order_acquire) == JitSlot::QUEUED)
        p->jit = jit().request(p->proto, true);      // queued earlier in auto mode: finish it now
    const bool use_jit = jit_mode != FMHIP_JIT_OFF && p->jit && p->jit->state.load(std::memory_order_acquire) == JitSlot::READY;
    // the specialised kernel may process a different number of elements per lane and pass than the interpreter variant
    const int64_t elems_per_pass = (int64_t)FM_BLOCK * (use_jit ? p->jit->elems : FM_VARIANT_ELEMS[args.variant]);
    const int64_t tiles = (n + elems_per_pass - 1) / elems_per_pass;
    if (tiles > int64_t(0x7fffffff)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "vector too long");
    // Passes per workgroup.  A workgroup has a fixed cost (launch, log-table copy, for reductions the wave/LDS combine, the
    // partial write and the arrival counter) and, run back to back, kernels pay one workgroup lifetime of ramp-up and
    // tail: measured on the bench program (64 rows x 1M paths, benchmarks/jit_knobs.py) 2048 / 4096 / 8192 / 16384
    // elements per workgroup = 223 / 199 / 193 / 199 µs with fused reductions (181 / 179 / 177 / 186 µs without; there one
    // pass per workgroup is kept: on the LMM op stream, whose launches have 8 rows, 8192 gained 2 % and lost it again
    // together with the prefetch below).  With reductions the span must not depend on the batch: it fixes the order in
    // which a row's partial sums are added, and a value must not depend on how many other rows shared its launch.
    static const int64_t ELEMS_PER_BLOCK_ENV = [] { const char* e = std::getenv("FMHIP_ELEMS_PER_BLOCK"); const long long v = e ? std::atoll(e) : 0; return v >= 1024 ? (int64_t)v : (int64_t)0; }();
    // A program that evaluates log copies the 8 KB log table into LDS once per workgroup: one pass (2048 elements, 8 KB per
    // vector) per workgroup would double its L2 traffic (`log` alone: 88 → 95 µs), so it also gets ≈ 8192 elements per
    // workgroup — unless the launch is too small to fill the chip that way.
    const bool log_table = (args.flags & FM_ARGS_LOG_TABLE) != 0;
    const int64_t elems_per_block = ELEMS_PER_BLOCK_ENV ? ELEMS_PER_BLOCK_ENV : ((n_red > 0 || log_table) ? 8192 : 0);      // the variable: for benchmarks/jit_knobs.py
    int64_t passes_per_block = std::max<int64_t>(1, elems_per_block / elems_per_pass);
    if (n_red == 0) while (passes_per_block > 1 && ((tiles + passes_per_block - 1) / passes_per_block) * batch < 4096) passes_per_block /= 2;
    int64_t span_blocks = 1;
    if (n_red > 0) {
        // The reduction tree (fm_kernel_parts.hpp) is defined on units of 2048 elements and spans of four units — more for vectors of
        // more than 65536 spans, so that a row never has more workgroups than that.  A workgroup takes a whole span, or, where the
        // launch would leave most of the chip idle that way (one row of 1 M paths: 123 spans on 256 CUs), a single unit: four times
        // the workgroups, the same tree, the same moments.
        const int64_t unit_tiles = std::max<int64_t>(1, FM_UNIT_ELEMS / elems_per_pass);
        const int64_t units = (n + FM_UNIT_ELEMS - 1) / FM_UNIT_ELEMS;
        const int64_t span_units = ELEMS_PER_BLOCK_ENV ? std::max<int64_t>(1, ELEMS_PER_BLOCK_ENV / FM_UNIT_ELEMS) : std::max<int64_t>(FM_SPAN_UNITS, (units + 65535) / 65536);
        if (!ELEMS_PER_BLOCK_ENV && unit_launch(n, batch)) span_blocks = FM_SPAN_UNITS;
        passes_per_block = span_units / span_blocks * unit_tiles;
    } else if ((tiles + passes_per_block - 1) / passes_per_block > 65536) passes_per_block = (tiles + 65535) / 65536;
    const int64_t bpr = (tiles + passes_per_block - 1) / passes_per_block;
    args.block_tiles = (uint32_t)passes_per_block;
    args.span_blocks = (uint32_t)span_blocks;
    args.n = n;
    args.tiles_per_row = (uint32_t)tiles;
    const size_t rw = args.row_words;
    const size_t table_bytes = (size_t)batch * rw * 8;
    const uint64_t* dev_rows = nullptr;
    std::vector<uint64_t> table((size_t)batch * rw, 0);
    for (int b = 0; b < batch; ++b) {
        uint64_t* r = table.data() + (size_t)b * rw;
        const RowSpec& rs = rows[b];
        for (int k = 0; k < p->n_in; ++k) r[k] = (uint64_t)(uintptr_t)rs.in[k];
        for (int k = 0; k < p->n_out; ++k) r[p->n_in + k] = (uint64_t)(uintptr_t)rs.out[k];
        for (int k = 0; k < n_red; ++k) { const double s = rs.shifts ? rs.shifts[k] : 0.0; std::memcpy(&r[p->n_in + p->n_out + k], &s, 8); }
        float* sc = (float*)(r + p->n_in + p->n_out + n_red);
        const float* src = rs.scalars ? rs.scalars : p->scalars.data();
        for (int k = 0; k < p->n_scal; ++k) sc[k] = src[k];
    }
    // Small batches travel in the kernel arguments (≤ FM_INLINE_WORDS 8-byte words): the table upload is an in-stream copy
    // kernel of ≈ 5 µs (rocprofv3 on the LMM op stream: 1 579 of them, 6 % of the stream time, before this).
    if ((size_t)batch * rw <= (size_t)FM_INLINE_WORDS) { args.use_inline = 1; std::memcpy(args.inline_row, table.data(), (size_t)batch * rw * 8); }
    else {
        args.use_inline = 0;
        // A loop that runs the same program over the same vectors (run_into: time stepping with fixed buffers, bench.py)
        // re-creates the same table every time: if the copy uploaded last time is still in the ring, use it again and
        // save the in-stream H2D copy (≈ 5-8 µs of stream time per launch).
        if (p->last_dev_rows && p->last_ring_generation == ring_generation_ && p->last_table == table) dev_rows = p->last_dev_rows;
        else {
            HostTimer t2(HostProfile::ROW_UPLOAD);
            const size_t ring_off = ring_reserve(table_bytes);          // may wrap (and bump the generation)
            std::memcpy((char*)ring_host_ + ring_off, table.data(), table_bytes);
            hip_check(hipMemcpyAsync((char*)ring_dev_ + ring_off, (char*)ring_host_ + ring_off, table_bytes, hipMemcpyHostToDevice, stream_), "row table H2D");
            dev_rows = (const uint64_t*)((char*)ring_dev_ + ring_off);
            p->last_table.swap(table);
            p->last_dev_rows = dev_rows;
            p->last_ring_generation = ring_generation_;
        }
    }

    RedLaunch red;
    if (n_red > 0) red_begin(red, batch, n_red, (size_t)bpr, host_moments, dev_moments);
    args.results = (double*)red.results;
    args.counters = counters_dev_;
    args.done_flag = const_cast<uint64_t*>(red.poll_flag); args.done_value = red.done_value;
    void* const partials = red.partials;
    try {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (profiling_) {
            hip_check(hipEventCreate(&ev0), "hipEventCreate"); hip_check(hipEventCreate(&ev1), "hipEventCreate");
            hip_check(hipEventRecord(ev0, stream_), "hipEventRecord");
        }
        HostTimer t3(HostProfile::LAUNCH_API);
        const bool used_jit = use_jit;
        if (use_jit) {
            const uint64_t* rows_arg = dev_rows; double* partials_arg = (double*)partials;
            void* params[] = { &args, &rows_arg, &partials_arg };
            hip_check(hipModuleLaunchKernel(args.use_inline ? p->jit->fn_inline : p->jit->fn_table, (unsigned)bpr, (unsigned)batch, 1, FM_BLOCK, 1, 1,
                                            0, stream_, params, nullptr), "launch specialised kernel");
            n_jit_launches_++;
        } else
            hip_check(launch_program(args, dev_rows, (double*)partials, (uint32_t)bpr, (uint32_t)batch, stream_), "launch fm_program_kernel");
        if (profiling_) { hip_check(hipEventRecord(ev1, stream_), "hipEventRecord"); profile_events_.push_back({ ev0, ev1 });
                          profile_tags_.push_back({ p->n_ops, p->n_in, p->n_out, n_red, batch, used_jit ? 1 : 0, n }); }
        n_launches_++; n_ops_executed_ += (int64_t)p->n_ops * batch;
        algorithmic_bytes_ += 4 * n * (int64_t)(p->n_in + p->n_out) * batch;
        bytes_written_ += 4 * n * (int64_t)p->n_out * batch;
        if (!used_jit) n_interpreter_launches_++;
        if (n_red > 0) {                     // the final combine ran inside the same launch (last workgroup of each row)
            if (defer_red_ && !defer_red_->pending && host_moments && red.on_host) {
                red.pending = true; red.batch = batch; red.n_red = n_red; red.host = host_moments;
                *defer_red_ = red;              // reduce() waits and releases
                return;
            }
            red_wait(red, batch, n_red, host_moments);
        }
    } catch (...) { red_release(red); throw; }
    red_release(red);
}

// The buffers a launch with fused reductions needs, and where its moments go.  Results wanted on the host only: the last workgroup of
// a row stores its 32 bytes straight into the pinned staging buffer (host memory is mapped into the device's address space) — no
// device-to-host copy command between the kernel and the wait (a `chain.getAverage()` through the C++ mirror at 100 / 5 000 paths:
// 19.6 → 17.7 / 22.9 → 21.2 µs, benchmarks/small_n_latency.cpp).  One row, results wanted on the host: the kernel raises a flag in
// pinned memory behind the results and the host POLLS it instead of synchronising the stream.  A caller that values one product after
// the other (finmath-lib's calibration: 144 getAverage() per objective evaluation) pays the wake-up of hipStreamSynchronize and,
// measured, a launch that takes 20–25 µs instead of 5 right after it, once per product.
double* Engine::arena_alloc(size_t count)
{
    const size_t need = count * 32;
    if (need > ARENA_BYTES) return nullptr;
    if (arena_off_ + need > ARENA_BYTES) {                     // full: everything written so far is collected, then it starts again
        wait_for_stream("hipStreamSynchronize(moments arena)");
        arena_collect();
        arena_off_ = 0;
    }
    volatile uint64_t* p = reinterpret_cast<volatile uint64_t*>(moments_arena_ + arena_off_);
    for (size_t i = 0; i < count * 4; ++i) p[i] = MOMENTS_SENTINEL;
    arena_off_ += need;
    return reinterpret_cast<double*>(const_cast<uint64_t*>(p));
}

void Engine::arena_assign(Node* nd, double* slot)
{
    nd->has_moments = false;
    nd->moments_slot = reinterpret_cast<volatile uint64_t*>(slot);
    arena_outstanding_.push_back({ nd->id, nd->moments_slot });
}

void Engine::arena_collect()
{
    for (const auto& o : arena_outstanding_) {
        Node* nd = nodes_.get(o.first);
        if (!nd || nd->moments_slot != o.second) continue;       // gone, asked for already, or written into since
        bool arrived = true;
        uint64_t v[4];
        for (int c = 0; c < 4; ++c) { v[c] = o.second[c]; arrived &= v[c] != MOMENTS_SENTINEL; }
        nd->moments_slot = nullptr;
        if (arrived) { std::memcpy(nd->moments, v, 32); nd->has_moments = true; }
    }
    arena_outstanding_.clear();
    for (auto& kv : tickets_) {                                   // tickets that wait for slots: what they wait for has arrived
        MomentsTicket& t = kv.second;
        for (size_t i = 0; i < t.slots.size(); ++i)
            if (volatile uint64_t* slot = t.slots[i]) { uint64_t v[4] = { slot[0], slot[1], slot[2], slot[3] }; std::memcpy(&t.ready[i], v, 32); t.slots[i] = nullptr; }
    }
}

// Waits for everything queued on the stream — and, while it waits, performs releases that other threads have queued (drain_late): the
// device is asked whether it is done between portions instead of being slept on.
void Engine::wait_for_stream(const char* what)
{
    if (has_late()) {
        for (;;) {
            const hipError_t q = hipStreamQuery(stream_);
            if (q == hipSuccess) return;
            if (q != hipErrorNotReady) hip_check(q, what);
            (void)hipGetLastError();
            if (!has_late()) break;
            drain_late(late_portion());
        }
    }
    hip_check(hipStreamSynchronize(stream_), what);
}

bool Engine::slot_wait(Node* nd)
{
    volatile uint64_t* slot = nd->moments_slot;
    if (!slot) return false;
    auto complete = [&]() { return slot[0] != MOMENTS_SENTINEL && slot[1] != MOMENTS_SENTINEL && slot[2] != MOMENTS_SENTINEL && slot[3] != MOMENTS_SENTINEL; };
    const auto t0 = std::chrono::steady_clock::now();
    bool arrived = complete();
    for (uint32_t spins = 1; !arrived; ++spins) {
        if (has_late()) drain_late(late_portion());           // the device is being waited for: queued releases are performed meanwhile, a few per look
        else {
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        arrived = complete();
        if (!arrived && (spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!arrived) { wait_for_stream("moments sync"); arrived = complete(); }
    std::atomic_thread_fence(std::memory_order_acquire);
    nd->moments_slot = nullptr;
    if (!arrived) return false;                                  // (the launch never took them: a failed launch)
    uint64_t v[4] = { slot[0], slot[1], slot[2], slot[3] };
    std::memcpy(nd->moments, v, 32);
    nd->has_moments = true;
    return true;
}

// A launch with fused reductions over `batch` rows of n elements gives every workgroup one unit of the reduction tree (instead of a span
// of four) when it has few spans in all: such a launch has as many workgroups as one without reductions, so a chain over many vectors
// need not fear it.
bool Engine::unit_launch(int64_t n, int64_t batch) const
{
    const int64_t units = (n + FM_UNIT_ELEMS - 1) / FM_UNIT_ELEMS, spans = (units + FM_SPAN_UNITS - 1) / FM_SPAN_UNITS;
    return spans <= 65536 && spans * batch <= unit_workgroups_;
}

void Engine::red_begin(RedLaunch& red, int batch, int n_red, size_t blocks_per_row, fmhip_moments* host_moments, void* dev_moments)
{
    red = RedLaunch();
    red.dev_moments = dev_moments;
    red.on_host = host_moments && !dev_moments;
    red.partials = pool_.alloc((size_t)batch * n_red * (blocks_per_row + 8) * 32, &red.partials_cap);       // + FM_COMBINE_GROUP_SLOTS group partials per row
    static const bool POLL = [] { const char* e = std::getenv("FMHIP_POLL"); return !(e && e[0] == '0'); }();
    try {
        if (dev_moments) red.results = dev_moments;
        else if (red.on_host && POLL && batch == 1 && n_red <= 2 && !free_slots_.empty()) {       // a slot of its own: results [0, 64), flag at 64
            red.slot = free_slots_.back(); free_slots_.pop_back();
            red.results = result_slots_ + (size_t)red.slot * 128;
        }
        else if (red.on_host) red.results = ensure_stage((size_t)batch * n_red * 32);
        else red.results = pool_.alloc((size_t)batch * n_red * 32, &red.results_cap);
    } catch (...) { pool_.release(red.partials, red.partials_cap); red.partials = nullptr; throw; }
    if (red.slot >= 0) {
        red.poll_flag = reinterpret_cast<volatile uint64_t*>((char*)red.results + 64);
        *red.poll_flag = 0;
        red.done_value = ++poll_sequence_;
    }
}

bool Engine::red_poll(const RedLaunch& red)
{
    if (!red.poll_flag) return false;
    const auto t0 = std::chrono::steady_clock::now();
    bool arrived = false;
    for (uint32_t spins = 1; !(arrived = *red.poll_flag == red.done_value); ++spins) {
#if defined(__x86_64__)
        _mm_pause();
#endif
        if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;      // a long kernel: wait the ordinary way
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return arrived;
}

void Engine::red_complete(RedLaunch& red, bool arrived)
{
    if (!red.pending) return;
    red.pending = false;
    try {
        if (!arrived) wait_for_stream("moments sync");
        std::memcpy(red.host, red.results, (size_t)red.batch * red.n_red * 32);
    } catch (...) { red_release(red); throw; }
    red_release(red);
}

void Engine::red_wait(RedLaunch& red, int batch, int n_red, fmhip_moments* host_moments)
{
    if (!host_moments) return;
    const size_t bytes = (size_t)batch * n_red * 32;
    void* src = red.results;
    if (!red.on_host) { src = ensure_stage(bytes); hip_check(hipMemcpyAsync(src, red.results, bytes, hipMemcpyDeviceToHost, stream_), "moments D2H"); }
    bool arrived = false;
    if (red.poll_flag && has_late()) {                      // (as red_poll, with queued releases performed between the looks)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 1; !(arrived = *red.poll_flag == red.done_value); ++spins) {
            if (has_late()) drain_late(late_portion());
            if ((spins & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!arrived && !red_poll(red)) wait_for_stream("moments sync");
    std::memcpy(host_moments, src, bytes);
}

void Engine::red_release(RedLaunch& red)
{
    if (red.partials) pool_.release(red.partials, red.partials_cap);
    if (red.results && !red.dev_moments && !red.on_host) pool_.release(red.results, red.results_cap);
    if (red.slot >= 0) { free_slots_.push_back(red.slot); red.slot = -1; }
    red.partials = nullptr; red.results = nullptr; red.poll_flag = nullptr;
}

// ---------------------------------------------------------------- lazy front-end

static int FUSION_MAX_WEIGHT = 40;    // pending ops below one node before it is executed on its own accord.  Measured on the LMM
                                       // calibration (same box, FMHIP_FUSION_MAX_WEIGHT=40 vs 1000): 6.3 s vs 6.8-7.6 s although the
                                       // larger value needs 3x fewer launches — executing early overlaps device work with the
                                       // recording of the next methods, and that path is host-bound

void fusion_max_weight_override(int v) { FUSION_MAX_WEIGHT = v; }
static const size_t SPECULATE_PENDING = [] { const char* e = std::getenv("FMHIP_SPECULATE_PENDING"); return e ? (size_t)std::atoll(e) : (size_t)2000; }();   // operations recorded since the last time step; 0 = off.  5000 until the second half of round 5, when the device was what the hint-free calibration waited for; with the
                                                         // swaptions of an exercise date merged (merge_families) it is the host, and what counts is how little is left to run when the caller asks for its first expectation: 1500 / 2000 / 2500 / 3000 / 4000 / 5000
                                                         // methods: 3.48 / 3.37–3.45 / 3.46–3.55 / 3.56 / 3.63 / 3.66–3.70 s on one box (profiles/round05b_merged_chains.txt)
static const size_t SPECULATE_IDLE_MIN = [] { const char* e = std::getenv("FMHIP_SPECULATE_IDLE_MIN"); return e ? (size_t)std::atoll(e) : (size_t)0; }();   // 0 (default) = the device's idleness is not looked at.  Measured, lmm_hip --finmath-like at 1 M paths on one box: off 4.68 s; 512 / 1024 / 2048 / 4096: 5.26 / 4.89 / 4.96 / 4.90 s —
                                                         // chains cut wherever the device happens to run dry are shapes that never repeat (70 kernels compiled instead of 38, 4–12 k launches on the interpreter)
static const size_t FUSION_SOFT_CAP = 32768;     // pending operations at which a SOFT hold (fmhip_fusion_hold(2)) executes everything
// Experiment knob (off): execute everything once this many operations are pending anywhere, instead of the per-handle weight rule.
// On the hint-free LMM calibration (lmm_hip --finmath-like) 1000 … 16000 gave 15-18 ms per evaluation against 22.8 with the
// weight rule and 12.5 with BrownianMotionHip's time-step grouping: cut points that do not coincide with time steps give graph
// shapes that never repeat, so every flush plans from scratch and nothing rolls.
static size_t FUSION_MAX_PENDING = [] { const char* e = std::getenv("FMHIP_FUSION_MAX_PENDING"); return e ? (size_t)std::atoll(e) : (size_t)0; }();   // 0 = off

fmhip_vec Engine::call(int opcode, int n_in, const fmhip_vec* in, double scalar, bool has_scalar) {
    HostTimer timer(HostProfile::CALL);
    require_init();
    const OpInfo inf = op_info(opcode);
    if (inf.n_vec == 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "unknown opcode " + std::to_string(opcode));
    if (inf.n_vec != n_in || inf.scalar != has_scalar)
        throw Error(FMHIP_ERR_INVALID_ARGUMENT, "opcode " + std::to_string(opcode) + " does not match this call shape");
    Node* ins[3] = { nullptr, nullptr, nullptr };
    for (int i = 0; i < n_in; ++i) {
        ins[i] = node(in[i]);
        if (ins[i]->discarded && !ins[i]->buf) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "the value of an operand does not exist: it was given up (fmhip_vec_give_up_values), or lost in a launch that failed");
    }
    if (!replicas_.empty())                                     // an operation on the root of a copy that exists as a description only: the copy becomes an expression first
        for (int i = 0; i < n_in; ++i) if (!ins[i]->buf && ins[i]->rep_copy) if (ReplicaGroup* g = replica_of(ins[i])) expand_replicas(g);
    for (int i = 1; i < n_in; ++i)
        if (ins[i]->n != ins[0]->n)
            throw Error(FMHIP_ERR_SIZE_MISMATCH, "operand sizes differ: " + std::to_string(ins[0]->n) + " vs " + std::to_string(ins[i]->n));
    // the caller uses these handles again: what the escape policy has learnt about their positions; a value that was left unstored is
    // computed from its recipe now and stored (once)
    for (int i = 0; i < n_in; ++i) if (ins[i]->watch || ins[i]->deferred) demand(ins[i]);
    if (fusion && group_steps > 0 && fusion_hold == 0)         // a caller's own hold (or eager mode) leaves nothing to group on its behalf
        for (int i = 0; i < n_in; ++i)
            if (ins[i]->bm_id && (ins[i]->bm_step != group_last_step_ || ins[i]->bm_id != group_bm_id_)) step_boundary(ins[i]);
    Node* nd = new_node(ins[0]->n);
    nd->opcode = opcode; nd->n_in = n_in; nd->scalar = scalar;
    int w = 1;
    for (int i = 0; i < n_in; ++i) { nd->in[i] = ins[i]; ins[i]->refs_int++; w += ins[i]->buf ? 0 : ins[i]->weight; }
    nd->weight = w;
    pend_insert(nd);
    ++ops_since_boundary_;
    const bool held = fusion_hold != 0 || group_hold_;
    if (!fusion || (w > FUSION_MAX_WEIGHT && !held)) {
        try { materialize({nd}); }
        catch (...) { nd->refs_ext = 0; nodes_.erase(nd->id); node_maybe_free(nd); throw; }
    } else if (SPECULATE_PENDING && group_hold_ && fusion_hold == 0 &&
               (ops_since_boundary_ > SPECULATE_PENDING ||
                // … or much sooner when the device has NOTHING to do (asked every 128 methods, from SPECULATE_IDLE_MIN pending ones on): the
                // caller records the payoffs behind a simulation whose launches have drained — whatever is complete enough to run keeps the
                // device busy while the recording goes on.  The count above is the cut for a device that is still busy: larger launches.
                (SPECULATE_IDLE_MIN && ops_since_boundary_ >= SPECULATE_IDLE_MIN && (ops_since_boundary_ & 127) == 0 && hipStreamQuery(stream_) == hipSuccess))) {
        // The engine's own hold (time steps being grouped), no new time step for thousands of operations, and a caller that keeps
        // recording without asking for anything (the payoffs of its products, behind the simulation): what is pending runs now, without
        // waiting for it, and the launches take the moments of their roots along — when the caller comes to ask for expectations
        // (Engine::reduce), the device has been at work while the caller was recording.
        const fmhip_vec id = nd->id;
        ops_since_boundary_ = 0;
        struct Mode { Engine* e; ~Mode() { e->want_root_moments_ = false; e->async_moments_ = false; } } mode{ this };
        want_root_moments_ = true; async_moments_ = true;
        try { flush_all(); }
        catch (...) {
            if (nodes_.get(id) == nd) { nd->refs_ext = 0; nodes_.erase(id); node_maybe_free(nd); }
            throw;
        }
    } else if ((FUSION_MAX_PENDING && !held && n_pending_ > FUSION_MAX_PENDING) || ((fusion_hold == 2 || (group_hold_ && fusion_hold == 0)) && n_pending_ > FUSION_SOFT_CAP)) {
        const fmhip_vec id = nd->id;
        try { flush_all(); }
        catch (...) {                       // the caller never receives this handle: take the node (and whatever still hangs below it) back
            if (nodes_.get(id) == nd) { nd->refs_ext = 0; nodes_.erase(id); node_maybe_free(nd); }
            throw;
        }
        return id;
    }
    return nd->id;
}

// The first use of a Brownian increment with a new time index: a time step of the caller's discretisation scheme begins (what
// BrownianMotionHip did for itself in round 2, now for every caller of fmhip_bm_generate's vectors, whatever class wraps them).
// Index 0, or an index below the last one, starts a new simulation; the last index ends the grouping — what follows the
// simulation is not the scheme's to group.
void Engine::step_boundary(const Node* inc) {
    // Per generation: the time index it was last used with.  A generation that goes BACK (or starts at index 0) begins a new simulation;
    // another generation at the time index just seen is the second Brownian motion of a hybrid model inside the same time step — not a
    // boundary; and two simulations that take turns (two threads, two models) do not reset each other's count any more: until round 4 the
    // state was one (generation, index) pair, every change of generation was a "restart", and such callers never reached a flush of
    // their own (results unaffected; the work ran at the 32768-operation cap instead, in shapes that never repeat).
    if (group_last_by_bm_.size() > 64) group_last_by_bm_.clear();
    auto known = group_last_by_bm_.find(inc->bm_id);
    const int32_t last_of_this = known == group_last_by_bm_.end() ? -1 : known->second;
    if (inc->bm_step == last_of_this) { group_bm_id_ = inc->bm_id; group_last_step_ = inc->bm_step; return; }      // this generation's current time step again (the generations of a hybrid model take turns)
    const bool restart = inc->bm_step == 0 || inc->bm_step < last_of_this;
    const bool same_time_index = !restart && inc->bm_id != group_bm_id_ && inc->bm_step == group_last_step_;
    group_last_by_bm_[inc->bm_id] = inc->bm_step;
    group_bm_id_ = inc->bm_id; group_last_step_ = inc->bm_step;
    if (same_time_index) return;
    ops_since_boundary_ = 0;
    if (restart) group_steps_pending_ = 0;
    if (group_steps_pending_ == 0) group_hold_ = true;
    if (++group_steps_pending_ > group_steps) { flush_all(); group_steps_pending_ = 1; }
    if (inc->bm_step == inc->bm_steps - 1) { group_steps_pending_ = 0; group_last_step_ = -1; group_bm_id_ = 0; group_hold_ = false; group_last_by_bm_.erase(inc->bm_id); }
}

// ---------------------------------------------------------------- replication of pending graphs (fmhip_graph_clone)

void Engine::collect_pending(const fmhip_vec* roots, int n_roots, std::vector<Node*>& graph) {
    if (n_roots <= 0 || !roots) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "no roots");
    if (!replicas_.empty())                                     // the root of a copy that exists as a description only: it becomes an expression first
        for (int r = 0; r < n_roots; ++r) { Node* root = node(roots[r]); if (!root->buf && root->rep_copy) if (ReplicaGroup* g = replica_of(root)) expand_replicas(g); }
    const uint64_t ep = ++epoch_;
    std::vector<Node*> stack;
    for (int r = 0; r < n_roots; ++r) {
        Node* root = node(roots[r]);
        if (root->buf || root->mark == ep) continue;
        root->mark = ep; stack.push_back(root);
        while (!stack.empty()) {
            Node* nd = stack.back(); stack.pop_back();
            graph.push_back(nd);
            for (int k = 0; k < nd->n_in; ++k) { Node* c = nd->in[k]; if (!c->buf && c->mark != ep) { c->mark = ep; stack.push_back(c); } }
        }
    }
    // recording order = ascending id, and a topological order: an operand exists before the operation that uses it
    std::sort(graph.begin(), graph.end(), [](const Node* a, const Node* b) { return a->id < b->id; });
}

int Engine::graph_scalars(const fmhip_vec* roots, int n_roots, double* out, int capacity) {
    require_init();
    std::vector<Node*> graph;
    collect_pending(roots, n_roots, graph);
    int count = 0;
    for (Node* nd : graph) if (op_info(nd->opcode).scalar) { if (out && count < capacity) out[count] = nd->scalar; ++count; }
    return count;
}

// The copies of a pending graph as ordinary nodes: copy j of the operations `graph` (recording order, marked ep_graph with tmp_id = position;
// substituted operands marked ep_leaf with tmp_id = i).  root_node(j, i) != nullptr: the node that is to carry operation i of copy j
// (a root that exists already: expand_replicas); otherwise a fresh node without a handle.  ids: id_of(j, i).
template <class RootNode, class IdOf, class ScalarOf, class LeafTo>
static void clone_nodes_impl(std::vector<Node*>& graph, uint64_t ep_graph, uint64_t ep_leaf, int n_copies, RootNode root_node, IdOf id_of, ScalarOf scalar_of, LeafTo leaf_to,
                             std::vector<Node*>& node_pool, std::vector<std::vector<Node*>>& copies, const std::function<void(Node*)>& pend_insert)
{
    copies.assign((size_t)n_copies, std::vector<Node*>(graph.size(), nullptr));
    for (int j = 0; j < n_copies; ++j) {
        std::vector<Node*>& copy = copies[(size_t)j];
        for (size_t i = 0; i < graph.size(); ++i) {
            const Node* src = graph[i];
            Node* nd = root_node(j, i);
            const bool fresh = nd == nullptr;
            if (fresh) {                                             // like new_node, but without a handle: inner values of a copy have none
                if (!node_pool.empty()) { nd = node_pool.back(); node_pool.pop_back(); *nd = Node(); } else nd = new Node();
                nd->id = id_of(j, i); nd->n = src->n;
            }
            nd->opcode = src->opcode; nd->n_in = src->n_in; nd->weight = src->weight;
            nd->scalar = scalar_of(j, i, src);
            for (int k = 0; k < src->n_in; ++k) {
                Node* c = src->in[k];
                Node* m = c->mark == ep_graph ? copy[(size_t)c->tmp_id] : (c->mark == ep_leaf ? leaf_to(j, c->tmp_id) : c);
                nd->in[k] = m; m->refs_int++;
            }
            if (fresh) pend_insert(nd);
            copy[i] = nd;
        }
    }
}

static const bool REPLICAS = [] { const char* e = std::getenv("FMHIP_REPLICAS"); return !(e && e[0] == '0'); }();     // =0: copies are always made of nodes (A/B measurement)

void Engine::graph_clone(const fmhip_vec* roots, int n_roots, int n_copies, const fmhip_vec* leaf_from, const fmhip_vec* leaf_to, int n_map,
                         const double* scalars, int n_scalars, fmhip_vec* out) {
    HostTimer timer(HostProfile::CLONE);
    require_init();
    if (n_copies < 0 || n_map < 0 || !out || (n_map > 0 && (!leaf_from || !leaf_to))) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad graph replication request");
    std::vector<Node*> graph;
    collect_pending(roots, n_roots, graph);
    const uint64_t ep_graph = ++epoch_;
    int count_scalars = 0;
    for (size_t i = 0; i < graph.size(); ++i) { graph[i]->mark = ep_graph; graph[i]->tmp_id = (int)i; count_scalars += op_info(graph[i]->opcode).scalar ? 1 : 0; }
    if (scalars && n_scalars != count_scalars)
        throw Error(FMHIP_ERR_INVALID_ARGUMENT, "the graph has " + std::to_string(count_scalars) + " scalar operands, the caller supplied " + std::to_string(n_scalars));
    const uint64_t ep_leaf = ++epoch_;
    std::vector<Node*> from((size_t)n_map);
    for (int i = 0; i < n_map; ++i) {
        Node* l = node(leaf_from[i]);
        touch(l);
        if (l->mark == ep_graph) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a substituted operand lies inside the graph to be replicated");
        l->mark = ep_leaf; l->tmp_id = i;
        from[(size_t)i] = l;
    }
    std::vector<Node*> root_nodes((size_t)n_roots);
    for (int r = 0; r < n_roots; ++r) { root_nodes[(size_t)r] = node(roots[r]); touch(root_nodes[(size_t)r]); }
    std::vector<Node*> to((size_t)n_map * (size_t)n_copies);
    for (int j = 0; j < n_copies; ++j)
        for (int i = 0; i < n_map; ++i) {
            Node* t = node(leaf_to[(size_t)j * n_map + i]);
            if (t->deferred) demand(t); else touch(t);
            if (t->n != from[(size_t)i]->n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "a substituted operand differs in size");
            to[(size_t)j * n_map + i] = t;
        }
    auto shared_root = [&](Node* root, int j) {                      // a root that is a vector already: the copy shares it (or its substitute)
        Node* c = root->mark == ep_leaf && n_map > 0 ? to[(size_t)j * n_map + root->tmp_id] : root;
        c->refs_ext++;
        if (c->refs_ext == 1 && !nodes_.get(c->id)) nodes_.put(c->id, c);
        return c->id;
    };
    // The copies as a DESCRIPTION (ReplicaGroup): only when every operand a copy would read exists as a vector already and no part of
    // the graph or of the operand map belongs to another live description.
    bool describe = REPLICAS && n_copies > 0 && !graph.empty();
    for (size_t i = 0; describe && i < graph.size(); ++i) describe = graph[i]->rep_id == 0 || replica_of(graph[i]) == nullptr;
    for (size_t i = 0; describe && i < from.size(); ++i) describe = from[i]->buf != nullptr && (from[i]->leaf_rep_id == 0 || replicas_.find(from[i]->leaf_rep_id) == replicas_.end());
    for (size_t i = 0; describe && i < to.size(); ++i) describe = to[i]->buf != nullptr;
    if (!describe) {
        std::vector<std::vector<Node*>> copies;
        clone_nodes_impl(graph, ep_graph, ep_leaf, n_copies,
                         [](int, size_t) -> Node* { return nullptr; }, [&](int, size_t) { return next_id_++; },
                         [&](int j, size_t, const Node* src) { return scalars && op_info(src->opcode).scalar ? 0.0 : src->scalar; },      // patched below (recording order)
                         [&](int j, int i) { return to[(size_t)j * n_map + i]; }, node_pool_, copies, [this](Node* nd) { pend_insert(nd); });
        for (int j = 0; j < n_copies; ++j) {
            if (scalars) { int k = 0; for (size_t i = 0; i < graph.size(); ++i) if (op_info(graph[i]->opcode).scalar) copies[(size_t)j][i]->scalar = scalars[(size_t)j * n_scalars + k++]; }
            for (int r = 0; r < n_roots; ++r) {
                Node* root = root_nodes[(size_t)r];
                if (root->mark != ep_graph) { out[(size_t)j * n_roots + r] = shared_root(root, j); continue; }
                Node* c = copies[(size_t)j][(size_t)root->tmp_id];
                c->refs_ext++;
                if (c->refs_ext == 1 && !nodes_.get(c->id)) nodes_.put(c->id, c);
                out[(size_t)j * n_roots + r] = c->id;
            }
            // copies of values nobody holds and nothing uses cannot exist: every graph node is below a root
        }
        return;
    }
    ReplicaGroup* g = new ReplicaGroup();
    g->id = next_replica_id_++;
    g->n_copies = n_copies; g->n_roots = n_roots; g->n_scalars = scalars ? n_scalars : 0;
    g->graph_size = (int)graph.size();
    g->id_base = next_id_; next_id_ += (int64_t)graph.size() * n_copies;
    g->scalar_slot.assign(graph.size(), -1);
    int slot = 0;
    for (size_t i = 0; i < graph.size(); ++i) {
        Node* nd = graph[i];
        nd->rep_id = g->id; nd->rep_index = (int32_t)i; nd->rep_root = -1; nd->rep_copy = 0;
        if (op_info(nd->opcode).scalar) g->scalar_slot[i] = slot++;
    }
    if (scalars) g->scalars.assign(scalars, scalars + (size_t)n_copies * n_scalars);
    g->leaf_from = from; g->leaf_to = to;
    for (size_t i = 0; i < from.size(); ++i) { from[i]->refs_int++; from[i]->leaf_rep_id = g->id; from[i]->leaf_rep_index = (int32_t)i; }
    for (Node* t : to) t->refs_int++;
    g->roots.assign((size_t)n_roots, nullptr);
    g->root_done.assign((size_t)n_roots, 1);
    g->copy_roots.assign((size_t)n_copies * n_roots, nullptr);
    for (int r = 0; r < n_roots; ++r) {
        Node* root = root_nodes[(size_t)r];
        if (root->mark != ep_graph) { for (int j = 0; j < n_copies; ++j) out[(size_t)j * n_roots + r] = shared_root(root, j); continue; }
        if (root->rep_root >= 0) {                                   // listed twice: the same copies again
            for (int j = 0; j < n_copies; ++j) { Node* c = g->copy_roots[(size_t)j * n_roots + root->rep_root]; c->refs_ext++; out[(size_t)j * n_roots + r] = c->id; }
            continue;
        }
        root->rep_root = r;
        root->refs_ext++;                                            // the group's hold on the original: the graph stays whole until it has run with its copies (or been expanded)
        g->roots[(size_t)r] = root; g->root_done[(size_t)r] = 0; g->remaining++;
        for (int j = 0; j < n_copies; ++j) {
            Node* c;
            if (!node_pool_.empty()) { c = node_pool_.back(); node_pool_.pop_back(); *c = Node(); } else c = new Node();
            c->id = g->id_base + (int64_t)j * g->graph_size + root->rep_index;
            c->n = root->n; c->refs_ext = 1; c->refs_int = 1;        // refs_int: the group's hold
            c->rep_id = g->id; c->rep_copy = (uint32_t)j + 1u; c->rep_root = r; c->rep_index = root->rep_index;
            nodes_.put(c->id, c);
            pend_insert(c);
            g->copy_roots[(size_t)j * n_roots + r] = c;
            out[(size_t)j * n_roots + r] = c->id;
        }
    }
    replicas_[g->id] = g;
    if (g->remaining == 0) destroy_replica_group(g);                 // (cannot happen: a non-empty graph lies below a pending root)
}

// Drops a group's holds on its operands and forgets it (its roots have all been executed, or expanded).
void Engine::destroy_replica_group(ReplicaGroup* g) {
    replicas_.erase(g->id);
    for (Node* l : g->leaf_from) { if (l->leaf_rep_id == g->id) { l->leaf_rep_id = 0; l->leaf_rep_index = -1; } node_unref_int(l); }
    for (Node* t : g->leaf_to) node_unref_int(t);
    delete g;
}

// The roots `done` (indices) have been executed together with all their copies: the copies' root nodes have their buffers.
void Engine::replica_roots_done(ReplicaGroup* g, const std::vector<int>& done) {
    for (int r : done) {
        if (g->root_done[(size_t)r]) continue;
        g->root_done[(size_t)r] = 1; g->remaining--;
        for (int j = 0; j < g->n_copies; ++j) {
            Node* c = g->copy_roots[(size_t)j * g->n_roots + r];
            c->rep_id = 0; c->rep_copy = 0; c->rep_root = -1; c->rep_index = -1;
            node_unref_int(c);
        }
        Node* root = g->roots[(size_t)r];
        root->rep_id = 0; root->rep_root = -1; root->rep_index = -1;
        if (--root->refs_ext == 0) { nodes_.erase(root->id); node_maybe_free(root); }
    }
    if (g->remaining == 0) destroy_replica_group(g);
}

// A launch sequence that was to execute these roots with their copies has FAILED midway (a device allocation, a HIP error): some
// parts or segments have committed buffers, the rest has not run.  A root whose ORIGINAL has its buffer is closed here like an executed
// one — its holds go, so the group can end and nothing leaks — and every copy of it that did NOT get its buffer is marked as lost:
// reading it is an error from now on, never a silently wrong value (a later expansion of the description skips a materialised original
// and would wire the copies' dependants to the original's vector).  Roots that have not run at all stay described, as before.
void Engine::replicas_after_failure(const std::vector<std::pair<ReplicaGroup*, std::vector<int>>>& done) {
    for (const auto& kv : done) {
        ReplicaGroup* g = kv.first;
        bool alive = false;
        for (const auto& r : replicas_) alive |= r.second == g;
        if (!alive) continue;
        std::vector<int> closed;
        for (int r : kv.second) {
            if (r < 0 || r >= g->n_roots || g->root_done[(size_t)r] || !g->roots[(size_t)r] || !g->roots[(size_t)r]->buf) continue;
            for (int j = 0; j < g->n_copies; ++j) { Node* c = g->copy_roots[(size_t)j * g->n_roots + (size_t)r]; if (c && !c->buf) c->discarded = true; }
            closed.push_back(r);
        }
        if (!closed.empty()) replica_roots_done(g, closed);
    }
}

// Fallback: the description becomes ordinary pending nodes (what graph_clone made before descriptions existed), for the roots that
// have not been executed yet.  The copies' root nodes — the handles are out — receive the root operations; every other operation of
// a copy becomes a fresh node with the id reserved for it.
void Engine::expand_replicas(ReplicaGroup* g) {
    std::vector<Node*> graph;
    {
        const uint64_t ep = ++epoch_;
        std::vector<Node*> stack;
        for (int r = 0; r < g->n_roots; ++r) {
            Node* root = g->roots[(size_t)r];
            if (!root || g->root_done[(size_t)r] || root->buf || root->mark == ep) continue;
            root->mark = ep; stack.push_back(root);
            while (!stack.empty()) {
                Node* nd = stack.back(); stack.pop_back();
                graph.push_back(nd);
                for (int k = 0; k < nd->n_in; ++k) { Node* c = nd->in[k]; if (!c->buf && c->mark != ep) { c->mark = ep; stack.push_back(c); } }
            }
        }
        std::sort(graph.begin(), graph.end(), [](const Node* a, const Node* b) { return a->id < b->id; });
    }
    const uint64_t ep_graph = ++epoch_;
    for (size_t i = 0; i < graph.size(); ++i) { graph[i]->mark = ep_graph; graph[i]->tmp_id = (int)i; }
    const uint64_t ep_leaf = ++epoch_;
    const int n_map = (int)g->leaf_from.size();
    for (int i = 0; i < n_map; ++i) { g->leaf_from[(size_t)i]->mark = ep_leaf; g->leaf_from[(size_t)i]->tmp_id = i; }
    std::vector<std::vector<Node*>> copies;
    clone_nodes_impl(graph, ep_graph, ep_leaf, g->n_copies,
                     [&](int j, size_t i) -> Node* { const Node* src = graph[i]; return src->rep_root >= 0 ? g->copy_roots[(size_t)j * g->n_roots + src->rep_root] : nullptr; },
                     [&](int j, size_t i) { return g->id_base + (int64_t)j * g->graph_size + graph[i]->rep_index; },
                     [&](int j, size_t i, const Node* src) { const int32_t sl = g->scalar_slot[(size_t)src->rep_index];
                                                             return (sl >= 0 && !g->scalars.empty()) ? g->scalars[(size_t)j * g->n_scalars + sl] : src->scalar; },
                     [&](int j, int i) { return g->leaf_to[(size_t)j * n_map + i]; }, node_pool_, copies, [this](Node* nd) { pend_insert(nd); });
    for (Node* nd : graph) { nd->rep_id = 0; nd->rep_index = -1; }   // (rep_root is cleared with the roots below)
    std::vector<int> all;
    for (int r = 0; r < g->n_roots; ++r) if (g->roots[(size_t)r] && !g->root_done[(size_t)r]) all.push_back(r);
    replica_roots_done(g, all);                                       // drops the holds; the copies' roots are ordinary pending expressions now
}

// Every live description the pending graph below `targets` touches is expanded: for the paths that execute a single expression
// (read, reduce, device_ptr, a chain passing the size threshold) instead of flushing everything.
void Engine::expand_replicas_below(const std::vector<Node*>& targets) {
    if (replicas_.empty()) return;
    for (bool again = true; again;) {
        again = false;
        const uint64_t ep = ++epoch_;
        std::vector<Node*> stack;
        for (Node* t : targets) { if (!t->buf && t->mark != ep) { t->mark = ep; stack.push_back(t); } }
        while (!stack.empty() && !again) {
            Node* nd = stack.back(); stack.pop_back();
            if (nd->rep_id) { if (ReplicaGroup* g = replica_of(nd)) { expand_replicas(g); again = true; break; } }
            for (int k = 0; k < nd->n_in; ++k) { Node* c = nd->in[k]; if (!c->buf && c->mark != ep) { c->mark = ep; stack.push_back(c); } }
        }
    }
}

// ---------------------------------------------------------------- escape policy (runtime.hpp: policy_state_)

static const bool ESCAPE_POLICY = [] { const char* e = std::getenv("FMHIP_ESCAPE_POLICY"); return !(e && e[0] == '0'); }();     // =0: whatever has a handle is stored (rounds 1–4; A/B measurement)

void Engine::policy_reset() { policy_state_.clear(); ++policy_gen_; }       // (shapes bind again, lazily: their generation no longer matches)

void Engine::policy_bind(ShapePolicy& p, size_t size) {
    if (p.gen != policy_gen_ || p.size != size) {
        if (policy_state_.size() + size > (size_t(1) << 28)) policy_reset();                // shapes that never repeat: start again
        p.base = (uint32_t)policy_state_.size(); p.size = (uint32_t)size; p.gen = policy_gen_;
        policy_state_.resize(policy_state_.size() + size, (uint8_t)POLICY_NEW);
        p.decided.assign(size, 0); p.flush = flush_seq_;
    } else if (p.flush != flush_seq_) { std::fill(p.decided.begin(), p.decided.end(), (uint8_t)0); p.flush = flush_seq_; }
}

// Is the value at `pos`, whose only claim to storage is a live handle, stored by this flush?  Decided once per position and flush: the
// members of a launch group must agree on their shape.  optimistic: the component comes from a caller that does not free its temporaries
// (most internally consumed values carry handles) — a position never seen is then assumed dead.
bool Engine::policy_store(ShapePolicy& p, size_t pos, bool optimistic) {
    uint8_t& d = p.decided[pos];
    if (d) return d == 1;
    uint8_t& st = policy_state_[p.base + pos];
    bool store = false;
    switch (st) {
    case POLICY_NEW:       if (optimistic) st = POLICY_DEFER; else { st = POLICY_OBSERVING; store = true; } break;
    case POLICY_OBSERVING: st = POLICY_DEFER; break;                // stored last time, and nobody came for it (touch() would have made it POLICY_STORE)
    case POLICY_STORE:     store = true; break;
    default:               break;
    }
    d = store ? 1 : 2;
    return store;
}

void Engine::defer_node(Node* nd, const ShapePolicy* p, size_t pos) {
    if (!nd->deferred) {
        pend_erase(nd);
        nd->deferred = true;
        nd->pend_prev = deferred_head_.pend_prev; nd->pend_next = &deferred_head_; deferred_head_.pend_prev->pend_next = nd; deferred_head_.pend_prev = nd;
        ++n_deferred_; ++n_deferred_total_;
    }
    if (p) watch_node(nd, *p, pos);
}

void Engine::demand(Node* nd) {
    touch(nd);
    if (!nd->buf) materialize({ nd });
}

void Engine::materialize_deferred() {
    std::vector<int64_t> ids;
    for (Node* nd = deferred_head_.pend_next; nd != &deferred_head_; nd = nd->pend_next) if (nd->refs_ext > 0) ids.push_back(nd->id);
    for (int64_t id : ids) {                        // (storing one dismantles its recipe: others may go away with it)
        Node* nd = nodes_.get(id);
        if (nd && nd->id == id && !nd->buf && nd->deferred && !nd->discarded) materialize({ nd });
    }
}

struct Engine::Dag {
    std::vector<Node*> roots;       // the values asked for
    std::vector<Node*> order;       // pending nodes, operands before users
    std::vector<Node*> leaves;      // distinct materialised inputs
    std::vector<Node*> outs;        // roots first, then escaping intermediates
    std::vector<SsaOp> ops;
    std::vector<int> out_ids;
    std::vector<float> scalars;
    std::string sig;
    // replica descriptions (ReplicaGroup): the stamp all pending nodes of this DAG share (0: none), whether they all share one, whether any has one
    uint32_t rep_id = 0;
    bool rep_uniform = true, rep_any = false;
};

// Linearise the pending expressions below `roots` into ONE program (roots may share intermediates: they become
// several outputs of the same launch).  Returns false when it cannot run as one launch.
bool Engine::build_dag(const std::vector<Node*>& roots, Dag& dag) {
    HostTimer timer(HostProfile::BUILD_DAG);
    dag.roots = roots;
    const uint64_t ep = ++epoch_;                   // nodes with mark == ep have been visited by THIS build
    std::vector<std::pair<Node*, int>> stack;
    auto visit = [&](Node* nd) { nd->mark = ep; nd->tmp_id = -1; nd->tmp_uses = 0; };
    bool first_pending = true;
    auto note_rep = [&](const Node* nd) {
        if (first_pending) { dag.rep_id = nd->rep_id; first_pending = false; } else if (nd->rep_id != dag.rep_id) dag.rep_uniform = false;
        dag.rep_any |= nd->rep_id != 0;
    };
    for (Node* root : roots) {
        if (root->mark == ep) continue;             // a root that is also an operand of an earlier root
        visit(root); note_rep(root);
        stack.push_back({ root, 0 });
        while (!stack.empty()) {                    // iterative post-order
            auto& top = stack.back();
            Node* nd = top.first;
            if (top.second < nd->n_in) {
                Node* c = nd->in[top.second++];
                if (c->buf) {
                    if (c->mark != ep) { visit(c); dag.leaves.push_back(c); }
                } else {
                    if (c->mark != ep) { visit(c); note_rep(c); stack.push_back({ c, 0 }); }
                    c->tmp_uses++;
                }
            } else {
                dag.order.push_back(nd);
                stack.pop_back();
                if ((int)dag.order.size() > FM_MAX_OPS) return false;       // too large for one launch: no point in walking the rest
            }
        }
    }
    if ((int)dag.leaves.size() > FM_MAX_IN || (int)dag.order.size() > FM_MAX_OPS) return false;
    const int n_in = (int)dag.leaves.size();
    for (int k = 0; k < n_in; ++k) dag.leaves[k]->tmp_id = k;
    // structural signature: a short byte string (opcode + operand ids per op); scalars and vectors are NOT part of it
    dag.sig.clear();
    dag.sig.reserve(dag.order.size() * 4 + 16);
    dag.sig.push_back((char)('0' + math_mode)); dag.sig.push_back((char)n_in);
    for (size_t i = 0; i < dag.order.size(); ++i) {
        Node* nd = dag.order[i];
        nd->tmp_id = n_in + (int)i;
        SsaOp op{ nd->opcode, -1, -1, -1, nd->scalar };
        int* slots[3] = { &op.a, &op.b, &op.c };
        for (int k = 0; k < nd->n_in; ++k) *slots[k] = nd->in[k]->tmp_id;
        dag.ops.push_back(op);
        if (op_info(nd->opcode).scalar) dag.scalars.push_back((float)nd->scalar);
        dag.sig.push_back((char)nd->opcode); dag.sig.push_back((char)(op.a + 1)); dag.sig.push_back((char)(op.b + 1)); dag.sig.push_back((char)(op.c + 1));
    }
    if (dag.scalars.empty()) dag.scalars.push_back(0.0f);
    // outputs: every root, plus every intermediate somebody else still needs (tmp_uses = consumers inside this DAG)
    for (Node* r : roots) if (r->tmp_uses >= 0) { dag.outs.push_back(r); dag.out_ids.push_back(r->tmp_id); r->tmp_uses = -1 - r->tmp_uses; }
    // … an intermediate with a consumer outside this DAG is a fact; one whose only claim is a live handle is the escape policy's decision
    // (1: stored, 2: a candidate, 3: not stored — deferred —, 4: a candidate the policy stores)
    const size_t n_ops = dag.order.size();
    std::vector<char> flag(n_ops, 0);
    size_t n_cand = 0, interior = 0;
    for (size_t i = 0; i < n_ops; ++i) {
        Node* nd = dag.order[i];
        if (nd->tmp_uses < 0) continue;             // already an output (root)
        ++interior;
        if (nd->refs_int > nd->tmp_uses) flag[i] = 1;
        else if (nd->refs_ext > 0) {
            if (nd->rep_id || !ESCAPE_POLICY) flag[i] = 1;
            else if (nd->deferred) flag[i] = 3;
            else { flag[i] = 2; ++n_cand; }
        }
    }
    ShapePolicy* policy = nullptr;
    if (n_cand) {
        if (dag_policies_.size() > 65536) { dag_policies_.clear(); }      // (shapes that never repeat)
        policy = &dag_policies_[dag.sig];                                  // the structural signature: nothing in it says who holds a handle
        policy_bind(*policy, n_ops);
        const bool optimistic = n_cand * 2 > interior;
        for (size_t i = 0; i < n_ops; ++i) if (flag[i] == 2) flag[i] = policy_store(*policy, i, optimistic) ? 4 : 3;
    }
    for (size_t i = 0; i < n_ops; ++i) if (flag[i] == 1 || flag[i] == 4) { dag.outs.push_back(dag.order[i]); dag.out_ids.push_back(dag.order[i]->tmp_id); }
    if ((int)dag.outs.size() > FM_MAX_OUT) return false;
    for (size_t i = 0; i < n_ops; ++i) {
        Node* nd = dag.order[i];
        if (flag[i] == 4) watch_node(nd, *policy, i);
        else if (flag[i] == 3) defer_node(nd, nd->deferred ? nullptr : policy, i);
        else if (flag[i] == 0 && nd->tmp_uses >= 0) pend_erase(nd);      // lives on as somebody's operand only: no flush starts from it
    }
    dag.sig.push_back((char)0xff);
    for (int v : dag.out_ids) dag.sig.push_back((char)(v + 1));
    return true;
}

// Execute structurally identical, mutually independent DAGs as ONE launch (one batch row per DAG).
// proto: the member that carries the structure (signature, operations); nullptr = dags[0].  Members that are copies existing as a
// description (flush_all: replica_dag) carry vectors, outputs and scalars only.
bool Engine::run_dags_plain(std::vector<Dag>& dags, const Dag* proto) {
    struct Off { bool& w; bool was; ~Off() { w = was; } } off{ want_root_moments_, want_root_moments_ };
    want_root_moments_ = false;
    return run_dags(dags, nullptr, nullptr, nullptr, proto);
}

bool Engine::run_dags(std::vector<Dag>& dags, const double* reduce_shift, fmhip_moments* host_moments, void* dev_moments, const Dag* proto) {
    HostTimer timer(HostProfile::RUN_DAGS);
    const Dag& d0 = proto ? *proto : dags[0];
    const int64_t n = dags[0].outs[0]->n;
    Program* prog = nullptr;
    if (reduce_shift && dags.size() != 1) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a fused expectation belongs to one expression");
    // a flush that collects the moments of all pending roots (Engine::reduce): expressions of one value each, all rows of this launch
    static const double no_shift = 0.0;
    std::vector<fmhip_moments> all;
    bool moments_only = false;
    double* async_slots = nullptr;                                  // … or, from a flush that does not wait, into slots of the pinned arena
    { static const bool batch_trace = std::getenv("FMHIP_BATCH_TRACE") != nullptr;
      if (batch_trace && want_root_moments_) std::fprintf(stderr, "[fmhip batch] launchable group of %zu, %zu ops, %zu outs, proto %d\n", dags.size(), d0.ops.size(), d0.out_ids.size(), proto ? 1 : 0); }
    if (want_root_moments_ && !reduce_shift && !host_moments && !dev_moments && d0.out_ids.size() == 1) {
        bool roots_only = true;
        for (const Dag& d : dags) roots_only &= d.outs.size() == 1 && d.outs[0]->refs_ext > 0 && !d.outs[0]->moments_blocked;
        if (roots_only && (dags.size() >= 4 || n * (int64_t)d0.leaves.size() <= (int64_t(1) << 21) || unit_launch(n, (int64_t)dags.size()))) {
            double* slots = async_moments_ ? arena_alloc(dags.size()) : nullptr;
            if (slots) { async_slots = slots; reduce_shift = &no_shift; dev_moments = slots; }
            else if (!async_moments_) { all.resize(dags.size()); reduce_shift = &no_shift; host_moments = all.data(); }
            // … and where the caller has given the values up (fmhip_vec_give_up_values: every member's root, nobody else holds it) the
            // launch takes the moments and stores NOTHING: a program without outputs
            // (the root of a copy that exists as a description carries one internal reference, its group's hold, until the launch is done)
            if (reduce_shift) {
                moments_only = true;
                for (const Dag& d : dags) { const Node* r = d.outs[0]; moments_only &= r->discard && r->refs_int == ((r->rep_id && r->rep_copy && replica_of(r)) ? 1 : 0); }
            }
        }
    }
    const std::string key = reduce_shift ? d0.sig + (moments_only ? "\xfeM" : "\xfeR") : d0.sig;       // the program that also reduces its root is a different program
    auto it = program_cache_.find(key);
    if (it != program_cache_.end()) prog = it->second;
    else {
        try { prog = compile(d0.ops, (int)d0.leaves.size(), moments_only ? std::vector<int>{} : d0.out_ids, reduce_shift ? std::vector<int>{ d0.out_ids[0] } : std::vector<int>{}, nullptr, false); }
        catch (const Error& e) {
            if (e.code != FMHIP_ERR_PROGRAM_LIMIT) throw;
            if (all.empty() && !async_slots) return false;
            all.clear(); async_slots = nullptr; reduce_shift = nullptr; host_moments = nullptr; dev_moments = nullptr;       // without the moments, then
            return run_dags_plain(dags, proto);
        }
        program_cache_[key] = prog;
    }
    std::vector<std::vector<Buffer*>> out_bufs(dags.size());
    std::vector<RowSpec> rows(dags.size());
    try {
        for (size_t i = 0; i < dags.size(); ++i) {
            for (Node* l : dags[i].leaves) rows[i].in.push_back(l->buf->ptr);
            for (size_t k = 0; k < dags[i].outs.size() && !moments_only; ++k) {
                Buffer* b = new_buffer(n);
                out_bufs[i].push_back(b);
                rows[i].out.push_back(b->ptr);
            }
            rows[i].scalars = dags[i].scalars.data();
            rows[i].shifts = reduce_shift;
        }
        launch(prog, n, rows, host_moments, dev_moments);
    } catch (...) {
        for (auto& v : out_bufs) for (Buffer* b : v) buffer_unref(b);
        throw;
    }
    if (async_slots) for (size_t i = 0; i < dags.size(); ++i) arena_assign(dags[i].outs[0], async_slots + i * 4);
    for (size_t i = 0; i < all.size(); ++i) {
        Node* r = dags[i].outs[0];
        r->moments[0] = all[i].sum; r->moments[1] = all[i].sumsq; r->moments[2] = all[i].min; r->moments[3] = all[i].max; r->has_moments = true;
    }
    if (moments_only) {         // nothing was stored: the roots keep their moments, give their expressions up and are no roots of later flushes
        for (size_t i = 0; i < dags.size(); ++i) { Node* r = dags[i].outs[0]; r->discarded = true; r->refs_int++; drop_expression(r); r->refs_int--; }
        return true;
    }
    // commit: outputs become materialised leaves; their expressions (and unreferenced intermediates) go away
    for (size_t i = 0; i < dags.size(); ++i)
        for (size_t k = 0; k < dags[i].outs.size(); ++k) commit_node(dags[i].outs[k], out_bufs[i][k]);
    // (all outputs are kept alive until every expression has been dismantled: one that nobody holds — stored for the sake of a launch shape
    // whose kernel exists, build_big — goes away with its last consumer)
    for (size_t i = 0; i < dags.size(); ++i) for (Node* nd : dags[i].outs) nd->refs_int++;
    for (size_t i = 0; i < dags.size(); ++i) for (Node* nd : dags[i].outs) drop_expression(nd);
    for (size_t i = 0; i < dags.size(); ++i) for (Node* nd : dags[i].outs) { nd->refs_int--; node_maybe_free(nd); }
    return true;
}

// ---------------------------------------------------------------- components larger than one launch
//
// A connected component of pending work that does not fit one launch (12 inputs / 8 outputs / 15 live values / 128
// micro-ops) is cut into consecutive SEGMENTS of its topological order.  Every prefix of a topological order is closed
// under dependencies, so a segment only reads materialised vectors and outputs of earlier segments; its outputs are the
// values somebody outside the segment still needs.  Each segment is the longest one that still fits (found once per
// component SHAPE and cached); components of identical shape — e.g. the same Euler step of several parameter sets — are
// cut identically and their segments run as rows of the same launches.

// What a copy that exists as a description (ReplicaGroup) needs to know about the ORIGINAL's component, per position of its order —
// taken from the nodes before anything runs (a segment's nodes are dismantled as soon as it has been committed).
struct Engine::ReplicaView {
    ReplicaGroup* g = nullptr;
    std::vector<int32_t> rep_index, rep_root;   // recording index (→ scalar slot) and root number (-1: an inner value) of the operation
    std::vector<double> scalar;                  // the original's scalar operand (copies without a scalar list use it)
};

struct Engine::BigDag {
    std::vector<Node*> roots;
    std::vector<Node*> order;       // all pending nodes of the component, operands before users
    std::vector<Node*> leaves;      // distinct materialised inputs, in discovery order (structural: equal for equal shapes)
    std::vector<char> escapes;      // per node of the order: needed outside the component (a handle, or a consumer elsewhere)
    std::string sig;                // shape: per op {opcode, operand ids (16 bit), escapes?}
    uint64_t hash = 0;              // of sig
    int64_t n = 0;                  // elements per vector
    uint32_t rep_id = 0;            // replica descriptions: as in Dag
    bool rep_uniform = true, rep_any = false;
    bool discard_root = false;      // its single root is wanted for its moments only (Node::discard; part of the signature)
    // A member WITHOUT nodes — a copy of another member's component that exists as a description: `leaves` holds the copy's operands
    // (same numbering as the original's), values produced by one launch for a later one live in `temp` (by position of the order),
    // root values also go to the copy's root nodes.
    std::shared_ptr<const ReplicaView> view;
    int copy = -1;
    std::vector<Buffer*> temp;
    bool described() const { return copy >= 0; }
    Buffer* value(size_t pos) const { return described() ? temp[pos] : order[pos]->buf; }        // nullptr: not computed (yet)
    float scalar_at(size_t pos) const {
        if (!described()) return (float)order[pos]->scalar;
        const ReplicaGroup* g = view->g;
        const int32_t slot = g->scalar_slot[(size_t)view->rep_index[pos]];
        return (float)((slot >= 0 && !g->scalars.empty()) ? g->scalars[(size_t)copy * g->n_scalars + slot] : view->scalar[pos]);
    }
};

bool Engine::build_big(const std::vector<Node*>& roots, BigDag& big) {
    HostTimer timer(HostProfile::BUILD_BIG);
    big.roots = roots;
    const uint64_t ep = ++epoch_;
    std::vector<std::pair<Node*, int>> stack;
    int n_leaves = 0;
    std::vector<Node*>& leaves = big.leaves;
    leaves.clear();
    auto visit = [&](Node* nd) { nd->mark = ep; nd->tmp_id = -1; nd->tmp_uses = 0; };
    bool first_pending = true;
    auto note_rep = [&](const Node* nd) {
        if (first_pending) { big.rep_id = nd->rep_id; first_pending = false; } else if (nd->rep_id != big.rep_id) big.rep_uniform = false;
        big.rep_any |= nd->rep_id != 0;
    };
    big.n = roots[0]->n;
    big.discard_root = roots.size() == 1 && roots[0]->discard && roots[0]->refs_int == 0;
    for (Node* root : roots) {
        if (root->mark == ep) continue;
        visit(root); note_rep(root);
        stack.push_back({ root, 0 });
        while (!stack.empty()) {
            auto& top = stack.back();
            Node* nd = top.first;
            if (top.second < nd->n_in) {
                Node* c = nd->in[top.second++];
                if (c->buf) { if (c->mark != ep) { visit(c); c->tmp_id = --n_leaves; leaves.push_back(c); } }      // leaves: -1, -2, …
                else { if (c->mark != ep) { visit(c); note_rep(c); stack.push_back({ c, 0 }); } c->tmp_uses++; }
            } else { big.order.push_back(nd); stack.pop_back(); }
        }
    }
    if (big.order.size() > 60000) return false;
    // The STRUCTURAL signature of a list of nodes: per node its opcode and its operands (positions in the list; leaves by order of
    // discovery) as one 8-byte word whose top byte — the flag: is the value stored? — stays zero; and its hash, as it is written (until
    // round 4 the flag was part of it, so that a shape's identity depended on who held handles when the flush came).
    auto sign = [&](const std::vector<Node*>& order, std::string& sig) {
        const size_t count = order.size();
        sig.resize(1 + count * 8);
        sig[0] = (char)('0' + math_mode);
        char* p = &sig[1];
        uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)math_mode;
        for (size_t i = 0; i < count; ++i, p += 8) {
            const Node* nd = order[i];
            uint64_t w = (uint64_t)(uint8_t)nd->opcode;
            for (int k = 0; k < 3; ++k) w |= (uint64_t)(uint16_t)(k < nd->n_in ? nd->in[k]->tmp_id + 32768 : 0) << (8 + 16 * k);
            std::memcpy(p, &w, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32;
        }
        return h;
    };
    const size_t m = big.order.size();
    for (size_t i = 0; i < m; ++i) big.order[i]->tmp_id = (int)i;
    // A shape seen before (same walk, same operands) is scheduled the way it was then: the schedule below is a function of the structure
    // up to ties, and any topological order computes the same values — a caller that asks for one expectation after the other (144
    // products per objective evaluation, each a graph of 50–250 nodes) pays for the walk and ONE signature, not for the scheduling pass
    // and a second one (≈ 6 → 3 µs per product, all of it time the device waits).  The memo also carries what the engine has learnt about
    // the shape's handles (escape policy).
    static const size_t MEMO_MAX_NODES = 16384;               // (a memo holds ≈ 30 bytes per node)
    const bool memoise = m <= MEMO_MAX_NODES;
    ScheduleMemo* memo = nullptr;
    if (memoise) {
        const uint64_t walk_hash = sign(big.order, walk_sig_);
        auto known = schedule_cache_.find(walk_hash);
        if (known != schedule_cache_.end() && known->second.walk_sig == walk_sig_) {
            memo = &known->second;
            std::vector<Node*> scheduled(m);
            for (size_t i = 0; i < m; ++i) { scheduled[i] = big.order[memo->perm[i]]; scheduled[i]->tmp_id = (int)i; }
            big.order.swap(scheduled);
        } else {
            if (schedule_cache_.size() >= 4096 || schedule_cache_bytes_ > (size_t(128) << 20)) { schedule_cache_.clear(); schedule_cache_bytes_ = 0; policy_reset(); }     // (shapes that never repeat: start again)
            memo = &schedule_cache_[walk_hash];
            schedule_cache_bytes_ -= std::min(schedule_cache_bytes_, memo->walk_sig.size() * 3 + memo->perm.size() * 6);       // (the same walk hash again: replaced)
            *memo = ScheduleMemo();
            memo->walk_sig = walk_sig_;
        }
    }
    if (!memo || memo->perm.empty()) {
        // Schedule: the DFS post-order above is A topological order, but not a good one to cut into launches — it lists a whole
        // dependency chain (e.g. the running factor sum over all LIBOR components of an Euler step) before the values that merely
        // consume one link of it, so every link would have to be materialised for a later segment.  List scheduling, consumers
        // first: emit a ready node, then prefer the nodes it has just made ready (LIFO; ties by creation order).  A value is
        // consumed as soon as possible after it is produced: short live ranges, few values crossing a cut — two Euler steps
        // recorded back to back come out component by component, both steps of a component adjacent, and the intermediate state
        // never touches HBM.  Every op computes the same thing in any topological order: results are unchanged bit for bit.
        std::vector<uint32_t> perm(m);
        {
            std::vector<int> indeg(m, 0), head(m + 1, 0);
            for (size_t i = 0; i < m; ++i)
                for (int k = 0; k < big.order[i]->n_in; ++k) { Node* c = big.order[i]->in[k]; if (!c->buf) { indeg[i]++; head[(size_t)c->tmp_id + 1]++; } }
            for (size_t i = 0; i < m; ++i) head[i + 1] += head[i];
            std::vector<int> consumers((size_t)head[m]), fill(head.begin(), head.end() - 1);
            for (size_t i = 0; i < m; ++i)
                for (int k = 0; k < big.order[i]->n_in; ++k) { Node* c = big.order[i]->in[k]; if (!c->buf) consumers[(size_t)fill[(size_t)c->tmp_id]++] = (int)i; }
            auto by_id_desc = [&](int a, int b) { return big.order[(size_t)a]->id > big.order[(size_t)b]->id; };
            std::vector<int> stack_ready;
            for (size_t i = 0; i < m; ++i) if (indeg[i] == 0) stack_ready.push_back((int)i);
            std::sort(stack_ready.begin(), stack_ready.end(), by_id_desc);             // oldest node on top
            std::vector<Node*> scheduled;
            scheduled.reserve(m);
            std::vector<int> fresh;
            while (!stack_ready.empty()) {
                const int i = stack_ready.back(); stack_ready.pop_back();
                scheduled.push_back(big.order[(size_t)i]);
                fresh.clear();
                for (int q = head[(size_t)i]; q < head[(size_t)i + 1]; ++q) if (--indeg[(size_t)consumers[(size_t)q]] == 0) fresh.push_back(consumers[(size_t)q]);
                std::sort(fresh.begin(), fresh.end(), by_id_desc);
                stack_ready.insert(stack_ready.end(), fresh.begin(), fresh.end());
            }
            if (scheduled.size() == m) big.order.swap(scheduled);                      // (always: the pending graph is acyclic)
            for (size_t i = 0; i < m; ++i) perm[i] = (uint32_t)big.order[i]->tmp_id;     // scheduled position → position in the walk
        }
        for (size_t i = 0; i < m; ++i) big.order[i]->tmp_id = (int)i;
        if (memo) {
            memo->perm.swap(perm);
            memo->sched_hash = sign(big.order, memo->sched_sig);
            schedule_cache_bytes_ += memo->walk_sig.size() * 3 + memo->perm.size() * 6;
        }
    }
    // Which values are stored.  A consumer outside the component, or being a root of this walk, is a fact; a value whose only claim is a
    // live handle is the escape policy's decision ('?' → 'X' stored and watched / 'D' deferred); 'd': deferred earlier, stays so.
    std::vector<char> flags(m);
    size_t n_cand = 0, interior = 0;
    for (size_t i = 0; i < m; ++i) {
        const Node* nd = big.order[i];
        // (a root whose VALUE the caller has given up — moments only, Node::discard — is not an output of the component: 'm', a shape of
        // its own, its peeled kernels do not store it; launches that cannot take the moments along — segments — store it all the same)
        char f = '.';
        if (nd->discard && nd->refs_int == 0) f = 'm';
        else if (nd->refs_int > nd->tmp_uses) f = 'x';
        else if (nd->refs_ext > 0) {
            if (nd->tmp_uses == 0 || nd->rep_id || !memo || !ESCAPE_POLICY) f = 'x';
            else if (nd->deferred) f = 'd';
            else { f = '?'; ++n_cand; }
        }
        interior += nd->tmp_uses > 0 ? 1 : 0;
        flags[i] = f;
    }
    const std::vector<char> facts = flags;                     // 'x' / 'm' here are facts; '?' and 'd' the policy's to decide, '.' nobody's
    if (n_cand) {
        policy_bind(memo->policy, m);
        const bool optimistic = n_cand * 2 > interior;
        for (size_t i = 0; i < m; ++i) if (flags[i] == '?') flags[i] = policy_store(memo->policy, i, optimistic) ? 'x' : '.';
    }
    for (size_t i = 0; i < m; ++i) if (flags[i] == 'd') flags[i] = '.';
    // the shape's full signature and hash: the structure with the flags in the top byte of every word
    auto finish = [&](const std::string& structural, uint64_t structural_hash, const std::vector<char>& fl, std::string& sig, uint64_t& hash) {
        sig = structural;
        uint64_t h = structural_hash;
        for (size_t i = 0; i < m; ++i) { sig[1 + i * 8 + 7] = fl[i]; h = (h ^ ((uint64_t)(uint8_t)fl[i] + (i << 8))) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
        hash = h;
    };
    if (memo) {
        // The variants of this shape — the same structure, other values stored — the engine has used: the escape policy moves a shape
        // through a few of them while it learns (stored for observation → left unstored → stored after all), and each needs its own loop
        // kernel.  A variant whose kernel is still being compiled does not run as segments on the interpreter meanwhile: the launch goes
        // through the variant used last whose kernel exists, as long as that one stores every value somebody outside the component needs
        // (the facts) — a value stored without need costs a write, one left unstored against the policy's wish is computed from its recipe
        // if somebody asks (and watched all the same).  The wanted variant's kernel is asked for now, and used once it is there.
        std::vector<ScheduleMemo::Variant>& vs = memo->variants;
        size_t at = vs.size();
        for (size_t v = 0; v < vs.size(); ++v) if (vs[v].flags == flags) { at = v; break; }
        if (at == vs.size()) {
            if (vs.size() >= 6) { vs.erase(vs.begin() + 1, vs.begin() + 2); at = vs.size(); }     // (the oldest but the first — the one that stores everything it was asked to at first sight)
            vs.emplace_back();
            vs[at].flags = flags;
            finish(memo->sched_sig, memo->sched_hash, flags, vs[at].sig, vs[at].hash);
            schedule_cache_bytes_ += vs[at].sig.size() + m;
        }
        size_t use = at;
        static const bool HYSTERESIS = [] { const char* e = std::getenv("FMHIP_VARIANT_HYSTERESIS"); return !(e && e[0] == '0'); }();
        if (HYSTERESIS && ESCAPE_POLICY && vs.size() > 1 && jit_mode == FMHIP_JIT_AUTO) {
            auto plan_of = [&](const ScheduleMemo::Variant& v) -> BigPlan* { auto it = plan_cache_.find(v.hash); return it != plan_cache_.end() && it->second.sig == v.sig ? &it->second : nullptr; };
            auto kernel_there = [&](const BigPlan& p) {
                if (!p.rolled.present) return true;
                const std::shared_ptr<JitSlot>& slot = p.rolled.peeled.present ? p.rolled.peeled.jit : p.rolled.jit;
                return slot && slot->state.load(std::memory_order_acquire) != JitSlot::QUEUED;
            };
            BigPlan* wanted = plan_of(vs[at]);
            if (!wanted && plan_cache_.count(vs[at].hash) == 0) {          // never planned: its loop is looked for and its kernels are asked for, nothing runs
                big.escapes.resize(m);
                for (size_t i = 0; i < m; ++i) big.escapes[i] = flags[i] == 'x' ? 1 : 0;
                BigPlan np;
                plan_loop(np, big);
                np.sig = vs[at].sig; np.discards_root = big.discard_root; np.segs_missing = true;
                wanted = &(plan_cache_[vs[at].hash] = std::move(np));
            }
            if (wanted && !kernel_there(*wanted)) {
                for (size_t back = memo->last_variant < vs.size() ? memo->last_variant : 0, tried = 0; tried < vs.size(); ++tried, back = (back + 1) % vs.size()) {
                    if (back == at) continue;
                    const ScheduleMemo::Variant& v = vs[back];
                    bool legal = true;
                    for (size_t i = 0; i < m && legal; ++i) legal = (facts[i] != 'x' || v.flags[i] == 'x') && ((facts[i] == 'm') == (v.flags[i] == 'm'));
                    if (!legal) continue;
                    const BigPlan* p = plan_of(v);
                    if (p && kernel_there(*p)) { use = back; break; }
                }
            }
        }
        memo->last_variant = use;
        if (use != at) flags = vs[use].flags;
        big.sig = vs[use].sig; big.hash = vs[use].hash;
    } else {
        std::string structural;
        const uint64_t h = sign(big.order, structural);
        finish(structural, h, flags, big.sig, big.hash);
    }
    // the nodes learn what has been decided: a handle whose value is stored is watched (a use makes its position a stored one for good), one
    // whose value is not is deferred; whatever lives on as somebody's operand only leaves the pending list — no flush starts from it
    big.escapes.resize(m);
    for (size_t i = 0; i < m; ++i) {
        Node* nd = big.order[i];
        const bool policy_position = facts[i] == '?' || facts[i] == 'd';
        if (flags[i] == 'x') { if (policy_position && facts[i] == '?') watch_node(nd, memo->policy, i); }
        else if (flags[i] == '.') {
            if (nd->refs_ext > 0 && nd->tmp_uses > 0 && nd->refs_int <= nd->tmp_uses) defer_node(nd, facts[i] == '?' ? &memo->policy : nullptr, i);
            else if (!nd->deferred) pend_erase(nd);
        }
        big.escapes[i] = flags[i] == 'x' ? 1 : 0;
    }
    return true;
}

// uses: per position of the component's order, the consumers INSIDE the component (structural: equal for all members of a group)
bool Engine::segment_dag(const BigDag& big, size_t s, size_t e, Dag& dag, const std::vector<int32_t>& uses) {
    const uint64_t ep = ++epoch_, ep_leaf = ++epoch_;          // mark == ep: produced inside the segment; == ep_leaf: registered input
    dag = Dag();
    for (size_t i = s; i < e; ++i) { Node* nd = big.order[i]; nd->mark = ep; nd->tmp_uses = 0; }
    for (size_t i = s; i < e; ++i) {
        Node* nd = big.order[i];
        for (int k = 0; k < nd->n_in; ++k) {
            Node* c = nd->in[k];
            if (c->mark == ep) { c->tmp_uses++; continue; }
            if (c->mark == ep_leaf) continue;
            if (!c->buf) return false;                          // reads a value that is neither materialised nor produced here: not a valid cut
            c->mark = ep_leaf;
            dag.leaves.push_back(c);
            if ((int)dag.leaves.size() > FM_MAX_IN) return false;
        }
    }
    if (e - s > (size_t)FM_MAX_OPS) return false;
    const int n_in = (int)dag.leaves.size();
    for (int k = 0; k < n_in; ++k) dag.leaves[(size_t)k]->tmp_id = k;
    dag.sig.push_back((char)('0' + math_mode)); dag.sig.push_back((char)n_in);
    for (size_t i = s; i < e; ++i) {
        Node* nd = big.order[i];
        nd->tmp_id = n_in + (int)(i - s);
        SsaOp op{ nd->opcode, -1, -1, -1, nd->scalar };
        int* slots[3] = { &op.a, &op.b, &op.c };
        for (int k = 0; k < nd->n_in; ++k) *slots[k] = nd->in[k]->tmp_id;
        dag.order.push_back(nd);
        dag.ops.push_back(op);
        if (op_info(nd->opcode).scalar) dag.scalars.push_back((float)nd->scalar);
        dag.sig.push_back((char)nd->opcode); dag.sig.push_back((char)(op.a + 1)); dag.sig.push_back((char)(op.b + 1)); dag.sig.push_back((char)(op.c + 1));
    }
    if (dag.scalars.empty()) dag.scalars.push_back(0.0f);
    for (size_t i = s; i < e; ++i) {
        Node* nd = big.order[i];
        // an output of the component (build_big's decision — a handle alone does not make one), a consumer in a later segment, or the
        // component's root (also where its value has been given up: a segment cannot take the moments along, it stores it all the same)
        if (big.escapes[i] || uses[i] > nd->tmp_uses || (uses[i] == 0 && nd->refs_ext > 0)) { dag.outs.push_back(nd); dag.out_ids.push_back(nd->tmp_id); }
    }
    if (dag.outs.empty() || (int)dag.outs.size() > FM_MAX_OUT) return false;
    dag.roots = dag.outs;
    dag.sig.push_back((char)0xff);
    for (int v : dag.out_ids) dag.sig.push_back((char)(v + 1));
    return true;
}

// A value of a member without nodes has been computed: it is kept for the launches that read it, and handed to the copy's root node
// if it is one of the replicated roots.
void Engine::commit_described(BigDag& big, size_t pos, Buffer* b) {
    big.temp[pos] = b;                                  // owns the buffer's first reference
    const int32_t root = big.view->rep_root[pos];
    if (root < 0) return;
    const ReplicaGroup* g = big.view->g;
    Node* c = g->copy_roots[(size_t)big.copy * g->n_roots + root];
    b->refs++;
    commit_node(c, b);
}

// One segment of a planned component for every member of a group: gather the row blocks by index, launch, commit.
void Engine::run_planned_segment(const BigPlan::Seg& seg, std::vector<BigDag>& group, size_t first, size_t count, ReduceRequest* rr, Program* prog_red) {
    const int64_t n = group[first].n;
    const size_t n_scal = seg.scal.empty() ? 1 : seg.scal.size();
    std::vector<RowSpec> rows(count);
    std::vector<float> scalars(count * n_scal, 0.0f);
    std::vector<Buffer*> out_bufs;
    out_bufs.reserve(count * seg.out.size());
    try {
        for (size_t c = 0; c < count; ++c) {
            BigDag& big = group[first + c];
            RowSpec& r = rows[c];
            r.in.reserve(seg.in.size()); r.out.reserve(seg.out.size());
            for (int32_t i : seg.in) {
                Buffer* b = i >= 0 ? big.value((size_t)i) : big.leaves[(size_t)(-1 - i)]->buf;
                if (!b) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "planned segment reads a value that has not been computed");
                r.in.push_back(b->ptr);
            }
            for (size_t k = 0; k < seg.out.size(); ++k) { Buffer* b = new_buffer(n); out_bufs.push_back(b); r.out.push_back(b->ptr); }
            float* sc = scalars.data() + c * n_scal;
            for (size_t k = 0; k < seg.scal.size(); ++k) sc[k] = big.scalar_at((size_t)seg.scal[k]);
            r.scalars = sc; r.shifts = rr ? &rr->shift : nullptr;
        }
        if (rr) { launch(prog_red, n, rows, rr->host_out, rr->dev_out); rr->done = true; }
        else launch(seg.prog, n, rows, nullptr, nullptr);
    } catch (...) { for (Buffer* b : out_bufs) buffer_unref(b); throw; }
    // commit (as run_dags): outputs become materialised leaves; their expressions (and unreferenced intermediates) go away
    for (size_t c = 0; c < count; ++c) {
        BigDag& big = group[first + c];
        for (size_t k = 0; k < seg.out.size(); ++k) {
            Buffer* b = out_bufs[c * seg.out.size() + k];
            if (big.described()) commit_described(big, (size_t)seg.out[k], b);
            else commit_node(big.order[(size_t)seg.out[k]], b);
        }
    }
    std::vector<Node*> done;
    for (size_t c = 0; c < count; ++c) {
        BigDag& big = group[first + c];
        if (big.described()) continue;
        for (size_t k = 0; k < seg.out.size(); ++k) done.push_back(big.order[(size_t)seg.out[k]]);
    }
    for (Node* nd : done) nd->refs_int++;       // keep alive while the expressions are dismantled
    for (Node* nd : done) drop_expression(nd);
    for (Node* nd : done) { nd->refs_int--; node_maybe_free(nd); }
}

// ---------------------------------------------------------------- rolled loops
//
// The scheduled order of a large component is often PERIODIC: the same few operations over one vector after another, each
// iteration feeding the next through a value or two — the running factor sum over the LIBOR components of an Euler step, a
// swap's backward induction over its periods.  Cut into launches of ≤ 12 inputs / 8 outputs such a stretch moves ≈ 1.5 vectors
// per iteration and step and costs a launch every six iterations.  Rolled up it is ONE launch: the body of one iteration is
// compiled (hiprtc) into a kernel that loops over the iterations, keeps the carried values in registers, loads each iteration's
// inputs while it computes the previous one and stores each result the moment it is final; iteration count, vector pointers and
// scalar operands come from the row table, so one kernel serves every component count.  Every operation is evaluated by the
// same ueval<> functions in the same order per element as in the segmented launches: results are bit-identical, and until the
// kernel is compiled (or with FMHIP_JIT=off / FMHIP_ROLL=0) the segmented launches run.

static bool MERGE_CHAINS_ON() { static const bool v = [] { const char* e = std::getenv("FMHIP_MERGE_CHAINS"); return !(e && e[0] == '0'); }(); return v; }
static inline uint64_t mix64(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdull; }

#define ROLL_TRACE(...) do { if (roll_trace) std::fprintf(stderr, __VA_ARGS__); } while (0)
bool Engine::detect_loop(const BigDag& g, const std::vector<std::array<int32_t, 3>>& operand, BigPlan::Rolled& ro, std::string* source, int* elems_out, RolledBody* body_out)
{
    static const bool roll_trace = std::getenv("FMHIP_ROLL_TRACE") != nullptr;
    const size_t n = g.order.size();
    ROLL_TRACE("[fmhip roll] component of %zu nodes, %zu leaves\n", n, g.leaves.size());
    const int MAX_PERIOD = 128, MIN_ITERATIONS = 5, GLOBAL_SPAN = MAX_PERIOD;      // an input of ONE iteration has all its uses less than a period apart
    if (n < 48) return false;
    std::vector<int32_t> first_use(g.leaves.size(), -1), last_leaf_use(g.leaves.size(), -1);
    std::vector<uint32_t> last_use(n, 0);                       // largest consumer index; n = needed outside the component
    for (size_t i = 0; i < n; ++i) {
        if (g.escapes[i]) last_use[i] = (uint32_t)n;
        for (int k = 0; k < g.order[i]->n_in; ++k) {
            const int32_t o = operand[i][(size_t)k];
            if (o < 0) { const size_t l = (size_t)(-1 - o); if (first_use[l] < 0) first_use[l] = (int32_t)i; last_leaf_use[l] = (int32_t)i; }
            else if (last_use[(size_t)o] < (uint32_t)i) last_use[(size_t)o] = (uint32_t)i;
        }
    }
    auto is_global = [&](size_t l) { return last_leaf_use[l] - first_use[l] >= GLOBAL_SPAN; };
    // position-independent signature of every node: what it does and how far back its operands are.  NOT whether it is stored: an
    // iteration that stores a value the others only pass on (a state one product reads, a handle the escape policy keeps for one
    // component and not for the next) is the same iteration — the loop stores that position in EVERY iteration (out_needed below is the
    // union over the iterations): a few vectors more written, against a stretch that would not roll at all.
    std::vector<uint64_t> sig(n);
    for (size_t i = 0; i < n; ++i) {
        const Node* nd = g.order[i];
        uint64_t h = mix64(0x1234, (uint64_t)nd->opcode * 8 + (uint64_t)nd->n_in * 2);
        for (int k = 0; k < nd->n_in; ++k) {
            const int32_t o = operand[i][(size_t)k];
            if (o >= 0) h = mix64(h, 0x100000000ull + (uint64_t)((int64_t)i - o));
            else { const size_t l = (size_t)(-1 - o); h = is_global(l) ? mix64(h, 0x200000000ull + l) : mix64(h, 0x300000000ull + (uint64_t)((int64_t)i - first_use[l])); }
        }
        sig[i] = h;
    }
    // the periodic stretch that covers the most nodes
    size_t best_start = 0, best_cover = 0; int best_period = 0;
    for (int P = 3; P <= MAX_PERIOD && (size_t)P * MIN_ITERATIONS <= n; ++P) {
        size_t run_start = 0, run = 0;
        for (size_t i = 0; i + (size_t)P <= n; ++i) {
            const bool match = i + (size_t)P < n && sig[i] == sig[i + (size_t)P];
            if (match) { if (run == 0) run_start = i; ++run; }
            if (!match || i + (size_t)P + 1 >= n) {
                if (run > 0) { const size_t cover = (run + (size_t)P) / (size_t)P * (size_t)P; if (cover > best_cover) { best_cover = cover; best_start = run_start; best_period = P; } }
                run = 0;
            }
        }
    }
    ROLL_TRACE("[fmhip roll]   best period %d, start %zu, cover %zu\n", best_period, best_start, best_cover);
    if (best_period == 0 || best_cover / (size_t)best_period < (size_t)MIN_ITERATIONS) return false;
    const uint32_t P = (uint32_t)best_period;
    // phase: any rotation of the period is periodic too; take the one with the fewest values crossing the iteration boundary
    uint32_t best_phase = 0; size_t best_carried = SIZE_MAX;
    for (uint32_t phase = 0; phase < P; ++phase) {
        const size_t b = best_start + P + phase;                // second detected iteration: its predecessors exist
        if (b + P > best_start + best_cover) break;
        std::unordered_set<int32_t> crossing;
        bool ok = true;
        for (uint32_t q = 0; q < P && ok; ++q)
            for (int k = 0; k < g.order[b + q]->n_in; ++k) {
                const int32_t o = operand[b + q][(size_t)k];
                if (o < 0) {                                    // an input of one iteration must not straddle the boundary either
                    const size_t l = (size_t)(-1 - o);
                    if (!is_global(l) && ((size_t)first_use[l] < b || (size_t)last_leaf_use[l] >= b + P)) { ok = false; break; }
                    continue;
                }
                const int64_t d = (int64_t)(b + q) - o;
                if (d > (int64_t)q) { if (d > (int64_t)q + P) { ok = false; break; } crossing.insert(o); }
            }
        if (ok && crossing.size() < best_carried) { best_carried = crossing.size(); best_phase = phase; }
    }
    ROLL_TRACE("[fmhip roll]   phase %u, %zu values cross the iteration boundary\n", best_phase, best_carried == SIZE_MAX ? (size_t)0 : best_carried);
    if (best_carried == SIZE_MAX) return false;
    const size_t begin = best_start + P + best_phase;
    const size_t R = (best_start + best_cover - begin) / P;
    if (R < (size_t)MIN_ITERATIONS - 1) return false;
    const size_t end = begin + R * P;
    // validate every iteration; collect the body's interface from the first one
    std::vector<char> out_needed(P, 0), final_needed(P, 0);
    for (size_t r = 0; r < R; ++r)
        for (uint32_t q = 0; q < P; ++q) {
            const size_t i = begin + r * P + q;
            if (sig[i] != sig[begin + q]) { ROLL_TRACE("[fmhip roll]   aperiodic at iteration %zu position %u\n", r, q); return false; }
            for (int k = 0; k < g.order[i]->n_in; ++k) {
                const int32_t o = operand[i][(size_t)k];
                if (o >= 0) { const int64_t d = (int64_t)i - o; if (d > (int64_t)q + P) { ROLL_TRACE("[fmhip roll]   operand further back than one iteration (iteration %zu position %u)\n", r, q); return false; } }
                else {
                    const size_t l = (size_t)(-1 - o);
                    if (!is_global(l) && ((size_t)first_use[l] < begin + r * P || (size_t)last_leaf_use[l] >= begin + (r + 1) * P)) {   // an input of exactly one iteration
                        ROLL_TRACE("[fmhip roll]   input used by more than one iteration (iteration %zu position %u, span %d)\n", r, q, last_leaf_use[l] - first_use[l]); return false; }
                }
            }
            // consumers in the same and in the next iteration are served from registers; anybody later (or outside) needs the vector
            const size_t reach = r + 1 < R ? begin + (r + 2) * P : end;
            if (last_use[i] >= reach) { if (r + 1 == R && !g.escapes[i]) final_needed[q] = 1; else out_needed[q] = 1; }     // last iteration only: stored once, behind the loop
        }
    ro = BigPlan::Rolled();
    ro.begin = (uint32_t)begin; ro.period = P; ro.iterations = (uint32_t)R;
    std::vector<int> carried_index(P, -1), global_index(g.leaves.size(), -1);
    std::vector<std::array<std::string, 3>> name(P);              // operand names of the body
    bool library_math = false, uses_log = false;
    int n_local_leaf = 0;
    std::unordered_map<size_t, int> local_leaf;                      // leaf -> per-iteration input number (first iteration's leaves)
    for (uint32_t q = 0; q < P; ++q) {
        const size_t i = begin + q;
        const Node* nd = g.order[i];
        library_math |= nd->opcode == FMHIP_OP_POW_S || nd->opcode == FMHIP_OP_SIN || nd->opcode == FMHIP_OP_COS || nd->opcode == FMHIP_OP_EXP || nd->opcode == FMHIP_OP_LOG;
        uses_log |= nd->opcode == FMHIP_OP_LOG && math_mode != FMHIP_MATH_FAST;
        if (op_info(nd->opcode).scalar) ro.scal_pos.push_back(q);
        if (out_needed[q]) ro.out_pos.push_back(q);
        else if (final_needed[q]) ro.final_pos.push_back(q);
        for (int k = 0; k < nd->n_in; ++k) {
            const int32_t o = operand[i][(size_t)k];
            if (o >= 0) {
                const int64_t d = (int64_t)i - o;
                if (d <= (int64_t)q) name[q][(size_t)k] = "v" + std::to_string(q - (uint32_t)d);
                else {
                    const uint32_t src = q + P - (uint32_t)d;
                    if (carried_index[src] < 0) { carried_index[src] = (int)ro.carried.size(); ro.carried.push_back(src); }
                    name[q][(size_t)k] = "c" + std::to_string(carried_index[src]);
                }
            } else {
                const size_t l = (size_t)(-1 - o);
                if (is_global(l)) {
                    if (global_index[l] < 0) { global_index[l] = (int)ro.global_leaf.size(); ro.global_leaf.push_back((int32_t)l); }
                    name[q][(size_t)k] = "g" + std::to_string(global_index[l]);
                } else {
                    auto it = local_leaf.find(l);
                    if (it == local_leaf.end()) { it = local_leaf.emplace(l, n_local_leaf++).first; ro.leaf_in.push_back({ q, (uint32_t)k }); }
                    name[q][(size_t)k] = "l" + std::to_string(it->second);
                }
            }
        }
    }
    const size_t G = ro.global_leaf.size(), CI = ro.carried.size(), CO = ro.final_pos.size(), LI = ro.leaf_in.size(), LO = ro.out_pos.size(), LS = ro.scal_pos.size();
    ROLL_TRACE("[fmhip roll]   begin %zu, %zu iterations of %u: %zu global, %zu carried, %zu in, %zu out, %zu scalars\n", begin, R, P, G, CI, LI, LO, LS);
    if (G > 8 || CI > 12 || CO > 12 || LI > 12 || LO > 12 || LO + CO == 0 || LS > 48) return false;
    ro.row_words = (uint32_t)(G + CI + CO + R * (LI + LO) + (R * LS + 1) / 2);
    ro.iter_leaf.resize(R * LI);
    for (size_t r = 0; r < R; ++r)
        for (size_t m2 = 0; m2 < LI; ++m2) ro.iter_leaf[r * LI + m2] = -1 - operand[begin + r * P + ro.leaf_in[m2].first][(size_t)ro.leaf_in[m2].second];
    // ---- the kernel
    // elements per lane: 8 keeps more bytes in flight per wave, 4 halves the registers (more waves per SIMD to overlap the loop's
    // load → compute → store with each other); FMHIP_ROLL_ELEMS overrides for measurements
    static const int ELEMS_ENV = [] { const char* e = std::getenv("FMHIP_ROLL_ELEMS"); const int v = e ? std::atoi(e) : 0; return (v == 4 || v == 8) ? v : 0; }();
    const int E = ELEMS_ENV ? ELEMS_ENV : (library_math ? 4 : 8);
    *elems_out = E;
    RolledBody body;
    body.elems = E; body.uses_log = uses_log; body.globals = (uint32_t)G; body.inputs = (uint32_t)LI;
    body.carried = ro.carried; body.final_pos = ro.final_pos; body.out_pos = ro.out_pos;
    for (uint32_t q = 0; q < P; ++q) {
        const Node* nd = g.order[begin + q];
        UVariant uv{};
        if (!variant_for(nd->opcode, 0, &uv)) return false;
        uint32_t uop = uv.uop;
        if (math_mode == FMHIP_MATH_FAST) { if (uop == U_EXP) uop = U_EXP_FAST; else if (uop == U_LOG) uop = U_LOG_FAST; }
        body.ops.push_back({ uop, name[q][0], uv.r1_pos >= 0 ? name[q][(size_t)uv.r1_pos] : std::string(), uv.r2_pos >= 0 ? name[q][(size_t)uv.r2_pos] : std::string(),
                             op_info(nd->opcode).scalar });
    }
    jit().record(jit_describe(body));
    *source = jit_generate_rolled_source(body);
    if (body_out) *body_out = body;
    return true;
}

// The PEELED form of a component with a rolled loop: everything in front of the loop and behind it in the same launch (jit.hpp:
// RolledBody::Peel).  Possible when both parts are short, read few vectors of their own, and the part behind the loop reads nothing
// of the loop but final values of its last iteration.
bool Engine::plan_peel(const BigDag& g, const std::vector<std::array<int32_t, 3>>& operand, BigPlan::Rolled& ro, const RolledBody& loop_body)
{
    static const bool PEEL = [] { const char* e = std::getenv("FMHIP_PEEL"); return !(e && e[0] == '0'); }();
    if (!PEEL) return false;
    const size_t n = g.order.size(), P = ro.period, R = ro.iterations, begin = ro.begin, end = begin + P * R;
    static const size_t MAX_OPS = [] { const char* e = std::getenv("FMHIP_PEEL_MAX_OPS"); return e ? (size_t)std::atoll(e) : (size_t)192; }();
    const size_t MAX_EXTRA = 16;
    if (begin > MAX_OPS || n - end > MAX_OPS || begin == 0) return false;
    RolledBody body = loop_body;
    RolledBody::Peel& pl = body.peel;
    BigPlan::Rolled::Peeled pe;
    pl.present = true;
    std::vector<int> global_of(g.leaves.size(), -1), extra_of(g.leaves.size(), -1);
    for (size_t k = 0; k < ro.global_leaf.size(); ++k) global_of[(size_t)ro.global_leaf[k]] = (int)k;
    auto leaf_name = [&](size_t l) {
        if (global_of[l] >= 0) return "g" + std::to_string(global_of[l]);
        if (extra_of[l] < 0) { extra_of[l] = (int)pe.extra_leaf.size(); pe.extra_leaf.push_back((int32_t)l); }
        return "x" + std::to_string(extra_of[l]);
    };
    std::vector<int> final_of(P, -1);
    for (size_t k = 0; k < ro.final_pos.size(); ++k) final_of[ro.final_pos[k]] = (int)k;
    auto make_op = [&](size_t i, bool behind, RolledBody::Op& out) {
        const Node* nd = g.order[i];
        UVariant uv{};
        if (!variant_for(nd->opcode, 0, &uv)) return false;
        if (nd->opcode == FMHIP_OP_POW_S || nd->opcode == FMHIP_OP_SIN || nd->opcode == FMHIP_OP_COS) return false;     // out-of-line library code: not in these kernels
        uint32_t uop = uv.uop;
        if (math_mode == FMHIP_MATH_FAST) { if (uop == U_EXP) uop = U_EXP_FAST; else if (uop == U_LOG) uop = U_LOG_FAST; }
        body.uses_log |= uop == U_LOG;
        std::string name[3];
        for (int k = 0; k < nd->n_in; ++k) {
            const int32_t o = operand[i][(size_t)k];
            if (o < 0) name[k] = leaf_name((size_t)(-1 - o));
            else if ((size_t)o < begin) name[k] = "p" + std::to_string(o);
            else if ((size_t)o >= end) { if (!behind) return false; name[k] = "q" + std::to_string((size_t)o - end); }
            else {                                                   // a value of the loop: only a final value of its LAST iteration, only from behind it
                const size_t it = ((size_t)o - begin) / P, q = ((size_t)o - begin) % P;
                if (!behind || it != R - 1 || final_of[q] < 0) return false;
                name[k] = "F" + std::to_string(final_of[q]);
            }
        }
        out = { uop, name[0], uv.r1_pos >= 0 ? name[(size_t)uv.r1_pos] : std::string(), uv.r2_pos >= 0 ? name[(size_t)uv.r2_pos] : std::string(), op_info(nd->opcode).scalar };
        return true;
    };
    for (size_t i = 0; i < begin; ++i) {
        RolledBody::Op op;
        if (!make_op(i, false, op)) return false;
        pl.pre.push_back(op);
        if (op.scalar) pe.pre_scal.push_back((uint32_t)i);
        if (g.escapes[i]) { pl.pre_out.push_back((uint32_t)i); pe.pre_out.push_back((uint32_t)i); }
    }
    pl.extra_pre = (uint32_t)pe.extra_leaf.size();
    for (size_t k = 0; k < ro.carried.size(); ++k) pl.carried_init.push_back("p" + std::to_string(begin - P + ro.carried[k]));
    for (size_t i = end; i < n; ++i) {
        RolledBody::Op op;
        if (!make_op(i, true, op)) return false;
        pl.post.push_back(op);
        if (op.scalar) pe.post_scal.push_back((uint32_t)i);
        if (g.escapes[i]) { pl.post_out.push_back((uint32_t)(i - end)); pe.post_out.push_back((uint32_t)i); }
    }
    pl.extra_post = (uint32_t)pe.extra_leaf.size() - pl.extra_pre;
    if (pe.extra_leaf.size() > MAX_EXTRA) return false;
    // a value of the loop that a later launch used to read (stored every iteration) must not be one the tail needed from an earlier
    // iteration: make_op has rejected those.  Final values are stored only where somebody outside the component reads them.
    for (size_t k = 0; k < ro.final_pos.size(); ++k) { const bool esc = g.escapes[begin + (R - 1) * P + ro.final_pos[k]] != 0; pl.final_store.push_back(esc ? 1u : 0u); pe.final_store.push_back(esc ? 1 : 0); }
    pe.n_pre_scal = (uint32_t)pe.pre_scal.size(); pe.n_post_scal = (uint32_t)pe.post_scal.size(); pe.n_ops = (uint32_t)n;
    const size_t NX = pe.extra_leaf.size(), G = ro.global_leaf.size(), CO = ro.final_pos.size(), NXO = pe.pre_out.size() + pe.post_out.size(), LI = ro.leaf_in.size(), LO = ro.out_pos.size(), LS = ro.scal_pos.size();
    pe.row_words = (uint32_t)(NX + G + CO + NXO + R * (LI + LO) + (pe.n_pre_scal + R * LS + pe.n_post_scal + 1) / 2);
    jit().record(jit_describe(body));
    pe.source = jit_generate_rolled_source(body);
    pe.elems = body.elems;
    pe.present = true;
    // the variant that also takes the moments of the component's root (its last operation), for `chain.getAverage()`
    if (body.elems == 8) {
        if (n > end) pl.reduce = "q" + std::to_string(n - 1 - end);
        else if (final_of[(n - 1 - begin) % P] >= 0) pl.reduce = "F" + std::to_string(final_of[(n - 1 - begin) % P]);
        if (!pl.reduce.empty()) { pe.desc_red = jit_describe(body); jit().record(pe.desc_red); pe.source_red = jit_generate_rolled_source(body);
                                  if (MERGE_CHAINS_ON()) pe.mergeable = merge_shape_index(pe.desc_red) >= 0 ? 1 : 0; }
    }
    ro.peeled = std::move(pe);
    return true;
}

// One launch for the rolled stretch of every member of a group: row tables by index, launch, commit.
void Engine::run_rolled(const BigPlan::Rolled& ro, std::vector<BigDag>& group, size_t first, size_t count)
{
    const size_t G = ro.global_leaf.size(), CI = ro.carried.size(), CO = ro.final_pos.size(), LI = ro.leaf_in.size(), LO = ro.out_pos.size(), LS = ro.scal_pos.size();
    const size_t R = ro.iterations, P = ro.period, rw = ro.row_words;
    const int64_t n = group[first].n;
    std::vector<uint64_t> table(count * rw, 0);
    std::vector<Buffer*> out_bufs;
    out_bufs.reserve(count * (R * LO + CO));
    auto ptr_of = [](const Buffer* b) -> uint64_t {
        if (!b) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "rolled loop reads a value that has not been computed");
        return (uint64_t)(uintptr_t)b->ptr;
    };
    try {
        for (size_t c = 0; c < count; ++c) {
            BigDag& big = group[first + c];
            uint64_t* row = table.data() + c * rw;
            for (size_t k = 0; k < G; ++k) row[k] = ptr_of(big.leaves[(size_t)ro.global_leaf[k]]->buf);
            for (size_t k = 0; k < CI; ++k) row[G + k] = ptr_of(big.value(ro.begin - P + ro.carried[k]));       // the iteration before the loop ran as ordinary launches
            float* sc = reinterpret_cast<float*>(row + G + CI + CO + R * (LI + LO));
            for (size_t r = 0; r < R; ++r) {
                uint64_t* ip = row + G + CI + CO + r * (LI + LO);
                const size_t base = ro.begin + r * P;
                for (size_t m = 0; m < LI; ++m) ip[m] = ptr_of(big.leaves[(size_t)ro.iter_leaf[r * LI + m]]->buf);
                for (size_t m = 0; m < LO; ++m) { Buffer* b = new_buffer(n); out_bufs.push_back(b); ip[LI + m] = (uint64_t)(uintptr_t)b->ptr; }
                for (size_t m = 0; m < LS; ++m) sc[r * LS + m] = big.scalar_at(base + ro.scal_pos[m]);
            }
            for (size_t k = 0; k < CO; ++k) { Buffer* b = new_buffer(n); out_bufs.push_back(b); row[G + CI + k] = (uint64_t)(uintptr_t)b->ptr; }    // after the per-iteration outputs, in this order
        }
        if (n > 0) {
            const int64_t elems_per_pass = (int64_t)FM_BLOCK * ro.jit->elems;
            const int64_t tiles = (n + elems_per_pass - 1) / elems_per_pass;
            DevRolledArgs args{};
            args.n = n; args.tiles_per_row = (uint32_t)tiles; args.row_words = (uint32_t)rw; args.iterations = (uint32_t)R;
            args.dump = (uint64_t)(uintptr_t)dump_dev_;
            const size_t table_bytes = table.size() * 8;
            const size_t ring_off = ring_reserve(table_bytes);
            std::memcpy((char*)ring_host_ + ring_off, table.data(), table_bytes);
            hip_check(hipMemcpyAsync((char*)ring_dev_ + ring_off, (char*)ring_host_ + ring_off, table_bytes, hipMemcpyHostToDevice, stream_), "rolled row table H2D");
            const uint64_t* rows_arg = (const uint64_t*)((char*)ring_dev_ + ring_off);
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (profiling_) { hip_check(hipEventCreate(&ev0), "hipEventCreate"); hip_check(hipEventCreate(&ev1), "hipEventCreate"); hip_check(hipEventRecord(ev0, stream_), "hipEventRecord"); }
            void* params[] = { &args, &rows_arg };
            hip_check(hipModuleLaunchKernel(ro.jit->fn_table, (unsigned)tiles, (unsigned)count, 1, FM_BLOCK, 1, 1, 0, stream_, params, nullptr), "launch rolled kernel");
            if (profiling_) { hip_check(hipEventRecord(ev1, stream_), "hipEventRecord"); profile_events_.push_back({ ev0, ev1 });
                              profile_tags_.push_back({ (int)(R * P), (int)(G + CI + R * LI), (int)(R * LO + CO), 0, (int)count, 2, n }); }
            n_launches_++; n_jit_launches_++; n_rolled_launches_++;
            n_ops_executed_ += (int64_t)(R * P) * (int64_t)count;
            algorithmic_bytes_ += 4 * n * (int64_t)(G + CI + CO + R * (LI + LO)) * (int64_t)count;
            bytes_written_ += 4 * n * (int64_t)(CO + R * LO) * (int64_t)count;
        }
    } catch (...) { for (Buffer* b : out_bufs) buffer_unref(b); throw; }
    // commit (as run_dags): outputs become materialised vectors; their expressions (and the inner values) go away
    size_t k = 0;
    std::vector<Node*> outs;
    outs.reserve(out_bufs.size());
    for (size_t c = 0; c < count; ++c) {
        BigDag& big = group[first + c];
        auto commit = [&](size_t pos) {
            Buffer* b = out_bufs[k++];
            if (big.described()) commit_described(big, pos, b);
            else { Node* nd = big.order[pos]; commit_node(nd, b); outs.push_back(nd); }
        };
        for (size_t r = 0; r < R; ++r)
            for (size_t m = 0; m < LO; ++m) commit(ro.begin + r * P + ro.out_pos[m]);
        for (size_t m = 0; m < CO; ++m) commit(ro.begin + (R - 1) * P + ro.final_pos[m]);
    }
    // (a stored value nobody holds — a position the loop stores in every iteration for the sake of one — goes away with its last consumer:
    // all of them are kept alive until every expression has been dismantled)
    for (Node* nd : outs) nd->refs_int++;
    for (Node* nd : outs) drop_expression(nd);
    for (Node* nd : outs) { nd->refs_int--; node_maybe_free(nd); }
}

// The whole component of every member of a group as ONE launch of its peeled kernel (plan_peel): row tables by index, launch, commit.
static const bool COMMON_ROWS = [] { const char* e = std::getenv("FMHIP_COMMON_ROWS"); return !(e && e[0] == '0'); }();     // =0: identical rows of a launch are all computed (A/B)

void Engine::run_peeled(const BigPlan::Rolled& ro, std::vector<BigDag>& group, size_t first, size_t count, ReduceRequest* rr, std::vector<uint32_t>* row_of_out)
{
    const BigPlan::Rolled::Peeled& pe = ro.peeled;
    const size_t NX = pe.extra_leaf.size(), G = ro.global_leaf.size(), CO = ro.final_pos.size(), NXO = pe.pre_out.size() + pe.post_out.size();
    const size_t LI = ro.leaf_in.size(), LO = ro.out_pos.size(), LS = ro.scal_pos.size(), NS0 = pe.n_pre_scal, NS2 = pe.n_post_scal;
    const size_t R = ro.iterations, P = ro.period, rw = pe.row_words;
    const size_t oG = NX, oCO = oG + G, oXO = oCO + CO, oIT = oXO + NXO;
    const int64_t n = group[first].n;
    std::vector<uint64_t> table(count * rw, 0);
    struct Out { size_t member, pos; Buffer* buf; };
    std::vector<Out> outs;
    outs.reserve(count * (R * LO + CO + NXO));
    auto ptr_of = [](const Buffer* b) -> uint64_t {
        if (!b) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "peeled loop reads a value that has not been computed");
        return (uint64_t)(uintptr_t)b->ptr;
    };
    // COMMON ROWS.  What a row computes is a function of the vectors it reads and of its scalars: members whose rows agree in both —
    // the parameter sets of a Jacobian batch up to the time step at which their bumped parameter is first used, which read the very same
    // vectors because THEIR predecessors were common rows too — are computed once; the others' values are the same vectors (shared
    // storage, copied if anybody writes into one in place: make_private).  A row's inputs and scalars are written first (output slots
    // zero), compared with the rows before it, and only a row that is new gets output vectors.
    std::vector<uint32_t> row_of(count);
    std::vector<size_t> member_of_row;                          // launch row → the member (offset from `first`) that it computes
    const bool dedup = COMMON_ROWS && count > 1 && (!rr || row_of_out);
    std::unordered_multimap<uint64_t, uint32_t> seen;
    std::vector<uint64_t> keys;                                 // the rows as they were compared: inputs and scalars, output slots still zero
    try {
        for (size_t c = 0; c < count; ++c) {
            BigDag& big = group[first + c];
            const size_t r_new = member_of_row.size();
            uint64_t* row = table.data() + r_new * rw;
            std::fill(row, row + rw, (uint64_t)0);
            for (size_t k = 0; k < NX; ++k) row[k] = ptr_of(big.leaves[(size_t)pe.extra_leaf[k]]->buf);
            for (size_t k = 0; k < G; ++k) row[oG + k] = ptr_of(big.leaves[(size_t)ro.global_leaf[k]]->buf);
            float* sc = reinterpret_cast<float*>(row + oIT + R * (LI + LO));
            for (size_t k = 0; k < NS0; ++k) sc[k] = big.scalar_at(pe.pre_scal[k]);
            for (size_t r = 0; r < R; ++r) {
                uint64_t* ip = row + oIT + r * (LI + LO);
                const size_t base = ro.begin + r * P;
                for (size_t m = 0; m < LI; ++m) ip[m] = ptr_of(big.leaves[(size_t)ro.iter_leaf[r * LI + m]]->buf);
                for (size_t m = 0; m < LS; ++m) sc[NS0 + r * LS + m] = big.scalar_at(base + ro.scal_pos[m]);
            }
            for (size_t k = 0; k < NS2; ++k) sc[NS0 + R * LS + k] = big.scalar_at(pe.post_scal[k]);
            if (dedup) {
                uint64_t h = 0x9e3779b97f4a7c15ull;
                for (size_t w = 0; w < rw; ++w) { h = (h ^ row[w]) * 0xff51afd7ed558ccdull; h ^= h >> 31; }
                bool common = false;
                auto range = seen.equal_range(h);
                for (auto it = range.first; it != range.second && !common; ++it)
                    if (std::memcmp(keys.data() + (size_t)it->second * rw, row, rw * 8) == 0) { row_of[c] = it->second; common = true; }
                if (common) { ++n_common_rows_; continue; }
                seen.emplace(h, (uint32_t)r_new);
                keys.insert(keys.end(), row, row + rw);
            }
            row_of[c] = (uint32_t)r_new;
            member_of_row.push_back(c);
            auto fresh = [&](size_t pos) { Buffer* b = new_buffer(n); outs.push_back({ first + c, pos, b }); return (uint64_t)(uintptr_t)b->ptr; };
            for (size_t k = 0; k < CO; ++k) if (pe.final_store[k]) row[oCO + k] = fresh(ro.begin + (R - 1) * P + ro.final_pos[k]);
            for (size_t k = 0; k < pe.pre_out.size(); ++k) row[oXO + k] = fresh(pe.pre_out[k]);
            for (size_t k = 0; k < pe.post_out.size(); ++k) row[oXO + pe.pre_out.size() + k] = fresh(pe.post_out[k]);
            for (size_t r = 0; r < R; ++r) {
                uint64_t* ip = row + oIT + r * (LI + LO);
                for (size_t m = 0; m < LO; ++m) ip[LI + m] = fresh(ro.begin + r * P + ro.out_pos[m]);
            }
        }
        const size_t rows = member_of_row.size();
        table.resize(rows * rw);
        if (n > 0) {
            const int64_t elems_per_pass = (int64_t)FM_BLOCK * pe.jit->elems;
            const int64_t tiles = (n + elems_per_pass - 1) / elems_per_pass;
            DevRolledArgs args{};
            args.n = n; args.tiles_per_row = (uint32_t)tiles; args.row_words = (uint32_t)rw; args.iterations = (uint32_t)R;
            args.dump = (uint64_t)(uintptr_t)dump_dev_;
            const size_t table_bytes = table.size() * 8;
            const bool inline_rows = table.size() <= (size_t)FM_INLINE_WORDS;      // few rows: the table travels in the kernel arguments
            const uint64_t* rows_arg = nullptr;
            if (inline_rows) std::memcpy(args.inline_row, table.data(), table_bytes);
            else {
                const size_t ring_off = ring_reserve(table_bytes);
                std::memcpy((char*)ring_host_ + ring_off, table.data(), table_bytes);
                hip_check(hipMemcpyAsync((char*)ring_dev_ + ring_off, (char*)ring_host_ + ring_off, table_bytes, hipMemcpyHostToDevice, stream_), "peeled row table H2D");
                rows_arg = (const uint64_t*)((char*)ring_dev_ + ring_off);
            }
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (profiling_) { hip_check(hipEventCreate(&ev0), "hipEventCreate"); hip_check(hipEventCreate(&ev1), "hipEventCreate"); hip_check(hipEventRecord(ev0, stream_), "hipEventRecord"); }
            void* params[] = { &args, &rows_arg };
            // Tiles per workgroup: ONE.  Measured (FMHIP_PEEL_TILES = 2 / 4, profiles/round04_peel_tiles_per_workgroup.txt): a workgroup
            // that walks two or four tiles one after the other — a quarter of the partials, arrival counts and lingering keeper waves of a
            // launch that takes the moments of its roots — is SLOWER on every kind of launch (valuation chains 5607 → 5527 → 5424 GB/s,
            // simulation components −2 % and −4 %): these kernels live on the number of independent tiles in flight.  Same moments either
            // way (the reduction tree is defined on the vector, fm_kernel_parts.hpp).
            static const int64_t PEEL_TILES_ENV = [] { const char* e = std::getenv("FMHIP_PEEL_TILES"); const long long v = e ? std::atoll(e) : 0; return (v == 1 || v == 2 || v == 4) ? (int64_t)v : (int64_t)0; }();
            int64_t tiles_per_wg = PEEL_TILES_ENV ? PEEL_TILES_ENV : 1;
            int64_t grid_x = (tiles + tiles_per_wg - 1) / tiles_per_wg;
            if ((tiles + grid_x - 1) / grid_x != tiles_per_wg) { tiles_per_wg = 1; grid_x = tiles; }       // (the kernel derives its stretch from the grid: it must come out as asked)
            RedLaunch red;
            std::vector<fmhip_moments> by_row;                                     // host moments arrive per ROW; the caller's array is per member
            if (rr) {                           // the kernel with the fused reduction of the root (rr->host_out: one entry per member; rr->dev_out: one slot per row)
                fmhip_moments* host_rows = rr->host_out;
                if (rr->host_out && rows != count) { by_row.resize(rows); host_rows = by_row.data(); }
                red_begin(red, (int)rows, 1, (size_t)tiles, host_rows, rr->dev_out);
                args.shift = rr->shift; args.partials = (double*)red.partials; args.results = (double*)red.results; args.counters = counters_dev_;
                args.done_flag = const_cast<uint64_t*>(red.poll_flag); args.done_value = red.done_value;
            }
            try {
                const JitSlot& slot = rr ? *pe.jit_red : *pe.jit;
                hip_check(hipModuleLaunchKernel(inline_rows ? slot.fn_inline : slot.fn_table, (unsigned)grid_x, (unsigned)rows, 1, FM_BLOCK, 1, 1, 0, stream_, params, nullptr), "launch peeled kernel");
                const size_t stored = R * LO + NXO + (size_t)std::count(pe.final_store.begin(), pe.final_store.end(), (char)1);
                if (profiling_) { hip_check(hipEventRecord(ev1, stream_), "hipEventRecord"); profile_events_.push_back({ ev0, ev1 });
                                  profile_tags_.push_back({ (int)pe.n_ops, (int)(NX + G + R * LI), (int)stored, rr ? 1 : 0, (int)rows, 2, n }); }
                n_launches_++; n_jit_launches_++; n_rolled_launches_++;
                n_ops_executed_ += (int64_t)pe.n_ops * (int64_t)rows;
                algorithmic_bytes_ += 4 * n * (int64_t)(NX + G + R * LI + stored) * (int64_t)rows;
                bytes_written_ += 4 * n * (int64_t)stored * (int64_t)rows;
                if (rr) {
                    rr->done = true;
                    if (count == 1 && defer_red_ && !defer_red_->pending && rr->host_out && red.on_host) {
                        red.pending = true; red.batch = 1; red.n_red = 1; red.host = rr->host_out;
                        *defer_red_ = red; red = RedLaunch();      // reduce() waits and releases
                    } else {
                        red_wait(red, (int)rows, 1, by_row.empty() ? rr->host_out : by_row.data());
                        if (!by_row.empty()) for (size_t c = 0; c < count; ++c) rr->host_out[c] = by_row[row_of[c]];
                    }
                }
            } catch (...) { red_release(red); throw; }
            red_release(red);
        } else if (rr) rr = nullptr;
    } catch (...) { for (Out& o : outs) buffer_unref(o.buf); throw; }
    if (row_of_out) *row_of_out = row_of;
    // the members of a common row receive the vectors its first member stored (one more reference each)
    if (member_of_row.size() != count) {
        std::vector<std::pair<size_t, size_t>> span(member_of_row.size(), { 0, 0 });   // per row: its outs [begin, end)
        { size_t k = 0;
          for (size_t r = 0; r < member_of_row.size(); ++r) { const size_t b = k; while (k < outs.size() && outs[k].member == first + member_of_row[r]) ++k; span[r] = { b, k }; } }
        const size_t n_first = outs.size();
        for (size_t c = 0; c < count; ++c) {
            const size_t r = row_of[c];
            if (member_of_row[r] == c) continue;
            for (size_t k = span[r].first; k < span[r].second && k < n_first; ++k) { Buffer* b = outs[k].buf; b->refs++; outs.push_back({ first + c, outs[k].pos, b }); }
        }
    }
    // commit: every stored value becomes a materialised vector; the rest of the component goes away with their expressions
    std::vector<Node*> done;
    done.reserve(outs.size());
    for (Out& o : outs) {
        BigDag& big = group[o.member];
        if (big.described()) commit_described(big, o.pos, o.buf);
        else { Node* nd = big.order[o.pos]; commit_node(nd, o.buf); done.push_back(nd); }
    }
    for (Node* nd : done) nd->refs_int++;
    for (Node* nd : done) drop_expression(nd);
    for (Node* nd : done) { nd->refs_int--; node_maybe_free(nd); }
}

// A component shape with a plan, for every member of a group: segment by segment (the rolled stretch as one launch once its kernel
// exists), ≤ 1024 members per launch.
// The root of a component that has exactly one, as its LAST operation — its node (for a copy that exists as a description: the copy's
// root node) or nullptr.  g0 = the member that carries the order.
Node* Engine::single_root(const BigDag& b, const BigDag& g0)
{
    if (!b.described()) return (b.roots.size() == 1 && !b.order.empty() && b.order.back() == b.roots[0]) ? b.roots[0] : nullptr;
    const std::vector<int32_t>& rep_root = b.view->rep_root;
    const size_t n = g0.order.size();
    if (rep_root.size() != n || n == 0 || rep_root[n - 1] < 0) return nullptr;
    for (size_t i = 0; i + 1 < n; ++i) if (rep_root[i] >= 0) return nullptr;
    const ReplicaGroup* g = b.view->g;
    return g->copy_roots[(size_t)b.copy * g->n_roots + (size_t)rep_root[n - 1]];
}

void Engine::run_plan(BigPlan& plan, std::vector<BigDag>& group, ReduceRequest* rr) {
    const size_t max_batch = 1024;
    { static const bool batch_trace = std::getenv("FMHIP_BATCH_TRACE") != nullptr;
      if (batch_trace && want_root_moments_) std::fprintf(stderr, "[fmhip batch] plan for a group of %zu, %zu nodes, %zu roots: rolled %d peeled %d segs %zu\n", group.size(), group[0].order.size(), group[0].roots.size(), plan.rolled.present ? 1 : 0, plan.rolled.peeled.present ? 1 : 0, plan.segs.size()); }
    if (plan.rolled.present && jit_mode != FMHIP_JIT_OFF && (!plan.rolled.jit || (jit_mode == FMHIP_JIT_SYNC && plan.rolled.jit->state.load(std::memory_order_acquire) == JitSlot::QUEUED)))
        plan.rolled.jit = jit().request_source(plan.rolled.source, plan.rolled.elems, jit_mode == FMHIP_JIT_SYNC);
    bool rolled = plan.rolled.present && jit_mode != FMHIP_JIT_OFF && plan.rolled.jit && plan.rolled.jit->state.load(std::memory_order_acquire) == JitSlot::READY;
    // The rolled launch carries one row table for all its members through the pinned ring (≈ 5 KB per row for the LMM step, but
    // iterations x (inputs + outputs) words in general): as many members per launch as fit; a single row that does not fit leaves the
    // stretch to its segments.
    size_t rolled_batch = max_batch;
    if (rolled) {
        const size_t rows_fit = ring_cap_ / ((size_t)plan.rolled.row_words * 8 + 256);
        if (rows_fit == 0) rolled = false; else rolled_batch = std::min(max_batch, rows_fit);
    }
    // The peeled form — head, loop and tail of the component in one launch.  A caller that values one product after the other (1 row x
    // 489 tiles) turns three launches of ≈ 9 + 20 + 8 µs into one; the lock-step batches of the native driver save the stores and loads
    // of the values between the parts (measured on the 1 M-path calibration, same box: 12 671 → 4 639 launches, 2.65 → 2.50 s of
    // kernel time with the head and the tail limited to 128 operations each).  FMHIP_PEEL_MAX_WORKGROUPS limits it to small launches.
    {
        BigPlan::Rolled::Peeled& pe = plan.rolled.peeled;
        if (pe.present && jit_mode != FMHIP_JIT_OFF && (!pe.jit || (jit_mode == FMHIP_JIT_SYNC && pe.jit->state.load(std::memory_order_acquire) == JitSlot::QUEUED)))
            pe.jit = jit().request_source(pe.source, pe.elems, jit_mode == FMHIP_JIT_SYNC);
        if (pe.present && jit_mode != FMHIP_JIT_OFF && pe.jit && pe.jit->state.load(std::memory_order_acquire) == JitSlot::READY && group[0].n > 0) {
            static const size_t PEEL_MAX_WORKGROUPS = [] { const char* e = std::getenv("FMHIP_PEEL_MAX_WORKGROUPS"); return e ? (size_t)std::atoll(e) : ~(size_t)0; }();
            const size_t tiles = (size_t)((group[0].n + (int64_t)FM_BLOCK * pe.jit->elems - 1) / ((int64_t)FM_BLOCK * pe.jit->elems));
            const size_t rows_fit = ring_cap_ / ((size_t)pe.row_words * 8 + 256);
            if (group.size() * tiles <= PEEL_MAX_WORKGROUPS && rows_fit >= group.size()) {
                struct TempGuard2 { Engine* e; std::vector<BigDag>& g; ~TempGuard2() { for (BigDag& b : g) if (b.described()) for (Buffer*& t : b.temp) if (t) { e->buffer_unref(t); t = nullptr; } } } guard2{ this, group };
                // `chain.getAverage()` on the component's root: the same launch takes the moments (a workgroup's tile is one unit of the
                // reduction tree) — a product of a caller that values one after the other is ONE launch, and its value is not read again
                ReduceRequest* fused = nullptr;
                if (rr && group.size() == 1 && !group[0].described() && !pe.source_red.empty() && group[0].order.back() == group[0].roots[0] &&
                    tiles <= (size_t)FM_SPAN_UNITS * 65536) {
                    if (!pe.jit_red || (jit_mode == FMHIP_JIT_SYNC && pe.jit_red->state.load(std::memory_order_acquire) == JitSlot::QUEUED))
                        pe.jit_red = jit().request_source(pe.source_red, pe.elems, jit_mode == FMHIP_JIT_SYNC);
                    if (pe.jit_red->state.load(std::memory_order_acquire) == JitSlot::READY) fused = rr;
                }
                // … and with a flush that collects the moments of all pending roots (Engine::reduce): of every member, as rows of this launch
                std::vector<fmhip_moments> all;
                ReduceRequest every{ 0.0, nullptr, nullptr, false };
                static const bool batch_trace = std::getenv("FMHIP_BATCH_TRACE") != nullptr;
                if (batch_trace && want_root_moments_) std::fprintf(stderr, "[fmhip batch] peeled group of %zu, %zu nodes: rr %d source_red %d roots %zu root_last %d\n", group.size(), group[0].order.size(), rr ? 1 : 0, pe.source_red.empty() ? 0 : 1, group[0].roots.size(), (int)(!group[0].order.empty() && !group[0].roots.empty() && group[0].order.back() == group[0].roots[0]));
                if (!fused && want_root_moments_ && !rr && !pe.source_red.empty() && tiles <= (size_t)FM_SPAN_UNITS * 65536) {
                    bool roots_only = true;
                    for (const BigDag& b : group) { Node* r = single_root(b, group[0]); roots_only &= r != nullptr && !r->moments_blocked && (!plan.discards_root || r->discard); }
                    if (roots_only) {
                        if (!pe.jit_red || (jit_mode == FMHIP_JIT_SYNC && pe.jit_red->state.load(std::memory_order_acquire) == JitSlot::QUEUED))
                            pe.jit_red = jit().request_source(pe.source_red, pe.elems, jit_mode == FMHIP_JIT_SYNC);
                        if (pe.jit_red->state.load(std::memory_order_acquire) == JitSlot::READY) {
                            if (async_moments_) { every.dev_out = arena_alloc(group.size()); if (every.dev_out) fused = &every; }
                            else { all.resize(group.size()); every.host_out = all.data(); fused = &every; }
                        }
                    }
                }
                // A plan whose root is wanted for its moments only has peeled kernels that do not store it: usable only when this launch
                // takes the moments of every member's root; otherwise the segments below run (they go by the nodes' references and store it)
                if (!plan.discards_root || fused == &every) {
                std::vector<Node*> root_nodes;
                if (fused == &every) for (const BigDag& b : group) root_nodes.push_back(single_root(b, group[0]));
                std::vector<uint32_t> row_of;
                run_peeled(plan.rolled, group, 0, group.size(), fused, &row_of);
                if (fused == &every && every.done && every.dev_out)
                    for (size_t i = 0; i < root_nodes.size(); ++i) arena_assign(root_nodes[i], (double*)every.dev_out + (size_t)row_of[i] * 4);      // (members of a common row: the same slot)
                else if (fused == &every && every.done)
                    for (size_t i = 0; i < root_nodes.size(); ++i) {
                        Node* r = root_nodes[i];
                        r->moments[0] = all[i].sum; r->moments[1] = all[i].sumsq; r->moments[2] = all[i].min; r->moments[3] = all[i].max; r->has_moments = true;
                    }
                if (plan.discards_root && every.done)       // moments taken, value not stored: not a root of later flushes; what it was computed from is let go
                    for (Node* r : root_nodes) if (!r->buf) { r->discarded = true; r->refs_int++; drop_expression(r); r->refs_int--; }
                return;
                }
            }
        }
    }
    if (plan.segs_missing) {
        // this shape has never run as segments: the members with nodes take the general path now, which writes the cuts into the plan
        std::vector<BigDag> described, with_nodes;
        for (BigDag& b : group) (b.described() ? described : with_nodes).push_back(std::move(b));
        if (with_nodes.empty()) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "a group of copies without their original");
        plan_segments(plan, with_nodes);
        plan.segs_missing = false;
        for (BigPlan::Seg& seg : plan.segs) seg.prog->refs++;              // the plan holds its programs (pool_purge drops both caches together)
        if (!described.empty()) run_plan(plan, described);
        return;
    }
    bool any_described = false;
    for (const BigDag& b : group) any_described |= b.described();
    auto release_temps = [&](const std::vector<int32_t>& positions) {
        if (!any_described) return;
        for (BigDag& b : group) {
            if (!b.described()) continue;
            for (int32_t pos : positions) if (Buffer* t = b.temp[(size_t)pos]) { b.temp[(size_t)pos] = nullptr; buffer_unref(t); }
        }
    };
    struct TempGuard {              // whatever happens, the values held for members without nodes go back to the pool
        Engine* e; std::vector<BigDag>& g;
        ~TempGuard() { for (BigDag& b : g) if (b.described()) for (Buffer*& t : b.temp) if (t) { e->buffer_unref(t); t = nullptr; } }
    } guard{ this, group };
    // The variant of the last segment that also reduces the component's root, for `chain.getAverage()` on a single large expression: one
    // launch and one read of the root less than a stand-alone reduction behind the segment.  Only when that variant accumulates like
    // the stand-alone reduction program does (8 elements per lane: the sums are then the same to the last bit).
    Program* prog_red = nullptr;
    if (rr && group.size() == 1 && !group[0].described() && !plan.segs.empty()) {
        BigPlan::Seg& last = plan.segs.back();
        const int32_t root_pos = (int32_t)group[0].order.size() - 1;
        size_t k_root = last.out.size();
        for (size_t k = 0; k < last.out.size(); ++k) if (last.out[k] == root_pos) k_root = k;
        // (the size gate of reduce(): a launch with a fused reduction has one workgroup per 8192 elements of a row — fine for a segment
        // over two input vectors, a starved launch for one over eleven)
        if (!(rolled && last.zone == 1) && !last.no_red && !last.ssa.empty() && k_root < last.out.size() && group[0].order[(size_t)root_pos] == group[0].roots[0] &&
            (group[0].n * (int64_t)last.n_in <= (int64_t(1) << 21) || unit_launch(group[0].n, 1))) {
            if (!last.prog_red) {
                try { last.prog_red = compile(last.ssa, last.n_in, last.out_ids, { last.out_ids[k_root] }, nullptr, false); }
                catch (const Error& e) { if (e.code != FMHIP_ERR_PROGRAM_LIMIT) throw; last.no_red = true; }
                if (last.prog_red && last.prog_red->proto.variant != 1u) { delete last.prog_red; last.prog_red = nullptr; last.no_red = true; }
            }
            prog_red = last.prog_red;
        }
    }
    bool rolled_done = false;
    for (size_t si = 0; si < plan.segs.size(); ++si) {
        const BigPlan::Seg& seg = plan.segs[si];
        if (rolled && seg.zone == 1) {              // the loop's stretch: one launch of the rolled kernel instead of its segments
            if (!rolled_done) for (size_t off = 0; off < group.size(); off += rolled_batch) run_rolled(plan.rolled, group, off, std::min(rolled_batch, group.size() - off));
            rolled_done = true;
        } else if (prog_red && si + 1 == plan.segs.size())
            run_planned_segment(seg, group, 0, 1, rr, prog_red);
        else
            for (size_t off = 0; off < group.size(); off += max_batch) run_planned_segment(seg, group, off, std::min(max_batch, group.size() - off));
        release_temps(seg.free_after);
    }
}

// The general path: the members WITH nodes of a group whose shape has no segments yet are cut into launches (longest segment that fits
// one launch, again and again; cuts forced at the ends of the loop plan.rolled describes, so that either form can run between them), run
// segment by segment, and the cuts are written into the plan.
void Engine::plan_segments(BigPlan& plan, std::vector<BigDag>& group) {
    const size_t n_ops = group[0].order.size();
    const size_t max_batch = 1024;
    // Node -> index in group[0] for the plan (segment_dag reuses the nodes' scratch fields)
    std::unordered_map<const Node*, int32_t> index_of;
    index_of.reserve(n_ops + group[0].leaves.size());
    for (size_t i = 0; i < n_ops; ++i) index_of[group[0].order[i]] = (int32_t)i;
    for (size_t i = 0; i < group[0].leaves.size(); ++i) index_of[group[0].leaves[i]] = -1 - (int32_t)i;
    std::vector<int32_t> uses(n_ops, 0);                    // consumers inside the component, per position (before anything runs)
    for (size_t i = 0; i < n_ops; ++i)
        for (int k = 0; k < group[0].order[i]->n_in; ++k) { const int32_t o = index_of.at(group[0].order[i]->in[k]); if (o >= 0) uses[(size_t)o]++; }
    size_t zone_begin = n_ops, zone_end = n_ops;
    if (plan.rolled.present) { zone_begin = plan.rolled.begin; zone_end = zone_begin + (size_t)plan.rolled.period * plan.rolled.iterations; }
    size_t s = 0;
    while (s < n_ops) {
        size_t e = 0;
        const size_t limit = s < zone_begin ? zone_begin : (s < zone_end ? zone_end : n_ops);      // a segment never crosses an end of the loop
        {
            // longest segment starting at s that fits one launch.  Inputs grow monotonically with the end; outputs and live
            // values do not: collect every end that passes the cheap checks, then take the largest one that also compiles.
            std::vector<size_t> cheap;
            for (size_t cand = s + 1; cand <= limit && cand - s <= (size_t)FM_MAX_OPS; ++cand) {
                Dag d;
                if (segment_dag(group[0], s, cand, d, uses)) cheap.push_back(cand);
                else if ((int)d.leaves.size() > FM_MAX_IN) break;
            }
            for (size_t k = cheap.size(); k-- > 0 && e == 0;) {
                Dag d;
                segment_dag(group[0], s, cheap[k], d, uses);
                if (!program_cache_.count(d.sig)) {
                    try { program_cache_[d.sig] = compile(d.ops, (int)d.leaves.size(), d.out_ids, {}, nullptr, false); }
                    catch (const Error& err) { if (err.code == FMHIP_ERR_PROGRAM_LIMIT) continue; throw; }
                }
                e = cheap[k];
            }
            if (e == 0) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "an operation does not fit one launch");
        }
        std::vector<Dag> dags(group.size());
        for (size_t c = 0; c < group.size(); ++c)
            if (!segment_dag(group[c], s, e, dags[c], uses)) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "inconsistent segment of a split component");
        {
            BigPlan::Seg seg;
            seg.prog = program_cache_.at(dags[0].sig);
            for (Node* l : dags[0].leaves) seg.in.push_back(index_of.at(l));
            for (Node* o : dags[0].outs) seg.out.push_back(index_of.at(o));
            for (size_t i = s; i < e; ++i) if (op_info(group[0].order[i]->opcode).scalar) seg.scal.push_back((int32_t)i);
            seg.zone = s < zone_begin ? 0 : (s < zone_end ? 1 : 2);
            if (e == n_ops) { seg.ssa = dags[0].ops; seg.out_ids = dags[0].out_ids; seg.n_in = (int)dags[0].leaves.size(); }
            plan.segs.push_back(std::move(seg));
        }
        for (size_t off = 0; off < dags.size(); off += max_batch) {
            std::vector<Dag> part(std::make_move_iterator(dags.begin() + off), std::make_move_iterator(dags.begin() + std::min(dags.size(), off + max_batch)));
            if (!run_dags(part)) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "segment of a split component does not fit one launch");
        }
        s = e;
    }
    // values no later segment reads: members without nodes give them back to the pool there
    {
        std::unordered_map<int32_t, size_t> last_reader;
        for (size_t k = 0; k < plan.segs.size(); ++k) for (int32_t i : plan.segs[k].in) if (i >= 0) last_reader[i] = k;
        for (const auto& kv : last_reader) plan.segs[kv.second].free_after.push_back(kv.first);
    }
}

// The loop of a component shape (detect_loop) and its peeled form (plan_peel), their kernels asked of the specialised tier; nothing runs.
void Engine::plan_loop(BigPlan& plan, const BigDag& g) {
    static const bool ROLL = [] { const char* e = std::getenv("FMHIP_ROLL"); return !(e && e[0] == '0'); }();
    if (!ROLL) return;
    const size_t n_ops = g.order.size();
    std::vector<std::array<int32_t, 3>> operand(n_ops);      // (build_big left every node's position, resp. -1 - leaf number, in tmp_id)
    std::unordered_map<const Node*, int32_t> index_of;
    index_of.reserve(n_ops + g.leaves.size());
    for (size_t i = 0; i < n_ops; ++i) index_of[g.order[i]] = (int32_t)i;
    for (size_t i = 0; i < g.leaves.size(); ++i) index_of[g.leaves[i]] = -1 - (int32_t)i;
    for (size_t i = 0; i < n_ops; ++i)
        for (int k = 0; k < g.order[i]->n_in; ++k) operand[i][(size_t)k] = index_of.at(g.order[i]->in[k]);
    std::string source; int elems = 0;
    RolledBody body;
    if (!detect_loop(g, operand, plan.rolled, &source, &elems, &body)) return;
    if (plan_peel(g, operand, plan.rolled, body) && jit_mode != FMHIP_JIT_OFF)
        plan.rolled.peeled.jit = jit().request_source(plan.rolled.peeled.source, plan.rolled.peeled.elems, jit_mode == FMHIP_JIT_SYNC);
    if (const char* dump = std::getenv("FMHIP_ROLL_DUMP")) { if (FILE* f = std::fopen(dump, "a")) { std::fputs(source.c_str(), f); std::fputs("\n// ----\n", f); std::fclose(f); } }
    plan.rolled.present = true;
    plan.rolled.source = source; plan.rolled.elems = elems;
    if (jit_mode != FMHIP_JIT_OFF) plan.rolled.jit = jit().request_source(std::move(source), elems, jit_mode == FMHIP_JIT_SYNC);
}

// ---------------------------------------------------------------- merged chains (runtime.hpp: merge_families; jit.hpp: RolledBody::chains)
//
// The 14 swaptions of one exercise date are 14 components of the same loop shape and different length — 14 launches by shape (each with the
// other exercise dates' swaptions of that tenor as its rows), every one of which reads the forward rates of its tenor: L_e[e] 14 times,
// L_e[e + 19] five times, 304 vector reads per exercise date where 61 vectors exist.  A FAMILY is a set of such components, found by
// their vectors: same shape of head, body and tail, the same tail inputs, and the head + loop inputs of each a suffix of the longest
// one's.  One launch per (shape, family size): a row per family, a step per vector of the longest chain, every chain joining at its own
// first step; per chain the same operations on the same operands in the same order as in its own launch, and its moments by the same
// tree — bit-identical results, a fifth of the bytes.  Nothing is assumed about the caller: the family is read off the pending graph.
static const bool MERGE_CHAINS = MERGE_CHAINS_ON();         // FMHIP_MERGE_CHAINS=0: off
static const bool MERGE_SMALL = [] { const char* e = std::getenv("FMHIP_MERGE_SMALL"); return !(e && e[0] == '0'); }();      // =0: components that fit one launch never join a family

struct Engine::SmallGroup { std::vector<Dag> members; Dag proto; const SmallMatch* match = nullptr; std::vector<std::pair<ReplicaGroup*, std::vector<int>>> done; };

// Is this single-launch component, position by position, head + R iterations of the body + tail of a mergeable loop shape?  (The walk
// that lists its operations — depth first from the root, operands in order — and the schedule of the large components of the same shape
// list a chain the same way; where they do not, the answer is no and the component runs on its own as before.)
const Engine::SmallMatch* Engine::match_small(const Dag& d)
{
    if (!MERGE_SMALL || merge_shapes_.empty()) return nullptr;
    auto known = small_match_.find(d.sig);
    if (known != small_match_.end()) {
        if (known->second.ok) return &known->second;
        if (known->second.shape == (int)merge_shapes_.size()) return nullptr;      // (no, with every shape known today)
    }
    if (small_match_.size() > 4096) small_match_.clear();
    SmallMatch& out = small_match_[d.sig];
    out = SmallMatch();
    out.shape = (int)merge_shapes_.size();                      // (looked at with these shapes known: asked again when another one appears)
    const size_t m = d.order.size(), n_in = d.leaves.size();
    if (d.roots.size() != 1 || d.outs.size() != 1 || m == 0 || d.order.back() != d.roots[0] || d.ops.size() != m) return nullptr;
    for (size_t si = 0; si < merge_shapes_.size() && !out.ok; ++si) {
        const RolledBody& B = merge_shape_bodies_[si];
        const RolledBody::Peel& PL = B.peel;
        const size_t n_pre = PL.pre.size(), n_post = PL.post.size(), P = B.ops.size(), NXa = PL.extra_pre, NXP = PL.extra_post;
        if (m < n_pre + n_post || (m - n_pre - n_post) % P != 0) continue;
        // the tail stores nothing but the component's root, or nothing at all
        if (!(PL.post_out.empty() || (PL.post_out.size() == 1 && PL.post_out[0] + 1 == n_post)) || PL.reduce != "q" + std::to_string(n_post - 1)) continue;
        const size_t R = (m - n_pre - n_post) / P;
        std::vector<int> seq(NXa + R, -1), post(NXP, -1);
        bool ok = true;
        // an operand by name → what it must be here: position of an operation (>= 0), or a sequence / tail vector (checked against the leaf)
        auto check = [&](const RolledBody::Op& op, size_t i, auto&& resolve) {
            const SsaOp& a = d.ops[i];
            UVariant uv{};
            if (!variant_for(a.opcode, 0, &uv)) return false;
            uint32_t uop = uv.uop;
            if (math_mode == FMHIP_MATH_FAST) { if (uop == U_EXP) uop = U_EXP_FAST; else if (uop == U_LOG) uop = U_LOG_FAST; }
            if (uop != op.uop || op_info(a.opcode).scalar != op.scalar) return false;
            const int ids[3] = { a.a, a.b, a.c };
            const std::string* names[3] = { &op.x0, &op.x1, &op.x2 };
            const int pos[3] = { 0, uv.r1_pos, uv.r2_pos };
            for (int k = 0; k < 3; ++k) {
                if (names[k]->empty()) { if (k > 0 && pos[k] >= 0) return false; continue; }
                if (pos[k] < 0 || ids[pos[k]] < 0) return false;
                if (!resolve(*names[k], ids[pos[k]])) return false;
            }
            return true;
        };
        auto is_op = [&](int id, size_t position) { return id >= (int)n_in && (size_t)(id - (int)n_in) == position; };
        auto is_leaf = [&](int id, int& slot) { if (id < 0 || id >= (int)n_in) return false; if (slot < 0) slot = id; return slot == id; };
        for (size_t i = 0; i < n_pre && ok; ++i)
            ok = check(PL.pre[i], i, [&](const std::string& nm, int id) {
                const size_t idx = (size_t)std::atoi(nm.c_str() + 1);
                if (nm[0] == 'x') return idx < NXa && is_leaf(id, seq[idx]);
                if (nm[0] == 'p') return idx < i && is_op(id, idx);
                return false; });
        for (size_t r = 0; r < R && ok; ++r)
            for (size_t q = 0; q < P && ok; ++q) {
                const size_t base = n_pre + r * P;
                ok = check(B.ops[q], base + q, [&](const std::string& nm, int id) {
                    const size_t idx = (size_t)std::atoi(nm.c_str() + 1);
                    if (nm[0] == 'v') return idx < q && is_op(id, base + idx);
                    if (nm[0] == 'c') {
                        if (idx >= B.carried.size()) return false;
                        if (r > 0) return is_op(id, base - P + B.carried[idx]);
                        const std::string& init = PL.carried_init[idx];
                        return init[0] == 'p' && is_op(id, (size_t)std::atoi(init.c_str() + 1)); }
                    if (nm == "l0") return is_leaf(id, seq[NXa + r]);
                    return false; });
            }
        for (size_t i = 0; i < n_post && ok; ++i) {
            const size_t base = n_pre + R * P;
            ok = check(PL.post[i], base + i, [&](const std::string& nm, int id) {
                const size_t idx = (size_t)std::atoi(nm.c_str() + 1);
                if (nm[0] == 'q') return idx < i && is_op(id, base + idx);
                if (nm[0] == 'x') return idx >= NXa && idx < NXa + NXP && is_leaf(id, post[idx - NXa]);
                if (nm[0] == 'F') {
                    if (idx >= B.final_pos.size()) return false;
                    if (R > 0) return is_op(id, base - P + B.final_pos[idx]);
                    for (size_t c = 0; c < B.carried.size(); ++c)
                        if (B.carried[c] == B.final_pos[idx]) { const std::string& init = PL.carried_init[c]; return init[0] == 'p' && is_op(id, (size_t)std::atoi(init.c_str() + 1)); }
                    return false; }
                return false; });
        }
        // every vector of the sequence is a vector of its own step (the kernel loads one per step), every leaf is accounted for
        for (int v : seq) ok = ok && v >= 0;
        for (int v : post) ok = ok && v >= 0;
        if (ok) { std::vector<int> all(seq); all.insert(all.end(), post.begin(), post.end()); std::sort(all.begin(), all.end()); ok = all.size() == n_in && std::adjacent_find(all.begin(), all.end()) == all.end(); }
        if (!ok) continue;
        out.ok = true; out.shape = (int)si; out.R = (uint32_t)R;
        out.seq_leaf.assign(seq.begin(), seq.end()); out.post_leaf.assign(post.begin(), post.end());
    }
    if (!out.ok) { out.shape = (int)merge_shapes_.size(); return nullptr; }
    return &out;
}

// The number of a mergeable loop shape (by its description), registered at its first sight — when its plan is made (plan_peel), so that
// single-launch components of the flush after can be recognised as its chains; -1: the shape has no merged form.
int Engine::merge_shape_index(const std::string& desc)
{
    if (desc.empty()) return -1;
    for (size_t i = 0; i < merge_shapes_.size(); ++i) if (merge_shapes_[i] == desc) return (int)i;
    RolledBody body;
    if (!jit_parse_description(desc, body)) return -1;
    RolledBody probe = body; probe.chains = 2; probe.shared_den = true;
    if (jit_generate_rolled_source(probe).empty()) return -1;
    merge_shapes_.push_back(desc); merge_shape_bodies_.push_back(std::move(body));
    return (int)merge_shapes_.size() - 1;
}

void Engine::merge_families(std::vector<std::vector<BigDag>>& groups, std::vector<SmallGroup>& small)
{
    if (!MERGE_CHAINS || !want_root_moments_ || jit_mode == FMHIP_JIT_OFF) return;
    // a chain: a large component (group, member: its plan says where its vectors and scalars are) or a small one (sgroup, smember: its match does)
    struct Chain { size_t group, member; BigPlan* plan; int shape; Node* root; uint32_t steps, R; const float* last; bool small; };
    std::vector<Chain> chains;
    auto shape_index = [&](BigPlan::Rolled::Peeled& pe) -> int { return merge_shape_index(pe.desc_red); };
    std::vector<BigPlan*> plan_of_shape;                         // a plan of every shape met in this flush (what its large chains are described by)
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        std::vector<BigDag>& g = groups[gi];
        if (g.empty() || g[0].described() || g[0].n <= 0) continue;
        auto planned = plan_cache_.find(g[0].hash);
        if (planned == plan_cache_.end() || planned->second.sig != g[0].sig) continue;
        BigPlan& plan = planned->second;
        BigPlan::Rolled& ro = plan.rolled;
        BigPlan::Rolled::Peeled& pe = ro.peeled;
        if (!ro.present || !pe.present || pe.desc_red.empty() || pe.elems != 8) continue;
        if (pe.mergeable < 0) pe.mergeable = shape_index(pe) >= 0 ? 1 : 0;
        if (!pe.mergeable) continue;
        const int shape = shape_index(pe);
        if (shape < 0) continue;
        if ((size_t)((g[0].n + FM_UNIT_ELEMS - 1) / FM_UNIT_ELEMS) > (size_t)FM_SPAN_UNITS * 65536) continue;
        if (plan_of_shape.size() <= (size_t)shape) plan_of_shape.resize((size_t)shape + 1, nullptr);
        if (!plan_of_shape[(size_t)shape]) plan_of_shape[(size_t)shape] = &plan;
        const uint32_t steps = (uint32_t)(merge_shape_bodies_[(size_t)shape].peel.extra_pre + ro.iterations);
        for (size_t mi = 0; mi < g.size(); ++mi) {
            const BigDag& b = g[mi];
            Node* r = single_root(b, g[0]);
            if (!r || r->moments_blocked || (plan.discards_root && !r->discard) || r->buf) continue;
            const int32_t last_leaf = ro.iter_leaf[(size_t)(ro.iterations - 1) * ro.leaf_in.size()];
            const Buffer* lb = b.leaves[(size_t)last_leaf]->buf;
            if (!lb) continue;
            chains.push_back({ gi, mi, &plan, shape, r, steps, ro.iterations, lb->ptr, false });
        }
    }
    for (size_t gi = 0; gi < small.size(); ++gi) {
        SmallGroup& sg = small[gi];
        if (!sg.match || sg.members.empty()) continue;
        const int shape = sg.match->shape;
        if ((size_t)shape >= plan_of_shape.size() || !plan_of_shape[(size_t)shape]) continue;      // no large chain of this shape in this flush: nobody to join
        const RolledBody& body = merge_shape_bodies_[(size_t)shape];
        const bool stores_root = !body.peel.post_out.empty();
        const uint32_t steps = (uint32_t)(body.peel.extra_pre + sg.match->R);
        for (size_t mi = 0; mi < sg.members.size(); ++mi) {
            const Dag& d = sg.members[mi];
            if (d.outs.size() != 1 || d.leaves.size() != sg.proto.leaves.size() || d.outs[0]->n <= 0) continue;
            Node* r = d.outs[0];
            // (as run_dags: a root that is held; given up — and held by nobody else, a copy's root but by its group — exactly when the shape stores nothing)
            const bool given_up = r->discard && r->refs_int == ((r->rep_id && r->rep_copy && replica_of(r)) ? 1 : 0);
            if (r->moments_blocked || r->buf || r->refs_ext <= 0 || (stores_root ? r->discard : !given_up)) continue;
            const Buffer* lb = d.leaves[(size_t)sg.match->seq_leaf.back()]->buf;
            if (!lb) continue;
            chains.push_back({ gi, mi, plan_of_shape[(size_t)shape], shape, r, steps, sg.match->R, lb->ptr, true });
        }
    }
    if (chains.size() < 2) return;
    struct Shape { const RolledBody* body = nullptr; size_t NXa = 0, NXP = 0, NS0 = 0, NS2 = 0, LS = 0, NXO = 0, P = 0; std::vector<uint32_t> shared_pre, shared_body; };
    std::unordered_map<int, Shape> shapes;
    auto shape_of = [&](int shape) -> Shape& {
        auto it = shapes.find(shape);
        if (it != shapes.end()) return it->second;
        Shape& sh = shapes[shape];
        sh.body = &merge_shape_bodies_[(size_t)shape];
        sh.NXa = sh.body->peel.extra_pre; sh.NXP = sh.body->peel.extra_post; sh.NXO = sh.body->peel.post_out.size(); sh.P = sh.body->ops.size();
        for (const RolledBody::Op& op : sh.body->peel.pre) sh.NS0 += op.scalar ? 1 : 0;
        for (const RolledBody::Op& op : sh.body->peel.post) sh.NS2 += op.scalar ? 1 : 0;
        for (const RolledBody::Op& op : sh.body->ops) sh.LS += op.scalar ? 1 : 0;
        jit_merged_shared_scalars(*sh.body, sh.shared_pre, sh.shared_body);
        return sh;
    };
    auto vec_ptr = [](const Node* leaf) -> const float* { return leaf->buf ? leaf->buf->ptr : nullptr; };
    auto chain_n = [&](const Chain& c) -> int64_t { return c.small ? small[c.group].members[c.member].outs[0]->n : groups[c.group][c.member].n; };
    // the vector a chain reads at step i of its own sequence (head inputs first, then one per iteration); the vectors of its tail
    auto seq_ptr = [&](const Chain& c, const Shape& sh, size_t i) -> const float* {
        if (c.small) { const SmallGroup& sg = small[c.group]; return vec_ptr(sg.members[c.member].leaves[(size_t)sg.match->seq_leaf[i]]); }
        const BigDag& b = groups[c.group][c.member];
        const BigPlan::Rolled& ro = c.plan->rolled;
        return vec_ptr(b.leaves[(size_t)(i < sh.NXa ? ro.peeled.extra_leaf[i] : ro.iter_leaf[(i - sh.NXa) * ro.leaf_in.size()])]);
    };
    auto post_ptr = [&](const Chain& c, const Shape& sh, size_t x) -> const float* {
        if (c.small) { const SmallGroup& sg = small[c.group]; return vec_ptr(sg.members[c.member].leaves[(size_t)sg.match->post_leaf[x]]); }
        return vec_ptr(groups[c.group][c.member].leaves[(size_t)c.plan->rolled.peeled.extra_leaf[sh.NXa + x]]);
    };
    // scalar number i of the chain's head / of iteration `it` / of its tail, in the order of the operations
    auto pre_scalar = [&](const Chain& c, const Shape&, size_t i) -> float {
        if (c.small) return small[c.group].members[c.member].scalars[i];
        return groups[c.group][c.member].scalar_at(c.plan->rolled.peeled.pre_scal[i]);
    };
    auto body_scalar = [&](const Chain& c, const Shape& sh, size_t it, size_t i) -> float {
        if (c.small) return small[c.group].members[c.member].scalars[sh.NS0 + it * sh.LS + i];
        const BigPlan::Rolled& ro = c.plan->rolled;
        return groups[c.group][c.member].scalar_at(ro.begin + it * ro.period + ro.scal_pos[i]);
    };
    auto post_scalar = [&](const Chain& c, const Shape& sh, size_t i) -> float {
        if (c.small) return small[c.group].members[c.member].scalars[sh.NS0 + (size_t)c.R * sh.LS + i];
        return groups[c.group][c.member].scalar_at(c.plan->rolled.peeled.post_scal[i]);
    };
    // candidates by (shape, last vector): longest first
    std::sort(chains.begin(), chains.end(), [&](const Chain& a, const Chain& b) {
        if (a.shape != b.shape) return a.shape < b.shape;
        if (a.last != b.last) return a.last < b.last;
        if (a.steps != b.steps) return a.steps > b.steps;
        if (a.small != b.small) return !a.small;
        if (a.group != b.group) return a.group < b.group;
        return a.member < b.member;
    });
    struct Family { std::vector<size_t> chain; };          // indices into `chains`, longest first
    std::vector<Family> families;
    for (size_t i = 0; i < chains.size();) {
        size_t j = i + 1;
        while (j < chains.size() && chains[j].last == chains[i].last && chains[j].shape == chains[i].shape) ++j;
        // [i, j): same shape, same last vector.  Those whose whole sequence is a suffix of the longest one's and whose tail inputs and
        // shared scalars agree with it form families of at most 16 (a family of small components only has nobody to carry it: skipped).
        // Chains of the SAME length that end in the same vector — the same product valued for several parameter sets whose simulations
        // were common rows up to this exercise date (run_peeled) — belong to different families: the m-th chain of every length forms
        // layer m; the layers are rows of one launch (and, reading the same vectors with the same scalars, one common row of it).
        std::vector<std::vector<size_t>> layers;
        { size_t occurrence = 0;
          for (size_t q = i; q < j; ++q) {
              occurrence = (q > i && chains[q].steps == chains[q - 1].steps) ? occurrence + 1 : 0;
              if (layers.size() <= occurrence) layers.resize(occurrence + 1);
              layers[occurrence].push_back(q);
          } }
        for (const std::vector<size_t>& layer : layers) {
        const Chain& lead = chains[layer[0]];
        const Shape& sh = shape_of(lead.shape);
        const bool any_shared = !(sh.shared_pre.empty() && sh.shared_body.empty());
        float s_star = 0.f;
        if (!sh.shared_pre.empty()) s_star = pre_scalar(lead, sh, sh.shared_pre[0]);
        else if (!sh.shared_body.empty() && lead.R > 0) s_star = body_scalar(lead, sh, 0, sh.shared_body[0]);
        uint32_t want; std::memcpy(&want, &s_star, 4);
        auto same = [&](float v) { uint32_t u; std::memcpy(&u, &v, 4); return u == want; };
        Family fam;
        for (size_t q : layer) {
            if (lead.small) break;
            const Chain& c = chains[q];
            bool ok = chain_n(c) == chain_n(lead);
            for (size_t x = 0; ok && x < sh.NXP; ++x) { const float* p = post_ptr(c, sh, x); ok = p != nullptr && p == post_ptr(lead, sh, x); }
            const size_t shift = lead.steps - c.steps;
            for (size_t t = 0; ok && t < c.steps; ++t) { const float* p = seq_ptr(c, sh, t); ok = p != nullptr && p == seq_ptr(lead, sh, shift + t); }
            if (ok && any_shared) {              // every scalar the shared denominators stand for carries the same bits
                for (uint32_t sl : sh.shared_pre) ok = ok && same(pre_scalar(c, sh, sl));
                for (size_t r = 0; ok && r < c.R; ++r) for (uint32_t sl : sh.shared_body) ok = ok && same(body_scalar(c, sh, r, sl));
            }
            if (!ok) continue;
            fam.chain.push_back(q);
            if (fam.chain.size() == 16) { families.push_back(std::move(fam)); fam = Family(); }
        }
        if (fam.chain.size() >= 2) families.push_back(std::move(fam));
        }
        i = j;
    }
    // (after a split at 16 the later part is a family of its own: its first chain is its longest, the others suffixes of it)
    families.erase(std::remove_if(families.begin(), families.end(), [&](const Family& f) { return f.chain.size() < 2 || chains[f.chain[0]].small; }), families.end());
    if (families.empty()) return;
    // kernels: one per (shape, family size); a family whose kernel does not exist yet runs as before
    struct Launch { std::shared_ptr<JitSlot> slot; std::vector<size_t> rows; };
    std::vector<Launch> launches;
    std::unordered_map<uint64_t, size_t> launch_of;
    for (size_t f = 0; f < families.size(); ++f) {
        const Chain& lead = chains[families[f].chain[0]];
        const size_t K = families[f].chain.size();
        const uint64_t lkey = ((uint64_t)lead.shape << 8) | K;
        auto known = launch_of.find(lkey);
        if (known == launch_of.end()) {
            const std::string key = merge_shapes_[(size_t)lead.shape] + " chains " + std::to_string(K) + " sden 1";
            std::shared_ptr<JitSlot>& slot = merged_kernels_[key];
            if (!slot || (jit_mode == FMHIP_JIT_SYNC && slot->state.load(std::memory_order_acquire) == JitSlot::QUEUED)) {
                RolledBody body = merge_shape_bodies_[(size_t)lead.shape];
                body.chains = (uint32_t)K; body.shared_den = true;
                std::string source = jit_generate_rolled_source(body);
                if (source.empty()) continue;
                jit().record(jit_describe(body));
                slot = jit().request_source(std::move(source), 8, jit_mode == FMHIP_JIT_SYNC);
            }
            known = launch_of.emplace(lkey, launches.size()).first;
            launches.push_back({ slot, {} });
        }
        launches[known->second].rows.push_back(f);
    }
    // the original of a replicated component and its copies go together or not at all: a copy left behind would have nobody to carry its
    // order (run_plan, run_dags) — if any set is split, nothing is merged in this flush
    std::vector<std::vector<char>> taken(groups.size()), staken(small.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) taken[gi].assign(groups[gi].size(), 0);
    for (size_t gi = 0; gi < small.size(); ++gi) staken[gi].assign(small[gi].members.size(), 0);
    auto mark = [&](const Chain& c) -> char& { return c.small ? staken[c.group][c.member] : taken[c.group][c.member]; };
    for (const Launch& l : launches) {
        if (!l.slot || l.slot->state.load(std::memory_order_acquire) != JitSlot::READY) continue;
        for (size_t f : l.rows) for (size_t q : families[f].chain) mark(chains[q]) = 1;
    }
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const std::vector<BigDag>& g = groups[gi];
        for (size_t mi = 0; mi < g.size(); ++mi) {
            if (g[mi].described()) continue;
            size_t e = mi + 1;
            while (e < g.size() && g[e].described()) ++e;
            for (size_t q = mi + 1; q < e; ++q) if (taken[gi][q] != taken[gi][mi]) return;
        }
    }
    for (size_t gi = 0; gi < small.size(); ++gi) {
        const std::vector<Dag>& g = small[gi].members;
        for (size_t mi = 0; mi < g.size(); ++mi) {
            if (g[mi].order.empty()) continue;                           // (a copy that exists as a description: vectors, outputs and scalars only)
            size_t e = mi + 1;
            while (e < g.size() && g[e].order.empty()) ++e;
            for (size_t q = mi + 1; q < e; ++q) if (staken[gi][q] != staken[gi][mi]) return;
        }
    }
    for (Launch& l : launches) {
        if (!l.slot || l.slot->state.load(std::memory_order_acquire) != JitSlot::READY || l.rows.empty()) continue;
        const Shape& sh = shape_of(chains[families[l.rows[0]].chain[0]].shape);
        const size_t K = families[l.rows[0]].chain.size();
        const int64_t n = chain_n(chains[families[l.rows[0]].chain[0]]);
        auto section_words = [&](const Chain& c) { return sh.NXO + (sh.NS0 + (size_t)c.R * sh.LS + sh.NS2 + 1) / 2; };
        size_t rw = 0;
        for (size_t f : l.rows) {
            size_t w = 1 + K + chains[families[f].chain[0]].steps + sh.NXP + 1;
            for (size_t q : families[f].chain) w += section_words(chains[q]);
            rw = std::max(rw, w);
        }
        const size_t max_rows = std::min((size_t)1024, ring_cap_ / (rw * 8 + 256));
        if (max_rows == 0) continue;
        // (rows of one launch have vectors of one length: families are looked for within a flush, whose components of a shape and length
        // share a group; a launch over rows of another length would be a different grid)
        std::vector<size_t> rows_n;
        for (size_t f : l.rows) if (chain_n(chains[families[f].chain[0]]) == n) rows_n.push_back(f);
        for (size_t off = 0; off < rows_n.size(); off += max_rows) {
            const size_t count = std::min(max_rows, rows_n.size() - off);
            std::vector<uint64_t> table(count * rw, 0);
            struct Out { size_t chain; Buffer* buf; };
            std::vector<Out> outs;
            size_t n_ops = 0, n_vec_in = 0;
            std::vector<fmhip_moments> all;
            ReduceRequest every{ 0.0, nullptr, nullptr, false };
            RedLaunch red;
            // common rows (as run_peeled): families that read the same vectors with the same scalars — the parameter sets of a Jacobian
            // batch at an exercise date before their bumped parameter matters — are ONE row; their chains share moments and stored values
            std::vector<uint32_t> row_of(count);
            std::vector<size_t> family_of_row;                           // launch row → index into rows_n (offset from `off`)
            std::unordered_multimap<uint64_t, uint32_t> seen;
            std::vector<uint64_t> keys;
            std::vector<std::pair<size_t, size_t>> out_span;             // per launch row: its outs [begin, end)
            try {
                for (size_t r = 0; r < count; ++r) {
                    const Family& fam = families[rows_n[off + r]];
                    const Chain& lead = chains[fam.chain[0]];
                    const size_t r_new = family_of_row.size();
                    uint64_t* row = table.data() + r_new * rw;
                    std::fill(row, row + rw, (uint64_t)0);
                    const size_t T = lead.steps;
                    row[0] = (uint64_t)T;
                    for (size_t t = 0; t < T; ++t) row[1 + K + t] = (uint64_t)(uintptr_t)seq_ptr(lead, sh, t);
                    for (size_t x = 0; x < sh.NXP; ++x) row[1 + K + T + x] = (uint64_t)(uintptr_t)post_ptr(lead, sh, x);
                    { float s_star = 0.f; bool any = false;
                      if (!sh.shared_pre.empty()) { any = true; s_star = pre_scalar(lead, sh, sh.shared_pre[0]); }
                      else if (!sh.shared_body.empty() && lead.R > 0) { any = true; s_star = body_scalar(lead, sh, 0, sh.shared_body[0]); }
                      uint32_t bits = 0; if (any) std::memcpy(&bits, &s_star, 4);
                      row[1 + K + T + sh.NXP] = bits; }
                    size_t at = 1 + K + T + sh.NXP + 1;
                    for (size_t k = 0; k < K; ++k) {
                        const Chain& c = chains[fam.chain[k]];
                        row[1 + k] = (uint64_t)(T - c.steps) | ((uint64_t)at << 32);
                        uint64_t* sec = row + at;
                        float* sc = reinterpret_cast<float*>(sec + sh.NXO);
                        for (size_t i = 0; i < sh.NS0; ++i) sc[i] = pre_scalar(c, sh, i);
                        for (size_t it = 0; it < c.R; ++it)
                            for (size_t m = 0; m < sh.LS; ++m) sc[sh.NS0 + it * sh.LS + m] = body_scalar(c, sh, it, m);
                        for (size_t i = 0; i < sh.NS2; ++i) sc[sh.NS0 + (size_t)c.R * sh.LS + i] = post_scalar(c, sh, i);
                        at += section_words(c);
                    }
                    if (COMMON_ROWS && count > 1) {
                        uint64_t h = 0x9e3779b97f4a7c15ull;
                        for (size_t w = 0; w < rw; ++w) { h = (h ^ row[w]) * 0xff51afd7ed558ccdull; h ^= h >> 31; }
                        bool common = false;
                        auto range = seen.equal_range(h);
                        for (auto it = range.first; it != range.second && !common; ++it)
                            if (std::memcmp(keys.data() + (size_t)it->second * rw, row, rw * 8) == 0) { row_of[r] = it->second; common = true; }
                        if (common) { ++n_common_rows_; continue; }
                        seen.emplace(h, (uint32_t)r_new);
                        keys.insert(keys.end(), row, row + rw);
                    }
                    row_of[r] = (uint32_t)r_new;
                    family_of_row.push_back(r);
                    n_vec_in += T + sh.NXP;
                    const size_t out_begin = outs.size();
                    for (size_t k = 0; k < K; ++k) {
                        const Chain& c = chains[fam.chain[k]];
                        uint64_t* sec = row + (size_t)(row[1 + k] >> 32);
                        for (size_t m = 0; m < sh.NXO; ++m) { Buffer* nb = new_buffer(n); outs.push_back({ fam.chain[k], nb }); sec[m] = (uint64_t)(uintptr_t)nb->ptr; }
                        n_ops += sh.body->peel.pre.size() + (size_t)c.R * sh.P + sh.body->peel.post.size();
                    }
                    out_span.push_back({ out_begin, outs.size() });
                }
                const size_t launch_rows = family_of_row.size();
                table.resize(launch_rows * rw);
                const int64_t tiles = (n + FM_UNIT_ELEMS - 1) / FM_UNIT_ELEMS;
                if (async_moments_) { every.dev_out = arena_alloc(launch_rows * K); if (!every.dev_out) { for (Out& o : outs) buffer_unref(o.buf); continue; } }
                else { all.resize(launch_rows * K); every.host_out = all.data(); }
                DevRolledArgs args{};
                args.n = n; args.tiles_per_row = (uint32_t)tiles; args.row_words = (uint32_t)rw; args.iterations = 0; args.pad = (uint32_t)K;
                args.dump = (uint64_t)(uintptr_t)dump_dev_;
                const size_t table_bytes = table.size() * 8;
                const size_t ring_off = ring_reserve(table_bytes);
                std::memcpy((char*)ring_host_ + ring_off, table.data(), table_bytes);
                hip_check(hipMemcpyAsync((char*)ring_dev_ + ring_off, (char*)ring_host_ + ring_off, table_bytes, hipMemcpyHostToDevice, stream_), "merged row table H2D");
                const uint64_t* rows_arg = (const uint64_t*)((char*)ring_dev_ + ring_off);
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                if (profiling_) { hip_check(hipEventCreate(&ev0), "hipEventCreate"); hip_check(hipEventCreate(&ev1), "hipEventCreate"); hip_check(hipEventRecord(ev0, stream_), "hipEventRecord"); }
                void* params[] = { &args, &rows_arg };
                red_begin(red, (int)launch_rows, (int)K, (size_t)tiles, every.host_out, every.dev_out);
                args.shift = 0.0; args.partials = (double*)red.partials; args.results = (double*)red.results; args.counters = counters_dev_;
                args.done_flag = const_cast<uint64_t*>(red.poll_flag); args.done_value = red.done_value;
                hip_check(hipModuleLaunchKernel(l.slot->fn_table, (unsigned)tiles, (unsigned)launch_rows, 1, FM_BLOCK, 1, 1, 0, stream_, params, nullptr), "launch merged kernel");
                if (profiling_) { hip_check(hipEventRecord(ev1, stream_), "hipEventRecord"); profile_events_.push_back({ ev0, ev1 });
                                  profile_tags_.push_back({ (int)(n_ops / launch_rows), (int)(n_vec_in / launch_rows), (int)(K * sh.NXO), (int)K, (int)launch_rows, 4, n }); }
                n_launches_++; n_jit_launches_++; n_rolled_launches_++; n_merged_launches_++; n_merged_chains_ += (int64_t)(count * K);
                n_ops_executed_ += (int64_t)n_ops;
                algorithmic_bytes_ += 4 * n * (int64_t)(n_vec_in + outs.size());
                bytes_written_ += 4 * n * (int64_t)outs.size();
                every.done = true;
                red_wait(red, (int)launch_rows, (int)K, every.host_out);
            } catch (...) { red_release(red); for (Out& o : outs) buffer_unref(o.buf); throw; }
            red_release(red);
            // the chains of a family that was a common row receive the vectors its first family stored (one more reference each)
            if (family_of_row.size() != count) {
                const size_t n_first = outs.size();
                for (size_t r = 0; r < count; ++r) {
                    const size_t lr = row_of[r];
                    if (family_of_row[lr] == r) continue;
                    const Family& fam = families[rows_n[off + r]];
                    size_t k = 0;
                    for (size_t o = out_span[lr].first; o < out_span[lr].second && o < n_first; ++o, ++k) { Buffer* b = outs[o].buf; b->refs++; outs.push_back({ fam.chain[k / std::max<size_t>(1, sh.NXO)], b }); }
                }
            }
            // the moments go to the chains' roots; stored values become vectors; expressions are dismantled
            for (size_t r = 0; r < count; ++r) {
                const Family& fam = families[rows_n[off + r]];
                for (size_t k = 0; k < K; ++k) {
                    Node* root = chains[fam.chain[k]].root;
                    if (every.dev_out) arena_assign(root, (double*)every.dev_out + ((size_t)row_of[r] * K + k) * 4);
                    else { const fmhip_moments& m = all[(size_t)row_of[r] * K + k]; root->moments[0] = m.sum; root->moments[1] = m.sumsq; root->moments[2] = m.min; root->moments[3] = m.max; root->has_moments = true; }
                }
            }
            std::vector<Node*> done;
            for (Out& o : outs) {                                       // (the one value a chain stores is its root)
                const Chain& c = chains[o.chain];
                if (c.small) { commit_node(c.root, o.buf); done.push_back(c.root); continue; }
                BigDag& b = groups[c.group][c.member];
                const size_t pos = c.plan->rolled.peeled.post_out[0];
                if (b.described()) commit_described(b, pos, o.buf);
                else { Node* nd = b.order[pos]; commit_node(nd, o.buf); done.push_back(nd); }
            }
            for (Node* nd : done) nd->refs_int++;
            for (Node* nd : done) drop_expression(nd);
            for (Node* nd : done) { nd->refs_int--; node_maybe_free(nd); }
            for (size_t r = 0; r < count; ++r)
                for (size_t q : families[rows_n[off + r]].chain) {
                    const Chain& c = chains[q];
                    if (sh.NXO == 0 && !c.root->buf) { c.root->discarded = true; c.root->refs_int++; drop_expression(c.root); c.root->refs_int--; }
                    if (!c.small) { BigDag& b = groups[c.group][c.member]; if (b.described()) for (Buffer*& t : b.temp) if (t) { buffer_unref(t); t = nullptr; } }
                    mark(c) = 2;                                        // has run
                }
        }
    }
    // what has run leaves its group (the others keep their order: the first member of a group carries the order for its copies)
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        std::vector<BigDag>& g = groups[gi];
        bool any = false;
        for (char t : taken[gi]) any |= t == 2;
        if (!any) continue;
        std::vector<BigDag> rest;
        for (size_t mi = 0; mi < g.size(); ++mi) if (taken[gi][mi] != 2) rest.push_back(std::move(g[mi]));
        g.swap(rest);
    }
    for (size_t gi = 0; gi < small.size(); ++gi) {
        std::vector<Dag>& g = small[gi].members;
        bool any = false;
        for (char t : staken[gi]) any |= t == 2;
        if (!any) continue;
        std::vector<Dag> rest;
        for (size_t mi = 0; mi < g.size(); ++mi) if (staken[gi][mi] != 2) rest.push_back(std::move(g[mi]));
        g.swap(rest);
    }
}

void Engine::run_big_group(std::vector<BigDag>& group, ReduceRequest* rr) {
    HostTimer timer(HostProfile::RUN_BIG);
    BigDag& g0 = group[0];
    auto planned = plan_cache_.find(g0.hash);
    if (planned != plan_cache_.end() && planned->second.sig != g0.sig) planned = plan_cache_.end();       // hash collision: general path, nothing cached
    const bool collision = planned == plan_cache_.end() && plan_cache_.count(g0.hash) != 0;
    if (planned != plan_cache_.end()) { run_plan(planned->second, group, rr); return; }
    // First component of this shape.  Its loop, if it has one, is found first: a periodic stretch of the order becomes a rolled loop (one
    // launch) as soon as its kernel exists; segments remain its fallback.
    BigPlan plan;
    size_t first_with_nodes = 0;
    while (first_with_nodes < group.size() && group[first_with_nodes].described()) ++first_with_nodes;
    if (first_with_nodes == group.size()) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "a group of copies without their original");
    plan_loop(plan, group[first_with_nodes]);
    plan.sig = g0.sig;
    plan.discards_root = g0.discard_root;
    plan.segs_missing = true;
    if (collision) {                                        // (practically never) nothing is cached: segments for the members with nodes, the plan they leave for the copies
        std::vector<BigDag> described, with_nodes;
        for (BigDag& b : group) (b.described() ? described : with_nodes).push_back(std::move(b));
        plan_segments(plan, with_nodes);
        plan.segs_missing = false;
        if (!described.empty()) run_plan(plan, described);
        return;
    }
    // The plan is written down WITHOUT its segments: where the whole component is one launch of a kernel that exists already (the peeled
    // form, from the code-object caches or the build-time pack) nothing else is ever needed; run_plan cuts the segments — by running the
    // component through the general path — the first time that launch is not available (until round 5 every shape's first occurrence ran as
    // segments on the interpreter, whatever kernels existed: a tenth of a second per calibration, and again for every variant of a shape
    // the escape policy moves through).
    BigPlan& cached = plan_cache_[g0.hash];
    cached = std::move(plan);
    run_plan(cached, group, rr);
}

bool Engine::try_fused(const std::vector<Node*>& roots) {
    std::vector<Dag> dags(1);
    if (!build_dag(roots, dags[0])) return false;
    return run_dags(dags);
}

void Engine::materialize(const std::vector<Node*>& targets) {
    ++flush_seq_;
    for (Node* t : targets)
        if (t->discarded && !t->buf) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "the value of this vector does not exist: it was given up (fmhip_vec_give_up_values: only its moments were taken), or lost in a launch that failed");
    expand_replicas_below(targets);                 // a single expression is executed, not everything pending: descriptions of copies it touches become nodes first
    for (Node* t : targets) {
        if (t->buf) continue;
        if (try_fused({ t })) continue;
        // Too large for one launch: cut the component into consecutive segments (run_big_group)
        std::vector<BigDag> one(1);
        if (!build_big({ t }, one[0])) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "expression too large");
        run_big_group(one);
    }
}

// Execute everything that is pending and still referenced.  Pending results that share intermediates form one
// connected component and run as ONE multi-output launch; components with identical structure (same ops, different
// vectors and scalars — e.g. all LIBOR components of an Euler step) are batched as rows of one launch.
void Engine::flush_all() {
    HostTimer timer(HostProfile::FLUSH);
    require_init();
    if (late_count() >= late_eager()) drain_late();
    ++flush_seq_;
    for (int round = 0; round < 1000000; ++round) {
        std::unique_ptr<HostTimer> t_components(new HostTimer(HostProfile::FLUSH_COMPONENTS));
        std::vector<Node*> roots;                   // live pending vectors nobody pending depends on
        for (Node* nd = pending_head_.pend_next; nd != &pending_head_; nd = nd->pend_next) if (nd->refs_int == 0 && nd->refs_ext > 0 && !nd->discarded) roots.push_back(nd);
        if (roots.empty()) {
            Node* live = nullptr;
            for (Node* nd = pending_head_.pend_next; nd != &pending_head_; nd = nd->pend_next) if (nd->refs_ext > 0 && !nd->discarded) { live = nd; break; }
            if (!live) return;
            materialize({ live });
            continue;
        }
        std::sort(roots.begin(), roots.end(), [](Node* a, Node* b) { return a->id < b->id; });
        // connected components of the pending graph (union-find over root indices)
        std::vector<int> parent(roots.size());
        for (size_t i = 0; i < roots.size(); ++i) parent[i] = (int)i;
        auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        const uint64_t ep = ++epoch_;
        std::vector<Node*> stack;
        for (size_t i = 0; i < roots.size(); ++i) {
            stack.push_back(roots[i]);
            roots[i]->mark = ep; roots[i]->tmp_id = (int)i;
            while (!stack.empty()) {
                Node* nd = stack.back(); stack.pop_back();
                for (int k = 0; k < nd->n_in; ++k) {
                    Node* c = nd->in[k];
                    if (c->buf) continue;
                    if (c->mark == ep) { const int a = find((int)i), b = find(c->tmp_id); if (a != b) parent[b] = a; }
                    else { c->mark = ep; c->tmp_id = (int)i; stack.push_back(c); }
                }
            }
        }
        std::unordered_map<int, std::vector<Node*>> comps;
        std::vector<int> comp_order;
        for (size_t i = 0; i < roots.size(); ++i) {
            const int c = find((int)i);
            if (!comps.count(c)) comp_order.push_back(c);
            comps[c].push_back(roots[i]);
        }
        t_components.reset();
        // one DAG per component; group identical structures
        std::unordered_map<std::string, std::vector<Dag>> groups;
        std::vector<std::string> group_order;
        std::vector<Node*> leftovers;
        bool expanded = false;
        for (int c : comp_order) {
            Dag d;
            if (!build_dag(comps[c], d)) { for (Node* r : comps[c]) leftovers.push_back(r); continue; }
            // A component of a replicated graph runs with its copies as further rows — if it still is exactly what was replicated: all
            // its operations belong to ONE live description and nothing but the replicated roots escapes from it.  Otherwise every
            // description it touches becomes ordinary nodes first, and the round starts again on the larger pending graph.
            if (d.rep_any) {
                ReplicaGroup* g = clean_replica_group(d.rep_id, d.rep_uniform);
                for (size_t k = 0; g && k < d.outs.size(); ++k) if (d.outs[k]->rep_root < 0 || d.outs[k]->rep_copy) g = nullptr;
                if (!g) { const size_t before = replicas_.size(); expand_replicas_below(comps[c]); expanded |= replicas_.size() != before; }
                if (expanded) break;                                 // (an expansion may release nodes other components were found through)
            }
            std::string key = d.sig; key.push_back('#'); key += std::to_string(d.roots[0]->n);
            if (!groups.count(key)) group_order.push_back(key);
            groups[key].push_back(std::move(d));
        }
        if (expanded) continue;
        std::vector<SmallGroup> waiting;                                  // groups whose components may be chains of a merged launch (merge_families): run behind the large components
        auto run_small = [&](std::vector<Dag>& members, const Dag& proto, std::vector<std::pair<ReplicaGroup*, std::vector<int>>>& done, bool late) {
            const size_t max_batch = 1024;
            bool ran = true;
            try {
                for (size_t off = 0; off < members.size(); off += max_batch) {
                    std::vector<Dag> part(std::make_move_iterator(members.begin() + off), std::make_move_iterator(members.begin() + std::min(members.size(), off + max_batch)));
                    if (!run_dags(part, nullptr, nullptr, nullptr, &proto)) {
                        ran = false;
                        for (Dag& d : part) for (Node* r : d.roots) { if (late) { if (!r->buf) materialize({ r }); } else leftovers.push_back(r); }
                    }
                }
            } catch (...) { replicas_after_failure(done); throw; }
            if (ran) for (auto& kv : done) replica_roots_done(kv.first, kv.second);
        };
        for (const std::string& key : group_order) {
            std::vector<Dag>& g = groups[key];
            // members: every component of this structure, each followed by its copies that exist as a description
            std::vector<Dag> members;
            std::vector<std::pair<ReplicaGroup*, std::vector<int>>> done;
            const Dag proto = g[0];                                   // carries the structure for launches that start with another member
            const SmallMatch* chain_of = (MERGE_CHAINS && want_root_moments_ && jit_mode != FMHIP_JIT_OFF) ? match_small(proto) : nullptr;
            for (Dag& d : g) {
                ReplicaGroup* rg = d.rep_any ? clean_replica_group(d.rep_id, d.rep_uniform) : nullptr;
                if (rg) {
                    std::vector<int> roots_done;
                    for (Node* o : d.outs) roots_done.push_back(o->rep_root);
                    done.push_back({ rg, std::move(roots_done) });
                    const size_t at = members.size();
                    members.push_back(std::move(d));
                    for (int j = 0; j < rg->n_copies; ++j) members.push_back(replica_dag(members[at], rg, j));
                } else members.push_back(std::move(d));
            }
            if (chain_of) {
                waiting.emplace_back();
                SmallGroup& sg = waiting.back();
                sg.members = std::move(members); sg.proto = proto; sg.match = chain_of; sg.done = std::move(done);
                continue;
            }
            run_small(members, proto, done, false);
        }
        // (nothing large in this flush: the waiting groups have no family to join)
        if (leftovers.empty()) { for (SmallGroup& sg : waiting) run_small(sg.members, sg.proto, sg.done, true); waiting.clear(); }
        // components that do not fit one launch: cut into segments; components of identical shape share the cuts and the launches
        if (!leftovers.empty()) {
            std::unordered_map<Node*, int> comp_of;                      // root -> component key
            for (int c : comp_order) for (Node* r : comps[c]) comp_of[r] = c;
            std::unordered_set<int> big_comps;
            std::vector<int> big_order;
            for (Node* r : leftovers) { const int c = comp_of[r]; if (big_comps.insert(c).second) big_order.push_back(c); }
            // grouped by a 64-bit hash of (shape signature, vector length): the signatures are ~10 KB strings, hashed ONCE in build_big
            std::unordered_map<uint64_t, std::vector<BigDag>> big_groups;
            std::vector<uint64_t> big_group_order;
            for (int c : big_order) {
                std::vector<Node*> roots;
                for (Node* r : comps[c]) if (!r->buf) roots.push_back(r);
                if (roots.empty()) continue;
                BigDag b;
                if (!build_big(roots, b)) throw Error(FMHIP_ERR_PROGRAM_LIMIT, "pending expression too large");
                if (b.rep_any) {                                         // as above
                    ReplicaGroup* g = clean_replica_group(b.rep_id, b.rep_uniform);
                    for (size_t i = 0; g && i < b.order.size(); ++i) if (b.escapes[i] && (b.order[i]->rep_root < 0 || b.order[i]->rep_copy)) g = nullptr;
                    if (!g) { const size_t before = replicas_.size(); expand_replicas_below(roots); expanded |= replicas_.size() != before; }
                    if (expanded) break;
                }
                uint64_t key = b.hash ^ ((uint64_t)roots[0]->n * 0x9E3779B97F4A7C15ull);
                auto& members = big_groups[key];
                if (members.empty()) big_group_order.push_back(key);
                else if (members[0].sig != b.sig || members[0].roots[0]->n != roots[0]->n) {       // hash collision (practically never): a group of its own
                    key = key * 0xff51afd7ed558ccdull + big_group_order.size() + 1;
                    big_group_order.push_back(key);
                    big_groups[key].push_back(std::move(b));
                    continue;
                }
                members.push_back(std::move(b));
            }
            if (expanded) { for (SmallGroup& sg : waiting) run_small(sg.members, sg.proto, sg.done, true); continue; }
            // members of every group: its components, each followed by its copies that exist as a description
            std::vector<std::vector<BigDag>> all_members(big_group_order.size());
            std::vector<std::vector<std::pair<ReplicaGroup*, std::vector<int>>>> all_done(big_group_order.size());
            for (size_t ki = 0; ki < big_group_order.size(); ++ki) {
                std::vector<BigDag>& originals = big_groups[big_group_order[ki]];
                std::vector<BigDag>& members = all_members[ki];
                std::vector<std::pair<ReplicaGroup*, std::vector<int>>>& done = all_done[ki];
                for (BigDag& b : originals) {
                    ReplicaGroup* rg = b.rep_any ? clean_replica_group(b.rep_id, b.rep_uniform) : nullptr;
                    if (!rg) { members.push_back(std::move(b)); continue; }
                    auto view = std::make_shared<ReplicaView>();
                    view->g = rg;
                    const size_t m = b.order.size();
                    view->rep_index.resize(m); view->rep_root.resize(m); view->scalar.resize(m);
                    std::vector<int> roots_done;
                    for (size_t i = 0; i < m; ++i) {
                        const Node* nd = b.order[i];
                        view->rep_index[i] = nd->rep_index; view->rep_root[i] = nd->rep_root; view->scalar[i] = nd->scalar;
                        if (nd->rep_root >= 0) roots_done.push_back(nd->rep_root);
                    }
                    done.push_back({ rg, std::move(roots_done) });
                    const size_t n_map = rg->leaf_from.size();
                    const size_t at = members.size();
                    members.push_back(std::move(b));
                    for (int j = 0; j < rg->n_copies; ++j) {
                        BigDag r;
                        r.n = members[at].n; r.view = view; r.copy = j;
                        r.leaves.reserve(members[at].leaves.size());
                        for (Node* l : members[at].leaves) r.leaves.push_back(l->leaf_rep_id == rg->id ? rg->leaf_to[(size_t)j * n_map + (size_t)l->leaf_rep_index] : l);
                        r.temp.assign(m, nullptr);
                        members.push_back(std::move(r));
                    }
                }
            }
            // components of one loop shape that read the same vectors, one a suffix of the other's: one launch per family size (merge_families)
            try { merge_families(all_members, waiting); }
            catch (...) { for (auto& done : all_done) replicas_after_failure(done); for (SmallGroup& sg : waiting) replicas_after_failure(sg.done); throw; }
            for (SmallGroup& sg : waiting) {
                if (!sg.members.empty()) run_small(sg.members, sg.proto, sg.done, true);
                else for (auto& kv : sg.done) replica_roots_done(kv.first, kv.second);
            }
            waiting.clear();
            for (size_t ki = 0; ki < big_group_order.size(); ++ki) {
                if (!all_members[ki].empty()) {
                    try { run_big_group(all_members[ki]); }
                    catch (...) { for (size_t kj = ki; kj < big_group_order.size(); ++kj) replicas_after_failure(all_done[kj]); throw; }
                }
                for (auto& kv : all_done[ki]) replica_roots_done(kv.first, kv.second);
            }
        }
    }
}

// A copy of the component `d` that exists as a description: the vectors it reads, the root nodes that receive its results, its scalars.
Engine::Dag Engine::replica_dag(const Dag& d, ReplicaGroup* g, int copy) {
    Dag r;
    const size_t n_map = g->leaf_from.size();
    r.leaves.reserve(d.leaves.size());
    for (Node* l : d.leaves) r.leaves.push_back(l->leaf_rep_id == g->id ? g->leaf_to[(size_t)copy * n_map + (size_t)l->leaf_rep_index] : l);
    r.outs.reserve(d.outs.size());
    for (Node* o : d.outs) r.outs.push_back(g->copy_roots[(size_t)copy * g->n_roots + (size_t)o->rep_root]);
    for (const Node* nd : d.order)
        if (op_info(nd->opcode).scalar) {
            const int32_t slot = g->scalar_slot[(size_t)nd->rep_index];
            r.scalars.push_back((float)((slot >= 0 && !g->scalars.empty()) ? g->scalars[(size_t)copy * g->n_scalars + slot] : nd->scalar));
        }
    if (r.scalars.empty()) r.scalars.push_back(0.0f);
    return r;
}

// ---------------------------------------------------------------- reductions

// The stand-alone reduction is the empty program with one fused reduction of its input (compiled once).
Program* Engine::reduce_program() {
    static const char* key = "__reduce1";
    auto it = program_cache_.find(key);
    if (it != program_cache_.end()) return it->second;
    Program* prog = compile({}, 1, {}, { 0 }, nullptr, true);
    if (jit_mode != FMHIP_JIT_OFF) prog->jit = jit().request(prog->proto, jit_mode == FMHIP_JIT_SYNC);      // every getAverage() runs it: specialised from the start (it is in the kernel pack)
    program_cache_[key] = prog;
    return prog;
}

void Engine::reduce(fmhip_vec h, double shift, fmhip_moments* host_out, void* dev_out, RedLaunch* hand_over) {
    require_init();
    end_step_group();
    Node* nd = node(h);
    auto cached = [&]() {
        if (!(nd->has_moments && shift == 0.0 && host_out && !dev_out)) return false;
        host_out->sum = nd->moments[0]; host_out->sumsq = nd->moments[1]; host_out->min = nd->moments[2]; host_out->max = nd->moments[3];
        return true;
    };
    if (cached()) return;
    if (nd->moments_slot && shift == 0.0 && host_out && !dev_out && slot_wait(nd) && cached()) return;
    ++flush_seq_;
    // One expectation is asked for while much else is pending (a caller that records the payoffs of all its products and then takes
    // their averages one by one — 144 per objective evaluation of the LIBOR market model calibration): everything pending runs NOW,
    // components of equal shape as rows of the same launches, and those launches take the moments of their roots along.  The other
    // products' getAverage() calls are answered from what is left with their nodes; the moments are those of the stand-alone
    // reduction to the last bit (one reduction tree per vector: fm_kernel_parts.hpp).
    static const size_t BATCH_PENDING = [] { const char* e = std::getenv("FMHIP_BATCH_EXPECTATIONS"); return e ? (size_t)std::atoll(e) : (size_t)256; }();   // 0 = off
    if (BATCH_PENDING && fusion && fusion_hold != 1 && !nd->buf && shift == 0.0 && host_out && !dev_out && n_pending_ >= BATCH_PENDING && n_pending_ >= 4 * (size_t)std::max(1, nd->weight)) {
        struct Want { bool& w; ~Want() { w = false; } } want{ want_root_moments_ };
        want_root_moments_ = true;
        flush_all();
    }
    if (cached()) return;
    if (nd->moments_slot && shift == 0.0 && host_out && !dev_out && slot_wait(nd) && cached()) return;
    touch(nd);
    RedLaunch deferred;
    struct Defer {                      // the launch that takes the moments hands its wait to this scope (RedLaunch::pending)
        Engine* e; RedLaunch* r; RedLaunch* hand_over;
        Defer(Engine* e_, RedLaunch* r_, RedLaunch* h_) : e(e_), r(r_), hand_over(h_) { e->defer_red_ = r; }
        ~Defer() { e->defer_red_ = nullptr; if (r->pending) { r->pending = false; (void)hipStreamSynchronize(e->stream_); e->red_release(*r); } }      // (an error behind the launch: its buffers go back when it has finished)
        // the moments: waited for here, or — results in a slot of their own, a caller that can wait without the engine lock — by the caller
        void finish() {
            e->defer_red_ = nullptr;
            if (!r->pending) return;
            if (hand_over && r->slot >= 0) { *hand_over = *r; r->pending = false; return; }
            e->red_wait(*r, r->batch, r->n_red, r->host); r->pending = false; e->red_release(*r);
        }
    } defer(this, &deferred, hand_over);
    if (!nd->buf) {
        // `chain.getAverage()`: the expectation of a pending expression that fits one launch is taken in THAT launch (the kernel's
        // fused reduction) instead of a second launch that reads the vector again — one launch and 4 B per path less.  A launch with a
        // fused reduction of a large row has one workgroup per 8192 elements (fine for a chain over two vectors, a starved launch for one
        // over eleven) unless it is small enough to take one UNIT of the reduction tree per workgroup (unit_launch).
        expand_replicas_below({ nd });
        std::vector<Dag> one(1);
        if (fusion && nd->weight <= 4 * FM_MAX_OPS && build_dag({ nd }, one[0]) && (nd->n * (int64_t)one[0].leaves.size() <= (int64_t(1) << 21) || unit_launch(nd->n, 1)) && run_dags(one, &shift, host_out, dev_out)) { defer.finish(); return; }
        // … and of one that takes several launches, in the LAST of them (when its plan exists: from the second time a shape is seen)
        if (fusion && !nd->buf && nd->n > 0) {
            std::vector<BigDag> big(1);
            if (build_big({ nd }, big[0])) {
                ReduceRequest rr{ shift, host_out, dev_out, false };
                run_big_group(big, &rr);
                if (rr.done) { defer.finish(); return; }
            }
        }
        defer.finish();
        defer_red_ = &deferred;
        if (!nd->buf) materialize({ nd });
    }
    Program* prog = reduce_program();
    std::vector<RowSpec> rows(1);
    rows[0].in.push_back(nd->buf->ptr);
    rows[0].scalars = nullptr;
    rows[0].shifts = &shift;
    launch(prog, nd->n, rows, host_out, dev_out);
    defer.finish();
}

int64_t Engine::reduce_batch_begin(const fmhip_vec* hs, int count, const double* shifts) {
    require_init();
    if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
    // Vectors that are still pending: the flush that computes them takes their moments along (rows of the launches that compute them
    // anyway, results into slots of the pinned arena) — no reduction launch, the vectors are not read again.  The ticket remembers the
    // slots; ending it waits for them.  (Vectors computed already, shifts, or a component whose launches cannot take moments: the
    // reduction launch below.)
    static const bool FROM_LAUNCHES = [] { const char* e = std::getenv("FMHIP_MOMENTS_FROM_LAUNCHES"); return !(e && e[0] == '0'); }();
    bool unshifted = true;
    for (int i = 0; shifts && i < count; ++i) unshifted &= shifts[i] == 0.0;
    if (FROM_LAUNCHES && fusion && unshifted && hs) return reduce_batch_begin_from_launches(hs, count);
    const size_t bytes = (size_t)count * 32;
    MomentsTicket t;
    for (size_t i = 0; i < free_tickets_.size(); ++i)
        if (free_tickets_[i].cap >= bytes) { t = free_tickets_[i]; free_tickets_[i] = free_tickets_.back(); free_tickets_.pop_back(); break; }
    try {
        if (!t.host) {
            t.cap = std::max(bytes, size_t(8192));
            hip_check(hipHostMalloc(&t.host, t.cap, hipHostMallocDefault), "hipHostMalloc(moments ticket)");
            hip_check(hipEventCreateWithFlags(&t.event, hipEventDisableTiming), "hipEventCreate(moments ticket)");
        }
        t.count = count;
        reduce_batch(hs, count, shifts, nullptr, t.host);     // the last workgroup of every row stores its moments straight into the block
        hip_check(hipEventRecord(t.event, stream_), "hipEventRecord(moments ticket)");
    } catch (...) { if (t.host) free_tickets_.push_back(t); throw; }
    const int64_t id = next_ticket_++;
    tickets_[id] = t;
    return id;
}

// fmhip_vec_give_up_values: the caller wants the EXPECTATIONS of these vectors and will never read their values.  A vector that is
// still pending and that nobody but the caller references is marked (Node::discard); a flush that takes the moments of its roots along
// (reduce_batch_begin, reduce) then computes it in a launch that takes its moments and does NOT store it (run_plan: a peeled component
// whose root is 'm' in the signature).  One 8 KB store per workgroup at the end of a read-only chain costs such a launch 8-10 % of its
// rate (benchmarks/read_pattern.hip: 6486 → 5893 GB/s; the valuation kernel in isolation 6145 → 6617): the memory system pays for
// turning a stream of reads around for a trickle of writes.  A marked vector that runs through a launch which cannot take its moments
// is stored like any other.
void Engine::give_up_values(const fmhip_vec* hs, int count) {
    require_init();
    if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
    static const bool DISCARD = [] { const char* e = std::getenv("FMHIP_DISCARD_VALUES"); return !(e && e[0] == '0'); }();      // =0: every value is stored (A/B measurement)
    std::vector<Node*> nds((size_t)count);
    for (int i = 0; i < count; ++i) nds[(size_t)i] = node(hs[i]);
    if (!DISCARD) return;
    // nobody but the caller references it — not counting the holds of a live replica description (fmhip_graph_clone) on the roots it
    // replicates (one external reference on the original's root) and on the roots of its copies (one internal reference each)
    auto sole_owner = [&](const Node* nd) {
        int ext = nd->refs_ext, in = nd->refs_int;
        if (nd->rep_id && replica_of(nd)) { if (nd->rep_copy) in -= 1; else if (nd->rep_root >= 0) ext -= 1; }
        return ext == 1 && in == 0;
    };
    for (Node* nd : nds) if (!nd->buf && !nd->moments_blocked && sole_owner(nd)) nd->discard = true;
}

// The expectations of vectors that may still be pending, every one through a slot of the pinned arena (or at hand already): the flush
// that computes the pending ones takes their moments along; what has none afterwards (computed earlier, a launch that could not take
// them along, a vector somebody writes into) is reduced by ONE launch of the reduction program into arena slots.
int64_t Engine::reduce_batch_begin_from_launches(const fmhip_vec* hs, int count) {
    end_step_group();
    std::vector<Node*> nds((size_t)count);
    for (int i = 0; i < count; ++i) nds[(size_t)i] = node(hs[i]);
    for (int i = 1; i < count; ++i)
        if (nds[(size_t)i]->n != nds[0]->n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
    bool pending = false;
    for (Node* nd : nds) pending |= !nd->buf && !nd->discarded;
    if (pending) {
        struct Mode { Engine* e; ~Mode() { e->want_root_moments_ = false; e->async_moments_ = false; } } mode{ this };
        want_root_moments_ = true; async_moments_ = true;
        flush_all();
    }
    std::vector<fmhip_vec> rest;
    std::vector<size_t> rest_index;
    for (int i = 0; i < count; ++i) {
        Node* nd = nds[(size_t)i];
        if (nd->has_moments || nd->moments_slot) continue;
        rest.push_back(hs[i]); rest_index.push_back((size_t)i);
    }
    MomentsTicket t;
    t.count = count; t.slots.resize((size_t)count, nullptr); t.ready.resize((size_t)count);
    if (!rest.empty()) {
        double* slots = arena_alloc(rest.size());
        if (!slots) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "too many expectations for one ticket");
        reduce_batch(rest.data(), (int)rest.size(), nullptr, nullptr, slots);
        for (size_t k = 0; k < rest.size(); ++k) t.slots[rest_index[k]] = reinterpret_cast<volatile uint64_t*>(slots + k * 4);
    }
    for (int i = 0; i < count; ++i) {
        Node* nd = nds[(size_t)i];
        if (t.slots[(size_t)i]) continue;
        if (nd->has_moments) t.ready[(size_t)i] = { nd->moments[0], nd->moments[1], nd->moments[2], nd->moments[3] };
        else t.slots[(size_t)i] = nd->moments_slot;
    }
    const int64_t id = next_ticket_++;
    tickets_[id] = std::move(t);
    return id;
}

// fmhip_reduce_moments_batch_device on vectors that may still be pending: as reduce_batch_begin_from_launches — the flush that computes them
// takes their moments along (values that were given up are not stored at all) — but the caller wants the 32-byte blocks in ONE device
// buffer, in the order asked (the send buffer of its RCCL exchange), not on the host: a one-wave kernel behind the launches collects them
// from their slots of the pinned arena, which the device reads through the same mapping it wrote them through.  Until round 4 a caller
// with a communicator of its own (lmm_hip --world N) had to flush first and pay a reduction launch that read every value again.
void Engine::reduce_batch_device_from_launches(const fmhip_vec* hs, int count, void* dev_out) {
    end_step_group();
    std::vector<Node*> nds((size_t)count);
    for (int i = 0; i < count; ++i) nds[(size_t)i] = node(hs[i]);
    for (int i = 1; i < count; ++i)
        if (nds[(size_t)i]->n != nds[0]->n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
    bool pending = false;
    for (Node* nd : nds) pending |= !nd->buf && !nd->discarded;
    if (pending) {
        struct Mode { Engine* e; ~Mode() { e->want_root_moments_ = false; e->async_moments_ = false; } } mode{ this };
        want_root_moments_ = true; async_moments_ = true;
        flush_all();
    }
    // one block of the arena for whatever has no slot yet, taken BEFORE the vectors are sorted: if the arena wraps here, the slots written so
    // far are collected into their nodes (has_moments) now and not between two looks at them
    double* block = arena_alloc((size_t)count);
    if (!block) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "too many expectations for one call");
    std::vector<fmhip_vec> rest;
    std::vector<uint64_t> src((size_t)count, 0);
    size_t used = 0;
    std::vector<size_t> rest_index;
    for (int i = 0; i < count; ++i) {
        Node* nd = nds[(size_t)i];
        if (nd->moments_slot) src[(size_t)i] = (uint64_t)(uintptr_t)nd->moments_slot;
        else if (nd->has_moments) { double* at = block + 4 * used++; std::memcpy(at, nd->moments, 32); src[(size_t)i] = (uint64_t)(uintptr_t)at; }
        else { rest.push_back(hs[i]); rest_index.push_back((size_t)i); }
    }
    if (!rest.empty()) {           // computed earlier, or by a launch that could not take the moments along: one reduction launch, into the block
        double* at = block + 4 * used;
        reduce_batch(rest.data(), (int)rest.size(), nullptr, nullptr, at);
        for (size_t k = 0; k < rest.size(); ++k) src[rest_index[k]] = (uint64_t)(uintptr_t)(at + 4 * k);
        used += rest.size();
    }
    { volatile uint64_t* tail = reinterpret_cast<volatile uint64_t*>(block + 4 * used); for (size_t i = 0; i < ((size_t)count - used) * 4; ++i) tail[i] = 0; }    // (unused slots: no sentinels left behind)
    for (int off = 0; off < count; off += FM_GATHER_MAX) {
        DevGatherArgs a{};
        a.count = (uint32_t)std::min(FM_GATHER_MAX, count - off);
        std::memcpy(a.src, src.data() + off, (size_t)a.count * 8);
        hip_check(launch_gather_moments(a, (double*)dev_out + (size_t)off * 4, stream_), "launch fm_gather_moments_kernel");
        n_launches_++;
    }
}

void Engine::reduce_batch_device(const fmhip_vec* hs, int count, const double* shifts, void* dev_out) {
    require_init();
    if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
    static const bool FROM_LAUNCHES = [] { const char* e = std::getenv("FMHIP_MOMENTS_FROM_LAUNCHES"); return !(e && e[0] == '0'); }();
    bool unshifted = true;
    for (int i = 0; shifts && i < count; ++i) unshifted &= shifts[i] == 0.0;
    bool worth = false;                                          // something is pending, given up, or has its moments already
    if (FROM_LAUNCHES && fusion && unshifted) for (int i = 0; i < count && !worth; ++i) { const Node* nd = node(hs[i]); worth = !nd->buf || nd->has_moments || nd->moments_slot; }
    if (worth) reduce_batch_device_from_launches(hs, count, dev_out);
    else reduce_batch(hs, count, shifts, nullptr, dev_out);
}

Engine::MomentsTicket Engine::ticket_take(int64_t id) {
    auto it = tickets_.find(id);
    if (it == tickets_.end()) throw Error(FMHIP_ERR_INVALID_HANDLE, "unknown (or already ended) expectation ticket");
    MomentsTicket t = std::move(it->second);
    tickets_.erase(it);
    for (size_t i = 0; i < t.slots.size(); ++i) {                 // moments taken by the launches that computed the vectors: wait for their slots
        volatile uint64_t* slot = t.slots[i];
        if (!slot) continue;
        auto complete = [&]() { return slot[0] != MOMENTS_SENTINEL && slot[1] != MOMENTS_SENTINEL && slot[2] != MOMENTS_SENTINEL && slot[3] != MOMENTS_SENTINEL; };
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = complete();
        for (uint32_t spins = 1; !arrived; ++spins) {
            if (has_late()) drain_late(late_portion());
            else {
#if defined(__x86_64__)
                _mm_pause();
#endif
            }
            arrived = complete();
            if (!arrived && (spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        // Not there after 2 ms of spinning: the launch that writes it is far down the queue.  Keep watching THIS slot, asleep in between —
        // never hipStreamSynchronize: that waits for everything queued behind as well (a driver that records batch b+1 before it asks
        // for batch b's expectations lost its overlap at every second batch that way: the device drained, then idled 2 ms per batch
        // while the host recorded the next one).  A stream that has run dry without the slot being written is an error.
        for (uint32_t naps = 1; !arrived; ++naps) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            arrived = complete();
            if (!arrived && (naps & 63u) == 0) {
                const hipError_t q = hipStreamQuery(stream_);
                if (q == hipSuccess) { arrived = complete(); break; }
                if (q != hipErrorNotReady) hip_check(q, "hipStreamQuery(moments)");
            }
        }
        if (!arrived) throw Error(FMHIP_ERR_HIP, "the moments of a vector never arrived");
        std::atomic_thread_fence(std::memory_order_acquire);
        uint64_t v[4] = { slot[0], slot[1], slot[2], slot[3] };
        std::memcpy(&t.ready[i], v, 32);
        t.slots[i] = nullptr;
    }
    return t;
}

void Engine::ticket_retire(MomentsTicket& t) {
    t.slots.clear(); t.ready.clear();
    if (t.host) free_tickets_.push_back(t);
    t = MomentsTicket();
}

void Engine::reduce_batch(const fmhip_vec* hs, int count, const double* shifts, fmhip_moments* host_out, void* dev_out) {
    HostTimer timer(HostProfile::REDUCE);
    require_init();
    end_step_group();
    if (count <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "count must be positive");
    std::vector<Node*> nds((size_t)count);
    for (int i = 0; i < count; ++i) nds[(size_t)i] = node(hs[i]);
    for (int i = 1; i < count; ++i)
        if (nds[(size_t)i]->n != nds[0]->n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "batched reduction over vectors of different size");
    bool pending = false;
    for (Node* nd : nds) { touch(nd); pending |= !nd->buf; }
    if (pending) flush_all();                                   // one batched flush instead of one launch per vector
    for (Node* nd : nds) if (!nd->buf) materialize({ nd });
    Program* prog = reduce_program();
    const int max_rows = 1024;
    for (int off = 0; off < count; off += max_rows) {
        const int m = std::min(max_rows, count - off);
        std::vector<RowSpec> rows((size_t)m);
        for (int i = 0; i < m; ++i) {
            rows[(size_t)i].in.push_back(nds[(size_t)(off + i)]->buf->ptr);
            rows[(size_t)i].scalars = nullptr;
            rows[(size_t)i].shifts = shifts ? &shifts[off + i] : nullptr;
        }
        launch(prog, nds[0]->n, rows, host_out ? host_out + off : nullptr, dev_out ? (char*)dev_out + (size_t)off * 32 : nullptr);
    }
}

// ---------------------------------------------------------------- explicit programs

fmhip_program Engine::program_create(const fmhip_prog_op* ops, int n_ops, int n_in, const int32_t* outs, int n_out,
                                     const int32_t* reds, int n_red) {
    require_init();
    if (n_ops < 0 || n_out < 0 || n_red < 0 || (n_ops > 0 && !ops) || (n_out > 0 && !outs) || (n_red > 0 && !reds))
        throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad program description");
    if (n_out == 0 && n_red == 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "a program needs an output or a reduction");
    std::vector<SsaOp> s(n_ops);
    for (int i = 0; i < n_ops; ++i) s[i] = { ops[i].opcode, ops[i].a, ops[i].b, ops[i].c, ops[i].scalar };
    Program* p = compile(s, n_in, std::vector<int>(outs, outs + n_out), std::vector<int>(reds, reds + n_red), nullptr, true);
    if (jit_mode != FMHIP_JIT_OFF) p->jit = jit().request(p->proto, jit_mode == FMHIP_JIT_SYNC);    // explicit = declared hot
    const int64_t id = next_id_++;
    programs_[id] = p;
    return id;
}

std::string Engine::program_source(const fmhip_prog_op* ops, int n_ops, int n_in, const int32_t* outs, int n_out, const int32_t* reds, int n_red) {
    if (n_ops < 0 || n_out < 0 || n_red < 0 || (n_ops > 0 && !ops) || (n_out > 0 && !outs) || (n_red > 0 && !reds))
        throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad program description");
    std::vector<SsaOp> s(n_ops);
    for (int i = 0; i < n_ops; ++i) s[i] = { ops[i].opcode, ops[i].a, ops[i].b, ops[i].c, ops[i].scalar };
    std::unique_ptr<Program> p(compile(s, n_in, std::vector<int>(outs, outs + n_out), std::vector<int>(reds, reds + n_red), nullptr, true));
    return jit_generate_source(p->proto);
}

Program* Engine::program(fmhip_program h) {
    auto it = programs_.find(h);
    if (it == programs_.end()) throw Error(FMHIP_ERR_INVALID_HANDLE, "invalid program handle " + std::to_string(h));
    return it->second;
}

void Engine::program_release(fmhip_program h) {
    require_init();
    Program* p = program(h);
    programs_.erase(h);
    if (--p->refs == 0) delete p;
}

void Engine::program_run(fmhip_program h, int batch, const fmhip_vec* inputs, fmhip_vec* outputs, bool into,
                         const double* shifts, fmhip_moments* moments, void* dev_moments) {
    require_init();
    Program* p = program(h);
    if (batch <= 0) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "batch must be positive");
    if (!inputs || (p->n_out > 0 && !outputs)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null handle array");
    std::vector<RowSpec> rows(batch);
    std::vector<Node*> in_nodes((size_t)batch * p->n_in);
    int64_t n = -1;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < p->n_in; ++k) {
            Node* nd = node(inputs[(size_t)b * p->n_in + k]);
            if (n < 0) n = nd->n;
            if (nd->n != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "program inputs differ in size");
            in_nodes[(size_t)b * p->n_in + k] = nd;
        }
    for (Node* nd : in_nodes) { touch(nd); if (!nd->buf) materialize({ nd }); }
    std::vector<Buffer*> fresh;
    std::vector<Node*> out_nodes;
    try {
        for (int b = 0; b < batch; ++b) {
            for (int k = 0; k < p->n_in; ++k) rows[b].in.push_back(in_nodes[(size_t)b * p->n_in + k]->buf->ptr);
            for (int k = 0; k < p->n_out; ++k) {
                if (into) {
                    Node* o = node(outputs[(size_t)b * p->n_out + k]);
                    if (o->n != n) throw Error(FMHIP_ERR_SIZE_MISMATCH, "program output differs in size");
                    touch(o);
                    if (!o->buf) materialize({ o });
                    if (o->refs_int > 0) { flush_all(); materialize_deferred(); }      // overwritten in place: whoever still reads the old contents (pending expressions, recipes of deferred values) is computed first
                    o->has_moments = false; o->moments_slot = nullptr;      // overwritten
                    make_private(o);
                    rows[b].out.push_back(o->buf->ptr);
                } else {
                    Buffer* bf = new_buffer(n);
                    fresh.push_back(bf);
                    rows[b].out.push_back(bf->ptr);
                }
            }
            rows[b].scalars = nullptr;
            rows[b].shifts = shifts;
        }
        launch(p, n, rows, moments, dev_moments);
    } catch (...) { for (Buffer* b : fresh) buffer_unref(b); throw; }
    if (!into)
        for (size_t i = 0; i < fresh.size(); ++i) { Node* nd = new_node(n); nd->buf = fresh[i]; outputs[i] = nd->id; }
}

// ---------------------------------------------------------------- brownian increments

void Engine::bm_generate(int64_t seed, int n_steps, int n_factors, int64_t n_paths, int64_t path_offset,
                         const double* dt, fmhip_vec* out) {
    require_init(); check_n(n_paths);
    if (n_steps <= 0 || n_factors <= 0 || !dt || !out || path_offset < 0)
        throw Error(FMHIP_ERR_INVALID_ARGUMENT, "bad Brownian motion description");
    if (n_factors > 32768) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "more than 32768 factors");      // one launch keeps whole steps together (grid.y)
    const int64_t n_streams = (int64_t)n_steps * n_factors;
    const int64_t stride = (n_paths + 63) & ~int64_t(63);              // every vector 256-B aligned
    Buffer* slab = new_buffer(std::max<int64_t>(stride, 64) * n_streams);
    slab->refs = 0;
    void* sq_dev = nullptr; size_t sq_cap = 0;
    try {
        float* st = (float*)ensure_stage((size_t)n_streams * 4);       // one entry per stream: the kernel reads it with a scalar load, no division
        for (int i = 0; i < n_steps; ++i) {
            if (!(dt[i] >= 0.0)) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "negative time step");
            const float sq = (float)std::sqrt(dt[i]);                  // (float)Math.sqrt(timeStep), BrownianMotionCudaWithRandomVariableCuda.java:170
            for (int f = 0; f < n_factors; ++f) st[(size_t)i * n_factors + f] = sq;
        }
        sq_dev = pool_.alloc((size_t)n_streams * 4, &sq_cap);
        hip_check(hipMemcpyAsync(sq_dev, st, (size_t)n_streams * 4, hipMemcpyHostToDevice, stream_), "sqrt_dt H2D");
        hip_check(hipStreamSynchronize(stream_), "sync");
        if (n_paths > 0) {
            const int64_t chunk = 32768 - (32768 % n_factors);         // grid.y limit; keep whole steps together
            for (int64_t s0 = 0; s0 < n_streams; s0 += chunk) {
                const int64_t ns = std::min(chunk, n_streams - s0);
                DevBmArgs a{};
                a.slab = slab->ptr + s0 * stride;
                a.sqrt_dt = (const float*)sq_dev + s0;
                a.stride_floats = stride; a.n_paths = n_paths; a.path_offset = path_offset;
                a.key0 = (uint32_t)(uint64_t)seed; a.key1 = (uint32_t)((uint64_t)seed >> 32);
                a.n_factors = (uint32_t)n_factors; a.stream0 = (uint32_t)s0;
                hipEvent_t ev0 = nullptr, ev1 = nullptr;            // fmhip_profile_enable: the generator's launches are bracketed like program launches
                if (profiling_) { hip_check(hipEventCreate(&ev0), "hipEventCreate"); hip_check(hipEventCreate(&ev1), "hipEventCreate"); hip_check(hipEventRecord(ev0, stream_), "hipEventRecord"); }
                hip_check(launch_bm(a, (uint32_t)ns, stream_), "launch fm_bm_kernel");
                if (profiling_) { hip_check(hipEventRecord(ev1, stream_), "hipEventRecord"); profile_events_.push_back({ ev0, ev1 });
                                  profile_tags_.push_back({ 0, 0, (int)ns, 0, 1, 3, n_paths }); }
                algorithmic_bytes_ += 4 * n_paths * ns;
                bytes_written_ += 4 * n_paths * ns;
                n_launches_++;
            }
        }
    } catch (...) {
        if (sq_dev) pool_.release(sq_dev, sq_cap);
        slab->refs = 1; buffer_unref(slab);
        throw;
    }
    pool_.release(sq_dev, sq_cap);
    const uint32_t bm_id = next_bm_id_++;
    for (int64_t s = 0; s < n_streams; ++s) {
        Buffer* v = new Buffer();
        v->ptr = slab->ptr + s * stride; v->cap = 0; v->refs = 1; v->parent = slab;
        slab->refs++;
        Node* nd = new_node(n_paths);
        nd->buf = v;
        nd->bm_id = bm_id; nd->bm_step = (int32_t)(s / n_factors); nd->bm_steps = n_steps;
        out[s] = nd->id;
    }
}

// ---------------------------------------------------------------- pool entry points

// (values that were left unstored are computed and stored first: their recipes may be all that keeps other vectors alive)
void Engine::pool_clean() { require_init(); if (has_late()) drain_late(); materialize_deferred(); hip_check(hipStreamSynchronize(stream_), "sync"); pool_.purge(); }

void Engine::pool_purge() {
    require_init();
    if (has_late()) drain_late();
    hip_check(hipStreamSynchronize(stream_), "sync");
    pool_.purge();
    for (auto& kv : plan_cache_) for (BigPlan::Seg& seg : kv.second.segs) { if (--seg.prog->refs == 0) delete seg.prog; delete seg.prog_red; }
    plan_cache_.clear();
    for (auto it = program_cache_.begin(); it != program_cache_.end();) { if (--it->second->refs == 0) delete it->second; it = program_cache_.erase(it); }
    schedule_cache_.clear(); schedule_cache_bytes_ = 0; dag_policies_.clear(); policy_reset();      // … and what was learnt about the shapes' handles
}

void Engine::engine_stats(fmhip_engine_stats_t* out) {
    require_init();
    if (!out) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null stats pointer");
    std::memset(out, 0, sizeof *out);
    out->size = (int64_t)sizeof *out;
    out->kernel_launches = n_launches_; out->specialised_launches = n_jit_launches_; out->interpreter_launches = n_interpreter_launches_;
    out->algorithmic_bytes = algorithmic_bytes_; out->algorithmic_bytes_written = bytes_written_;
    out->values_deferred = n_deferred_total_; out->values_deferred_now = (int64_t)n_deferred_; out->values_demanded = n_demanded_;
    out->pending_operations = (int64_t)n_pending_;
    out->peak_bytes_reserved = pool_.peak_reserved;
    out->late_releases_while_waiting = n_late_waiting_; out->late_releases_at_once = n_late_at_once_; out->late_release_nanoseconds = late_ns_;
    out->merged_launches = n_merged_launches_; out->merged_chains = n_merged_chains_; out->common_rows = n_common_rows_;
}

void Engine::pool_stats(fmhip_pool_stats_t* out) {
    require_init();
    if (!out) throw Error(FMHIP_ERR_INVALID_ARGUMENT, "null stats pointer");
    if (has_late()) drain_late();
    size_t fr = 0, tot = 0;
    hip_check(hipMemGetInfo(&fr, &tot), "hipMemGetInfo");
    out->bytes_reserved = pool_.reserved; out->bytes_in_use = pool_.in_use; out->bytes_cached = pool_.cached;
    out->device_bytes_free = (int64_t)fr; out->device_bytes_total = (int64_t)tot;
    out->n_alloc_hits = pool_.hits; out->n_alloc_misses = pool_.misses;
    out->n_live_vectors = (int64_t)nodes_.size();
    out->n_kernel_launches = n_launches_; out->n_ops_executed = n_ops_executed_;
}

// ---------------------------------------------------------------- measurement

void Engine::profile_enable(bool on) {
    require_init();
    if (!on) { for (auto& p : profile_events_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); } profile_events_.clear(); profile_tags_.clear(); }
    profiling_ = on;
}

void Engine::profile_read(double* ms_total, int64_t* n) {
    require_init();
    hip_check(hipStreamSynchronize(stream_), "sync");
    double total = 0.0;
    // FMHIP_PROFILE_DUMP=1: per program shape (ops/in/out/red/rows/tier), launches, device time and algorithmic GB/s
    const bool dump = std::getenv("FMHIP_PROFILE_DUMP") != nullptr;
    struct Agg { long long launches = 0; double ms = 0.0, bytes = 0.0; };
    std::map<std::string, Agg> agg;
    for (size_t i = 0; i < profile_events_.size(); ++i) {
        auto& p = profile_events_[i];
        float ms = 0.0f;
        hip_check(hipEventElapsedTime(&ms, p.first, p.second), "hipEventElapsedTime");
        total += ms;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
        if (dump && i < profile_tags_.size()) {
            const ProfileTag& t = profile_tags_[i];
            char key[128];
            std::snprintf(key, sizeof key, "ops %3d in %2d out %d red %d rows %4d n %9lld %s", t.n_ops, t.n_in, t.n_out, t.n_red, t.batch, (long long)t.n, t.tier == 4 ? "merged chains" : t.tier == 3 ? "fm_bm_kernel" : t.tier ? "specialised" : "interpreter");
            Agg& a = agg[key];
            a.launches++; a.ms += ms; a.bytes += 4.0 * (double)t.n * (t.n_in + t.n_out) * t.batch;
        }
    }
    if (dump) {
        std::vector<std::pair<std::string, Agg>> v(agg.begin(), agg.end());
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.ms > b.second.ms; });
        std::fprintf(stderr, "[fmhip profile] %zu launches, %.3f ms of kernel time\n", profile_events_.size(), total);
        for (const auto& kv : v)
            std::fprintf(stderr, "  %-72s %7lld launches %9.3f ms %8.1f us each %8.0f GB/s\n", kv.first.c_str(), kv.second.launches, kv.second.ms,
                         kv.second.ms / kv.second.launches * 1e3, kv.second.ms > 0 ? kv.second.bytes / (kv.second.ms * 1e-3) / 1e9 : 0.0);
    }
    profile_tags_.clear();
    if (ms_total) *ms_total = total;
    if (n) *n = (int64_t)profile_events_.size();
    profile_events_.clear();
}

} // namespace fm
